"""The N > 1 inference path on CPU: two real processes over gloo (the GPU box uses the same code over RCCL).
Ranks take whole videos (no data-path collective); rank 0 ends up with every frame's result in global order."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hvrnet_amd import sharding, window  # noqa: E402


def test_partition_follows_the_reference_greedy_fill():
    # ceil(40 / 2) = 20: rank 0 takes videos while the running frame count stays <= 20
    assert sharding.partition_videos([8, 7, 5, 9, 11], 2) == [[0, 1, 2], [3, 4]]
    # the last rank absorbs the remainder even past the average (imagenet_vid_sequence.py:138-141)
    assert sharding.partition_videos([10, 10, 10, 10, 1], 2) == [[0, 1], [2, 3, 4]]
    assert sharding.partition_videos([3, 3, 3], 1) == [[0, 1, 2]]
    parts = sharding.partition_videos([5] * 17, 8)
    assert sorted(v for p in parts for v in p) == list(range(17)) and all(parts)


class _EchoModel(object):
    """Stands in for the detector: backbone_feat returns the frame id, forward_feat returns the window."""

    def __call__(self, img=None, img_meta=None, backbone_feat=False, forward_feat=False, x=None, **kw):
        return (img,) if backbone_feat else list(x)


def retry_rendezvous(test):
    """A rendezvous on 127.0.0.1 can lose its port to another process between _free_port() and init_process_group (seen once
    in ~50 runs on the build container): one more attempt with a fresh port, then the failure stands."""
    import functools

    @functools.wraps(test)
    def run(*a, **k):
        try:
            return test(*a, **k)
        except Exception:          # noqa: BLE001 -- second attempt reports the real error
            return test(*a, **k)
    return run


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _reap(procs, timeout=180):
    """Both ranks have DELIVERED their results when this is called (the queue reads above would have timed out otherwise): what is
    left is the process group's teardown, which on a loaded host can outlast any fixed wait and can end rank > 0 with a reset
    connection (see _shutdown).  So: wait, end what still runs, and fail only for a rank 0 that exited with an error code of its
    own -- `p.join(60)` + `exitcode == 0` failed once under an 8-job compile on the build container with every result correct."""
    for p in procs:
        p.join(timeout)
    for r, p in enumerate(procs):
        if p.is_alive():
            p.terminate()
            p.join(30)
        elif r == 0:
            assert p.exitcode == 0, 'rank 0 exited with %r after delivering its result' % (p.exitcode,)


def _shutdown():
    # rank 0 hosts the rendezvous store: when it goes away first, the other rank's teardown can see a reset connection
    # (seen as a rare non-zero exit code on a loaded build container) -- the results are already on the queue by then
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def _worker(rank, world, port, lengths, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        def run_video(vid):
            frames = [vid * 1000 + i for i in range(lengths[vid])]
            res = window.VideoWindowRunner(_EchoModel(), 5).run_video(frames, [dict() for _ in frames])
            return [res[i] for i in range(lengths[vid])]

        out = sharding.run_sharded(lengths, run_video, rank, world)
        # the barrier + max-over-ranks timing pattern of bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        q.put((rank, out, float(t.item())))
    finally:
        _shutdown()


@pytest.mark.timeout(600)
@retry_rendezvous
def test_two_ranks_over_gloo_cover_every_frame_once():
    lengths = [6, 9, 4, 7, 5]
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, out, tmax = q.get(timeout=240)
        got[rank] = (out, tmax)
    _reap(procs)
    assert got[1][0] is None and got[0][1] == got[1][1] == 2.0
    out = got[0][0]
    assert len(out) == sum(lengths)
    k = 0
    for vid, n in enumerate(lengths):
        for f in range(n):
            centre = out[k][2]  # window of 5: centre entry is the frame the detection belongs to
            assert centre == vid * 1000 + f
            assert all(vid * 1000 <= w < vid * 1000 + n for w in out[k])  # windows never mix videos
            k += 1


# ------------------------------------------------------------------------------- training step's exchange (config 5)
def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hvrnet_amd import dist_train
        torch.manual_seed(0)  # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
        flat = dist_train.FlatParams(net)
        # parameters and their gradients are views of the flat buffers: nothing to flatten or copy back
        assert all(p.data_ptr() >= flat.flat.data_ptr() for p in net.parameters())
        flat.zero_grad()
        x = torch.full((4, 6), float(rank + 1))
        net(x).sum().backward()                       # rank-dependent gradients, accumulated in place into flat.grad
        mine = [p.grad.clone() for p in net.parameters()]
        world_seen = flat.allreduce_grads()           # one all_reduce of the whole gradient buffer
        dist.barrier()                                # both ranks are done talking before either tears the group down
        # numpy arrays, not tensors: a tensor travels through a torch.multiprocessing queue as a shared-memory handle the RECEIVER fetches from
        # the sender, and a sender that has already exited (this worker, on a loaded host) resets that connection -- by value instead
        q.put((rank, world_seen, [t.numpy().copy() for t in mine], float(flat.grad.sum()), [p.grad.detach().numpy().copy() for p in net.parameters()]))
    finally:
        _shutdown()


@pytest.mark.timeout(300)
@retry_rendezvous
def test_flat_gradient_allreduce_over_gloo_sums_the_replicas():
    """dist_train.FlatParams: the reference's allreduce_grads (dist_utils.py:9-41) as ONE all_reduce over a flat buffer
    the parameters' .grad tensors are views of; the division by world_size is applied by the update kernel."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, world_seen, mine, summed, views = q.get(timeout=240)
        got[rank] = (world_seen, [torch.from_numpy(a) for a in mine], summed, [torch.from_numpy(a) for a in views])
    _reap(procs)
    total = [a + b for a, b in zip(got[0][1], got[1][1])]
    for r in range(world):
        assert got[r][0] == world
        for view, want in zip(got[r][3], total):      # the parameters' .grad views see the reduced values
            assert torch.allclose(view, want)
        assert abs(got[r][2] - sum(float(t.sum()) for t in total)) < 1e-4   # (alignment padding stays zero)
    assert not torch.allclose(got[0][1][0], got[1][1][0])
