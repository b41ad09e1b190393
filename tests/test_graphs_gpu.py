"""hipGraph replays (hvrnet_amd/graphs.py) against the eager path they were captured from: the same kernels in the same
per-stream order, so the per-class detection arrays must be bit-identical -- clip mode (frame groups, RPN side stream and
read-out side stream inside the capture) and stream mode (per-frame cache, padded first / last windows), both heads, several
replays with changing inputs; and a window with a short frame must come back through the exact eager path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hvrnet_amd  # noqa: E402
from hvrnet_amd import synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from hvrnet_amd.graphs import GraphedClip, GraphedStream  # noqa: E402

DEV = 'cuda:0'
HW, PAD = (150, 250), (160, 256)


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


def _check(kind, got, want):
    if kind == 'hvr':
        assert len(got) == len(want) == 2 and all(_same(g, w) for g, w in zip(got, want))
    else:
        assert _same(got, want)


@pytest.mark.parametrize('kind', ['hvr', 'selsa'])
def test_graphed_clip_equals_the_eager_window(kind):
    T, n_prop = 5, 24
    make = hvr_config if kind == 'hvr' else selsa_config
    model = hvrnet_amd.build_model(make(frame_interval=T // 2, nms_post=n_prop), S.synth_state_dict(kind), torch.bfloat16, DEV)
    metas = [S.synth_meta(HW, PAD) for _ in range(T)]
    clips = [torch.cat([S.synth_frame(10 * c + i, img_hw=HW, pad_hw=PAD) for i in range(T)], 0).to(DEV) for c in range(3)]
    g = GraphedClip(model, clips[0], metas, rescale=True)
    n_det = 0
    for rep in range(2):
        for clip in clips:
            pend = g.run(clip)
            got = pend.result()
            assert not pend.respeculated
            with torch.no_grad():
                c4 = model(img=clip, img_meta=metas, backbone_feat=True)[0]
                want = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
            _check(kind, got, want)
            n_det += sum(len(r) for r in (got[-1] if kind == 'hvr' else got))
    assert n_det > 0


@pytest.mark.parametrize('kind', ['hvr', 'selsa'])
def test_two_graphed_clips_in_flight_on_two_streams_equal_the_eager_windows(kind):
    """bench.py's headline region: windows replayed from hipGraphs on two HIP streams in turn, two DIFFERENT clips in flight at a
    time.  Each lane has its own graph, output buffers and per-stream scratch, so every window must equal the eager result of
    its own clip whatever the other lane is running."""
    T, n_prop = 5, 24
    make = hvr_config if kind == 'hvr' else selsa_config
    model = hvrnet_amd.build_model(make(frame_interval=T // 2, nms_post=n_prop), S.synth_state_dict(kind), torch.bfloat16, DEV)
    metas = [S.synth_meta(HW, PAD) for _ in range(T)]
    clips = [torch.cat([S.synth_frame(10 * c + i, img_hw=HW, pad_hw=PAD) for i in range(T)], 0).to(DEV) for c in range(4)]
    want = []
    with torch.no_grad():
        for clip in clips:
            c4 = model(img=clip, img_meta=metas, backbone_feat=True)[0]
            want.append(model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True))
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    gcs = []
    for st in lanes:
        with torch.cuda.stream(st):
            gcs.append(GraphedClip(model, clips[0], metas, rescale=True, n_out=1))
    torch.cuda.synchronize()
    pend = [None, None]
    order = [0, 1, 2, 3, 3, 2, 1, 0, 1, 3]
    for i, c in enumerate(order):
        k = i % 2
        if pend[k] is not None:
            _check(kind, pend[k][0].result(), want[pend[k][1]])
        with torch.cuda.stream(lanes[k]):
            lanes[k].wait_stream(torch.cuda.current_stream())
            pend[k] = (gcs[k].run(clips[c]), c)
    for k in range(2):
        assert not pend[k][0].respeculated
        _check(kind, pend[k][0].result(), want[pend[k][1]])


def test_graphed_clip_with_two_clips_per_graph_equals_the_eager_windows():
    """GraphedClip(windows=2): the frames of two independent clips go through the backbone as ONE batch (30 frames give layer 3's
    convs a chip-covering grid of 288 x 256 tiles), res5 / RPN / RoIAlign / head / read-out run per clip; each clip's detections
    equal its own eager window."""
    T, n_prop = 5, 24
    model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n_prop), S.synth_state_dict('hvr'), torch.bfloat16, DEV)
    metas = [S.synth_meta(HW, PAD) for _ in range(T)]
    clips = [torch.cat([S.synth_frame(10 * c + i, img_hw=HW, pad_hw=PAD) for i in range(T)], 0).to(DEV) for c in range(4)]
    want = []
    with torch.no_grad():
        for clip in clips:
            c4 = model(img=clip, img_meta=metas, backbone_feat=True)[0]
            want.append(model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True))
    g = GraphedClip(model, torch.cat(clips[:2], 0), metas + metas, rescale=True, windows=2)
    for a, b in ((0, 1), (2, 3), (3, 0)):
        pend = g.run(torch.cat([clips[a], clips[b]], 0))
        assert len(pend) == 2
        _check('hvr', pend[0].result(), want[a])
        _check('hvr', pend[1].result(), want[b])


def test_graphed_clip_with_a_short_frame_takes_the_exact_path():
    """A harsh RPN NMS leaves some frame with fewer than nms_post proposals: the replay's speculative result is discarded and
    the window is re-run through the exact (ragged) eager path -- same answer as eager forward_feat(speculate=False)."""
    T = 3
    cfg = hvr_config(frame_interval=1, nms_post=400)
    cfg.test_cfg.rpn.nms_thr = 0.02
    model = hvrnet_amd.build_model(cfg, S.synth_state_dict('hvr'), torch.float32, DEV)
    metas = [S.synth_meta() for _ in range(T)]
    clip = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(DEV)
    g = GraphedClip(model, clip, metas, rescale=True, warmup=1)
    pend = g.run()
    got = pend.result()
    assert pend.respeculated
    with torch.no_grad():
        c4 = model(img=clip, img_meta=metas, backbone_feat=True)[0]
        want = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True, speculate=False)
    _check('hvr', got, want)


@pytest.mark.parametrize('kind', ['hvr', 'selsa'])
def test_graphed_stream_equals_the_cached_frame_loop(kind):
    """GraphedStream (graph F per arriving frame, graph W per emitted window) driven in the reference loop's shape --
    the first frame pushed (T+1)/2 times, the last one repeated while the remaining centres are emitted
    (tools/test.py:201-212,257-300) -- against VideoWindowRunner(cache_frames=True) on the same video."""
    from hvrnet_amd.window import VideoWindowRunner
    fi, n_prop = 2, 24
    T = 2 * fi + 1
    make = hvr_config if kind == 'hvr' else selsa_config
    model = hvrnet_amd.build_model(make(frame_interval=fi, nms_post=n_prop), S.synth_state_dict(kind), torch.bfloat16, DEV)
    frames = [S.synth_frame(i, img_hw=HW, pad_hw=PAD).to(DEV) for i in range(8)]
    meta = S.synth_meta(HW, PAD)
    with torch.no_grad():
        want = VideoWindowRunner(model, T, cache_frames=True).run_video(frames, [meta] * len(frames))
    gs = GraphedStream(model, frames[0], meta, rescale=True, fewrow_split=False)
    got = {}
    # first frame: the deque is padded with copies until it holds (T + 1) / 2 entries
    gs.push(frames[0])
    for _ in range((T + 1) // 2 - 1):
        gs.repeat_last()
    filled = (T + 1) // 2
    emitted = 0
    for i in range(1, len(frames) - 1):
        gs.push(frames[i])
        filled += 1
        if filled >= T:
            got[emitted] = gs.emit().result()
            emitted += 1
    # last frame: pad to T - 1, then append + emit the remaining centres
    gs.push(frames[-1])
    filled += 1
    while filled < T:
        gs.repeat_last()
        filled += 1
    got[emitted] = gs.emit().result()
    emitted += 1
    while emitted < len(frames):
        gs.repeat_last()
        got[emitted] = gs.emit().result()
        emitted += 1
    assert sorted(got) == sorted(want) == list(range(len(frames)))
    for off in want:
        _check(kind, got[off], want[off])


@pytest.mark.parametrize('window_cus', [None, 96])
@pytest.mark.parametrize('kind', ['hvr', 'selsa'])
def test_pipelined_stream_equals_the_sequential_stream(kind, window_cus):
    """push_async / commit: frame i + 1's per-frame part (graph FC, own stream / pool / scratch) runs beside window i's
    relation stages and read-out (graph W).  Same kernels on the same rows: every emitted window equals the one the
    sequential push() / emit() stream gives, bit for bit, over two passes of the video.  window_cus = 96: graph W replayed on a
    stream confined to 96 CUs (native.cu_masked_stream) -- the same graph, ordered against commit() by events."""
    fi, n_prop = 2, 24
    T = 2 * fi + 1
    make = hvr_config if kind == 'hvr' else selsa_config
    model = hvrnet_amd.build_model(make(frame_interval=fi, nms_post=n_prop), S.synth_state_dict(kind), torch.bfloat16, DEV)
    frames = [S.synth_frame(i, img_hw=HW, pad_hw=PAD).to(DEV) for i in range(7)]
    meta = S.synth_meta(HW, PAD)
    seq = GraphedStream(model, frames[0], meta, rescale=True, fewrow_split=False)
    want = []
    for rep in range(2):
        for f in frames:
            seq.push(f)
            want.append(seq.emit().result())
    gs = GraphedStream(model, frames[0], meta, rescale=True, fewrow_split=False, window_cus=window_cus)
    order = frames + frames
    got, pend = [], None
    gs.push_async(order[0])
    for i in range(len(order)):
        gs.commit()                       # frame i's rows enter the window
        if i + 1 < len(order):
            gs.push_async(order[i + 1])   # frame i + 1 computes beside window i
        nxt = gs.emit()
        if pend is not None:
            got.append(pend.result())
        pend = nxt
    got.append(pend.result())
    assert len(got) == len(want)
    for g, w in zip(got, want):
        _check(kind, g, w)


@pytest.mark.parametrize('lanes,window_cus', [(2, 256), (3, None)])
def test_frame_lanes_give_the_windows_of_the_sequential_stream(lanes, window_cus):
    """GraphedStream(frame_lanes=L): L frames in flight, each with its own graph FC / stream / staging rows / graph C, taken in turn
    by push_async() and committed in arrival order (tools/stream_bench.py: one frame's chain of ~150 small launches leaves most of the
    chip idle, two chains side by side overlap).  Every emitted window equals the sequential push() / emit() stream's bit for bit,
    over two passes of the video; the lanes are primed with L frames before the first commit."""
    fi, n_prop = 2, 24
    model = hvrnet_amd.build_model(hvr_config(frame_interval=fi, nms_post=n_prop), S.synth_state_dict('hvr'), torch.bfloat16, DEV)
    frames = [S.synth_frame(i, img_hw=HW, pad_hw=PAD).to(DEV) for i in range(7)]
    meta = S.synth_meta(HW, PAD)
    seq = GraphedStream(model, frames[0], meta, rescale=True, fewrow_split=False)
    order = frames + frames
    want = []
    for f in order:
        seq.push(f)
        want.append(seq.emit().result())
    gs = GraphedStream(model, frames[0], meta, rescale=True, fewrow_split=False, window_cus=window_cus, frame_lanes=lanes)
    assert len(gs._lanes) == lanes
    got, pend, fed = [], None, 0
    for _ in range(lanes):
        gs.push_async(order[fed])
        fed += 1
    with pytest.raises(AssertionError):
        gs.push_async(order[0])           # every lane holds a frame: commit() first
    for i in range(len(order)):
        gs.commit()                       # the OLDEST frame in flight enters the window
        if fed < len(order):
            gs.push_async(order[fed])     # the freed lane takes the next frame
            fed += 1
        nxt = gs.emit()
        if pend is not None:
            got.append(pend.result())
        pend = nxt
    got.append(pend.result())
    assert len(got) == len(want)
    for g, w in zip(got, want):
        _check('hvr', g, w)


@pytest.mark.parametrize('kind', ['hvr', 'selsa'])
def test_graphed_stream_with_look_ahead_batches_gives_the_same_frames(kind):
    """`lookahead` frames through the per-frame part in one batch (graph FB), then one `advance(i)` + `emit()` per output
    frame: the rows of a frame do not depend on what else is in its batch, so every emitted window equals the
    one-frame-at-a-time stream's bit for bit."""
    fi, n_prop, B = 1, 24, 4
    T = 2 * fi + 1
    make = hvr_config if kind == 'hvr' else selsa_config
    model = hvrnet_amd.build_model(make(frame_interval=fi, nms_post=n_prop), S.synth_state_dict(kind), torch.bfloat16, DEV)
    frames = torch.cat([S.synth_frame(i, img_hw=HW, pad_hw=PAD) for i in range(2 * B)], 0).to(DEV)
    meta = S.synth_meta(HW, PAD)
    one = GraphedStream(model, frames[0:1], meta, rescale=True, fewrow_split=False)
    want = []
    for i in range(frames.shape[0]):
        one.push(frames[i:i + 1])
        if i >= T - 1:
            want.append(one.emit().result())
    look = GraphedStream(model, frames[0:1], meta, rescale=True, lookahead=B, fewrow_split=False)
    got, seen = [], 0
    for b0 in range(0, frames.shape[0], B):
        look.push_batch(frames[b0:b0 + B])
        for i in range(B):
            look.advance(i)
            seen += 1
            if seen >= T:
                got.append(look.emit().result())
    assert len(got) == len(want) == frames.shape[0] - T + 1
    for g, w in zip(got, want):
        _check(kind, g, w)


def test_one_frame_split_k_path_tracks_the_unsplit_kernels():
    """GraphedStream's default one-frame graphs run the few-row split-K kernels (native.fewrow_split): one 600x1000 frame through
    backbone / res5 / RPN / proposals / RoIAlign / fc_new_1 with the split on and off.  Same products, f32 sums in slice order:
    the C4 map agrees to bf16 rounding noise (not bit for bit -- that is why the switch is opt-in), the proposal sets overlap,
    and a replayed stream with the split on returns detections for every window."""
    from hvrnet_amd import native
    T, n_prop = 3, 64
    model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n_prop), S.synth_state_dict('hvr'), torch.bfloat16, DEV)
    meta = S.synth_meta()
    frames = [S.synth_frame(i).to(DEV) for i in range(4)]
    with torch.no_grad():
        c4_ref = model(img=frames[0], img_meta=[meta], backbone_feat=True)[0]
        e_ref = model.frame_tensors(c4_ref, meta)
        with native.fewrow_split(True):
            c4 = model(img=frames[0], img_meta=[meta], backbone_feat=True)[0]
            e = model.frame_tensors(c4, meta)
    assert not torch.equal(c4, c4_ref), 'the split-K route was not taken'
    scale = float(c4_ref.float().abs().max())
    assert float((c4.float() - c4_ref.float()).abs().max()) < 0.03 * scale
    assert float((c4.float() - c4_ref.float()).abs().mean()) < 0.002 * scale
    # proposals: the boxes reappear (mean best IoU > 0.9) -- with random weights the RPN's scores are near-ties and its top-k / NMS
    # order flips for a few of them
    a, b = e['props'][:, :4].float(), e_ref['props'][:, :4].float()
    lt, rb = torch.max(a[:, None, :2], b[None, :, :2]), torch.min(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt + 1).clamp(min=0).prod(-1)
    area = lambda x: (x[:, 2] - x[:, 0] + 1) * (x[:, 3] - x[:, 1] + 1)
    iou = inter / (area(a)[:, None] + area(b)[None] - inter)
    best = iou.max(1).values
    assert float(best.mean()) > 0.9 and float((best > 0.6).float().mean()) > 0.9, best
    gs = GraphedStream(model, frames[0], meta, rescale=True)
    assert gs.fewrow_split
    for i in range(T):
        gs.push(frames[i])
    for f in frames:
        gs.push(f)
        res = gs.emit().result()
        assert len(res) == 2 and all(len(r) == model.bbox_head.num_classes - 1 for r in res)


def test_look_ahead_stream_also_takes_single_frames_and_guards_its_buffers():
    """A look-ahead object can still be fed one frame at a time (graph F was captured against scratch buffers sized by the
    look-ahead batch, not by a one-frame run that a later batch would have outgrown); an output slot cannot be replayed before its
    previous window was read; a graph whose packed weights were rebuilt refuses to replay."""
    fi, n_prop, B = 1, 24, 4
    T = 2 * fi + 1
    model = hvrnet_amd.build_model(hvr_config(frame_interval=fi, nms_post=n_prop), S.synth_state_dict('hvr'), torch.bfloat16, DEV)
    frames = torch.cat([S.synth_frame(i, img_hw=HW, pad_hw=PAD) for i in range(B)], 0).to(DEV)
    meta = S.synth_meta(HW, PAD)
    one = GraphedStream(model, frames[0:1], meta, rescale=True, fewrow_split=False)
    look = GraphedStream(model, frames[0:1], meta, rescale=True, lookahead=B, fewrow_split=False)
    look.push_batch(frames)
    for i in range(B):
        look.advance(i)
    via_batch = look.emit().result()
    for i in range(B):                       # the same frames again, one push() at a time, through graph F of the look-ahead object
        look.push(frames[i:i + 1])
        one.push(frames[i:i + 1])
    _check('hvr', look.emit().result(), via_batch)
    _check('hvr', one.emit().result(), via_batch)
    # the caller's batch buffer may be reused after push_batch: the object keeps its own copy for re-runs
    buf = frames.clone()
    look.push_batch(buf)
    buf.zero_()
    look.advance(0)
    assert torch.equal(look._hist[-1], frames[0:1])
    # n_out = 2: a third emit() without reading the first raises instead of overwriting its pinned buffers
    p1 = one.emit()
    p2 = one.emit()
    with pytest.raises(RuntimeError):
        one.emit()
    p1.result(); p2.result()
    one.emit().result()
    # stale graphs
    g = GraphedClip(model, torch.cat([frames[:T]], 0), [meta] * T, rescale=True, n_out=1)
    g.run().result()
    pend = g.run()
    with pytest.raises(RuntimeError):
        g.run()                              # slot 0 unread
    pend.result()
    hvrnet_amd.set_compute_dtype(model, torch.bfloat16)     # drops (and later rebuilds) every packed weight buffer
    with pytest.raises(RuntimeError):
        g.run()
    with pytest.raises(RuntimeError):
        one.push(frames[0:1])


def test_one_chain_stream_graphs_survive_launches_between_replays():
    """Stream-mode graph F captured as ONE chain (HVR_RPN_SIDE=0: the RPN branch in line, its one-frame proposal call in the chip-wide
    form) replayed with ordinary launches between the replays and no host synchronisation -- the loop of GraphedStream.push().  With a
    hipMemsetAsync node in that chain this sequence took a GPU memory fault on ROCm 7.2 (profiles/r04_graph_memset_fault.txt); the
    zeroing is a kernel now.  In a child process: a fault kills the process that owns the queue."""
    import os, subprocess, sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import torch, sys
sys.path.insert(0, %r)
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedStream
T, dev = 5, 'cuda:0'
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=64), S.synth_state_dict('hvr'), torch.bfloat16, dev)
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev); meta = S.synth_meta()
with torch.no_grad():
    gs = GraphedStream(model, frames[0:1], meta, rescale=True)
    scratch = torch.zeros(1024, device=dev)
    for r in range(3):
        for i in range(T):
            gs.push(frames[i:i + 1]); scratch.add_(1)
        res = gs.emit().result()
    torch.cuda.synchronize()
    print('OK', sum(len(c) for c in res))
""" % root
    r = subprocess.run([_sys.executable, '-c', code], env=dict(os.environ, HVR_RPN_SIDE='0'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
