"""Headline benchmark: VID key-frame detections / second, R101 Faster-RCNN + HVR (or SELSA) head,
1000x600 frames (padded 608x1008), 300 proposals / frame, T = 15 frames / window.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = one window in CLIP MODE (BASELINE.json configs[2]): all T frames go through
backbone -> res5 -> RPN -> proposals -> RoIAlign -> relation head -> read-out, nothing cached between
steps, and one key-frame detection comes out (per-class arrays on the host, as the reference's
bbox2result returns them).  Frames are already resident in HBM when timing starts.  Ranks run
independent clips (no data-path collective): weak scaling, value = N * steps / max-over-ranks time.

Extra JSON keys: `roofline` (relation core, measured with HIP events inside the timed region),
`kernel_classes` (per-class time / achieved rate from one extra instrumented window after the timed
region) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload, rank 0, N=1).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TF = {'bf16': 2500.0, 'f32': 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--head', default='hvr', choices=['hvr', 'selsa'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--frames', type=int, default=15)
    ap.add_argument('--proposals', type=int, default=300)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train-step', action='store_true', help='skip the training-step side measurement (tools/train_bench.py)')
    ap.add_argument('--inflight', type=int, default=int(os.environ.get('HVR_INFLIGHT', '1')),
                    help='independent windows enqueued on that many HIP streams in turn (throughput mode)')
    ap.add_argument('--breakdown', action='store_true', help='print per-shape conv / gemm times of one window to stderr')
    return ap.parse_args()


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup quota). The GPU box shows 256 logical CPUs under a
    16-CPU cgroup quota; oversubscribing it makes the CPU path ~200x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (IOError, OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(head, T, n_prop, sd):
    """CPU oracle ("port": PyTorch-CPU restatement, oracle/hvr_oracle.py) on a bounded sample of one clip-mode window:
    3 of the T frames through backbone/res5/RPN/proposals/RoIAlign (scaled by T/3) + the full-size relation head
    (M = T * n_prop rows) + read-out, all host cores."""
    from hvrnet_amd import synthetic as S
    from oracle import hvr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    ns = min(3, T)
    imgs = [S.synth_frame(i) for i in range(ns)]
    metas = [S.synth_meta() for _ in range(ns)]
    with torch.no_grad():
        O.resnet_c4(imgs[0][:, :, :128, :128], sd)  # warm-up of the thread pool / allocator
        t0 = time.time()
        c4 = [O.resnet_c4(im, sd) for im in imgs]
        x = torch.cat(c4, 0)
        c5 = O.shared_head(x, sd)
        cls, reg = O.rpn_forward(x, sd)
        base = O.gen_base_anchors(16, [4, 8, 16, 32], [0.5, 1.0, 2.0])
        anchors = O.grid_anchors(base, cls.shape[-2:], 16)
        cfg = dict(O.RPN_TEST_CFG, nms_post=n_prop, max_num=n_prop)
        props = [O.rpn_get_bboxes_single(cls[i], reg[i], anchors, metas[i]['img_shape'], cfg) for i in range(ns)]
        rois = [O.bbox2roi([p]) for p in props]
        feats = torch.cat([O.roi_align(c5[i:i + 1], rois[i], 7, 1.0 / 16, 2) for i in range(ns)], 0)
        t_frames = time.time() - t0
        M = T * n_prop
        g = torch.Generator().manual_seed(0)
        roi_feats = torch.rand((M, 256, 7, 7), generator=g)
        key = T // 2
        cur = dict(start=key * n_prop, length=n_prop)
        t0 = time.time()
        if head == 'hvr':
            cs, rs = O.hvr_head_forward_test(roi_feats, sd, cur, n_prop, T)
        else:
            c, r = O.selsa_head_forward(roi_feats, sd, cur, n_prop, T)
            cs, rs = [c], [r]
        key_rois = torch.cat([torch.zeros(n_prop, 1), props[0][:n_prop, :4]], 1) if props[0].shape[0] >= n_prop else \
            torch.cat([torch.zeros(n_prop, 1), torch.rand(n_prop, 4) * 500], 1)
        for c, r in zip(cs, rs):
            O.get_det_bboxes(key_rois, c, r, (600, 1000, 3), 1.0, True, O.RCNN_TEST_CFG)
        t_head = time.time() - t0
    window_s = t_frames * (T / float(ns)) + t_head
    return dict(value=1.0 / window_s, unit='frames/s', cores=cores, kind='port',
                sample='%d of %d frames through backbone+res5+RPN+proposals+RoIAlign (%.2f s, scaled x%.1f) + full relation head '
                       'M=%d and read-out (%.2f s); torch %d threads' % (ns, T, t_frames, T / float(ns), M, t_head, cores),
                window_seconds=window_s)


def train_step_side_measurement(head):
    """configs[4] beside the headline, never as `value`: one training iteration of the same detector family (HNMBRCNN: 5 videos x 3
    frames in, 3 chosen; SelsaRCNN: 1 key + 2 reference frames) at 600x1000 / 300 proposals, bf16 operands with f32 master weights,
    measured by tools/train_bench.py in its own process after the timed region.  None if that run fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train_bench.py'), '--head', head, '--steps', '5', '--warmup', '2'],
                           capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        d = json.loads(line)
        return dict(iterations_per_s=d['value'], ms_per_iteration=d['ms_per_step'], input_frames_per_s=d['frames_per_s'], dtype=d['dtype'],
                    trainable_params=d['params'], what=d['metric'])
    except Exception as exc:   # noqa: BLE001 -- a side measurement must not take the headline down
        sys.stderr.write('train_step side measurement skipped: %r\n' % (exc,))
        return None


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world)
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import hvrnet_amd
    from hvrnet_amd import native, synthetic as S
    from hvrnet_amd.config import hvr_config, selsa_config

    T, n_prop = args.frames, args.proposals
    assert T % 2 == 1
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    cfg = (hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=T // 2, nms_post=n_prop)
    sd = S.synth_state_dict(args.head)
    model = hvrnet_amd.build_model(cfg, sd, dt, dev)
    # each rank works on its own clip: different synthetic frames per rank
    frames = torch.cat([S.synth_frame(rank * 1000 + i) for i in range(T)], 0).to(dev)  # [T,3,608,1008] resident in HBM
    metas = [S.synth_meta() for _ in range(T)]
    n_keys = []

    lanes = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.inflight))] if args.inflight > 1 else [None]
    if args.inflight > 1 and 'HVR_FRAME_GROUPS' not in os.environ:
        type(model).frame_groups = 1  # the second stream's work comes from the other window instead
    turn = [0]

    def step(prev=None):
        """Enqueues one window; collects the PREVIOUS window's results afterwards (its single host sync), so the host is
        never waiting on the window it has just launched.  Every window's results are read inside the timed region.
        With --inflight N > 1 consecutive windows go to N HIP streams in turn: they are independent clips, and the
        latency-bound phases of one (proposals, read-out) run under the dense phases of another."""
        lane = lanes[turn[0] % len(lanes)]
        turn[0] += 1
        with torch.no_grad(), (torch.cuda.stream(lane) if lane is not None else contextlib.nullcontext()):
            c4 = model(img=frames, img_meta=metas, backbone_feat=True)[0]       # backbone on all T frames
            pend = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True, defer=True)
        if prev is not None:
            prev.result()
        return pend

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pend = None
    for _ in range(args.warmup):
        pend = step(pend)
    if pend is not None:
        pend.result()
    sync()
    native.profile_begin(tags=('relation_full', 'relation_key'))
    t0 = time.perf_counter()
    pend = None
    for _ in range(args.steps):
        pend = step(pend)
    res = pend.result()
    sync()
    elapsed = time.perf_counter() - t0
    rel = native.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the reference's own steady-state loop (tools/test.py:214-250): ONE new frame through the backbone per output
    # frame, the other T-1 C4 maps come from the deque; reported next to the clip-mode headline, never as `value`
    with torch.no_grad():
        c4_all = model(img=frames, img_meta=metas, backbone_feat=True)[0]
        window = [c4_all[i:i + 1] for i in range(T)]
    n_loop = max(3, min(args.steps, 10))
    sync()
    t1 = time.perf_counter()
    for i in range(n_loop):
        with torch.no_grad():
            new = model(img=frames[i % T:i % T + 1], img_meta=[metas[0]], backbone_feat=True)[0]
            window = window[1:] + [new]
            model(x=window, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
    sync()
    ref_loop_fps = n_loop / (time.perf_counter() - t1)

    # the same loop with the per-frame cache: res5 / RPN / RoIAlign / fc_new_1 run once per incoming frame
    # (model.frame_tensors), a window runs the relation stages and the read-out on the T cached entries; the results
    # are bit-identical to the two loops above (tests/test_parity_gpu.py::test_cached_frame_loop_matches_clip_mode)
    with torch.no_grad():
        entries = [model.frame_tensors(c, m) for c, m in zip(window, metas)]
    sync()
    t2 = time.perf_counter()
    pend = None
    for i in range(n_loop):
        with torch.no_grad():
            new = model(img=frames[i % T:i % T + 1], img_meta=[metas[0]], backbone_feat=True)[0]
            window = window[1:] + [new]
            entries = entries[1:] + [model.frame_tensors(new, metas[0])]
            nxt = model.forward_feat_frames(entries, c4s=window, rescale=True, defer=True)
        if pend is not None:
            pend.result()
        pend = nxt
    pend.result()
    sync()
    cached_loop_fps = n_loop / (time.perf_counter() - t2)

    # clip mode again with TWO independent windows in flight on two HIP streams (frame groups off): the latency-bound
    # phases of one window (proposals, read-out) run under the dense phases of the other.  Reported beside the headline:
    # the headline run stays single-lane so that the HIP-event times around the relation core are that kernel's own
    overlap2_fps = None
    if args.inflight == 1 and world == 1:
        lanes[:] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        groups0 = type(model).frame_groups
        type(model).frame_groups = 1
        pend = None
        for _ in range(2):
            pend = step(pend)
        pend.result()
        sync()
        n2 = max(4, min(args.steps, 12))
        t3 = time.perf_counter()
        pend = None
        for _ in range(n2):
            pend = step(pend)
        pend.result()
        sync()
        overlap2_fps = n2 / (time.perf_counter() - t3)
        lanes[:] = [None]
        type(model).frame_groups = groups0

    # per-class breakdown from one extra, fully instrumented window (outside the timed region)
    native.profile_begin(tags=('*',))
    step()
    classes = native.profile_end()
    if args.breakdown and rank == 0:
        native.profile_begin(tags=('*',), detail=True)
        step()
        for tag, d in sorted(native.profile_end().items(), key=lambda kv: -kv[1]['ms']):
            rate = d['work'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
            sys.stderr.write('%-44s calls %3d  %8.3f ms  %8.1f T(FLOP|B)/s\n' % (tag, d['calls'], d['ms'], rate))

    if rank == 0:
        branch = res[-1] if args.head == 'hvr' else res
        n_det = int(sum(len(r) for r in branch))
        full = rel.get('relation_full', dict(calls=0, ms=0.0, work=0.0))
        peak = MFMA_PEAK_TF[args.dtype]
        roofline = None
        traffic = None  # HBM-side bytes per launch come from the committed PMC passes (bench.py cannot collect PMC itself)
        tpath = os.path.join(ROOT, 'profiles', 'r01_relation_traffic.json')
        if os.path.exists(tpath) and T * n_prop == 4500 and args.dtype == 'bf16':
            traffic = json.load(open(tpath))['traffic_bytes_per_launch']
        if full['calls']:
            ach = full['work'] / (full['ms'] * 1e-3) / 1e12
            roofline = dict(kernel='relation core (scores incl. V^T copy + apply: 2 launches), Mq=Mk=%d D=1024' % (T * n_prop), bound='mfma',
                            achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=traffic,
                            launches=full['calls'], avg_ms=round(full['ms'] / full['calls'], 4),
                            flops_per_launch=full['work'] / full['calls'])
        kc = {}
        for tag, d in classes.items():
            e = dict(calls=d['calls'], ms=round(d['ms'], 4))
            if tag in ('gemm', 'conv', 'stem', 'relation_full', 'relation_key') and d['ms'] > 0:
                e['tflops'] = round(d['work'] / (d['ms'] * 1e-3) / 1e12, 2)
                e['frac_mfma_peak'] = round(e['tflops'] / peak, 4)
            elif d['ms'] > 0:
                e['gbs'] = round(d['work'] / (d['ms'] * 1e-3) / 1e9, 1)
                e['frac_hbm_peak'] = round(e['gbs'] / HBM_PEAK_GBS, 4)
            kc[tag] = e
        out = dict(metric='VID frames/sec, R101 Faster-RCNN+%s, 1000x600, %d props, T=%d' % (args.head.upper(), n_prop, T),
                   value=round(world * args.steps / elapsed, 3), unit='frames/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype=args.dtype, data='synthetic',
                   config=dict(workload='configs[2]: faster_rcnn_r101_hrnmp_c5 inference, clip mode' if args.head == 'hvr'
                               else 'configs[1]: faster_rcnn_r101_selsa_c5 inference, clip mode',
                               frames_per_window=T, proposals_per_frame=n_prop, input='3x600x1000 padded to 608x1008',
                               mode='clip (all T frames through backbone+res5+RPN+RoIAlign+head every step)',
                               parallelism='dp%d independent clips, no collectives' % world, windows_in_flight=args.inflight,
                               key_frame_detections=n_det),
                   roofline=roofline, kernel_classes=kc,
                   ref_loop=dict(frames_per_s_per_gpu=round(ref_loop_fps, 2), steps=n_loop,
                                 what='tools/test.py steady state: 1 new backbone frame + res5/RPN/RoIAlign/head on all T per output frame'),
                   cached_loop=dict(frames_per_s_per_gpu=round(cached_loop_fps, 2), steps=n_loop,
                                    what='the same loop with per-frame caching of res5/RPN/RoIAlign/fc_new_1 (identical detections)'))
        if overlap2_fps is not None:
            out['two_in_flight'] = dict(frames_per_s_per_gpu=round(overlap2_fps, 2),
                                        what='clip mode, two independent windows in flight on two HIP streams (--inflight 2)')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.head, T, n_prop, sd)
        if world == 1 and not args.no_train_step:
            ts = train_step_side_measurement(args.head)
            if ts is not None:
                out['train_step'] = ts
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
