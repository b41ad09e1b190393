"""Per-call breakdown of one clip-mode window of configs[2] in one compute mode: every C-ABI call of the window tagged with its
shape (native.profile_begin(detail=True): HIP events around each call on the launch stream), summed per tag, sorted by time.

    python tools/window_breakdown.py --mode f16x2 [--iters 3] [--json out.json]

Eager, one window in flight (the `single_lane` loop of bench.py), so a row is that call's own duration, not a share of an
overlapped interval.  Used to decide which kernel of a mode to work on next; bench.py stays the contract bench.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from precision_ladder import apply_mode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='f16x2')
    ap.add_argument('--head', default='hvr')
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--frames', type=int, default=15)
    ap.add_argument('--json', default=None)
    ap.add_argument('--clips', type=int, default=1, help='W independent clips per call (the batched window: detectors.window_device_outputs(clips=W)); rows are per CALL, i.e. for W windows')
    ap.add_argument('--sequence', action='store_true', help='also print the last window call by call, in issue order')
    ap.add_argument('--side-streams', action='store_true', help='keep the RPN side stream (the product form): rows of calls issued beside it are then intervals of shared chip time, not kernel times')
    args = ap.parse_args()
    T, N, dev = args.frames, 300, 'cuda:0'
    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=T // 2, nms_post=N),
                                   S.synth_state_dict(args.head), None, dev)
    apply_mode(model, args.mode)
    if not args.side_streams:   # RPN branch on the main stream: a conv row is the call's own duration
        model.rpn_side_stream = False
    W = args.clips
    fr = torch.cat([S.synth_frame(i) for i in range(T * W)], 0).to(dev)
    metas = [S.synth_meta() for _ in range(T * W)]

    def window():
        c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
        if W > 1:
            out = model.window_device_outputs(c4, metas, rescale=True, clips=W)
            torch.cuda.synchronize()
            return out
        return model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)

    with torch.no_grad():
        window()  # packs the weights
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            window()
        torch.cuda.synchronize()
        wall = (time.time() - t0) / args.iters * 1e3
        native.profile_begin(('*',), detail=True)
        for _ in range(args.iters):
            window()
        prof = native.profile_end()
        seq = None
        if args.sequence:
            native.profile_begin(('*',), detail=True)
            window()
            seq = native.profile_end(raw=True)
    rows = sorted(((d['ms'] / args.iters, d['calls'] // args.iters, d['work'] / args.iters, tag) for tag, d in prof.items()), reverse=True)
    total = sum(r[0] for r in rows)
    print('mode %s, %d clip(s) per call: %.2f ms per call (wall, un-profiled) = %.2f ms per window; %.2f ms summed over the tagged calls' % (args.mode, W, wall, wall / W, total))
    print('%9s %6s %9s %9s  %s' % ('ms', 'calls', 'us/call', 'work/s', 'call'))
    for ms, calls, work, tag in rows:
        unit = 'GB/s' if tag.startswith(('conv_expand', 'roi_align', 'rpn_proposals')) else 'TF/s'
        rate = work / ms / (1e6 if unit == 'GB/s' else 1e9) if ms > 0 else 0.0
        print('%9.3f %6d %9.1f %7.0f %s  %s' % (ms, calls, ms * 1e3 / max(calls, 1), rate, unit, tag))
    if seq:
        print('\nthe window call by call (us):')
        for tag, work, ms in seq:
            print('%9.1f  %s' % (ms * 1e3, tag))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(dict(mode=args.mode, wall_ms=wall, tagged_ms=total,
                           rows=[dict(tag=t, ms=m, calls=c, work=w) for m, c, w, t in rows]), f, indent=1)


if __name__ == '__main__':
    main()
