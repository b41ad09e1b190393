"""Layer 3's identity block tail, fused against separate: relu(h W3^T + b3 + x) and the next block's conv1 relu(y Wn^T + bn) as ONE launch
(hvr_bottleneck_tail_next, expand.hip NX = 16) against the expand conv + the reducing 1x1 as two launches; microseconds per block-call and
GB/s of algorithmic traffic (h, x, y once; hn once; the fused form never reads y back).

    python tools/l3_fused_bench.py [--frames 15,60] [--hint 0|16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
ap = argparse.ArgumentParser()
ap.add_argument('--frames', default='15,60')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16'])
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
H, W, C1, Cout, Cn = 38, 63, 256, 1024, 256


def timed(f, iters):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B in [int(b) for b in args.frames.split(',')]:
    g = torch.Generator().manual_seed(3)
    h = torch.randn((B, H, W, C1), generator=g).relu().to(dt).cuda()
    x = torch.randn((B, H, W, Cout), generator=g).relu().to(dt).cuda()
    w3 = (torch.randn((Cout, C1), generator=g) * 0.05).to(dt).cuda()
    b3 = torch.randn(Cout, generator=g).cuda()
    wn = (torch.randn((Cn, Cout), generator=g) * 0.03).to(dt).cuda()
    bn = torch.randn(Cn, generator=g).cuda()
    M = B * H * W
    assert native.bottleneck_tail_next_supported(h, None, x, w3, b3, 1, wn, bn)
    y, hn = native.bottleneck_tail_next(h, None, x, w3, b3, wn, bn)
    for name, hint in (('default', 0), ('throughput hint', native.BIG_TILE_HINT)):
        def separate():
            yy = native.conv2d_nhwc(h, w3.view(Cout, 1, 1, C1), b3, x, relu=True, tile=hint)
            return yy, native.conv2d_nhwc(yy, wn.view(Cn, 1, 1, Cout), bn, relu=True, tile=hint)
        ys, hs = separate()
        same_y = torch.equal(ys, y)
        dh = float((hs.float() - hn.float()).abs().max())
        t_sep = timed(separate, args.iters)
        t_exp = timed(lambda: native.conv2d_nhwc(h, w3.view(Cout, 1, 1, C1), b3, x, relu=True, tile=hint), args.iters)
        t_fus = timed(lambda: native.bottleneck_tail_next(h, None, x, w3, b3, wn, bn), args.iters)
        by_sep = (M * (C1 + 2 * Cout) + M * (Cout + Cn)) * 2
        by_fus = (M * (C1 + 2 * Cout) + M * Cn) * 2
        print('%3d frames  %-15s separate %7.1f us (expand %6.1f + reduce %6.1f; %4.0f GB/s)   fused %7.1f us (%4.0f GB/s, %4.0f TF/s)   y identical: %s, max |dhn| %.3g'
              % (B, name, t_sep, t_exp, t_sep - t_exp, by_sep / t_sep / 1e3, t_fus, by_fus / t_fus / 1e3, 4.0 * M * Cout * C1 / t_fus / 1e6, same_y, dh), flush=True)
