"""The Bottleneck's closing 1x1 conv + BN shift + identity + ReLU (resnet.py:248-264) as its own launch: microseconds and GB/s of
algorithmic traffic (X, residual, output once, W once) per (stage, frames, mode, hint).

    python tools/expand_bench.py [--frames 15,60] [--modes bf16,f16x2] [--stages l3,l2,l1,res5]
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
ap.add_argument('--modes', default='bf16,f16x2')
ap.add_argument('--stages', default='l3')
ap.add_argument('--iters', type=int, default=30)
args = ap.parse_args()
STAGES = dict(l1=(152, 252, 64, 256), l2=(76, 126, 128, 512), l3=(38, 63, 256, 1024), res5=(38, 63, 512, 2048))
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT}
for st in args.stages.split(','):
    H, W, Cin, Cout = STAGES[st]
    for mode in args.modes.split(','):
        dt = DT[mode]
        for B in [int(b) for b in args.frames.split(',')]:
            g = torch.Generator().manual_seed(1)
            x = native.cast(torch.randn((B, H, W, Cin), generator=g).relu().cuda(), dt) if dt == native.SPLIT else torch.randn((B, H, W, Cin), generator=g).relu().cuda().to(dt)
            r = native.cast(torch.randn((B, H, W, Cout), generator=g).relu().cuda(), dt) if dt == native.SPLIT else torch.randn((B, H, W, Cout), generator=g).relu().cuda().to(dt)
            w = native.as_operand(torch.randn((Cout, 1, 1, Cin), generator=g).cuda() * 0.05, dt)
            bias = torch.randn(Cout, generator=g).cuda()
            es = 4 if dt == native.SPLIT else 2
            nbytes = (B * H * W * (Cin + 2 * Cout) + Cout * Cin) * es
            for name, hint in (('default', 0), ('throughput hint', native.BIG_TILE_HINT)):
                f = lambda: native.conv2d_nhwc(x, w, bias, r, relu=True, tile=hint)  # noqa: E731
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.iters):
                    f()
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / args.iters * 1e3
                print('%-4s %-5s %3d frames  %-15s %8.1f us  %6.0f GB/s  (%.1f us per 15 frames)' % (st, mode, B, name, us, nbytes / us / 1e3, us * 15 / B), flush=True)
