"""The 288 x 256 big-tile kernel (csrc/bigtile.hip) on the stride-16 convs, as its own launches: microseconds per call against the
number of tile rounds -- tiles = ceil(M / 288) x N / 256 on 256 CUs, one workgroup per CU -- to separate the per-K-step cost from the
per-tile fixed cost (prologue, epilogue, store drain):   t(call) ~ rounds x (a + ksteps x b).

    python tools/bigtile_probe.py [--frames 15,30,60,120] [--shapes reduce,c3,expand,res5c3,rpn] [--mode bf16]
    HVR_BENCH_LIB=abtest/libhvr_bgclk.so python tools/bigtile_probe.py --clk --frames 60 --shapes reduce,c3     (-DHVR_DBG_BG_CLK build: per-phase stamps)
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
ap.add_argument('--frames', default='15,30,60,120')
ap.add_argument('--shapes', default='reduce,c3,expand,res5c3')
ap.add_argument('--mode', default='bf16')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--clk', action='store_true')
ap.add_argument('--hint', default='force', choices=['force', 'default', 'throughput'])
args = ap.parse_args()
# name: (Cin, Cout, k, dil, residual)
SHAPES = dict(reduce=(1024, 256, 1, 1, False), c3=(256, 256, 3, 1, False), expand=(256, 1024, 1, 1, True), res5c3=(512, 512, 3, 2, False),
              rpn=(1024, 512, 3, 1, False), res5reduce=(2048, 512, 1, 1, False), fc1=(12544, 1024, 1, 1, False), qk=(1024, 2048, 1, 1, False))
# fc1 / qk: the head's linear layers on 300 proposals per frame (M = frames x 300 rows)
H, W = 38, 63
BIG_FORCE = native.BIG_TILE_HINT + 1   # gemm_params.h: kBigForce
hint = dict(force=BIG_FORCE, default=0, throughput=native.BIG_TILE_HINT)[args.hint]
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT}[args.mode]


def operand(t):
    return native.cast(t.cuda(), dt) if dt == native.SPLIT else t.cuda().to(dt)


for name in args.shapes.split(','):
    Cin, Cout, k, dil, res = SHAPES[name]
    for B in [int(b) for b in args.frames.split(',')]:
        g = torch.Generator().manual_seed(1)
        H, W = (1, 300) if name in ('fc1', 'qk') else (38, 63)
        x = operand(torch.randn((B, H, W, Cin), generator=g).relu())
        r = operand(torch.randn((B, H, W, Cout), generator=g).relu()) if res else None
        w = native.as_operand(torch.randn((Cout, k, k, Cin), generator=g).cuda() * 0.05, dt)
        bias = torch.randn(Cout, generator=g).cuda()
        pad = dil * (k - 1) // 2
        f = lambda: native.conv2d_nhwc(x, w, bias, r, relu=True, pad=pad, dil=dil, tile=hint)  # noqa: E731
        M = B * H * W
        tiles = (M + 287) // 288 * (Cout // 256)
        ksteps = k * k * Cin // (32 if dt == native.SPLIT else 64)
        if args.clk:
            f(); torch.cuda.synchronize()
            print('CLK %s B %d tiles %d ksteps %d' % (name, B, tiles, ksteps), flush=True)
            f(); torch.cuda.synchronize()
            continue
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
        gf = 2.0 * M * Cout * k * k * Cin / 1e9
        rounds = tiles / 256.0
        print('%-10s %-5s %3d frames  M %6d  tiles %4d (%.2f rounds of 256)  ksteps %3d  %8.1f us  %6.0f TF/s   %.1f us per whole round'
              % (name, args.mode, B, M, tiles, rounds, ksteps, us, gf / us / 1e3, us / max(1.0, -(-tiles // 256))), flush=True)
