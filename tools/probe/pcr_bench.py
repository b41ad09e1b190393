"""pcr_gemm.hip (tile hint 18) against the tile engine / the automatic choice on layer 3's N = 256 convs: equality and HIP-event timings.
    python tools/probe/pcr_bench.py [--dtype bf16|f16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--frames', type=int, default=15)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}[args.dtype]
B = args.frames
PCR = 18
SHAPES = [('l3.conv1 1024->256', 38, 63, 1024, 256, 1, 0, 1), ('l3.conv2 3x3 256', 38, 63, 256, 256, 3, 1, 1),
          ('res5.conv1 2048->512', 38, 63, 2048, 512, 1, 0, 1), ('res5.conv2 3x3 512 d2', 38, 63, 512, 512, 3, 2, 2),
          ('res5.ext 2048->256', 38, 63, 2048, 256, 1, 0, 1), ('rpn 3x3 1024->512', 38, 63, 1024, 512, 3, 1, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
for name, H, W, Cin, Cout, k, pad, dil in SHAPES:
    x = native.as_operand(torch.randn((B, H, W, Cin), device='cuda', generator=g), DT)
    w = native.as_operand(torch.randn((Cout, k, k, Cin), device='cuda', generator=g) * 0.05, DT)
    bias = torch.randn(Cout, device='cuda', generator=g)
    line = '%-24s' % name
    ref = None
    for hint in (0, 11, PCR):
        try:
            y = native.conv2d_nhwc(x, w, bias, None, relu=True, pad=pad, dil=dil, tile=hint)
            ref = y if ref is None else ref
            for _ in range(3):
                native.conv2d_nhwc(x, w, bias, None, relu=True, pad=pad, dil=dil, tile=hint)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                native.conv2d_nhwc(x, w, bias, None, relu=True, pad=pad, dil=dil, tile=hint)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / args.iters * 1e3
            line += '  hint %2d: %7.1f us %6.0f TF/s %s' % (hint, us, 2.0 * B * H * W * Cout * k * k * Cin / us / 1e6, 'same' if torch.equal(y, ref) else 'DIFF')
        except native.HvrError as exc:
            line += '  hint %2d: %s' % (hint, str(exc)[:40])
    print(line, flush=True)
