"""The three convs of ONE layer-3 Bottleneck (resnet.py:220-266) on a 15-frame batch, each launched `--iters` times, for profiler passes
that should see exactly these kernels: conv1 1x1 1024->256, conv2 3x3 256->256, conv3 1x1 256->1024 + residual + ReLU.
    python tools/l3_block.py [--dtype bf16|f16|f16x2] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--frames', type=int, default=15)
args = ap.parse_args()
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}[args.dtype]
B, H, W = args.frames, 38, 63
g = torch.Generator(device='cuda').manual_seed(0)
act = lambda *s: native.cast(torch.randn(s, device='cuda', generator=g), DT)  # noqa: E731
wgt = lambda *s: native.as_operand(torch.randn(s, device='cuda', generator=g) * 0.03, DT)  # noqa: E731
x, w1, w2, w3 = act(B, H, W, 1024), wgt(256, 1, 1, 1024), wgt(256, 3, 3, 256), wgt(1024, 1, 1, 256)
b1, b2, b3 = (torch.randn(n, device='cuda', generator=g) for n in (256, 256, 1024))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = [0.0, 0.0, 0.0]
for it in range(args.iters + 2):
    ev[0].record()
    h1 = native.conv2d_nhwc(x, w1, b1, relu=True)
    ev[1].record()
    h2 = native.conv2d_nhwc(h1, w2, b2, relu=True, pad=1)
    ev[2].record()
    y = native.conv2d_nhwc(h2, w3, b3, resid=x, relu=True)
    ev[3].record()
    torch.cuda.synchronize()
    if it >= 2:
        for k in range(3):
            tot[k] += ev[k].elapsed_time(ev[k + 1])
M = B * H * W
gf = [2.0 * M * 1024 * 256, 2.0 * M * 2304 * 256, 2.0 * M * 256 * 1024]
print('layer-3 block, %s, %d frames (M = %d): ' % (args.dtype, B, M) + ' | '.join(
    '%s %.1f us (%.0f TF/s)' % (n, t / args.iters * 1e3, f / (t / args.iters * 1e-3) / 1e12) for n, t, f in zip(('conv1', 'conv2', 'conv3+res'), tot, gf)))
