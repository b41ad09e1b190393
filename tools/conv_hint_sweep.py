"""Tile-shape sweep of the window's convs in one operand format: does the cost model (fitted on bf16) pick well for half / split half?
    python tools/conv_hint_sweep.py --dtype f16x2"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='f16x2')
ap.add_argument('--frames', type=int, default=15)
ap.add_argument('--hints', default=','.join(str(h) for h in range(1, 13)))
ap.add_argument('--cold', type=int, default=0, help='MB written by a fill between the timed calls (0: back-to-back calls on the same operands)')
args = ap.parse_args()
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}[args.dtype]
B = args.frames
SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad, dil, resid
    ('l1.conv1 256->64', 152, 252, 256, 64, 1, 1, 0, 1, False), ('l1.conv2 3x3 64', 152, 252, 64, 64, 3, 1, 1, 1, False),
    ('l1.conv3 64->256 +res', 152, 252, 64, 256, 1, 1, 0, 1, True), ('l2.conv1 256->128 s2', 152, 252, 256, 128, 1, 2, 0, 1, False),
    ('l2.conv2 3x3 128', 76, 126, 128, 128, 3, 1, 1, 1, False), ('l2.conv3 128->512 +res', 76, 126, 128, 512, 1, 1, 0, 1, True),
    ('l3.conv1 1024->256', 38, 63, 1024, 256, 1, 1, 0, 1, False), ('l3.conv2 3x3 256', 38, 63, 256, 256, 3, 1, 1, 1, False),
    ('l3.conv3 256->1024 +res', 38, 63, 256, 1024, 1, 1, 0, 1, True), ('res5.conv2 3x3 512 d2', 38, 63, 512, 512, 3, 1, 2, 2, False),
    ('res5.conv3 512->2048 +res', 38, 63, 512, 2048, 1, 1, 0, 1, True), ('rpn 3x3 1024->512', 38, 63, 1024, 512, 3, 1, 1, 1, False),
    ('res5.conv1 2048->512', 38, 63, 2048, 512, 1, 1, 0, 1, False), ('res5.ds 1024->2048', 38, 63, 1024, 2048, 1, 1, 0, 1, False),
    ('res5.ext 2048->256', 38, 63, 2048, 256, 1, 1, 0, 1, False)]
g = torch.Generator(device='cuda').manual_seed(0)
FLUSH = torch.empty(args.cold << 18, device='cuda') if args.cold else None
for name, H, W, Cin, Cout, k, st, pad, dil, res in SHAPES:
    x = native.as_operand(torch.randn((B, H, W, Cin), device='cuda', generator=g), DT)
    w = native.as_operand(torch.randn((Cout, k, k, Cin), device='cuda', generator=g) * 0.05, DT)
    bias = torch.randn(Cout, device='cuda', generator=g)
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    r = native.as_operand(torch.randn((B, OH, OW, Cout), device='cuda', generator=g), DT) if res else None
    out = []
    for hint in [0] + [int(h) for h in args.hints.split(',') if h]:
        try:
            for _ in range(2):
                native.conv2d_nhwc(x, w, bias, r, relu=True, stride=st, pad=pad, dil=dil, tile=hint)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.cold:   # every timed call after a fill that evicts the operands from the L2 and the Infinity Cache
                tot = 0.0
                for _ in range(5):
                    FLUSH.fill_(1.0)
                    s.record()
                    native.conv2d_nhwc(x, w, bias, r, relu=True, stride=st, pad=pad, dil=dil, tile=hint)
                    e.record()
                    torch.cuda.synchronize()
                    tot += s.elapsed_time(e)
                out.append((hint, tot / 5 * 1e3))
                continue
            s.record()
            for _ in range(5):
                native.conv2d_nhwc(x, w, bias, r, relu=True, stride=st, pad=pad, dil=dil, tile=hint)
            e.record()
            torch.cuda.synchronize()
            out.append((hint, s.elapsed_time(e) / 5 * 1e3))
        except native.HvrError:
            out.append((hint, float('nan')))
    best = min([(t, h) for h, t in out[1:] if t == t] or [(out[0][1], 0)])
    print('%-26s auto %7.1f us | best hint %2d %7.1f us | %s' % (name, out[0][1], best[1], best[0], ' '.join('%d:%.0f' % (h, t) for h, t in out[1:])))
