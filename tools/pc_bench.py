"""Producer / consumer tile kernel (pc_gemm.hip) against the tile engine on the path's plain GEMM shapes: equality of the
results (same MFMA order per output element -> bit-identical) and HIP-event timings.

    python tools/pc_bench.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])

PC128, PC256 = 14, 15   # gemm_params.h: kPcHint128 / kPcHint256

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--only', type=int, default=-1, help='shape index')
args = ap.parse_args()
torch.manual_seed(0)
shapes = [('apply-shaped 4500x1024x4608', 4500, 1024, 4608), ('fc_new_1 4500x1024x12544', 4500, 1024, 12544), ('fc_new_k 4500x1024x1024', 4500, 1024, 1024),
          ('qk proj 4500x2048x1024', 4500, 2048, 1024), ('l3.conv1 35910x256x1024', 35910, 256, 1024), ('l3.conv2 as gemm 35910x256x2304', 35910, 256, 2304),
          ('r5.conv1 35910x512x2048', 35910, 512, 2048), ('rpn as gemm 35910x512x9216', 35910, 512, 9216)]
for si, (name, M, N, K) in enumerate(shapes):
    if args.only >= 0 and si != args.only:
        continue
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda').bfloat16()
    ref = native.gemm(a, w, bias, res, relu=True)
    line = '%-34s' % name
    for tile in (0, PC128, PC256):
        out = native.gemm(a, w, bias, res, relu=True, tile=tile)
        same = torch.equal(out, ref)
        err = (out.float() - ref.float()).abs().max().item()
        for _ in range(3):
            native.gemm(a, w, bias, res, relu=True, tile=tile)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            native.gemm(a, w, bias, res, relu=True, tile=tile)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / args.iters
        line += '  tile %2d: %7.4f ms %7.1f TF/s %s' % (tile, ms, 2.0 * M * N * K / ms / 1e9, 'same' if same else 'DIFF %.3g' % err)
    print(line, flush=True)
