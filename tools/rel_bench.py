"""Full-stage relation core only (Mq = Mk = 4500, D = 1024), for profiler passes that should not mix shapes.

    python tools/rel_bench.py [--iters 10] [--mq 4500] [--mk 4500]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--mq', type=int, default=4500)
ap.add_argument('--mk', type=int, default=4500)
ap.add_argument('--d', type=int, default=1024)
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f16x2', 'f32'])
args = ap.parse_args()
torch.manual_seed(0)
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}[args.dtype]
q = native.as_operand(torch.randn(args.mq, args.d, device='cuda'), DT)
k = native.as_operand(torch.randn(args.mk, args.d, device='cuda'), DT)
v = native.as_operand(torch.randn(args.mk, args.d, device='cuda'), DT)
for _ in range(3):
    native.relation_fwd(q, k, v, 1 / 32, staging=1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    native.relation_fwd(q, k, v, 1 / 32, staging=1)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / args.iters
print('relation %s Mq=%d Mk=%d D=%d  %.4f ms  %.1f TF/s' % (args.dtype, args.mq, args.mk, args.d, ms, 4.0 * args.mq * args.mk * args.d / ms / 1e9))
