"""Full-stage relation core only (Mq = Mk = 4500, D = 1024), for profiler passes that should not mix shapes.

    python tools/rel_bench.py [--iters 10] [--mq 4500] [--mk 4500] [--groups G] [--check]

--groups G > 1: G independent problems through hvr_relation_fwd_grouped (the windows a batched head has in flight); the time printed
is per call and per group.  --check: the grouped result against G single calls and against an f32 torch softmax on the same operands.
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
ap.add_argument('--groups', type=int, default=1)
ap.add_argument('--check', action='store_true')
args = ap.parse_args()
torch.manual_seed(0)
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}[args.dtype]
G = args.groups
q = native.as_operand(torch.randn(G * args.mq, args.d, device='cuda'), DT)
k = native.as_operand(torch.randn(G * args.mk, args.d, device='cuda'), DT)
v = native.as_operand(torch.randn(G * args.mk, args.d, device='cuda'), DT)


def call():
    return native.relation_fwd_grouped(q, k, v, 1 / 32, G, staging=1) if G > 1 else native.relation_fwd(q, k, v, 1 / 32, staging=1)


if args.check:
    o = call().float()
    worst_single = worst_ref = 0.0
    for g in range(G):
        qs, ks, vs = (t[g * n:(g + 1) * n] for t, n in ((q, args.mq), (k, args.mk), (v, args.mk)))
        single = native.relation_fwd(qs, ks, vs, 1 / 32, staging=1).float()
        ref = torch.softmax(qs.float() @ ks.float().t() / 32, dim=1) @ vs.float()
        og = o[g * args.mq:(g + 1) * args.mq]
        worst_single = max(worst_single, float((og - single).abs().max()))
        worst_ref = max(worst_ref, float((og - ref).abs().max()))
        print('group %d: |grouped - single| %.3e  |grouped - f32 torch| %.3e  |single - f32 torch| %.3e  (|ref| max %.3f)'
              % (g, float((og - single).abs().max()), float((og - ref).abs().max()), float((single - ref).abs().max()), float(ref.abs().max())))
    o2 = call().float()
    print('repeatable: %s' % bool((o2 == o).all()))
for _ in range(3):
    call()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    call()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / args.iters
print('relation %s G=%d Mq=%d Mk=%d D=%d  %.4f ms per call  %.4f ms per group  %.1f TF/s' % (args.dtype, G, args.mq, args.mk, args.d, ms, ms / G, 4.0 * G * args.mq * args.mk * args.d / ms / 1e9))
