"""Relation core forward + HIP backward at window size (Mq = Mk = 4500, D = 1024, bf16): HIP-event times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import ops  # noqa: E402

M, D = 4500, 1024
torch.manual_seed(0)
q, k, v = [(torch.randn(M, D, device='cuda') * 1.2).bfloat16().requires_grad_(True) for _ in range(3)]
go = torch.randn(M, D, device='cuda').bfloat16()


def run():
    o = ops.relation(q, k, v, 1 / 32)
    o.backward(go)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print('relation fwd + bwd  %.3f ms  (%.0f TF/s on 3 x 4 M^2 D flops)' % (ms, 12.0 * M * M * D / ms / 1e9))
