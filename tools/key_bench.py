"""Key-frame-only relation stage (Mq = 300 queries against Mk = 4 500 keys, D = 1 024; hrnmp_bbox_head.py:269-278,888-891): time per
hvr_relation_fwd call and, with --dump PATH, the output of seeded operands (uint16 view of the bf16 tensor).
--repeat N: the call repeated N times, every output compared with the first."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native
if os.environ.get('HVR_BENCH_LIB'): native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('--mq', type=int, default=300)
ap.add_argument('--mk', type=int, default=4500)
ap.add_argument('--dump')
ap.add_argument('--repeat', type=int, default=0)
ap.add_argument('--iters', type=int, default=50)
a = ap.parse_args()
dev = 'cuda:0'
g = torch.Generator().manual_seed(11)
D = 1024
q = (torch.randn((a.mq, D), generator=g) * 2).to(dev).to(torch.bfloat16)
k = (torch.randn((a.mk, D), generator=g) * 2).to(dev).to(torch.bfloat16)
v = torch.randn((a.mk, D), generator=g).to(dev).to(torch.bfloat16)
o = native.relation_fwd(q, k, v, 1.0 / 32)
torch.cuda.synchronize()
if a.dump:
    np.save(a.dump, o.view(torch.int16).cpu().numpy())
if a.repeat:
    first = o.clone()
    bad = 0
    for _ in range(a.repeat):
        bad += int(not torch.equal(native.relation_fwd(q, k, v, 1.0 / 32), first))
    print('repeat %d: %d differ from the first' % (a.repeat, bad))
for _ in range(5): native.relation_fwd(q, k, v, 1.0 / 32)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters): native.relation_fwd(q, k, v, 1.0 / 32)
e.record(); torch.cuda.synchronize()
print('Mq %d Mk %d: %.1f us per call' % (a.mq, a.mk, s.elapsed_time(e) / a.iters * 1000))
