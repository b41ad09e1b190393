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
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16x2'])
ap.add_argument('--groups', type=int, default=0, help='> 1: hvr_relation_fwd_grouped over that many clips (one launch per pass over all of them) against as many single calls')
a = ap.parse_args()
dev = 'cuda:0'
g = torch.Generator().manual_seed(11)
D = 1024
op = (lambda t: native.cast(t.to(dev), native.SPLIT)) if a.dtype == 'f16x2' else (lambda t: t.to(dev).to(torch.bfloat16))
q = op(torch.randn((a.mq, D), generator=g) * 2)
k = op(torch.randn((a.mk, D), generator=g) * 2)
v = op(torch.randn((a.mk, D), generator=g))
o = native.relation_fwd(q, k, v, 1.0 / 32)
torch.cuda.synchronize()
if a.dump:
    np.save(a.dump, o.view(torch.int16).cpu().numpy() if a.dtype == 'bf16' else native.cast(o, torch.float32).cpu().numpy())
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

if a.groups > 1:
    G = a.groups
    qg, kg, vg = op(torch.randn((G * a.mq, D), generator=g) * 2), op(torch.randn((G * a.mk, D), generator=g) * 2), op(torch.randn((G * a.mk, D), generator=g))
    og = native.relation_fwd_grouped(qg, kg, vg, 1.0 / 32, G)
    eq = (lambda x, y: torch.equal(native.cast(x, torch.float32), native.cast(y, torch.float32))) if a.dtype == 'f16x2' else torch.equal
    same = all(eq(og[i * a.mq:(i + 1) * a.mq], native.relation_fwd(qg[i * a.mq:(i + 1) * a.mq], kg[i * a.mk:(i + 1) * a.mk], vg[i * a.mk:(i + 1) * a.mk], 1.0 / 32)) for i in range(G))
    for name, f in (('grouped call', lambda: native.relation_fwd_grouped(qg, kg, vg, 1.0 / 32, G)),
                    ('%d single calls' % G, lambda: [native.relation_fwd(qg[i * a.mq:(i + 1) * a.mq], kg[i * a.mk:(i + 1) * a.mk], vg[i * a.mk:(i + 1) * a.mk], 1.0 / 32) for i in range(G)])):
        for _ in range(5): f()
        torch.cuda.synchronize()
        s.record()
        for _ in range(a.iters): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / a.iters * 1000
        print('G %d x (Mq %d, Mk %d), %s: %.1f us = %.1f us per clip (%.0f TF/s); equal to the single calls bit for bit: %s' % (G, a.mq, a.mk, name, us, us / G, 4.0 * G * a.mq * a.mk * D / us / 1e6, same))
