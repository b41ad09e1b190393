"""Practical streaming ceiling on this box for the 'expand + residual' conv shapes: torch elementwise kernels on the same tensors."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N in ((35910, 1024), (35910, 2048), (143640, 512), (574560, 256)):
    a = torch.randn(M, N, device='cuda').bfloat16(); b = torch.randn(M, N, device='cuda').bfloat16(); o = torch.empty_like(a)
    us_add = t(lambda: torch.add(a, b, out=o))
    us_cp = t(lambda: o.copy_(a))
    nb = M * N * 2
    print('M%d N%d  add %.1f us (%.2f TB/s)   copy %.1f us (%.2f TB/s)' % (M, N, us_add, 3 * nb / us_add / 1e6, us_cp, 2 * nb / us_cp / 1e6))
