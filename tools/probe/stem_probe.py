import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from hvrnet_amd import native
if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
wpk = (torch.randn(64, 7, 32, device='cuda') * 0.05).bfloat16(); bias = torch.randn(64, device='cuda')
ref = None
for B in (15, 8, 7):
    img = torch.randn(B, 3, 608, 1008, device='cuda') * 50
    out = native.stem_fused(img, wpk, bias)
    print('B=%d  %.1f us  checksum %.6e' % (B, t(lambda: native.stem_fused(img, wpk, bias)), float(out.float().sum())))
