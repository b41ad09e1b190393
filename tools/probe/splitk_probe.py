import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from hvrnet_amd import native
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N, K in ((7182, 256, 2304), (7182, 256, 1024), (7182, 1024, 256), (7182, 512, 4608), (21546, 512, 4608), (7182, 2048, 512), (900, 1024, 1024), (900, 1024, 12544)):
    a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    g = t(lambda: native.gemm(a, w))
    sk = t(lambda: native.gemm_splitk(a, w))
    nb = native.lib().hvr_gemm_splitk_workspace_bytes(M, N, K, 1)
    ideal = 2.0 * M * N * K / 2.5e15 * 1e6
    print('M%d N%d K%d  gemm %.1f us  splitk %.1f us (slices ws %d MB)  ideal %.1f us' % (M, N, K, g, sk, nb >> 20, ideal))
