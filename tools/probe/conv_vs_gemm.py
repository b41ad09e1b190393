"""The same 1x1 product as a conv (implicit-GEMM row decomposition in the loader set-up) and as a plain GEMM: what the set-up costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
g = torch.Generator(device='cuda').manual_seed(0)
for name, H, W, Cin, Cout in (('l3.conv1', 38, 63, 1024, 256), ('l2.conv1', 76, 126, 512, 128), ('l1.conv1', 152, 252, 256, 64), ('res5.conv1', 38, 63, 2048, 512)):
    x = torch.randn((15, H, W, Cin), device='cuda', generator=g).bfloat16()
    w = (torch.randn((Cout, 1, 1, Cin), device='cuda', generator=g) * 0.05).bfloat16()
    b = torch.randn(Cout, device='cuda', generator=g)
    for hint in (0, 11):
        tc = timed(lambda: native.conv2d_nhwc(x, w, b, relu=True, tile=hint))
        tg = timed(lambda: native.gemm(x.view(-1, Cin), w.view(Cout, Cin), b, relu=True, tile=hint))
        print('%-12s hint %2d: conv %.1f us, plain GEMM %.1f us' % (name, hint, tc, tg), flush=True)
