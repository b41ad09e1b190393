import sys, torch
sys.path.insert(0, '/root/repo')
from hvrnet_amd import native
torch.manual_seed(0)
dev = 'cuda:0'
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
def check(B, H, W, Cin, Cout, k, stride, pad, dil, res=False):
    x = (torch.randn(B, H, W, Cin, device=dev)).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.03).bfloat16()
    b = torch.randn(Cout, device=dev)
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn(B, OH, OW, Cout, device=dev).bfloat16() if res else None
    f = lambda t: native.conv2d_nhwc(x, w, b, r, relu=True, stride=stride, pad=pad, dil=dil, tile=t)
    a, g = f(11), f(18)
    print('%2d %3d %3d %4d->%4d k%d %s  engine %6.1f us  big144 %6.1f us  big288 %6.1f us' % (B, H, W, Cin, Cout, k, 'equal' if torch.equal(a, g) else 'DIFF %.4f' % (a.float() - g.float()).abs().max().item(),
          timed(lambda: f(11)), timed(lambda: f(18)), timed(lambda: f(17))), flush=True)
check(15, 38, 63, 256, 256, 3, 1, 1, 1)
check(15, 38, 63, 1024, 256, 1, 1, 0, 1)
check(15, 76, 126, 512, 256, 1, 2, 0, 1)
check(13, 37, 61, 256, 256, 3, 1, 1, 1)
check(15, 38, 63, 512, 512, 3, 1, 2, 2)
check(15, 38, 63, 256, 1024, 1, 1, 0, 1, res=True)
check(15, 38, 63, 1024, 512, 3, 1, 1, 1)
check(21, 38, 63, 256, 256, 3, 1, 1, 1)
