import sys, torch
sys.path.insert(0, '/root/repo')
from hvrnet_amd import native
torch.manual_seed(0)
dev = 'cuda:0'
def check(B, H, W, Cin, Cout, relu=True):
    x = (torch.randn(B, H, W, Cin, device=dev)).bfloat16()
    w = (torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05).bfloat16()
    b = torch.randn(Cout, device=dev)
    r = torch.randn(B, H, W, Cout, device=dev).bfloat16()
    a = native.conv2d_nhwc(x, w, b, r, relu=relu, tile=11)
    g = native.conv2d_nhwc(x, w, b, r, relu=relu, tile=17)
    print(B, H, W, Cin, Cout, 'equal' if torch.equal(a, g) else 'DIFF max %.4f' % (a.float() - g.float()).abs().max().item(), flush=True)
check(15, 38, 63, 256, 1024)
check(15, 38, 63, 512, 2048)
check(13, 37, 61, 256, 1024, relu=False)
check(2, 19, 32, 128, 512)
