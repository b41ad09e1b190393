import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from hvrnet_amd import native
torch.manual_seed(0)
dev = 'cuda:0'
def check(B, H, W, Cin, Cout, k, stride, pad, dil):
    x = (torch.randn(B, H, W, Cin, device=dev)).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.03).bfloat16()
    b = torch.randn(Cout, device=dev)
    a = native.conv2d_nhwc(x, w, b, None, relu=True, stride=stride, pad=pad, dil=dil)
    g = native.conv2d_nhwc(x, w, b, None, relu=True, stride=stride, pad=pad, dil=dil, tile=16)
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=stride, padding=pad, dilation=dil)).permute(0, 2, 3, 1)
    print(B, H, W, Cin, Cout, k, 'equal' if torch.equal(a, g) else 'DIFF max %.4f' % (a.float() - g.float()).abs().max().item(),
          'vs f32 ref max err %.4f' % (g.float() - ref).abs().max().item(), flush=True)
check(15, 38, 63, 256, 256, 3, 1, 1, 1)
check(15, 38, 63, 1024, 256, 1, 1, 0, 1)
check(15, 38, 63, 512, 512, 3, 1, 2, 2)
check(15, 38, 63, 1024, 512, 3, 1, 1, 1)
check(15, 38, 63, 2048, 512, 1, 1, 0, 1)
check(7, 37, 61, 256, 256, 3, 1, 1, 1)
check(15, 76, 126, 512, 256, 1, 2, 0, 1)
