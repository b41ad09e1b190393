import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from hvrnet_amd import native, backbone as B
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for (ci, co, k) in ((64, 64, 3), (256, 1024, 1), (512, 512, 3)):
        conv = nn.Conv2d(ci, co, k, bias=False).cuda()
        bn = nn.BatchNorm2d(co).cuda().eval()
        with torch.no_grad():
            conv.weight.mul_(0.05); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.01, 2.0)
        w_new, b_new = B.fold_conv_bn(conv, bn, dt)
        w = conv.weight.detach().float()
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w_old = native.as_operand((w * scale[:, None, None, None]).permute(0, 2, 3, 1), dt)
        print(dt, ci, co, k, 'weights equal:', bool(torch.equal(w_new, w_old)), 'bias equal:', bool(torch.equal(b_new, bias)),
              'max diff', float((w_new.float() - w_old.float()).abs().max()))
