"""Parity of hvr_relation_fwd against a torch f32 reference at window size (tuning probe)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
for Mq, Mk in [(4500, 4500), (4500, 4400), (3000, 4500), (4400, 4477)]:
    torch.manual_seed(Mq + Mk)
    q = (torch.randn(Mq, 1024, device='cuda') * 1.5).bfloat16()
    k = (torch.randn(Mk, 1024, device='cuda') * 1.5).bfloat16()
    v = torch.randn(Mk, 1024, device='cuda').bfloat16()
    k[Mk - 3] = (q[5].float() * 3).bfloat16()
    ref = torch.softmax((q.float() @ k.float().t()) / 32, dim=1) @ v.float()
    out = native.relation_fwd(q, k, v, 1 / 32, staging=1).float()
    err = (out - ref).abs()
    print('Mq=%d Mk=%d max abs err %.4g  mean %.4g  ref absmax %.3g finite %s' % (Mq, Mk, err.max().item(), err.mean().item(), ref.abs().max().item(), torch.isfinite(out).all().item()))
