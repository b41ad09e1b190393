"""debug: packed forward after an SGD step vs a fresh model from the updated state dict"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import selsa_train_config
from hvrnet_amd.dist_train import FlatParams, train_detector_iteration
DEV = 'cuda:0'
n_post, n_sel, T = 24, 16, 3
cfg = selsa_train_config(nms_post=n_post, rcnn_sampler_num=n_sel, t_dim=T)
model = hvrnet_amd.enable_training(hvrnet_amd.build_model(cfg, S.synth_state_dict('selsa'), torch.float32, DEV))
g = torch.Generator().manual_seed(94)
hw = (128, 192)
imgs = (torch.randn((T, 3) + hw, generator=g) * 50.0).to(DEV)
metas = [dict(img_shape=hw + (3,), pad_shape=hw + (3,), scale_factor=1.0, flip=False) for _ in range(T)]
gt_b = torch.tensor([[16., 24., 90., 100.], [100., 30., 170., 110.]]).to(DEV)
gt_l = torch.tensor([5, 12]).to(DEV)
keys = dict(rpn=torch.rand((hw[0] // 16) * (hw[1] // 16) * 12, generator=g).to(DEV), rcnn=[torch.rand(2 + n_post, generator=g).to(DEV) for _ in range(T)])
data = dict(img=imgs, img_meta=metas, return_loss=True, gt_bboxes=[gt_b] * T, gt_labels=[gt_l] * T, keys=keys)
def fwd(m, tag):
    with torch.no_grad():
        c4 = m(img=imgs, img_meta=metas, backbone_feat=True)[0]
        c5 = m.shared_head(c4)
    torch.cuda.synchronize()
    print(tag, 'c4 absmax %.4g mean %.4g  c5 absmax %.4g' % (c4.float().abs().max().item(), c4.float().abs().mean().item(), c5.float().abs().max().item()), flush=True)
    return c4.float().clone(), c5.float().clone()
c40, c50 = fwd(model, 'before')
sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
fresh0 = hvrnet_amd.build_model(cfg, sd0, torch.float32, DEV)
fwd(fresh0, 'fresh0')
flat = FlatParams(model)
print('packed modules tracked:', len(flat._packed_modules))
c4a, c5a = fwd(model, 'after FlatParams')
print('c4 same after flat:', torch.equal(c40, c4a))
log = train_detector_iteration(model, flat, data, lr=float(os.environ.get('LR', '5e-3')), momentum=0.9, weight_decay=1e-4, max_norm=35.0)
print({k: float(v) for k, v in log.items()})
c41, c51 = fwd(model, 'after step')
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
d = max((sd[k].float() - sd0[k].float()).abs().max().item() for k in sd0 if sd0[k].dtype.is_floating_point)
print('max param change', d, 'nonfinite params:', [k for k, v in sd.items() if v.dtype.is_floating_point and not torch.isfinite(v).all()][:5])
fresh = hvrnet_amd.build_model(cfg, sd, torch.float32, DEV)
c4f, c5f = fwd(fresh, 'fresh')
print('c4 diff', (c41 - c4f).abs().max().item(), 'c5 diff', (c51 - c5f).abs().max().item())
for m in model.modules():
    if hasattr(m, '_drop_packed'):
        m._drop_packed()
c42, c52 = fwd(model, 'after dropping ALL packed')
print('c4 diff vs fresh', (c42 - c4f).abs().max().item())
