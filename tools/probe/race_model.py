import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
from precision_ladder import apply_mode
T, N, dev = 15, 300, 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=N), S.synth_state_dict('hvr'), None, dev)
apply_mode(model, mode)
fr = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
with torch.no_grad():
    c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
    w = model.window_tensors(c4, metas)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
outs = []
with torch.no_grad():
    for it in range(4):
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(model(img=fr, img_meta=metas, backbone_feat=True)[0])
torch.cuda.synchronize()
print('backbone: %d of %d concurrent C4 maps differ' % (sum(0 if torch.equal(o, c4) else 1 for o in outs), len(outs)))
# stem only
p = model.backbone.packed(fr.device)
ref = native.stem_fused(fr, p['fused'], p['fused_bias'])
torch.cuda.synchronize()
outs = []
for it in range(8):
    for st in streams:
        with torch.cuda.stream(st):
            outs.append(native.stem_fused(fr, p['fused'], p['fused_bias']))
torch.cuda.synchronize()
print('stem: %d of %d differ' % (sum(0 if torch.equal(o, ref) else 1 for o in outs), len(outs)))
# head only from fixed c4
with torch.no_grad():
    ref = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
    outs = []
    for it in range(4):
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True))
torch.cuda.synchronize()
import numpy as np
def same(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y)) for ba, bb in zip(a, b) for x, y in zip(ba, bb))
print('head: %d of %d concurrent windows differ' % (sum(0 if same(o, ref) else 1 for o in outs), len(outs)))
