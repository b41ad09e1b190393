import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedClip
from precision_ladder import apply_mode
T, N, dev = 15, 300, 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=N), S.synth_state_dict('hvr'), None, dev)
apply_mode(model, mode)
fr = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
with torch.no_grad():
    c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
    want = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
print('eager detections', [sum(len(r) for r in b) for b in want])
g = GraphedClip(model, fr, metas, rescale=True, n_out=1, throughput=False)
for i in range(3):
    p = g.run()
    got = p.result()
    print('replay', i, 'respeculated', p.respeculated, 'detections', [sum(len(r) for r in b) for b in got])
lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
gcs = []
for st in lanes:
    with torch.cuda.stream(st):
        gcs.append(GraphedClip(model, fr, metas, rescale=True, n_out=1, throughput=True))
pend = [None, None]
for i in range(6):
    k = i % 2
    if pend[k] is not None:
        r = pend[k].result()
        print('lane', k, 'respeculated', pend[k].respeculated, 'detections', [sum(len(x) for x in b) for b in r])
    with torch.cuda.stream(lanes[k]):
        pend[k] = gcs[k].run()
for k in range(2):
    r = pend[k].result()
    print('drain lane', k, 'detections', [sum(len(x) for x in b) for b in r])
