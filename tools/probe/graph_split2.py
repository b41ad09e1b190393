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
mode, thr, sync = sys.argv[1], sys.argv[2] == '1', sys.argv[3] == '1'
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=N), S.synth_state_dict('hvr'), None, dev)
apply_mode(model, mode)
fr = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
gcs = []
for st in lanes:
    with torch.cuda.stream(st):
        gcs.append(GraphedClip(model, fr, metas, rescale=True, n_out=1, throughput=thr))
torch.cuda.synchronize()
out = []
pend = [None, None]
for i in range(8):
    k = i % 2
    if pend[k] is not None:
        r = pend[k].result()
        out.append([sum(len(x) for x in b) for b in r])
    with torch.cuda.stream(lanes[k]):
        pend[k] = gcs[k].run()
    if sync:
        torch.cuda.synchronize()
for k in range(2):
    out.append([sum(len(x) for x in b) for b in pend[k].result()])
print(mode, 'throughput', thr, 'serialised', sync, out)
