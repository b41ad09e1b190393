"""Stream mode's per-frame work, eagerly (what GraphedStream captures as the frame graph: one new frame through backbone / res5 /
RPN / RoIAlign / fc_new_1 with the few-row split-K forms), for a rocprofv3 --kernel-trace --stats run: which kernels own a frame."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
T, dev = 15, torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frame = S.synth_frame(0).to(dev)
meta = S.synth_meta()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def one():
    with torch.no_grad(), native.fewrow_split(True):
        c4 = model(img=frame, img_meta=[meta], backbone_feat=True)[0]
        return model.frame_tensors(c4, meta)
for _ in range(3): one()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n): one()
e.record(); torch.cuda.synchronize()
print('one frame, eager: %.3f ms' % (s.elapsed_time(e) / n))
