"""Replays the clip-mode window from its hipGraph a few times (profiler target: rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedClip
T, n = 15, 300
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).cuda()
metas = [S.synth_meta() for _ in range(T)]
g = GraphedClip(model, frames, metas, rescale=True)
torch.cuda.synchronize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g.run().result()
torch.cuda.synchronize()
