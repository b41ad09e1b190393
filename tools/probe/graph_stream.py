"""Replays the stream-mode loop (graph F + graph W per output frame) a few times (profiler target)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedStream
T, n = 15, 300
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).cuda()
meta = S.synth_meta()
g = GraphedStream(model, frames[0:1], meta, rescale=True)
for i in range(T):
    g.push(frames[i:i + 1])
torch.cuda.synchronize()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g.push(frames[i % T:i % T + 1])
    g.emit().result()
torch.cuda.synchronize()
