"""Clip-mode windows replayed from hipGraphs on L HIP streams in turn (L windows in flight): frames/s per L."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedClip
T, n = 15, 300
dev = torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n), S.synth_state_dict('hvr'), torch.bfloat16, dev)
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for L in [int(x) for x in os.environ.get("HVR_LANES_LIST", "1,2,3,4").split(",")]:
    lanes = [torch.cuda.Stream(device=dev) for _ in range(L)]
    gcs = []
    for s in lanes:
        with torch.cuda.stream(s):
            gcs.append(GraphedClip(model, frames, metas, rescale=True, n_out=1, throughput=os.environ.get("HVR_THR", "1") == "1"))
    torch.cuda.synchronize()
    pend = [None] * L
    def go(i):
        k = i % L
        if pend[k] is not None:
            pend[k].result()
        with torch.cuda.stream(lanes[k]):
            pend[k] = gcs[k].run()
    for i in range(2 * L):
        go(i)
    for p in pend:
        p.result()
    pend = [None] * L
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        go(i)
    for p in pend:
        if p is not None:
            p.result()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('graph lanes %d: %.2f frames/s  %.3f ms/window' % (L, steps / el, el / steps * 1e3), flush=True)
    del gcs
