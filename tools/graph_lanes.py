"""Clip-mode windows replayed from hipGraphs: L HIP streams in turn x W clips per graph (their W * T frames go through the
backbone as one batch): frames/s per (L, W).   HVR_LW="1x1,2x1,1x2,2x2"  HVR_THR=1"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedClip
T, n = 15, 300
dev = torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n), S.synth_state_dict('hvr'), torch.bfloat16, dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
thr = os.environ.get('HVR_THR', '1') == '1'
for lw in os.environ.get('HVR_LW', '1x1,2x1,1x2,2x2,1x3,3x1').split(','):
    L, W = [int(x) for x in lw.split('x')]
    frames = torch.cat([S.synth_frame(i) for i in range(T * W)], 0).to(dev)
    metas = [S.synth_meta() for _ in range(T * W)]
    lanes = [torch.cuda.Stream(device=dev) for _ in range(L)]
    gcs = []
    for s in lanes:
        with torch.cuda.stream(s):
            gcs.append(GraphedClip(model, frames, metas, rescale=True, n_out=1, throughput=thr, windows=W))
    torch.cuda.synchronize()
    pend = [None] * L
    def read(p):
        for q in (p if isinstance(p, list) else [p]):
            q.result()
    def go(i):
        k = i % L
        if pend[k] is not None:
            read(pend[k])
        with torch.cuda.stream(lanes[k]):
            pend[k] = gcs[k].run()
    for i in range(2 * L):
        go(i)
    for p in pend:
        read(p)
    pend = [None] * L
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(2, steps // W)
    for i in range(reps):
        go(i)
    for p in pend:
        if p is not None:
            read(p)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('lanes %d x %d clips per graph: %.2f frames/s  %.3f ms/window' % (L, W, reps * W / el, el / (reps * W) * 1e3), flush=True)
    del gcs
