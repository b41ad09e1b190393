"""Clip-mode windows replayed from hipGraphs: L HIP streams in turn x W clips per graph (their W * T frames go through the
backbone as one batch; HVR_BATCH_HEAD=1 (default): res5 / RPN / RoIAlign / head batched too, the relation core in grouped calls): frames/s
per (L, W).   HVR_LW="1x1,2x1,1x2,2x2"  HVR_THR=1  HVR_MODE=bf16|f16|f16x2"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
from hvrnet_amd.graphs import GraphedClip
T, n = 15, 300
dev = torch.device('cuda:0')
from hvrnet_amd import native
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT}[os.environ.get('HVR_MODE', 'bf16')]
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=n), S.synth_state_dict('hvr'), DT, dev)
batch_head = os.environ.get('HVR_BATCH_HEAD', '1') == '1'
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
            gcs.append(GraphedClip(model, frames, metas, rescale=True, n_out=1, throughput=thr and L > 1, windows=W, batch_head=batch_head))
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
    print('lanes %d x %d clips per graph (batch_head %d): %.2f frames/s  %.3f ms/window' % (L, W, batch_head, reps * W / el, el / (reps * W) * 1e3), flush=True)
    del gcs
