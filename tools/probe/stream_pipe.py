"""Stream mode pieces: graph FC + C alone, graph W alone, sequential push/emit, pipelined push_async/commit/emit."""
import os, sys, time, torch
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
g.emit().result()
N = 30
def timeit(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print('%-34s %.3f ms / frame' % (name, (time.perf_counter() - t0) / N * 1e3), flush=True)
def fc_only():
    for i in range(N):
        g.push_async(frames[i % T:i % T + 1]); g.commit()
def w_only():
    p = None
    for i in range(N):
        q = g.emit()
        if p is not None: p.result()
        p = q
    p.result()
def seq():
    p = None
    for i in range(N):
        g.push(frames[i % T:i % T + 1]); q = g.emit()
        if p is not None: p.result()
        p = q
    p.result()
def pipe():
    p = None
    g.push_async(frames[0:1])
    for i in range(N):
        g.commit(); g.push_async(frames[(i + 1) % T:(i + 1) % T + 1]); q = g.emit()
        if p is not None: p.result()
        p = q
    p.result(); g.commit()
def host_only():
    t0 = time.perf_counter()
    for i in range(N):
        g.commit(); g.push_async(frames[(i + 1) % T:(i + 1) % T + 1]); q = g.emit()
    print('host enqueue per frame (pipelined) %.3f ms' % ((time.perf_counter() - t0) / N * 1e3)); g.commit(); torch.cuda.synchronize()
for rep in range(2):
    timeit('graph FC + C alone', fc_only); timeit('graph W alone', w_only); timeit('sequential push / emit', seq); timeit('pipelined', pipe)
g.push_async(frames[0:1]); host_only()
