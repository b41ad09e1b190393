"""How long the host needs to ENQUEUE one clip-mode window (no waiting on the GPU): if this approaches the window's GPU
time the loop is host-bound and launch batching (HIP graphs) would be the lever; if it is far below, it is not."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import hvrnet_amd
from hvrnet_amd import synthetic as S
from hvrnet_amd.config import hvr_config
dev = torch.device('cuda', 0)
T = 15
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, dev)
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
metas = [S.synth_meta() for _ in range(T)]
def enqueue():
    with torch.no_grad():
        c4 = model(img=frames, img_meta=metas, backbone_feat=True)[0]
        return model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True, defer=True)
for _ in range(3):
    enqueue().result()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); p = enqueue(); t1 = time.perf_counter()
    p.result(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
ts.sort()
print('host enqueue of one window: median %.2f ms (min %.2f); enqueue + drain: median %.2f ms' % (ts[5][0] * 1e3, ts[0][0] * 1e3, sorted(t[1] for t in ts)[5] * 1e3))
