"""RPN proposal path on the bench model's own RPN outputs: time per call for T = 15 and T = 1, boxes walked by the NMS until the
nms_post-th survivor.  HVR_BENCH_LIB=path picks another build of the library."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
if os.environ.get("HVR_BENCH_LIB"): native.LIB_PATH = os.path.abspath(os.environ["HVR_BENCH_LIB"])

T, dev = 15, torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
meta = S.synth_meta()
cap = {}
orig = native.rpn_proposals
def spy(*a, **k):
    cap['a'], cap['k'] = a, k
    return orig(*a, **k)
native.rpn_proposals = spy
import hvrnet_amd.rpn_head as RH
RH.native.rpn_proposals = spy
with torch.no_grad():
    model.simple_test(frames, [meta] * T)
torch.cuda.synchronize()
native.rpn_proposals = orig
a, k = cap['a'], cap['k']
cls, reg = a[0], a[1]
print('cls', tuple(cls.shape), cls.dtype, 'logit range', float(cls.min()), float(cls.max()), 'std', float(cls.float().std()))

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000

for t in (15, 1):
    c, r = cls[:t], reg[:t]
    us = timed(lambda: orig(c, r, *a[2:], **k))
    props, counts = orig(c, r, *a[2:], **k)
    torch.cuda.synchronize()
    H, W, A = cls.shape[1:]
    npre = min(int(a[7]), H * W * A)
    ws = native._workspace(1, dev, 'rpn')
    walked = None
    if ws is not None:
        al = lambda x: (x + 255) // 256 * 256
        off = al(t * npre * 5 * 4)
        keep = ws[off:off + t * npre * 8].view(torch.int64).view(t, npre)
        nk = ws[off + al(t * npre * 8):off + al(t * npre * 8) + 4 * t].view(torch.int32)
        walked = [int(keep[f, int(nk[f]) - 1]) for f in range(t)]
    print('T=%d  %.1f us per call  counts %s  walked %s' % (t, us, counts.tolist()[:4], walked), flush=True)
    print('  checksum', float(props.double().sum()), flush=True)
