"""hvr_nms_first timing in three overlap regimes (n = 6000, cap 300): sparse boxes (the cap is reached in the first chunks),
RPN-like (neighbouring anchors), heavy overlap (few survivors, the sweep walks everything).  HVR_NMS_MASK=1: mask + sweep path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
if os.environ.get("HVR_BENCH_LIB"): native.LIB_PATH = os.path.abspath(os.environ["HVR_BENCH_LIB"])

dev = 'cuda:0'
g = torch.Generator().manual_seed(1)
n = 6000


def boxes(span, size):
    xy = torch.rand((n, 2), generator=g) * span
    wh = torch.rand((n, 2), generator=g) * size + 4
    sc = torch.sort(torch.rand(n, generator=g), descending=True).values[:, None]
    return torch.cat([xy, xy + wh, sc], 1).to(dev)


for name, d in [('sparse', boxes(5000.0, 30.0)), ('rpn-like', boxes(900.0, 200.0)), ('heavy', boxes(60.0, 300.0))]:
    keep = native.nms_first(d, 0.7, 300)
    torch.cuda.synchronize()
    keep_t = torch.empty(n, dtype=torch.long, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = native._workspace(native.lib().hvr_nms_workspace_bytes(n), d.device, 'nms')
    def run():
        native._check(native.lib().hvr_nms_first(native._ptr(d), n, 0.7, 1, 300, native._ptr(keep_t), native._ptr(cnt), native._ptr(ws), ws.numel(), native._stream()), 'x')
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print('%-9s survivors %4d   %.1f us' % (name, keep.numel(), s.elapsed_time(e) / 20 * 1000), flush=True)
    if os.environ.get('HVR_BENCH_LIB'):
        print('   clocks: load %d  sweep %d  compact %d   chunks %d kept %d   diag %d resolve %d suppress %d' % tuple(keep_t[4000:4008].cpu().tolist()))
