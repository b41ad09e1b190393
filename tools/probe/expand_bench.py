"""Expand convs (expand.hip) alone, 15 frames: l1 / l2 / l3 / res5 shapes.  HVR_BENCH_LIB A/Bs a debug build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
T = 15
dev = 'cuda:0'
only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, H, W, Cin, Cout in [('l1', 152, 252, 64, 256), ('l2', 76, 126, 128, 512), ('l3', 38, 63, 256, 1024), ('r5', 38, 63, 512, 2048)]:
    if only and only != name:
        continue
    x = (torch.randn(T, H, W, Cin, device=dev)).bfloat16()
    w = (torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05).bfloat16()
    b = torch.zeros(Cout, device=dev)
    res = torch.randn(T, H, W, Cout, device=dev).bfloat16()
    out = torch.empty(T, H, W, Cout, device=dev, dtype=torch.bfloat16)
    fn = lambda: native.conv2d_nhwc(x, w, b, res, relu=True, staging=1, out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    by = (x.numel() + 2 * res.numel() + w.numel()) * 2
    print('expand %s %dx%d %d->%d  %.1f us  %.2f TB/s' % (name, H, W, Cin, Cout, ms * 1e3, by / ms / 1e9), flush=True)
