import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
if os.environ.get("HVR_BENCH_LIB"): native.LIB_PATH = os.path.abspath(os.environ["HVR_BENCH_LIB"])
from hvrnet_amd.box_ops import AnchorGenerator
T, A, H, W = 15, 12, 38, 63
g = torch.Generator().manual_seed(3)
cls = (torch.randn((T, H, W, A), generator=g) * 1.5).cuda()
reg = (torch.randn((T, H, W, 4 * A), generator=g) * 0.3).cuda()
gen = AnchorGenerator(16, [4, 8, 16, 32], [0.5, 1.0, 2.0])
for _ in range(2):
    native.rpn_proposals(cls, reg, gen.base_anchors, 16, (0., 0., 0., 0.), (1., 1., 1., 1.), (600, 1000), 6000, 300, 300, 0.7)
torch.cuda.synchronize()
