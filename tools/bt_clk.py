"""Per-workgroup wall-clock stamps of the persistent scores kernel (debug build -DHVR_DBG_BT_CLK, tools/build_dbg.sh btclk "-DHVR_DBG_BT_CLK";
HVR_BENCH_LIB=abtest/libhvr_btclk.so): V^T, then per tile loop start / loop end / epilogue end, in 100 MHz ticks from the workgroup's start.

    HVR_BENCH_LIB=abtest/libhvr_btclk.so python tools/bt_clk.py [--groups 4] > bt_clk.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
ap = argparse.ArgumentParser()
ap.add_argument('--groups', type=int, default=4)
ap.add_argument('--m', type=int, default=4500)
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16x2'])
args = ap.parse_args()
G = args.groups
torch.manual_seed(0)
q, k, v = (native.as_operand(torch.randn(G * args.m, 1024, device='cuda'), torch.bfloat16 if args.dtype == 'bf16' else native.SPLIT) for _ in range(3))
os.environ.setdefault('HVR_QUIET', '1')
for i in range(2):   # the second call's lines are the warm ones
    print('CALL %d' % i, flush=True)
    native.relation_fwd_grouped(q, k, v, 1 / 32, G, staging=1) if G > 1 else native.relation_fwd(q, k, v, 1 / 32, staging=1)
    torch.cuda.synchronize()
