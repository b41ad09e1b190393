"""Stream mode's per-frame work call by call (what GraphedStream captures as the frame graph: one new frame through backbone / res5 /
RPN / RoIAlign / fc_new_1 with the few-row split-K forms): every C-ABI call tagged with its shape, HIP events around each
(native.profile_begin(detail=True)), summed per tag.  Eager, so a row is that call's own duration including its reduce launch.

    python tools/frame_breakdown.py [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
T, dev = 15, torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frame = S.synth_frame(0).to(dev)
meta = S.synth_meta()


def one():
    with torch.no_grad(), native.fewrow_split(True):
        c4 = model(img=frame, img_meta=[meta], backbone_feat=True)[0]
        return model.frame_tensors(c4, meta)


for _ in range(3):
    one()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    one()
e.record()
torch.cuda.synchronize()
wall = s.elapsed_time(e) / args.iters
native.profile_begin(('*',), detail=True)
for _ in range(args.iters):
    one()
prof = native.profile_end()
rows = sorted(((d['ms'] / args.iters, d['calls'] // args.iters, d['work'] / args.iters, tag) for tag, d in prof.items()), reverse=True)
print('one frame, eager: %.3f ms wall (un-profiled); %.3f ms summed over the tagged calls' % (wall, sum(r[0] for r in rows)))
print('%9s %6s %9s %9s  %s' % ('ms', 'calls', 'us/call', 'work/s', 'call'))
for ms, calls, work, tag in rows:
    unit = 'GB/s' if tag.startswith(('conv_expand', 'roi_align', 'rpn_proposals')) else 'TF/s'
    rate = work / ms / (1e6 if unit == 'GB/s' else 1e9) if ms > 0 else 0.0
    print('%9.3f %6d %9.1f %7.0f %s  %s' % (ms, calls, ms * 1e3 / max(calls, 1), rate, unit, tag))
