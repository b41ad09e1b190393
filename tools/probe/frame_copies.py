"""Which Python lines issue the device-to-device copies of a stream-mode frame (torch.profiler, CPU + device activities, stacks):
prints, per source line, how many aten::copy_ / clone / contiguous / cat calls it makes per frame."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hvrnet_amd
from hvrnet_amd import native, synthetic as S
from hvrnet_amd.config import hvr_config
T, dev = 15, torch.device('cuda:0')
model = hvrnet_amd.build_model(hvr_config(frame_interval=T // 2, nms_post=300), S.synth_state_dict('hvr'), torch.bfloat16, 'cuda:0')
frame, meta = S.synth_frame(0).to(dev), S.synth_meta()
def one():
    with torch.no_grad(), native.fewrow_split(True):
        c4 = model(img=frame, img_meta=[meta], backbone_feat=True)[0]
        return model.frame_tensors(c4, meta)
for _ in range(3): one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one(); torch.cuda.synchronize()
ev = prof.events()
names = collections.Counter(e.name for e in ev if e.device_type == torch.autograd.DeviceType.CUDA)
print('device events of one frame:', sum(names.values()))
for n, c in names.most_common(12): print('  %4d  %s' % (c, n[:110]))
by_line = collections.Counter()
for e in ev:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::zeros', 'aten::fill_', 'aten::zero_', 'aten::_to_copy', 'aten::empty_like') and e.stack:
        own = [s for s in e.stack if 'hvrnet_amd' in s]
        by_line[(e.name, own[0].strip() if own else e.stack[0].strip())] += 1
print('aten copy-like ops by first hvrnet_amd frame:')
for (n, l), c in by_line.most_common(40): print('  %3d  %-18s %s' % (c, n, l[-120:]))
