"""Where the launches of one training iteration come from (~1 460 / 1 700 dispatches per iteration).

    python tools/train_census.py [--head selsa|hvr] [--top N]

Two censuses of ONE iteration after warm-up:
  * C-ABI calls (`hvr_*` through ctypes): a proxy around the loaded library counts every call by (symbol, nearest caller line under
    hvrnet_amd/ that is not native.py);
  * torch's own device work: torch.profiler (CPU side, with_stack) -- every aten op that launched at least one kernel or memcpy, by
    (op, nearest hvrnet_amd / tools source line).
Prints both tables to stdout.  Measurement tooling only (not on the product path)."""
import argparse
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_train_config, selsa_train_config  # noqa: E402
from hvrnet_amd.dist_train import FlatParams, train_detector_iteration  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site(frames, skip=('native.py',)):
    """nearest frame under the repo that is not in `skip`: 'file:line func'"""
    for fr in reversed(frames):
        fn = fr.filename
        if fn.startswith(ROOT) and os.path.basename(fn) not in skip and 'train_census' not in fn:
            return '%s:%d %s' % (os.path.relpath(fn, ROOT), fr.lineno, fr.name)
    return '?'


class LibProxy(object):
    def __init__(self, handle, counts):
        self._h, self._c, self._w = handle, counts, {}

    def __getattr__(self, name):
        w = self._w.get(name)
        if w is None:
            fn = getattr(self._h, name)
            counts = self._c

            def w(*a, _fn=fn, _name=name):
                if counts['on']:
                    counts['n'][(_name, site(traceback.extract_stack(limit=14)[:-1]))] += 1
                return _fn(*a)
            self._w[name] = w
        return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--head', choices=['selsa', 'hvr'], default='hvr')
    ap.add_argument('--top', type=int, default=70)
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    hw, pad = (600, 1000), (608, 1008)
    gt_b = torch.tensor([[120., 80., 420., 330.], [296., 136., 359., 199.], [500., 100., 780., 300.], [820., 420., 865., 460.]]).to(dev)
    gt_l = torch.tensor([3, 17, 9, 22]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    if args.head == 'selsa':
        T = 3
        cfg = selsa_train_config(nms_post=300, rcnn_sampler_num=128, t_dim=T)
    else:
        T = 15
        cfg = hvr_train_config(nms_post=300, rcnn_sampler_num=128)
    model = hvrnet_amd.enable_training(hvrnet_amd.build_model(cfg, S.synth_state_dict(args.head), torch.bfloat16, dev))
    flat = FlatParams(model)
    imgs = torch.cat([S.synth_frame(i, img_hw=hw, pad_hw=pad) for i in range(T)], 0).to(dev)
    metas = [S.synth_meta(hw, pad) for _ in range(T)]
    data = dict(img=imgs, img_meta=metas, return_loss=True, gt_bboxes=[gt_b] * T, gt_labels=[gt_l] * T, generator=gen)

    def step():
        return train_detector_iteration(model, flat, data, lr=1e-4, momentum=0.9, weight_decay=1e-4, max_norm=35.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()

    counts = dict(on=False, n=collections.Counter())
    native._lib = LibProxy(native.lib(), counts)
    from torch.profiler import profile, ProfilerActivity
    from torch.utils._python_dispatch import TorchDispatchMode
    VIEW = ('view', 'slice', 'select', 'permute', 'transpose', 'expand', 'as_strided', 'detach', 'alias', 'unsqueeze', 'squeeze', 'reshape',
            'split', 'narrow', 't.default', 'unbind', 'size', 'stride', 'numel', 'is_', 'empty', '_unsafe_view', 'lift_fresh', 'sym_')
    disp = collections.Counter()

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(v in name for v in VIEW):
                disp[(name, site(traceback.extract_stack(limit=16)[:-1]))] += 1
            return func(*args, **(kwargs or {}))

    counts['on'] = True
    with torch.autograd.set_multithreading_enabled(False):     # backward on this thread: the mode (thread-local) sees its ops too
        with Census():
            step()
            torch.cuda.synchronize()
    counts['on'] = False
    print('# %s training iteration: aten ops seen by a TorchDispatchMode (views left out), by (op, nearest repo frame); total %d' % (args.head, sum(disp.values())))
    for (name, where), n in disp.most_common(args.top + 40):
        print('%5d  %-36s %s' % (n, name[:36], where))
    print()
    counts['n'].clear()
    counts['on'] = True
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    counts['on'] = False

    print('# %s training iteration: C-ABI calls by (symbol, caller)' % args.head)
    tot = sum(counts['n'].values())
    print('# total %d' % tot)
    for (name, where), n in counts['n'].most_common(args.top):
        print('%5d  %-34s %s' % (n, name, where))

    # aten ops that launched device work
    ops = collections.Counter()
    kern = collections.Counter()
    for e in prof.events():
        if str(e.device_type).endswith('CPU') and e.kernels:
            where = '?'
            for fr in (e.stack or []):
                if ROOT in fr and 'train_census' not in fr:
                    where = fr.replace(ROOT + '/', '')
                    break
            ops[(e.name, where)] += len(e.kernels)
            kern[e.name] += len(e.kernels)
    print('\n# torch ops with device work: launches by (op, nearest repo frame); total %d' % sum(ops.values()))
    for (name, where), n in ops.most_common(args.top):
        print('%5d  %-40s %s' % (n, name[:40], where))
    print('\n# by op')
    for name, n in kern.most_common(30):
        print('%5d  %s' % (n, name))


if __name__ == '__main__':
    main()
