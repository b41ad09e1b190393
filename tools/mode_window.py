"""K clip-mode windows of configs[2] in one compute mode (tools/precision_ladder.py's mode names), for profiling:

    rocprofv3 --kernel-trace --stats -d /tmp/p -o w -- python tools/mode_window.py --mode f16x2 --iters 3
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from precision_ladder import apply_mode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='f16x2')
    ap.add_argument('--head', default='hvr')
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--frames', type=int, default=15)
    args = ap.parse_args()
    T, N, dev = args.frames, 300, 'cuda:0'
    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=T // 2, nms_post=N),
                                   S.synth_state_dict(args.head), None, dev)
    apply_mode(model, args.mode)
    fr = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
    metas = [S.synth_meta() for _ in range(T)]
    ts = []
    with torch.no_grad():
        for _ in range(args.iters + 1):
            torch.cuda.synchronize()
            t0 = time.time()
            c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
            model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
    print('mode %s: %s ms per window' % (args.mode, ['%.2f' % (t * 1e3) for t in ts[1:]]))


if __name__ == '__main__':
    main()
