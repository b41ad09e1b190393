"""Which detection of a window's read-out differs between the device and the CPU oracle, and why (tools/tol_clip_probe.py's counterpart for the
read-out's own discontinuity: multiclass NMS at IoU 0.5 over score-ordered candidates, bbox_nms.py:6-66).   python tools/readout_probe.py --clip 1 --mode f16x2 [--branch 0]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config  # noqa: E402
from tools.tol_clip_probe import iou64  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clip', type=int, default=1)
    ap.add_argument('--mode', choices=['f16x2', 'f32'], default='f16x2')
    ap.add_argument('--branch', type=int, default=0)
    ap.add_argument('--one-level', action='store_true', help='the RPN conv on one accumulator in every mode')
    ap.add_argument('--two-level', action='store_true', help='... on the two-level kernel in split half too')
    args = ap.parse_args()
    from oracle import hvr_oracle as O   # checker
    from bench import host_cores
    torch.set_num_threads(host_cores())
    T, n_prop, dev = 15, 300, 'cuda:0'
    sd = S.synth_state_dict('hvr')
    dt = native.SPLIT if args.mode == 'f16x2' else torch.float32
    model = hvrnet_amd.build_model(hvr_config(), sd, dt, dev)
    if args.one_level:
        model.rpn_head.two_level_dtypes = ()
    if args.two_level:
        model.rpn_head.two_level_dtypes = (torch.float32, native.SPLIT)
    metas = [S.synth_meta() for _ in range(T)]
    ids = [5000 * args.clip + i for i in range(T)]
    imgs = [S.synth_frame(i) for i in ids]
    with torch.no_grad():
        want, inter = O.clip_forward(imgs, metas, sd, 'hvr', T // 2, n_prop, T, rpn_cfg=dict(O.RPN_TEST_CFG, nms_post=n_prop, max_num=n_prop), return_intermediates=True)
        c4 = model(img=torch.cat(imgs, 0).to(dev), img_meta=metas, backbone_feat=True)[0]
        got = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
    g, w = got[args.branch], want[args.branch]
    print('clip %d mode %s branch %d (%s RPN sum)' % (args.clip, args.mode, args.branch, 'one-level' if (args.one_level or (args.mode == 'f16x2' and not args.two_level)) else 'two-level'))
    for c in range(len(w)):
        a, b = np.asarray(g[c], np.float64).reshape(-1, 5), np.asarray(w[c], np.float64).reshape(-1, 5)
        if a.shape != b.shape:
            print('  class %d: device keeps %d, oracle %d' % (c, len(a), len(b)))
        n = min(len(a), len(b))
        for i in range(n):
            d = np.abs(a[i, :4] - b[i, :4]).max()
            if d > 1e-2 or abs(a[i, 4] - b[i, 4]) > 1e-3:
                io = iou64(a[i, :4], b[i:i + 1, :4])[0]
                print('  class %d rank %d: device [%.3f %.3f %.3f %.3f] %.7f | oracle [%.3f %.3f %.3f %.3f] %.7f | box diff %.3f px, IoU between them %.4f'
                      % (c, i, *a[i, :4], a[i, 4], *b[i, :4], b[i, 4], d, io))
                # is the oracle's box among the device's detections of this class at another rank (and vice versa)?
                da = np.abs(a[:, :4] - b[i, :4]).max(axis=1)
                db = np.abs(b[:, :4] - a[i, :4]).max(axis=1)
                print('      oracle box on the device side: nearest rank %d (%.4f px, score %.7f); device box on the oracle side: nearest rank %d (%.4f px, score %.7f)'
                      % (int(da.argmin()), da.min(), a[int(da.argmin()), 4], int(db.argmin()), db.min(), b[int(db.argmin()), 4]))


if __name__ == '__main__':
    main()
