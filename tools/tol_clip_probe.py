"""Why a clip's RPN proposal lists differ between the device and the CPU oracle (tolerance claim over many clips, profiles/r06_bench_tol24.json).

    python tools/tol_clip_probe.py --clips 12 --mode f16x2        # clip c = synthetic frames 5000 c + 0 .. 14 (bench.py's tol_clip_ids)

For every frame whose lists differ it prints the boxes only one side holds with their rank and score, the nearest box of the other side,
the other side's last kept score (a box at the top-300 cut differs by a RANK tie, not by an NMS decision), and for every one-sided box the
higher-scored box of the other side with the IoU nearest 0.7 -- computed in float64 from EITHER side's coordinates (a one-pixel-high
sliver's IoU moves by 1e-4 with 1e-4 px).  Then the window's final distances, as they are and with the oracle's lists injected.
Measurement tooling: imports the oracle as the checker, like bench.py's parity legs."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, parity, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config  # noqa: E402


def iou64(a, b):
    """float64 IoU with the reference's +1 convention (nms_cpu.cpp:30-45) of box a against rows of b"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(-1, b.shape[-1])
    xx1, yy1 = np.maximum(a[0], b[:, 0]), np.maximum(a[1], b[:, 1])
    xx2, yy2 = np.minimum(a[2], b[:, 2]), np.minimum(a[3], b[:, 3])
    inter = np.maximum(0.0, xx2 - xx1 + 1) * np.maximum(0.0, yy2 - yy1 + 1)
    aa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (aa + ab - inter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, nargs='+', default=[12])
    ap.add_argument('--mode', choices=['f16x2', 'f32'], default='f16x2')
    ap.add_argument('--tol', type=float, default=1e-2)
    args = ap.parse_args()
    from oracle import hvr_oracle as O   # checker
    from bench import host_cores, parity_object
    torch.set_num_threads(host_cores())
    T, n_prop, dev = 15, 300, 'cuda:0'
    sd = S.synth_state_dict('hvr')
    dt = native.SPLIT if args.mode == 'f16x2' else torch.float32
    model = hvrnet_amd.build_model(hvr_config(), sd, dt, dev)
    metas = [S.synth_meta() for _ in range(T)]
    rpn_cfg = dict(O.RPN_TEST_CFG, nms_post=n_prop, max_num=n_prop)
    for c in args.clips:
        ids = [5000 * c + i for i in range(T)] if c > 0 else list(range(T))
        imgs = [S.synth_frame(i) for i in ids]
        with torch.no_grad():
            want, inter = O.clip_forward(imgs, metas, sd, 'hvr', T // 2, n_prop, T, rpn_cfg=rpn_cfg, return_intermediates=True)
            wp = [p_.numpy() for p_ in inter['proposals']]
            fr = torch.cat(imgs, 0).to(dev)
            c4 = model(img=fr, img_meta=metas, backbone_feat=True)[0]
            got = model(x=c4, img=None, img_meta=metas, forward_feat=True, return_loss=False, rescale=True)
            gp = [p_.cpu().numpy() for p_ in model.window_tensors(c4, metas)['proposals']]
            got_i = model(x=c4, img=None, img_meta=metas, proposals=[torch.from_numpy(p_).to(dev) for p_ in wp], forward_feat=True, return_loss=False, rescale=True)
        p0, p1 = parity_object('hvr', args.mode, got, want), parity_object('hvr', args.mode, got_i, want)
        print('== clip %d, mode %s: as it is: flips %d, score %.2g, box %.5f px; oracle proposals injected: flips %d, score %.2g, box %.5f px'
              % (c, args.mode, p0['class_flips'], p0['max_score_err'], p0['max_box_err'], p1['class_flips'], p1['max_score_err'], p1['max_box_err']))
        eq = parity.proposal_lists_equal(gp, wp, args.tol)
        for f, same in enumerate(eq):
            g, w = np.asarray(gp[f], np.float64), np.asarray(wp[f], np.float64)
            # largest coordinate distance between the sorted lists: how "equal" an equal frame is
            if same:
                continue
            print('  frame %d: device keeps %d, oracle %d; last kept score device %.7f oracle %.7f' % (f, len(g), len(w), g[-1, 4], w[-1, 4]))
            for side, a, b in (('device', g, w), ('oracle', w, g)):
                for i in parity._only_in(a, b, args.tol):
                    d = np.abs(b[:, :4] - a[i, :4]).max(axis=1)
                    j = int(np.argmin(d))
                    line = '    only %s: rank %3d score %.7f box [%.3f %.3f %.3f %.3f]; nearest on the other side: rank %3d, %.4f px away, score %.7f' % (
                        side, i, a[i, 4], a[i, 0], a[i, 1], a[i, 2], a[i, 3], j, d[j], b[j, 4])
                    sup = b[b[:, 4] >= a[i, 4] - 1e-6]
                    if len(sup):
                        io = iou64(a[i, :4], sup[:, :4])
                        k = int(np.argmin(np.abs(io - 0.7)))
                        # the same pair from the suppressor's own side's coordinates of the box, when that side has the box among its pre-NMS candidates
                        line += '; higher-scored box of the other side nearest IoU 0.7: rank %d, IoU(f64) %.7f' % (int(np.where((b == sup[k]).all(axis=1))[0][0]), io[k])
                    cut = b[-1, 4]
                    line += '; other side cut score %.7f (%+.2g)' % (cut, a[i, 4] - cut)
                    print(line)
        sys.stdout.flush()


if __name__ == '__main__':
    main()
