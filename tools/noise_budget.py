"""Where the f32 rounding noise of a full-size window comes from (SELSA head: one read-out branch): the oracle in float64 is the truth;
the oracle in float32 (CPU) and the device's exact-f32 mode are two f32 evaluations of it.  Per stage, the relative error of each
against the f64 value WITH THE STAGE'S INPUT TAKEN FROM THE F64 RUN (rounded to f32): the stage's own noise, not what it inherits.

    python tools/noise_budget.py [--head selsa]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from oracle import hvr_oracle as O  # noqa: E402  (a measurement tool, not the product)

ap = argparse.ArgumentParser()
ap.add_argument('--head', default='selsa')
args = ap.parse_args()
T, N, KEY, dev = 15, 300, 7, 'cuda:0'
from bench import host_cores  # noqa: E402
torch.set_num_threads(host_cores())
frames = [S.synth_frame(i) for i in range(T)]
metas = [S.synth_meta() for _ in range(T)]
sd = S.synth_state_dict(args.head)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
cfg = dict(O.RPN_TEST_CFG, nms_post=N, max_num=N)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


with torch.no_grad():
    c4_64 = [O.resnet_c4(f.double(), sd64) for f in frames]
    c4_32 = [O.resnet_c4(f, sd) for f in frames]
    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=KEY, nms_post=N), sd, torch.float32, dev)
    c4_dev = model(img=torch.cat(frames, 0).to(dev), img_meta=metas, backbone_feat=True)[0]
    print('backbone (frames -> C4):            cpu f32 %.3g   device f32 %.3g   (relative to max |C4|)' % (rel(torch.cat(c4_32), torch.cat(c4_64)), rel(c4_dev, torch.cat(c4_64))))
    # from here on every evaluation starts from the f64 C4 maps rounded to f32
    c4_in = [c.float() for c in c4_64]
    res64, i64 = O.window_forward(c4_64, metas, sd64, args.head, KEY, N, T, rpn_cfg=cfg, return_intermediates=True)
    res32, i32 = O.window_forward(c4_in, metas, sd, args.head, KEY, N, T, rpn_cfg=cfg, return_intermediates=True)
    x_dev = torch.cat(c4_in, 0).to(dev).contiguous(memory_format=torch.channels_last)
    w = model.window_tensors(x_dev, metas)
    print('res5 (C4 -> C5):                    cpu f32 %.3g   device f32 %.3g' % (rel(i32['c5'], i64['c5']), rel(w['c5'], i64['c5'])))
    same = all(torch.equal(a.cpu()[:, :4].float().round(decimals=2), b[:, :4].float().round(decimals=2)) for a, b in zip(w['proposals'], i64['proposals']))
    print('proposal lists equal the f64 run\'s (to 1e-2 px): device %s   cpu f32 %s' % (same, all(torch.equal(a[:, :4].round(decimals=2), b[:, :4].float().round(decimals=2)) for a, b in zip(i32['proposals'], i64['proposals']))))
    print('RoIAlign (C5 + rois -> features):   cpu f32 %.3g   device f32 %.3g' % (rel(i32['roi_feats'], i64['roi_feats']), rel(w['roi_feats'], i64['roi_feats'])))
    # head alone on the f64 RoI features (rounded)
    rf = i64['roi_feats'].float()
    cur = i64['cur_range']
    if args.head == 'selsa':
        c32, r32 = O.selsa_head_forward(rf, sd, cur, N, T)
        cd, rd = model.bbox_head(rf.to(dev), cur, key_dim=KEY)[:2]
        c64, r64 = i64['cls_scores'][0], i64['bbox_preds'][0]
    else:
        c32, r32 = [t[-1] for t in O.hvr_head_forward_test(rf, sd, cur, N, T)]
        cs, rs = model.bbox_head.forward_test(rf.to(dev), [cur], key_dim=KEY)
        cd, rd = cs[-1], rs[-1]
        c64, r64 = i64['cls_scores'][-1], i64['bbox_preds'][-1]
    print('head (features -> box deltas):      cpu f32 %.3g   device f32 %.3g   abs: cpu %.3g device %.3g (|deltas| max %.3g)'
          % (rel(r32, r64), rel(rd, r64), float((r32.double() - r64).abs().max()), float((rd.double().cpu() - r64).abs().max()), float(r64.abs().max())))
    print('head (features -> class logits):    cpu f32 %.3g   device f32 %.3g' % (rel(c32, c64), rel(cd, c64)))
    # decode alone on the f64 deltas (rounded)
    from hvrnet_amd import parity
    key_rois = O.bbox2roi([i64['proposals'][KEY]])
    b32, _ = O.get_det_bboxes(key_rois.float(), c64.float(), r64.float(), metas[0]['img_shape'], metas[0]['scale_factor'], True, None)
    b64, _ = O.get_det_bboxes(key_rois, c64, r64, metas[0]['img_shape'], metas[0]['scale_factor'], True, None)
    sc, bd = model.bbox_head.get_det_bboxes(key_rois.float().to(dev), c64.float().to(dev), r64.float().to(dev), metas[0]['img_shape'], metas[0]['scale_factor'], rescale=True, cfg=None)
    print('decode (deltas -> boxes, px):       cpu f32 %.3g   device f32 %.3g   (absolute, max over %d x 4 coordinates)'
          % (float((b32.double() - b64).abs().max()), float((sc.double().cpu() - b64).abs().max()) if sc.shape == b64.shape else float((bd.double().cpu() - b64).abs().max()), b64.shape[0]))
