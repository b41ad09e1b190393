"""Which stage's f32 rounding noise ends up in the final box coordinates?  The oracle in float64 is the truth; per experiment ONE part of the
chain runs on the device (compute mode --mode: f32 = exact-f32 MFMA, f16x2 = split half), everything in front of and behind it in float64 on
the host, and the decoded key-frame boxes [300, 4] (get_det_bboxes without NMS: no discontinuity) are compared with the all-f64 boxes:

    backbone      frames -> C4 on the device, the rest f64
    rpn           f64 C4 -> the device's RPN convs / proposals, res5 + RoIAlign + head + decode f64 on THOSE proposals
    res5          f64 C4 -> device res5 (C5), f64 proposals / RoIAlign / head
    head chain    f64 C4 + f64 proposals injected -> device res5, RoIAlign, head, decode
    head          f64 RoI features -> device head -> f64 decode
    everything    the device's own window

    python tools/noise_contrib.py [--mode f16x2] [--head selsa] [--clip 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvrnet_amd  # noqa: E402
from hvrnet_amd import native, parity, synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from oracle import hvr_oracle as O  # noqa: E402  (a measurement tool, not the product)

ap = argparse.ArgumentParser()
ap.add_argument('--head', default='selsa')
ap.add_argument('--mode', default='f16x2', choices=['f32', 'f16x2'])
ap.add_argument('--clip', type=int, default=0)
ap.add_argument('--built-two-level', action='store_true', help='the RPN conv on the built two-level kernel (tile_hint 18) in this mode too (split half keeps the big tiles by default)')
ap.add_argument('--rpn-two-level', action='store_true', help='one more row: the RPN 3x3 conv as nine per-tap products with f32 outputs summed in f32 (what two-level accumulation of its K = 9 216 sum would give: VERDICT r05 item 1d)')
args = ap.parse_args()
T, N, KEY, dev = 15, 300, 7, 'cuda:0'
from bench import host_cores  # noqa: E402
torch.set_num_threads(host_cores())   # the cgroup's CPU quota, not the host's core count (oversubscribed f64 runs never finish)
dt = torch.float32 if args.mode == 'f32' else native.SPLIT
frames = [S.synth_frame(5000 * args.clip + i) for i in range(T)]
metas = [S.synth_meta() for _ in range(T)]
sd = S.synth_state_dict(args.head)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
cfg = dict(O.RPN_TEST_CFG, nms_post=N, max_num=N)
meta0 = metas[0]


def to_mode(x_nchw_f32):
    """f32 NCHW host tensor -> the model's activation format on the device (logical NCHW over physical NHWC)"""
    x = x_nchw_f32.to(dev).permute(0, 2, 3, 1).contiguous()
    if dt != torch.float32:
        x = native.cast(x, dt)
    return x.permute(0, 3, 1, 2)


def to_f64(x_dev_nchw):
    x = x_dev_nchw.permute(0, 2, 3, 1)
    if x.dtype != torch.float32:
        x = native.cast(x.contiguous(), torch.float32)
    return x.permute(0, 3, 1, 2).double().cpu()


def head64(roi_feats, cur):
    if args.head == 'selsa':
        return O.selsa_head_forward(roi_feats, sd64, cur, N, T)
    c, r = O.hvr_head_forward_test(roi_feats, sd64, cur, N, T)
    return c[-1], r[-1]


def boxes64(key_props, c, r):
    key_rois = O.bbox2roi([key_props])
    b, _ = O.get_det_bboxes(key_rois, c, r, meta0['img_shape'], meta0['scale_factor'], True, None)
    return b


def tail64(c5, proposals):
    """f64: RoIAlign on c5 at `proposals` (list of T [n,5] tensors), head, decode of the key frame"""
    proposals = [p.double() for p in proposals]
    rois = [O.bbox2roi([p]) for p in proposals]
    feats = torch.cat([O.roi_align(c5[i:i + 1], rois[i], 7, 1.0 / 16, 2) for i in range(T)], 0)
    cur = dict(start=int(sum(p.shape[0] for p in proposals[:KEY])), length=proposals[KEY].shape[0])
    c, r = head64(feats, cur)
    return boxes64(proposals[KEY], c, r)


def rest64(c4_list):
    x = torch.cat(c4_list, 0)
    c5 = O.shared_head(x, sd64)
    cls, reg = O.rpn_forward(x, sd64)
    base = O.gen_base_anchors(O.ANCHOR_CFG['base_size'], O.ANCHOR_CFG['scales'], O.ANCHOR_CFG['ratios'])
    anchors = O.grid_anchors(base, cls.shape[-2:], O.ANCHOR_CFG['stride']).double()
    props = [O.rpn_get_bboxes_single(cls[i], reg[i], anchors, metas[i]['img_shape'], cfg) for i in range(T)]
    return c5, props, tail64(c5, props)


def same_lists(a, b):
    return all(parity.proposal_lists_equal([p.detach().cpu().numpy() for p in a], [p.detach().cpu().numpy() for p in b]))


def report(name, boxes, ref, note=''):
    d = (boxes.double().cpu() - ref).abs()
    print('%-14s max |dbox| %.3g px   rms %.3g px   (99th pct %.3g)  %s' % (name, float(d.max()), float(d.pow(2).mean().sqrt()), float(d.flatten().kthvalue(int(d.numel() * 0.99)).values), note), flush=True)


with torch.no_grad():
    c4_64 = [O.resnet_c4(f.double(), sd64) for f in frames]
    c5_64, props_64, ref = rest64(c4_64)
    print('mode %s, head %s, clip %d: all-f64 reference built (%d boxes, extent %.0f px)' % (args.mode, args.head, args.clip, ref.shape[0], float(ref.abs().max())), flush=True)
    # the CPU's own f32 evaluation, for scale
    c4_32 = [O.resnet_c4(f, sd) for f in frames]
    res32, i32 = O.window_forward(c4_32, metas, sd, args.head, KEY, N, T, rpn_cfg=cfg, return_intermediates=True)
    b32 = boxes64(i32['proposals'][KEY].double(), i32['cls_scores'][-1].double(), i32['bbox_preds'][-1].double())
    report('cpu f32 (all)', b32, ref, 'proposal lists equal the f64 run\'s: %s' % same_lists(i32['proposals'], props_64))

    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=KEY, nms_post=N), sd, dt, dev)
    if args.built_two_level:
        model.rpn_head.two_level_dtypes = (torch.float32, native.SPLIT)
    fr_dev = torch.cat(frames, 0).to(dev)
    # ---- backbone on the device, the rest f64
    c4_dev = model(img=fr_dev, img_meta=metas, backbone_feat=True)[0]
    c4_dev64 = to_f64(c4_dev)
    _, props_b, boxes_b = rest64([c4_dev64[i:i + 1] for i in range(T)])
    report('backbone', boxes_b, ref, 'proposal lists equal: %s' % same_lists(props_b, props_64))
    # ---- from here on the device starts from the f64 C4 maps (rounded to the mode's format)
    x_dev = to_mode(torch.cat(c4_64, 0).float())
    w = model.window_tensors(x_dev, metas)
    props_dev = [p.double().cpu() for p in w['proposals']]
    eq = same_lists(props_dev, props_64)
    dprop = max(float((a[:, :4] - b[:, :4]).abs().max()) for a, b in zip(props_dev, props_64)) if eq else float('nan')
    report('rpn', tail64(c5_64, props_dev), ref, 'proposal lists equal: %s; max |d proposal coordinate| %.3g px' % (eq, dprop))
    if args.rpn_two_level:
        # the same with the RPN's 3x3 conv (K = 9 Cin = 9 216: one running accumulator per output on the device) summed in TWO levels: nine
        # per-tap 1x1 products (K = 1 024 each) with f32 outputs, added in f32 -- emulated with the shipped kernels, a measurement only
        import torch.nn.functional as F_
        from hvrnet_amd.backbone import as_logical, as_nhwc
        rpn = model.rpn_head
        one_level = rpn.forward_single

        def two_level(x):
            pk = rpn.packed(x.device)
            A_ = rpn.num_anchors
            xn = as_nhwc(x, rpn.compute_dtype)
            w3, b3 = pk['conv']
            H_, W_ = xn.shape[1], xn.shape[2]
            xp = F_.pad(xn, (0, 0, 1, 1, 1, 1))
            acc = None
            for ky in range(3):
                for kx in range(3):
                    part = native.conv2d_nhwc(xp[:, ky:ky + H_, kx:kx + W_, :].contiguous(), w3[:, ky:ky + 1, kx:kx + 1, :].contiguous(), None, relu=False, out_f32=True)
                    acc = part if acc is None else acc + part
            y32 = torch.relu(acc + b3)
            y = y32 if rpn.compute_dtype == torch.float32 else native.cast(y32, rpn.compute_dtype)
            o = native.conv2d_nhwc(y, pk['heads'][0], pk['heads'][1], relu=False, out_f32=True)
            return as_logical(o[..., :A_]), as_logical(o[..., A_:5 * A_])

        c1, r1 = one_level(x_dev)
        c2, r2 = two_level(x_dev)
        print('               (per-tap form against the shipped conv on the same input: max |d objectness logit| %.3g, max |d delta| %.3g)'
              % (float((c1 - c2).abs().max()), float((r1 - r2).abs().max())), flush=True)
        rpn.forward_single = two_level
        try:
            w2 = model.window_tensors(x_dev, metas)
        finally:
            del rpn.forward_single
        props_2 = [p.double().cpu() for p in w2['proposals']]
        eq2 = same_lists(props_2, props_64)
        dprop2 = max(float((a[:, :4] - b[:, :4]).abs().max()) for a, b in zip(props_2, props_64)) if eq2 else float('nan')
        report('rpn, two-level', tail64(c5_64, props_2), ref, 'proposal lists equal: %s; max |d proposal coordinate| %.3g px' % (eq2, dprop2))
    c5_dev64 = to_f64(w['c5'])
    report('res5', tail64(c5_dev64, props_64), ref)
    # ---- device res5 + RoIAlign + head + decode on the f64 proposals
    wi = model.window_tensors(x_dev, metas, proposals=[p.float().to(dev) for p in props_64])
    if args.head == 'selsa':
        cd, rd = model.bbox_head(wi['roi_feats'], wi['cur_range'], key_dim=KEY)[:2]
    else:
        cs, rs = model.bbox_head.forward_test(wi['roi_feats'], [wi['cur_range']], key_dim=KEY)
        cd, rd = cs[-1], rs[-1]
    report('head chain', boxes64(props_64[KEY], cd.double().cpu(), rd.double().cpu()), ref, '(device res5 + RoIAlign + head on the f64 proposals, f64 decode)')
    # ---- head alone on the f64 RoI features
    rois64 = [O.bbox2roi([p]) for p in props_64]
    rf64 = torch.cat([O.roi_align(c5_64[i:i + 1], rois64[i], 7, 1.0 / 16, 2) for i in range(T)], 0)
    cur = dict(start=KEY * N, length=N)
    rf_dev = rf64.float().to(dev)
    if args.head == 'selsa':
        cd, rd = model.bbox_head(rf_dev, cur, key_dim=KEY)[:2]
    else:
        cs, rs = model.bbox_head.forward_test(rf_dev, [cur], key_dim=KEY)
        cd, rd = cs[-1], rs[-1]
    report('head', boxes64(props_64[KEY], cd.double().cpu(), rd.double().cpu()), ref, '(f64 RoI features -> device head, f64 decode)')
    # ---- everything on the device (decoded without NMS)
    wa = model.window_tensors(c4_dev, metas)
    if args.head == 'selsa':
        cd, rd = model.bbox_head(wa['roi_feats'], wa['cur_range'], key_dim=KEY)[:2]
    else:
        cs, rs = model.bbox_head.forward_test(wa['roi_feats'], [wa['cur_range']], key_dim=KEY)
        cd, rd = cs[-1], rs[-1]
    pa = [p.double().cpu() for p in wa['proposals']]
    report('everything', boxes64(pa[KEY], cd.double().cpu(), rd.double().cpu()), ref, 'proposal lists equal: %s' % same_lists(pa, props_64))
