"""Seeded inputs shared by the golden-vector generator (make_golden.py, runs the reference) and
the tests that replay the same inputs through the oracle / the HIP path."""
import torch


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def delta2bbox_case(n=256, seed=101):
    g = _gen(seed)
    xy = torch.rand((n, 2), generator=g) * torch.tensor([900.0, 520.0])
    wh = torch.rand((n, 2), generator=g) * 300 + 1
    rois = torch.cat([xy, xy + wh], 1)
    deltas = torch.randn((n, 4), generator=g) * 3.0
    deltas[:8, 2:] = 50.0     # clamp edge (wh_ratio_clip)
    deltas[8:16, 2:] = -50.0
    deltas[16:24, :2] = 40.0  # pushes boxes out of the image -> clip to img_shape
    return rois, deltas


def boxes(n, seed, span=(900.0, 520.0), size=(8.0, 160.0)):
    g = _gen(seed)
    xy = torch.rand((n, 2), generator=g) * torch.tensor(span)
    wh = torch.rand((n, 2), generator=g) * (size[1] - size[0]) + size[0]
    sc = torch.rand((n, 1), generator=g)
    return torch.cat([xy, xy + wh, sc], 1)


def nms_cases():
    """(name, dets, thr).  Scores are distinct; `tie*` cases have IoU exactly == thr (>= semantics)."""
    out = [('doc', torch.tensor([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9], [49.2, 31.8, 51.0, 35.4, 0.5],
                                 [35.1, 11.5, 39.1, 15.7, 0.5], [35.6, 11.8, 39.3, 14.2, 0.5], [35.3, 11.5, 39.9, 14.5, 0.4],
                                 [35.2, 11.7, 39.7, 15.7, 0.3]]), 0.7)]
    for n in (1, 64, 65, 300, 6000):
        for thr in (0.3, 0.7):
            span = (900.0, 520.0) if n < 1000 else (990.0, 590.0)
            out.append(('n%d_t%02d' % (n, int(thr * 10)), boxes(n, 1000 + n, span=span), thr))
    # IoU exactly 0.5: boxes [0,0,9,9] (area 100) and [0,0,9,4] (area 50): inter 50 / union 100
    out.append(('tie_iou_half', torch.tensor([[0., 0., 9., 9., 0.9], [0., 0., 9., 4., 0.8], [20., 20., 29., 29., 0.7],
                                              [20., 20., 29., 24., 0.95]]), 0.5))
    return out


def relation_input(m=96, d=1024, seed=201):
    return torch.randn((m, d), generator=_gen(seed))


def roi_feat_input(m=96, seed=202):
    return torch.randn((m, 256, 7, 7), generator=_gen(seed)).abs()  # post-ReLU features are non-negative


def det_case(r=300, ncls=31, seed=203):
    g = _gen(seed)
    b = boxes(r, seed + 1)
    rois = torch.cat([torch.zeros(r, 1), b[:, :4]], 1)
    cls = torch.randn((r, ncls), generator=g) * 2.0
    reg = torch.randn((r, 4), generator=g)
    return rois, cls, reg


def small_image(seed=204):
    return torch.randn((1, 3, 64, 96), generator=_gen(seed)) * 50.0


def head_train_case(n=32, ncls=31, seed=205):
    """Targets for the key frame's n sampled RoIs of the SELSA head's training step (bbox_head.py:87-130 argument order:
    labels, label_weights, bbox_targets, bbox_weights): a third of the RoIs are positives (class 1..30, weight-1 boxes)."""
    g = _gen(seed)
    labels = torch.zeros(n, dtype=torch.long)
    npos = n // 3
    labels[:npos] = torch.randint(1, ncls, (npos,), generator=g)
    label_weights = torch.ones(n)
    bbox_targets = torch.zeros(n, 4)
    bbox_targets[:npos] = torch.randn((npos, 4), generator=g) * 0.8   # both branches of the smooth-L1 (|d| < 1 and > 1)
    bbox_weights = torch.zeros(n, 4)
    bbox_weights[:npos] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights


def target_case(n_prop=300, ncls=31, seed=206):
    """Inputs of the target-generation steps of one training iteration on a 600x1000 key frame: ground truth (one box
    equal to an anchor so an IoU of exactly 1 occurs; one tiny box whose best anchor stays under pos_iou_thr, the
    "each gt keeps its best boxes" branch), RPN head maps of the key frame, `n_prop` proposals (jittered copies of the ground truth + background) and the
    head's outputs on them for the OHEM ranking."""
    g = _gen(seed)
    gt_bboxes = torch.tensor([[120., 80., 420., 330.],      # large
                              [296., 136., 359., 199.],      # == the 64x64 anchor centred on cell (10, 20): IoU exactly 1
                              [700., 400., 716., 420.],      # tiny: best anchor IoU < min_pos_iou, gets no anchor
                              [500., 100., 780., 300.],
                              [820., 420., 865., 460.]])     # best anchor between min_pos_iou and pos_iou_thr: low-quality match
    gt_labels = torch.tensor([3, 17, 30, 9, 22])
    rpn_cls = torch.randn((1, 12, 38, 63), generator=g) * 1.5
    rpn_reg = torch.randn((1, 48, 38, 63), generator=g) * 0.3
    props = []
    for k in range(5):
        b = gt_bboxes[k]
        w, h = b[2] - b[0], b[3] - b[1]
        jit = (torch.rand((24, 4), generator=g) - 0.5) * torch.stack([w, h, w, h]) * 0.5
        props.append(b[None] + jit)
    props.append(boxes(n_prop - 120, seed + 1, span=(900.0, 520.0), size=(10.0, 200.0))[:, :4])
    props = torch.cat(props, 0)
    props[:, 0::2] = props[:, 0::2].clamp(0, 999)
    props[:, 1::2] = props[:, 1::2].clamp(0, 599)
    props = torch.cat([props[:, :2], torch.max(props[:, 2:], props[:, :2] + 1)], 1)
    props = props[torch.randperm(n_prop, generator=g)]
    proposals = torch.cat([props, torch.rand((n_prop, 1), generator=g)], 1)
    cls_score = torch.randn((n_prop + 5, ncls), generator=g) * 2.0
    bbox_pred = torch.randn((n_prop + 5, 4), generator=g) * 0.7
    return dict(gt_bboxes=gt_bboxes, gt_labels=gt_labels, rpn_cls=rpn_cls, rpn_reg=rpn_reg, proposals=proposals,
                cls_score=cls_score, bbox_pred=bbox_pred)


RPN_TRAIN_CFG = dict(assigner=dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3), allowed_border=0, pos_weight=-1,
                     sampler=dict(num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False))
RCNN_TRAIN_CFG = dict(assigner=dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5), pos_weight=-1,
                      sampler=dict(num=300, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                      ohem=dict(num=128, pos_fraction=0.25, neg_pos_ub=-1))


def mining_case(mq=96, mk=288, d=64, ncls=6, seed=207):
    """Scaled affinities of `mq` key-frame proposals against `mk` proposals of the video with class labels (0 = background,
    a third of the rows), one class present only once among the keys and one query class absent from them."""
    g = _gen(seed)
    q, k = torch.randn((mq, d), generator=g), torch.randn((mk, d), generator=g)
    labels = torch.randint(0, ncls, (mq,), generator=g)
    labels[torch.rand(mq, generator=g) < 0.33] = 0
    all_labels = torch.randint(0, ncls, (mk,), generator=g)
    all_labels[all_labels == ncls - 1] = 1
    all_labels[7] = ncls - 1          # a class with a single key
    labels[3] = ncls + 3              # a query class no key carries: its "same label" set is empty
    aff_scale = (q @ k.t()) * (1.0 / d ** 0.5)
    return labels, all_labels, aff_scale


def hvr_train_case(videos=3, frames=3, n=16, ncls=31, seed=215):
    """Inputs of the HVR head's TRAINING forward (hrnmp_bbox_head.py:609-795, dynamic=False): per video the RoI features of its
    `frames` frames (key frame's n rows first), the key rows' labels of all videos (`others`: about a third background, a few
    foreground classes so that every foreground row has same-label and different-label keys among the videos * n key rows) and
    BBoxHead.loss targets for them.  -> (feats list, cur_range_s, labels, label_weights, bbox_targets, bbox_weights)."""
    g = _gen(seed)
    feats = [torch.randn((frames * n, 256, 7, 7), generator=g).abs() for _ in range(videos)]
    cur = [dict(start=0, length=n) for _ in range(videos)]
    m = videos * n
    labels = torch.randint(1, 5, (m,), generator=g)
    labels[torch.rand(m, generator=g) < 0.33] = 0
    labels[1], labels[n + 2] = ncls - 1, ncls - 1     # a class carried by exactly two rows (in different videos)
    label_weights = torch.ones(m)
    bbox_targets = torch.randn((m, 4), generator=g) * 0.8
    bbox_weights = (labels > 0).float()[:, None].expand(m, 4).contiguous()
    bbox_targets = bbox_targets * bbox_weights
    return feats, cur, labels, label_weights, bbox_targets, bbox_weights


def eval_case(n_img=24, n_cls=5, seed=208):
    """Per-class detection arrays + annotations for the mAP evaluation: detections are jittered copies of the ground truth
    (some duplicated -> later claims of a covered box are false positives), background boxes, equal scores, images without
    ground truth or without detections, a class without any ground truth, and ignore flags on a few boxes."""
    import numpy as np
    rng = np.random.RandomState(seed)
    dets, gtb, gtl, gti = [], [], [], []
    for i in range(n_img):
        k = 0 if i % 7 == 3 else rng.randint(1, 6)
        xy = rng.rand(k, 2) * np.array([800.0, 450.0])
        wh = rng.rand(k, 2) * 180 + 12
        boxes = np.hstack([xy, xy + wh]).astype(np.float32)
        labels = rng.randint(1, n_cls, size=k)          # class n_cls never occurs in the ground truth
        gtb.append(boxes)
        gtl.append(labels)
        gti.append(rng.rand(k) < 0.15)
        per_cls = []
        for c in range(1, n_cls + 1):
            mine = boxes[labels == c]
            rows = []
            for b in mine:
                for _ in range(rng.randint(0, 3)):
                    jit = (rng.rand(4) - 0.5) * (b[2] - b[0]) * rng.choice([0.1, 0.6])
                    rows.append(np.append(b + jit, np.round(rng.rand(), 2)))       # two-decimal scores: ties occur
            for _ in range(rng.randint(0, 3) if i % 5 else 0):
                o = rng.rand(2) * np.array([800.0, 450.0])
                rows.append(np.array([o[0], o[1], o[0] + 40, o[1] + 60, np.round(rng.rand(), 2)]))
            per_cls.append(np.array(rows, dtype=np.float32).reshape(-1, 5))
        dets.append(per_cls)
    return dets, gtb, gtl, gti
