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
