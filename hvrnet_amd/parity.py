"""Comparison of two detection results (lists of per-class [k,5] arrays, what bbox2result returns,
mmdet/core/bbox/transforms.py:181-199) -- host numpy, used by the full-size parity tests and by bench.py's
`parity` object.  Nothing here computes a detection; it only measures how far two sets of them are apart.

Two measures:
  * `strict(got, want)`: north_star's bar -- every class holds the same number of detections in the same order
    (class indices exact), largest |score difference| and |coordinate difference| over all of them;
  * `track(got, want)`: for results that may differ in WHICH boxes survive the discontinuous steps (top-k, NMS) --
    each reference detection above a score floor is matched to the candidate of the same class with the highest IoU.
"""
import numpy as np

# north_star's tolerance on a window's detections against the CPU reference path (oracle.clip_forward on the same frames): class
# indices exact, scores within 1e-3, box coordinates within 1e-3 px.  ONE definition for bench.py, tools/precision_ladder.py, the
# full-size tests and smoke().  The box bar carries two f32 ulps at 1000 px (2 x 6.1e-5 = 1.2e-4): the oracle and the device round the
# decode's f32 arithmetic in different orders and coordinates reach 1000 -- the oracle's OWN f32 evaluation sits 4.3e-4 - 4.9e-4 px
# from its f64 evaluation on the benchmark's clips (bench.py: `oracle_noise_floor`).  Nothing else is added.
TOL_SCORE = 1e-3
TOL_BOX_PX = 1e-3 + 1.2e-4


def within_tolerance(st):
    """north_star's bar on a strict() result (or any dict with class_flips / max_score_err / max_box_err)."""
    return bool(st is not None and st['class_flips'] == 0 and st['max_score_err'] < TOL_SCORE and st['max_box_err'] < TOL_BOX_PX)



def _iou_one_to_many(box, cand):
    """IoU with the reference's +1 pixel convention (mmdet/core/bbox/geometry.py:34-45)."""
    x1 = np.maximum(cand[:, 0], box[0]); y1 = np.maximum(cand[:, 1], box[1])
    x2 = np.minimum(cand[:, 2], box[2]); y2 = np.minimum(cand[:, 3], box[3])
    inter = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
    area_c = (cand[:, 2] - cand[:, 0] + 1) * (cand[:, 3] - cand[:, 1] + 1)
    area_b = (box[2] - box[0] + 1) * (box[3] - box[1] + 1)
    return inter / (area_c + area_b - inter)


def strict(got, want):
    """-> dict(class_flips, n, max_score_err, max_box_err).  class_flips = sum over classes of |count difference|
    (0 = every detection carries the reference's class index); the errors are taken over the detections both sides
    hold at the same (class, rank) position."""
    assert len(got) == len(want)
    flips, n, es, eb = 0, 0, 0.0, 0.0
    for g, w in zip(got, want):
        g, w = np.asarray(g, dtype=np.float64).reshape(-1, 5), np.asarray(w, dtype=np.float64).reshape(-1, 5)
        flips += abs(len(g) - len(w))
        k = min(len(g), len(w))
        if k:
            es = max(es, float(np.abs(g[:k, 4] - w[:k, 4]).max()))
            eb = max(eb, float(np.abs(g[:k, :4] - w[:k, :4]).max()))
            n += k
    return dict(class_flips=int(flips), n=int(n), max_score_err=es, max_box_err=eb)


def track(got, want, score_floor=0.05, iou_match=0.9):
    """-> dict(n_ref, matched, same_class_frac, max_score_err, mean_score_err, max_box_err, missing): reference detections
    with score >= score_floor, the fraction that has a same-class candidate with IoU > iou_match, and the score / coordinate
    errors over those matches."""
    n_ref = matched = 0
    ds, db = [], []
    for g, w in zip(got, want):
        g, w = np.asarray(g, dtype=np.float64).reshape(-1, 5), np.asarray(w, dtype=np.float64).reshape(-1, 5)
        for box in w:
            if box[4] < score_floor:
                continue
            n_ref += 1
            if len(g) == 0:
                continue
            iou = _iou_one_to_many(box, g)
            j = int(np.argmax(iou))
            if iou[j] > iou_match:
                matched += 1
                ds.append(abs(g[j, 4] - box[4]))
                db.append(float(np.abs(g[j, :4] - box[:4]).max()))
    return dict(n_ref=int(n_ref), matched=int(matched), same_class_frac=(matched / float(n_ref)) if n_ref else 1.0,
                max_score_err=float(max(ds)) if ds else 0.0, mean_score_err=float(np.mean(ds)) if ds else 0.0,
                max_box_err=float(max(db)) if db else 0.0, missing=int(n_ref - matched))


def proposal_overlap(got, want, iou_match=0.9):
    """Per-frame proposal sets ([n,>=4] arrays): fraction of the reference's proposals that have a candidate with IoU >
    iou_match, averaged over frames, and the smallest per-frame fraction."""
    fr = []
    for g, w in zip(got, want):
        g, w = np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64)
        if len(w) == 0:
            fr.append(1.0)
            continue
        if len(g) == 0:
            fr.append(0.0)
            continue
        hit = sum(1 for box in w if _iou_one_to_many(box, g).max() > iou_match)
        fr.append(hit / float(len(w)))
    return dict(mean=float(np.mean(fr)), min=float(np.min(fr)))
