"""mAP evaluation of per-class detection arrays: the consumer of `bbox2result`'s output.

Same entry point and result structure as the reference's `eval_map` (mmdet/core/evaluation/mean_ap.py:475-586) on the path
`tools/vid_eval.py:11-52` takes: `dataset` is the tuple of 30 VID class names there, which selects `tpfp_default`
(mean_ap.py:376-438) and the area-under-curve AP (mean_ap.py:9-53).  Host-side numpy -- this runs once per evaluation over
a few hundred thousand boxes, not per frame -- written as array operations instead of the reference's per-detection Python
loop: for one image and class, the detections are ranked by score, each takes its best-overlapping ground-truth box, and a
true positive is the FIRST detection in rank order that claims a (non-ignored) box.
"""
import numpy as np


def bbox_overlaps(bboxes1, bboxes2):
    """IoU matrix [n, k] with the +1 pixel convention, float32 (mmdet/core/evaluation/bbox_overlaps.py:4-49, mode 'iou')."""
    b1, b2 = np.asarray(bboxes1, dtype=np.float32), np.asarray(bboxes2, dtype=np.float32)
    if b1.shape[0] == 0 or b2.shape[0] == 0:
        return np.zeros((b1.shape[0], b2.shape[0]), dtype=np.float32)
    area1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    area2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    w = np.maximum(np.minimum(b1[:, None, 2], b2[None, :, 2]) - np.maximum(b1[:, None, 0], b2[None, :, 0]) + 1, 0)
    h = np.maximum(np.minimum(b1[:, None, 3], b2[None, :, 3]) - np.maximum(b1[:, None, 1], b2[None, :, 1]) + 1, 0)
    inter = (w * h).astype(np.float32)
    return inter / (area1[:, None] + area2[None, :] - inter)


def _area(boxes):
    return (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)


def tpfp_default(det_bboxes, gt_bboxes, gt_ignore, iou_thr, area_ranges=None):
    """(tp, fp) float32 [num_scales, num_dets] for one image and class (mean_ap.py:376-438)."""
    ranges = [(None, None)] if area_ranges is None else list(area_ranges)
    n, k = det_bboxes.shape[0], gt_bboxes.shape[0]
    tp = np.zeros((len(ranges), n), dtype=np.float32)
    fp = np.zeros((len(ranges), n), dtype=np.float32)
    det_area = _area(det_bboxes) if n else np.zeros(0, dtype=np.float32)
    if k == 0:   # no ground truth: every detection (inside the area range) is a false positive
        for s, (lo, hi) in enumerate(ranges):
            fp[s, :] = 1 if lo is None else ((det_area >= lo) & (det_area < hi))
        return tp, fp
    if n == 0:
        return tp, fp
    ious = bbox_overlaps(det_bboxes, gt_bboxes)
    best, claim = ious.max(axis=1), ious.argmax(axis=1)
    order = np.argsort(-det_bboxes[:, -1])
    hit = best[order] >= iou_thr                      # in rank order
    gt_of = claim[order]
    # the first detection (in rank order) claiming each box
    first = np.zeros(n, dtype=bool)
    ranks = np.nonzero(hit)[0]
    if ranks.size:
        _, idx = np.unique(gt_of[ranks], return_index=True)
        first[ranks[idx]] = True
    gt_ignore = np.asarray(gt_ignore).astype(bool)
    gt_area = _area(gt_bboxes)
    for s, (lo, hi) in enumerate(ranges):
        dead = gt_ignore if lo is None else (gt_ignore | (gt_area < lo) | (gt_area >= hi))
        # "first claim" must be counted among the boxes that are alive at this scale only (a dead box never gets covered)
        counted = hit & ~dead[gt_of]
        first_s = first if lo is None and not gt_ignore.any() else np.zeros(n, dtype=bool)
        if first_s is not first:
            r = np.nonzero(counted)[0]
            if r.size:
                _, idx = np.unique(gt_of[r], return_index=True)
                first_s[r[idx]] = True
        in_range = np.ones(n, dtype=bool) if lo is None else ((det_area[order] >= lo) & (det_area[order] < hi))
        tp[s, order] = counted & first_s
        fp[s, order] = (counted & ~first_s) | (~hit & in_range)
    return tp, fp


def average_precision(recalls, precisions, mode='area'):
    """mean_ap.py:9-53: area under the monotone precision envelope ('area') or the 11-point average."""
    r, p = np.atleast_2d(recalls), np.atleast_2d(precisions)
    assert r.shape == p.shape
    ap = np.zeros(r.shape[0], dtype=np.float32)
    if mode == 'area':
        pad0, pad1 = np.zeros((r.shape[0], 1), dtype=r.dtype), np.ones((r.shape[0], 1), dtype=r.dtype)
        mrec = np.hstack((pad0, r, pad1))
        mpre = np.maximum.accumulate(np.hstack((pad0, p, pad0))[:, ::-1], axis=1)[:, ::-1]   # envelope from the right
        for i in range(r.shape[0]):
            step = np.nonzero(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, step + 1] - mrec[i, step]) * mpre[i, step + 1])
    elif mode == '11points':
        for i in range(r.shape[0]):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                sel = p[i, r[i] >= thr]
                ap[i] += sel.max() if sel.size else 0
            ap /= 11   # (inside the scale loop, as the reference has it, mean_ap.py:47)
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if np.ndim(recalls) == 1 else ap


def eval_map(det_results, gt_bboxes, gt_labels, gt_ignore=None, scale_ranges=None, iou_thr=0.5, dataset=None, print_summary=True):
    """det_results: per image a list of per-class [k, 5] arrays (bbox2result); gt_bboxes / gt_labels (1-based) per image.
    -> (mAP, [dict(num_gts, num_dets, recall, precision, ap) per class])   (mean_ap.py:475-586)."""
    assert len(det_results) == len(gt_bboxes) == len(gt_labels)
    if dataset in ('det', 'vid'):
        raise NotImplementedError("tpfp_imagenet (dataset 'det' / 'vid') is not the path tools/vid_eval.py takes: it passes class names")
    if gt_ignore is not None:
        assert len(gt_ignore) == len(gt_labels) and all(len(a) == len(b) for a, b in zip(gt_labels, gt_ignore))
    area_ranges = None if scale_ranges is None else [(lo ** 2, hi ** 2) for lo, hi in scale_ranges]
    num_scales = 1 if scale_ranges is None else len(scale_ranges)
    labels = [np.asarray(l) if np.asarray(l).ndim == 1 else np.asarray(l)[:, 0] for l in gt_labels]
    results = []
    eps = np.finfo(np.float32).eps
    for c in range(len(det_results[0])):
        dets = [np.asarray(d[c], dtype=np.float32).reshape(-1, 5) for d in det_results]
        tps, fps = [], []
        num_gts = np.zeros(num_scales, dtype=int)
        for j, boxes in enumerate(gt_bboxes):
            boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
            mine = labels[j] == c + 1
            g = boxes[mine] if boxes.shape[0] else boxes
            ign = np.zeros(g.shape[0], dtype=bool) if gt_ignore is None else np.asarray(gt_ignore[j])[mine].astype(bool)
            t, f = tpfp_default(dets[j], g, ign, iou_thr, area_ranges)
            tps.append(t)
            fps.append(f)
            if area_ranges is None:
                num_gts[0] += int((~ign).sum())
            else:
                ga = _area(g)
                for s, (lo, hi) in enumerate(area_ranges):
                    num_gts[s] += int((~ign & (ga >= lo) & (ga < hi)).sum())
        all_dets = np.vstack(dets)
        order = np.argsort(-all_dets[:, -1])
        tp = np.cumsum(np.hstack(tps)[:, order], axis=1)
        fp = np.cumsum(np.hstack(fps)[:, order], axis=1)
        recalls = tp / np.maximum(num_gts[:, None], eps)
        precisions = tp / np.maximum(tp + fp, eps)
        if scale_ranges is None:
            recalls, precisions, n_gt = recalls[0], precisions[0], int(num_gts[0])
        else:
            n_gt = num_gts
        ap = average_precision(recalls, precisions, 'area' if dataset != 'voc07' else '11points')
        results.append(dict(num_gts=n_gt, num_dets=all_dets.shape[0], recall=recalls, precision=precisions, ap=ap))
    if scale_ranges is None:
        aps = [r['ap'] for r in results if r['num_gts'] > 0]
        mean_ap = float(np.mean(aps)) if aps else 0.0
    else:
        all_ap = np.vstack([r['ap'] for r in results])
        all_gt = np.vstack([r['num_gts'] for r in results])
        mean_ap = [float(all_ap[all_gt[:, s] > 0, s].mean()) if (all_gt[:, s] > 0).any() else 0.0 for s in range(num_scales)]
    if print_summary:
        print_map_summary(mean_ap, results, dataset)
    return mean_ap, results


VID_CLASSES = ('airplane', 'antelope', 'bear', 'bicycle', 'bird', 'bus', 'car', 'cattle', 'dog', 'domestic_cat', 'elephant', 'fox',
               'giant_panda', 'hamster', 'horse', 'lion', 'lizard', 'monkey', 'motorcycle', 'rabbit', 'red_panda', 'sheep', 'snake',
               'squirrel', 'tiger', 'train', 'turtle', 'watercraft', 'whale', 'zebra')   # tools/vid_eval.py:33-40


def print_map_summary(mean_ap, results, dataset=None):
    """Plain-text table of mean_ap.py:588-640's columns (class, gts, dets, recall, ap); no terminaltables dependency."""
    names = list(dataset) if isinstance(dataset, (list, tuple)) else [str(i + 1) for i in range(len(results))]
    rows = ['%-16s %8s %8s %8s %8s' % ('class', 'gts', 'dets', 'recall', 'ap')]
    for name, r in zip(names, results):
        rec = np.atleast_2d(r['recall'])
        rows.append('%-16s %8s %8d %8.3f %8.3f' % (name, np.sum(r['num_gts']), r['num_dets'], float(rec[0, -1]) if rec.size else 0.0,
                                                   float(np.atleast_1d(r['ap'])[0])))
    rows.append('%-16s %35.3f' % ('mAP', float(np.atleast_1d(mean_ap)[0])))
    print('\n'.join(rows))


def vid_eval(det_results, annotations, iou_thr=0.5, print_summary=True):
    """tools/vid_eval.py:11-52 without the dataset object: annotations = per image dict(bboxes, labels[, bboxes_ignore,
    labels_ignore]) (the reference's `dataset.get_ann_info(i)`)."""
    gt_bboxes, gt_labels, gt_ignore = [], [], []
    for ann in annotations:
        boxes, labels = np.asarray(ann['bboxes'], dtype=np.float32).reshape(-1, 4), np.asarray(ann['labels'])
        if 'bboxes_ignore' in ann:
            extra = np.asarray(ann['bboxes_ignore'], dtype=np.float32).reshape(-1, 4)
            gt_ignore.append(np.concatenate([np.zeros(boxes.shape[0], dtype=bool), np.ones(extra.shape[0], dtype=bool)]))
            boxes = np.vstack([boxes, extra])
            labels = np.concatenate([labels, np.asarray(ann['labels_ignore'])])
        gt_bboxes.append(boxes)
        gt_labels.append(labels)
    return eval_map(det_results, gt_bboxes, gt_labels, gt_ignore=gt_ignore or None, scale_ranges=None, iou_thr=iou_thr,
                    dataset=VID_CLASSES, print_summary=print_summary)
