"""Box utilities with the reference's names and argument meaning (mmdet/core).

  AnchorGenerator      mmdet/core/anchor/anchor_generator.py:4-98   (host; constant per shape)
  delta2bbox           mmdet/core/bbox/transforms.py:34-111          (device kernel)
  bbox2roi             mmdet/core/bbox/transforms.py:149-168         (memory plumbing)
  bbox2result          mmdet/core/bbox/transforms.py:181-199         (D2H + split per class)
  multiclass_nms       mmdet/core/post_processing/bbox_nms.py:6-66   (device kernel)
"""
import numpy as np
import torch

from . import native


class AnchorGenerator(object):
    """Base anchors are 12 numbers: they are computed on the host and handed to the RPN kernel,
    which derives every grid anchor on the fly (the reference rebuilds a [28 728, 4] tensor per call,
    anchor_head.py:258-263)."""

    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.Tensor(scales)
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        w = h = self.base_size
        x_ctr, y_ctr = (0.5 * (w - 1), 0.5 * (h - 1)) if self.ctr is None else self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * self.scales[None, :]).view(-1)
        else:
            ws = (w * self.scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * self.scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)],
                           dim=-1).round()

    def grid_anchors(self, featmap_size, stride=16, device='cuda'):
        """(y, x, anchor)-ordered grid; kept for API parity (the fused RPN kernel does not need it)."""
        base = self.base_anchors.to(device)
        feat_h, feat_w = featmap_size
        sx = torch.arange(0, feat_w, device=device) * stride
        sy = torch.arange(0, feat_h, device=device) * stride
        xx = sx.repeat(len(sy))
        yy = sy.view(-1, 1).repeat(1, len(sx)).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 4)


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None, wh_ratio_clip=16 / 1000):
    """rois [N,4], deltas [N,4] (class-agnostic) -> boxes [N,4] on the device."""
    assert deltas.shape[1] == 4, 'only class-agnostic deltas are on the HVR hot path'
    n = rois.shape[0]
    rois5 = torch.cat([rois.new_zeros((n, 1)), rois[:, :4].float()], dim=1)
    logits = torch.cat([deltas.float().new_zeros((n, 1)), deltas.float()], dim=1).contiguous()
    _, boxes = native.det_decode(logits, 0, 1, 1, rois5, means, stds, max_shape, 0.0, wh_ratio_clip)
    return boxes


def bbox2roi(bbox_list):
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            img_inds = bboxes.new_full((bboxes.size(0), 1), img_id)
            rois_list.append(torch.cat([img_inds, bboxes[:, :4]], dim=-1))
        else:
            rois_list.append(bboxes.new_zeros((0, 5)))
    return torch.cat(rois_list, 0)


def bbox2result(bboxes, labels, num_classes):
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes - 1)]
    bboxes = bboxes.cpu().numpy()
    labels = labels.cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes - 1)]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """Returns (bboxes [k,5], labels [k]) like the reference; one host read of k."""
    if score_factors is not None or multi_bboxes.shape[1] != 4:
        raise NotImplementedError('class-specific boxes / score_factors are outside the HVR hot path')
    nms_cfg_ = dict(nms_cfg)
    if nms_cfg_.pop('type', 'nms') != 'nms':
        raise NotImplementedError('only greedy nms is on the HVR hot path (configs use type="nms")')
    iou_thr = nms_cfg_.pop('iou_thr')
    R, nfg = multi_bboxes.shape[0], multi_scores.shape[1] - 1
    boxes, scores = multi_bboxes.float(), multi_scores.float()
    if max_num < 0:
        # The reference's default max_num = -1 is not "no cap": `bboxes.shape[0] > max_num` is always true, so the survivors are
        # sorted by score and `inds[:max_num]` drops the last |max_num| of them (bbox_nms.py:55-59).  Reproduced as written:
        # count the survivors first (uncapped pass, class order), then cut to count + max_num in score order.
        dets, labels, n = native.multiclass_nms(boxes, scores, score_thr, iou_thr, max(R * nfg, 1))
        k = int(n.item()) + int(max_num)
        if k <= 0:
            return dets[:0], labels[:0]
        dets, labels, n = native.multiclass_nms(boxes, scores, score_thr, iou_thr, k)
        return dets[:k], labels[:k]
    dets, labels, n = native.multiclass_nms(boxes, scores, score_thr, iou_thr, max(int(max_num), 1))
    k = int(n.item())
    return dets[:k], labels[:k]
