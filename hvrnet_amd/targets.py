"""Training-target generation on the device: the reference's assigner / sampler / target API over the HIP kernels.

Mirror of `mmdet/core/bbox` + `mmdet/core/anchor/anchor_target.py` for the pieces SelsaRCNN.forward_train touches
(selsa_rcnn.py:126-232) -- same class names, constructor arguments and return objects, so a training config's
`train_cfg.rpn` / `train_cfg.rcnn` dicts build the same objects:

  MaxIoUAssigner.assign          max_iou_assigner.py:48-173   -> native.max_iou_assign   (hvr_max_iou_assign)
  RandomSampler.sample           base_sampler.py:32-78        -> native.sample_pos_neg   (hvr_sample_pos_neg)
  OHEMHNLSampler.get_ohem_weights ohem_hnl_sampler.py:86-113  -> native.sample_pos_neg on keys = -loss
  anchor_target                  anchor_target.py:7-155       -> assign + sample + native.box_targets(scatter)
  bbox_target                    bbox_target.py:7-62          -> native.box_targets

One deliberate difference: the random subset is not drawn by a host-side numpy shuffle (random_sampler.py:19-35, which
needs the candidate indices on the host).  Each box carries a random key and the `expected` smallest keys of a class win;
with uniform keys that is the same distribution, it needs no device-to-host copy, and a caller that passes `keys` gets a
reproducible (and, against the reference, replayable) choice.  Outside the implemented envelope the classes raise
NotImplementedError instead of approximating: ignore boxes (ignore_iof_thr > 0), gt_max_assign_all=False.
"""
import torch

from . import native


_arange1 = {}


def _one_based(k, device):
    """arange(1, k + 1) int64 on `device`, one tensor per (k, device): the gt rows' own assignments in add_gt_."""
    key = (int(k), str(device))
    if key not in _arange1:
        _arange1[key] = torch.arange(1, k + 1, dtype=torch.long, device=device)
    return _arange1[key]


class AssignResult(object):
    """assigners/assign_result.py:4-19.  (`max_overlaps` after add_gt_ is put together on first use: nothing on the training path
    reads it, and a frame-by-frame loop pays two launches per frame for it otherwise.)"""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self._max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels
        self._ones_in_front = 0

    @property
    def max_overlaps(self):
        if self._ones_in_front:
            self._max_overlaps = torch.cat([self._max_overlaps.new_ones(self._ones_in_front), self._max_overlaps])
            self._ones_in_front = 0
        return self._max_overlaps

    @max_overlaps.setter
    def max_overlaps(self, v):
        self._max_overlaps, self._ones_in_front = v, 0

    def add_gt_(self, gt_labels):
        self.gt_inds = torch.cat([_one_based(self.num_gts, self.gt_inds.device), self.gt_inds])
        self._ones_in_front += self.num_gts
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


class MaxIoUAssigner(object):
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, gpu_assign_thr=-1):
        if not gt_max_assign_all:
            raise NotImplementedError('gt_max_assign_all=False is not on the HIP path (the HVRNet configs keep the default)')
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.ignore_iof_thr = ignore_iof_thr
        self._ext = (None, None)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, valid=None):
        """-> AssignResult (gt_inds: -1 ignore, 0 background, g+1 assigned to gt g).  `valid` [n] restricts the boxes that
        take part (anchor_target_single's inside_flags) without compacting them."""
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            raise NotImplementedError('ignore boxes (ignore_iof_thr > 0) are not on the HIP path; both configs set -1')
        gt_inds, max_ov = native.max_iou_assign(bboxes, gt_bboxes, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou, valid)
        labels = None
        if gt_labels is not None:   # max_iou_assigner.py:156-163: label of the assigned gt, 0 for background / ignored boxes
            if self._ext[0] is not gt_labels:     # [0 | gt_labels], kept while the same label tensor comes back (the frames of a video)
                self._ext = (gt_labels, torch.cat([gt_labels.new_zeros(1), gt_labels]))
            labels = self._ext[1][gt_inds.clamp(min=0)]
        return AssignResult(gt_bboxes.shape[0], gt_inds, max_ov, labels)


class SamplingResult(object):
    """samplers/sampling_result.py:4-24.  `inds` / `counts` are the kernel's raw outputs (positives first); the attribute
    views below need the two counts on the host (one copy, made on first use)."""

    def __init__(self, inds, counts, bboxes, gt_bboxes, assign_result, gt_flags):
        """gt_flags: the uint8 flag vector, or the number of ground-truth rows in front (the vector is then built on first use)."""
        self.inds, self.counts, self.all_bboxes, self.gt_bboxes = inds, counts, bboxes, gt_bboxes
        self.assign_result, self._gt_flags, self.num_gts = assign_result, gt_flags, gt_bboxes.shape[0]
        self._n = None
        self._bboxes = None

    @property
    def gt_flags(self):
        if isinstance(self._gt_flags, int):
            f = self.all_bboxes.new_zeros((self.all_bboxes.shape[0],), dtype=torch.uint8)
            f[:self._gt_flags] = 1
            self._gt_flags = f
        return self._gt_flags

    def _counts(self):
        if self._n is None:
            self._n = tuple(int(v) for v in self.counts.tolist())
        return self._n

    @staticmethod
    def resolve(results):
        """The (#pos, #neg) pairs of several results in ONE host read (each result's first attribute access otherwise makes its own:
        a device round trip per frame in a loop that assigns and samples frame by frame)."""
        todo = [r for r in results if r._n is None]
        if len(todo) > 1:
            for r, row in zip(todo, torch.stack([r.counts for r in todo]).tolist()):
                r._n = (int(row[0]), int(row[1]))
        return results

    pos_inds = property(lambda self: self.inds[:self._counts()[0]])
    neg_inds = property(lambda self: self.inds[self._counts()[0]:sum(self._counts())])
    pos_bboxes = property(lambda self: self.all_bboxes[self.pos_inds])
    neg_bboxes = property(lambda self: self.all_bboxes[self.neg_inds])
    pos_is_gt = property(lambda self: self.gt_flags[self.pos_inds])
    pos_assigned_gt_inds = property(lambda self: self.assign_result.gt_inds[self.pos_inds] - 1)
    pos_gt_bboxes = property(lambda self: self.gt_bboxes[self.pos_assigned_gt_inds, :])
    pos_gt_labels = property(lambda self: None if self.assign_result.labels is None else self.assign_result.labels[self.pos_inds])
    @property
    def bboxes(self):
        """cat(pos_bboxes, neg_bboxes): the sampled boxes in the kernel's order (gathered once)."""
        if self._bboxes is None:
            self._bboxes = self.all_bboxes[self.inds[:sum(self._counts())]]
        return self._bboxes


class BaseSampler(object):
    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub, self.add_gt_as_proposals = num, pos_fraction, neg_pos_ub, add_gt_as_proposals


class RandomSampler(BaseSampler):
    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, keys=None, generator=None, **kwargs):
        """base_sampler.py:32-78.  keys: f32 [n (+ num_gts with add_gt_as_proposals)], drawn uniformly when None."""
        bboxes = bboxes[:, :4]
        gt_flags = 0                                   # (the flag vector itself is built when somebody reads it)
        if self.add_gt_as_proposals:
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = int(gt_bboxes.shape[0])
        n = assign_result.gt_inds.numel()
        if keys is None:
            keys = torch.rand(n, device=bboxes.device, generator=generator)
        assert keys.numel() == n, 'one key per box (ground-truth rows first when they are added as proposals)'
        inds, counts = native.sample_pos_neg(assign_result.gt_inds.contiguous(), (keys if keys.dtype == torch.float32 else keys.float()).contiguous(), self.num,
                                             int(self.num * self.pos_fraction), self.neg_pos_ub)
        return SamplingResult(inds, counts, bboxes.contiguous(), gt_bboxes, assign_result, gt_flags)


class PseudoSampler(object):
    def __init__(self, **kwargs):
        raise NotImplementedError('PseudoSampler (sampling=False heads: RetinaNet-style) is outside the HVR training path')


class OHEMHNLSampler(BaseSampler):
    """ohem_hnl_sampler.py: the second, loss-ranked stage of the SELSA config's sampler list (selsa_rcnn.py:207-222)."""

    def __init__(self, num, pos_fraction, context=None, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        super(OHEMHNLSampler, self).__init__(num, pos_fraction, neg_pos_ub, add_gt_as_proposals)

    def select(self, labels, loss):
        """-> (inds int64 [num]: hardest positives then hardest negatives, counts int32 [2]); no host copy."""
        return native.sample_pos_neg(labels.contiguous(), (-loss).float().contiguous(), self.num, int(self.num * self.pos_fraction),
                                     self.neg_pos_ub)

    def get_ohem_weights(self, labels, label_weights, bbox_weights, loss):
        """ohem_hnl_sampler.py:86-113 -> (label_weights, bbox_weights, pos_inds, neg_inds); weights rewritten in place."""
        inds, counts = self.select(labels, loss)
        np_, nn_ = (int(v) for v in counts.tolist())
        pos_inds, neg_inds = inds[:np_], inds[np_:np_ + nn_]
        label_weights[...] = 0.
        label_weights[pos_inds] = 1.0
        label_weights[neg_inds] = 1.0
        bbox_weights[...] = 0
        bbox_weights[pos_inds] = 1.0
        return label_weights, bbox_weights, pos_inds, neg_inds


_ASSIGNERS = dict(MaxIoUAssigner=MaxIoUAssigner)
_SAMPLERS = dict(RandomSampler=RandomSampler, OHEMHNLSampler=OHEMHNLSampler, PseudoSampler=PseudoSampler)


def _from_dict(cfg, table, what, **default_args):
    args = dict(cfg)
    name = args.pop('type')
    if name not in table:
        raise NotImplementedError('%s %r is not part of the HVR training path (have: %s)' % (what, name, ', '.join(sorted(table))))
    for k, v in default_args.items():
        args.setdefault(k, v)
    return table[name](**args)


def build_assigner(cfg, **kwargs):
    """assign_sampling.py:6-13."""
    return cfg if isinstance(cfg, MaxIoUAssigner) else _from_dict(cfg, _ASSIGNERS, 'assigner', **kwargs)


def build_sampler(cfg, **kwargs):
    """assign_sampling.py:16-30: a dict builds one sampler, a list of dicts a list (the SELSA config's [Random, OHEMHNL])."""
    if isinstance(cfg, BaseSampler):
        return cfg
    if isinstance(cfg, (list, tuple)):
        return [_from_dict(c, _SAMPLERS, 'sampler', **kwargs) for c in cfg]
    return _from_dict(cfg, _SAMPLERS, 'sampler', **kwargs)


def assign_and_sample(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels, cfg, keys=None, valid=None):
    """assign_sampling.py:33-40."""
    assign_result = build_assigner(cfg['assigner']).assign(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels, valid=valid)
    sampling_result = build_sampler(cfg['sampler']).sample(assign_result, bboxes, gt_bboxes, gt_labels, keys=keys)
    return assign_result, sampling_result


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """anchor_target.py:158-170 -> uint8 [N]."""
    img_h, img_w = img_shape[:2]
    if allowed_border < 0:
        return valid_flags
    inside = ((flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border)
              & (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border))
    return inside.to(torch.uint8) if valid_flags is None else (valid_flags.to(torch.uint8) & inside.to(torch.uint8))


def anchor_target_single(flat_anchors, valid_flags, gt_bboxes, img_meta, target_means, target_stds, cfg, keys=None, generator=None):
    """anchor_target_single (anchor_target.py:92-155) with sampling=True, gt_labels=None, unmap_outputs=True, entirely on the
    device: -> (labels, label_weights, bbox_targets, bbox_weights) over ALL anchors and counts int32 [2] = (#pos, #neg)
    sampled.  The anchors outside the image are masked instead of compacted, so nothing is unmapped afterwards."""
    inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2], cfg['allowed_border'])
    assign_result = build_assigner(cfg['assigner']).assign(flat_anchors, gt_bboxes, None, None, valid=inside)
    sampler = build_sampler(cfg['sampler'])
    if not isinstance(sampler, RandomSampler) or sampler.add_gt_as_proposals:
        raise NotImplementedError('anchor targets use a RandomSampler with add_gt_as_proposals=False (train_cfg.rpn.sampler)')
    n = flat_anchors.shape[0]
    if keys is None:
        keys = torch.rand(n, device=flat_anchors.device, generator=generator)
    inds, counts = native.sample_pos_neg(assign_result.gt_inds, keys.float().contiguous(), sampler.num,
                                         int(sampler.num * sampler.pos_fraction), sampler.neg_pos_ub)
    out = native.box_targets(flat_anchors, gt_bboxes, None, assign_result.gt_inds, inds, counts, target_means, target_stds,
                             cfg['pos_weight'], scatter=True)
    return out + (counts,)


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds, cfg, gt_bboxes_ignore_list=None,
                  gt_labels_list=None, label_channels=1, sampling=True, unmap_outputs=True, keys_list=None):
    """anchor_target (anchor_target.py:7-76) for single-level anchors: -> (labels_list, label_weights_list,
    bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg), the *_list entries indexed by level ([num_imgs, N])."""
    if not sampling or gt_labels_list is not None and any(g is not None for g in gt_labels_list) or not unmap_outputs:
        raise NotImplementedError('RPN-style targets only: sampling=True, no gt_labels, unmap_outputs=True')
    if any(len(a) != 1 for a in anchor_list):
        raise NotImplementedError('single-level anchors only (anchor_strides=[16])')
    per_img = []
    for i, meta in enumerate(img_metas):
        valid = valid_flag_list[i][0] if valid_flag_list is not None else None
        per_img.append(anchor_target_single(anchor_list[i][0], valid, gt_bboxes_list[i], meta, target_means, target_stds, cfg,
                                            keys=None if keys_list is None else keys_list[i]))
    counts = torch.stack([p[4] for p in per_img]).clamp(min=1).sum(0).tolist()   # anchor_target.py:66-67
    return ([torch.stack([p[0] for p in per_img])], [torch.stack([p[1] for p in per_img])], [torch.stack([p[2] for p in per_img])],
            [torch.stack([p[3] for p in per_img])], int(counts[0]), int(counts[1]))


def bbox_target(sampling_results, gt_bboxes_list, gt_labels_list, cfg, reg_classes=1, target_means=(.0, .0, .0, .0),
                target_stds=(1.0, 1.0, 1.0, 1.0), concat=True):
    """BBoxHead.get_target + bbox_target (bbox_head.py:80-96, bbox_target.py:7-62) from SamplingResults: rows in
    cat(pos, neg) order -> (labels, label_weights, bbox_targets, bbox_weights)."""
    if reg_classes != 1:
        raise NotImplementedError('class-agnostic regression only (reg_class_agnostic=True in both configs)')
    outs = []
    for res, gt_b, gt_l in zip(sampling_results, gt_bboxes_list, gt_labels_list):
        n = sum(res._counts())
        full = native.box_targets(res.all_bboxes, gt_b, gt_l, res.assign_result.gt_inds, res.inds, res.counts, target_means,
                                  target_stds, cfg['pos_weight'], scatter=False)
        outs.append(tuple(t[:n] for t in full))
    if not concat:
        return tuple(list(x) for x in zip(*outs))
    return tuple(torch.cat(x, 0) for x in zip(*outs))
