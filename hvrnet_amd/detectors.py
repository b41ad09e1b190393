"""Video detectors: orchestration of one window, same public surface as the reference.

  BaseDetector.forward dispatch        mmdet/models/detectors/base.py:106-132
  TwoStageDetector ctor / extract_feat mmdet/models/detectors/two_stage.py:20-97
  SelsaRCNN                            mmdet/models/detectors/selsa_rcnn.py:13-83, 281-338
  HNMBRCNN                             mmdet/models/detectors/hnmb_rcnn.py:19-48, 195-222, 571-613

What differs from the reference is HOW a window executes, not what it computes:
  * RPN proposals for all T frames come from one device pipeline (where the reference synchronises
    inside every per-frame NMS),
  * RoIAlign runs once over all frames (batch index = frame) instead of T launches on split maps
    (hnmb_rcnn.py:596-598),
  * the read-out (softmax, decode, 30-class NMS) stays on the device,
  * a window has ONE host synchronisation, at its end (`PendingWindow.result`): the per-frame proposal
    counts are not read mid-window -- the window runs on the assumption that every frame kept `nms_post`
    proposals (the normal case) and is re-run through the exact ragged path if the counts, read together
    with the detections, say otherwise.  `forward_feat(..., defer=True)` hands the PendingWindow to the
    caller, which can enqueue the next window before collecting this one.
The reference dump's defects are implemented as intended: SelsaRCNN takes
`[:2]` of the head's 3-tuple (selsa_rcnn.py:306), `collections.Sequence` -> `collections.abc`.
Training: SelsaRCNN.forward_train (selsa_rcnn.py:85-279) and HNMBRCNN.forward_train (hnmb_rcnn.py:224-434; its triplet term
through a documented stand-in) run on the HIP path end to end.
"""
import collections.abc
import os

import numpy as np
import torch
import torch.nn as nn

from . import registry
from .box_ops import bbox2result, bbox2roi
from .registry import DETECTORS


class BaseDetector(nn.Module):

    def __init__(self):
        super(BaseDetector, self).__init__()
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_shared_head(self):
        return hasattr(self, 'shared_head') and self.shared_head is not None

    @property
    def with_bbox(self):
        return hasattr(self, 'bbox_head') and self.bbox_head is not None

    @property
    def with_mask(self):
        return False

    def forward(self, img, img_meta, return_loss=True, backbone_feat=False, forward_feat=False, **kwargs):
        """base.py:106-132: backbone_feat -> C4 features; forward_feat -> one window; else train / test."""
        if backbone_feat:
            if isinstance(img, list):
                assert len(img) == len(img_meta), 'img and img_meta should have same number!'
                return [self.extract_feat(im_) for im_ in img]
            return self.extract_feat(img)
        if forward_feat:
            if isinstance(img_meta[0], list) and len(img_meta[0]) != 0:
                raise NotImplementedError('multi-scale test-time augmentation (forward_feat_aug) is outside the hot path')
            return self.forward_feat(img_meta=img_meta, **kwargs)
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)

    def forward_train(self, img, img_meta, **kwargs):
        raise NotImplementedError('%s has no training step on the HIP path (SelsaRCNN and HNMBRCNN do)' % type(self).__name__)

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError('{} must be a list, but got {}'.format(name, type(var)))
        if len(imgs) != 1:
            raise NotImplementedError('test-time augmentation is outside the hot path')
        assert imgs[0].size(0) == 1
        return self.simple_test(imgs[0], img_metas[0], **kwargs)


class TwoStageDetector(BaseDetector):

    def __init__(self, backbone, neck=None, shared_head=None, rpn_head=None, bbox_roi_extractor=None, bbox_head=None,
                 mask_roi_extractor=None, mask_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super(TwoStageDetector, self).__init__()
        if neck is not None or mask_head is not None or mask_roi_extractor is not None:
            raise NotImplementedError('necks / mask heads are outside the HVR hot path')
        self.backbone = registry.build_backbone(backbone)
        if shared_head is not None:
            self.shared_head = registry.build_shared_head(shared_head)
        if rpn_head is not None:
            self.rpn_head = registry.build_head(rpn_head)
        self.feat_from_shared_head = False
        if bbox_head is not None:
            bbox_roi_extractor = dict(bbox_roi_extractor)
            self.feat_from_shared_head = bbox_roi_extractor.pop('feat_from_shared_head', False)  # two_stage.py:46
            self.bbox_roi_extractor = registry.build_roi_extractor(bbox_roi_extractor)
            self.bbox_head = registry.build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    @property
    def with_rpn(self):
        return hasattr(self, 'rpn_head') and self.rpn_head is not None

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_shared_head:
            self.shared_head.init_weights(pretrained=pretrained)
        if self.with_rpn:
            self.rpn_head.init_weights()
        if self.with_bbox:
            self.bbox_roi_extractor.init_weights()
            self.bbox_head.init_weights()

    def extract_feat(self, img):
        """backbone on a batch of frames.  The kernels address their operands with 32-bit byte offsets (tensors below 2 GiB): a batch
        whose largest map would pass that -- W clips of 15 frames in a 4-byte mode: layer 1's output is 39 MB per 608 x 1008 frame, the
        f32 mode's stem patch matrix 98 MB -- goes through in equal chunks, one after the other on the same stream, each writing its
        slice of ONE C4 map (frames are independent through the backbone: the chunking changes no result).  (Rounds 1-4 could also
        cut a batch into groups on several HIP streams; it stopped paying in round 2 -- 114.6 vs 114.7 us per layer-3 block -- and was
        removed in round 5.)"""
        bb = self.backbone
        shape_of = getattr(bb, 'out_shape_nhwc', None)
        if img.is_cuda and img.dim() == 4 and shape_of is not None and shape_of(img.shape[0], img.shape[2], img.shape[3]) is not None:
            es = 2 if bb.compute_dtype in (torch.bfloat16, torch.float16) else 4
            H, Wd = img.shape[2], img.shape[3]
            per_frame = (H // 4) * (Wd // 4) * 256 * es
            if bb.compute_dtype == torch.float32:
                per_frame = max(per_frame, (H // 2) * (Wd // 2) * 192 * es)   # the generic stem's patch matrix
            max_b = max(1, (2 ** 31 - 2 ** 20) // per_frame)
            B = img.shape[0]
            if B > max_b:
                n = -(-B // max_b)
                whole = torch.empty(shape_of(B, H, Wd), dtype=bb.compute_dtype, device=img.device)
                bounds = [round(i * B / n) for i in range(n + 1)]
                for a, b in zip(bounds[:-1], bounds[1:]):
                    bb(img[a:b], out=whole[a:b])
                return (whole.permute(0, 3, 1, 2),)
        return self.backbone(img)

    def simple_test_rpn(self, x, img_meta, rpn_test_cfg):
        """test_mixins.py:9-13."""
        rpn_outs = self.rpn_head(x)
        return self.rpn_head.get_bboxes(*(rpn_outs + (img_meta, rpn_test_cfg)))


def _sampled_rois(results):
    """bbox2roi of the frames' sampled boxes (transforms.py:114-136): [sum n_i, 5] = (frame index, box).  The frame-index column comes from
    one repeat_interleave when every frame kept the same number of boxes (the sampler's `num`: the usual case), the boxes from one cat."""
    boxes = [r.bboxes for r in results]
    n = [int(b.shape[0]) for b in boxes]
    allb = torch.cat(boxes, 0)
    if len(set(n)) == 1 and n[0] > 0:
        col = torch.arange(len(n), device=allb.device, dtype=allb.dtype).repeat_interleave(n[0])
    else:
        col = torch.cat([allb.new_full((k,), float(i)) for i, k in enumerate(n)], 0)
    return torch.cat([col[:, None], allb], 1), n


class PendingWindow(object):
    """A window whose kernels are enqueued but whose results have not been read.  `result()` is the window's single
    host synchronisation: detections, labels, counts and the per-frame proposal counts arrive in one batch of
    asynchronous copies into pinned memory behind one event."""

    def __init__(self, branches, counts_dev, full_count, num_classes, exact, single=False):
        # branches: list of (dets [max,5], labels [max], n [1]) device tensors; exact: () -> results, the ragged re-run;
        # single: the detector has one read-out branch and returns its result bare (SelsaRCNN)
        self._num_classes, self._exact, self._full, self._single = num_classes, exact, full_count, single
        self._host = [[self._pinned(t) for t in b] for b in branches]
        self._counts = self._pinned(counts_dev) if counts_dev is not None else None
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(branches[0][0].device))
        self._result = None
        self.respeculated = False  # True once result() had to take the exact path

    @staticmethod
    def _pinned(t):
        return torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True)

    def result(self):
        if self._result is None:
            self._event.synchronize()
            if self._counts is not None and any(int(c) != self._full for c in self._counts.tolist()):
                self.respeculated = True
                self._result = self._exact()
            else:
                out = []
                for dets, labels, n in self._host:
                    k = int(n[0])
                    out.append(bbox2result(dets[:k], labels[:k], self._num_classes))
                self._result = out[0] if self._single else out
            self._host = self._counts = self._exact = None
        return self._result


class _WindowDetector(TwoStageDetector):
    """forward_feat / simple_test_bboxes shared by SelsaRCNN and HNMBRCNN."""

    def get_roi_feat(self, x, rois):
        if not self.feat_from_shared_head:
            raise NotImplementedError('per-RoI shared head (feat_from_shared_head=False) is outside the hot path')
        return self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)

    rpn_side_stream = os.environ.get('HVR_RPN_SIDE', '1') != '0'

    def _side_stream(self, device):
        streams = self.__dict__.setdefault('_side_streams', {})
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)  # one RPN side stream per main stream
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

    @staticmethod
    def _cat_frames(x):
        """torch.cat(deque, 0) that keeps the physical NHWC layout (hnmb_rcnn.py:200)."""
        if isinstance(x, torch.Tensor):
            return x
        assert isinstance(x, collections.abc.Sequence) and isinstance(x[0], torch.Tensor)
        if all(not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous() for t in x):
            return torch.cat([t.permute(0, 2, 3, 1) for t in x], dim=0).permute(0, 3, 1, 2)
        return torch.cat(tuple(x), dim=0)

    def window_tensors(self, x, img_meta, proposals=None, rescale=False, speculate=False, clips=1):
        """Runs one window up to the head outputs; returns a dict of device tensors (used by forward_feat and tests).
        speculate: do not read the proposal counts; build the RoIs as if every frame kept all `nms_post` proposals and
        return the counts tensor as `counts_dev` for the caller to check.
        clips = W > 1 (speculative only): x / img_meta hold W independent clips of T frames back to back; res5, the RPN, its
        proposals and RoIAlign take all W * T frames as one batch; `cur_range` addresses the key rows inside every clip and
        `key_rois` is a list of W tensors."""
        xc = self._cat_frames(x)
        assert xc.shape[0] == len(img_meta) and len(img_meta) % clips == 0
        if clips > 1 and not (speculate and proposals is None):
            raise NotImplementedError('several clips per call run the speculative path (every frame keeps nms_post proposals); run the '
                                      'exact path one clip at a time')
        feats, proposal_list, rois, counts_dev, counts_h = self._c5_and_rois(xc, img_meta, proposals, speculate)
        key = self.key_dim
        start = int(np.sum(counts_h[:key]))
        cur_range = dict(start=start, length=int(counts_h[key]))
        roi_feats = self.get_roi_feat(feats, rois.contiguous())
        per_clip = rois.shape[0] // clips
        key_rois = []
        for w in range(clips):
            kr = rois[w * per_clip + start:w * per_clip + start + cur_range['length']].clone()
            kr[:, 0] = 0  # the reference's rois carry batch index 0 (hnmb_rcnn.py:582-584)
            key_rois.append(kr)
        return dict(c5=feats[0], proposals=proposal_list, rois=rois, roi_feats=roi_feats, cur_range=cur_range,
                    key_rois=key_rois[0] if clips == 1 else key_rois,
                    counts_dev=counts_dev, full_count=int(counts_h[0]) if counts_dev is not None else None)

    def _c5_and_rois(self, xc, img_meta, proposals=None, speculate=False):
        """res5 and the RPN proposals of the frames in xc -> (feats, proposal_list, rois [n,5], counts_dev, counts_h)."""
        if proposals is None:
            # The RPN branch (3x3 conv, heads, select / NMS: a few latency-bound workgroups) and res5 both
            # depend only on C4: run the RPN on a second HIP stream underneath res5.
            if self.rpn_side_stream:
                main, side = torch.cuda.current_stream(xc.device), self._side_stream(xc.device)
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    rpn_outs = self.rpn_head([xc])
                    props, counts = self.rpn_head.get_bboxes_batched(rpn_outs[0], rpn_outs[1], img_meta, self.test_cfg.rpn)
                    done = torch.cuda.Event()
                    done.record(side)
                feats = [self.shared_head(xc)] if self.feat_from_shared_head else [xc]
                main.wait_event(done)
                props.record_stream(main)
                counts.record_stream(main)
            else:   # one chain (HVR_RPN_SIDE=0): no fork for a captured graph's replay to spread over further hardware queues
                rpn_outs = self.rpn_head([xc])
                props, counts = self.rpn_head.get_bboxes_batched(rpn_outs[0], rpn_outs[1], img_meta, self.test_cfg.rpn)
                feats = [self.shared_head(xc)] if self.feat_from_shared_head else [xc]
            T, mx = props.shape[0], props.shape[1]
            counts_dev = counts if speculate else None
            counts_h = [mx] * T if speculate else counts.tolist()  # exact path: a mid-window host read of T integers
            frame = torch.arange(T, device=props.device, dtype=props.dtype).view(T, 1, 1).expand(T, mx, 1)
            rois = torch.cat([frame, props[..., :4]], dim=-1)
            if all(c == mx for c in counts_h):
                rois = rois.reshape(T * mx, 5)
            else:
                rois = torch.cat([rois[i, :c] for i, c in enumerate(counts_h)], dim=0)
            proposal_list = [props[i, :c] for i, c in enumerate(counts_h)]
        else:
            feats = [self.shared_head(xc)] if self.feat_from_shared_head else [xc]
            proposal_list = list(proposals)
            counts_dev = None
            counts_h = [p.shape[0] for p in proposal_list]
            rois = bbox2roi([p for p in proposal_list])  # batch index = frame index
        return feats, proposal_list, rois, counts_dev, counts_h

    # ---- per-frame cache -------------------------------------------------------------------
    # Everything up to the fc_new_1 rows is a function of ONE frame: res5, the RPN and its proposals, RoIAlign and
    # fc_new_1 (hnmb_rcnn.py:195-222 recomputes them for all T frames of every window).  `frame_tensors` computes them
    # once per frame, `forward_feat_frames` assembles a window from T such entries and runs only what mixes frames
    # (the relation stages and the read-out).  Bit-identical to `forward_feat` on the same frames: every kernel on the
    # per-frame part accumulates an output element in the same order whatever the batch size.
    def frame_tensors(self, c4, img_meta):
        """c4: the frame's backbone map [1,1024,h,w]; -> dict(props [mx,5], count [1] int32 device, f1 [mx,1024]).
        The proposal count is not read here: a short frame (count < mx) is detected when the window is read out."""
        feats, proposal_list, rois, counts_dev, _ = self._c5_and_rois(c4, [img_meta], None, speculate=True)
        f1 = self.bbox_head.fc1_rows(self.get_roi_feat(feats, rois.contiguous()))
        return dict(props=proposal_list[0], count=counts_dev, f1=f1, meta=img_meta)

    def frames_tensors(self, c4, img_metas):
        """frame_tensors for a BATCH of B frames (frames are independent up to fc_new_1: one launch sequence for all of them):
        -> dict(props [B,mx,5], count [B] int32 device, f1 [B * mx, 1024]); frame i's rows are f1[i * mx:(i + 1) * mx]."""
        feats, proposal_list, rois, counts_dev, _ = self._c5_and_rois(c4, list(img_metas), None, speculate=True)
        f1 = self.bbox_head.fc1_rows(self.get_roi_feat(feats, rois.contiguous()))
        return dict(props=torch.stack(proposal_list, 0), count=counts_dev, f1=f1, meta=img_metas[0])

    def _frames_window(self, entries):
        mx = entries[0]['props'].shape[0]
        key = self.key_dim
        f1 = torch.cat([e['f1'] for e in entries], dim=0)
        cur_range = dict(start=key * mx, length=mx)
        key_rois = torch.cat([entries[key]['props'].new_zeros((mx, 1)), entries[key]['props'][:, :4]], dim=1)
        counts_dev = torch.cat([e['count'] for e in entries])
        return f1, cur_range, key_rois, counts_dev, mx

    # ---- device-only forms (no host read, no PendingWindow): what graphs.py captures into a hipGraph ----
    def _head_branches(self, feats, from_f1, cur_range, key_rois, meta0, rescale, clips=1):
        """relation head + read-out -> list of (dets [max,5], labels [max], n [1]) device tensors, one per output branch; with
        clips = W > 1 (key_rois a list of W tensors, meta0 a list of W metas) a list of W such lists."""
        hvr = type(self).__name__ == 'HNMBRCNN'
        if hvr:
            head = self.bbox_head.forward_from_f1 if from_f1 else self.bbox_head.forward_test
            cls_score, bbox_pred = head(feats, [cur_range], key_dim=self.key_dim, all_res=False, clips=clips)
        else:
            head = self.bbox_head.forward_from_f1 if from_f1 else self.bbox_head
            cls_score, bbox_pred = head(feats, cur_range, key_dim=self.key_dim, all_res=False, clips=clips)[:2]
        l = int(cur_range['length'])
        out = []
        for w in range(clips):
            rows = slice(w * l, (w + 1) * l)
            kr = key_rois if clips == 1 else key_rois[w]
            meta = meta0 if clips == 1 else meta0[w]
            if hvr:
                cs, bp = ([c[rows] for c in cls_score], [b[rows] for b in bbox_pred]) if clips > 1 else (cls_score, bbox_pred)
                branches, _ = self.bbox_head.get_det_bboxes(kr, cs, bp, meta['img_shape'], meta['scale_factor'],
                                                            rescale=rescale, cfg=self.test_cfg.rcnn, defer=True)
                out.append(list(branches))
            else:
                cs, bp = (cls_score[rows], bbox_pred[rows]) if clips > 1 else (cls_score, bbox_pred)
                branch, _ = self.bbox_head.get_det_bboxes(kr, cs, bp, meta['img_shape'], meta['scale_factor'],
                                                          rescale=rescale, cfg=self.test_cfg.rcnn, defer=True)
                out.append([branch])
        return out[0] if clips == 1 else out

    def window_device_outputs(self, x, img_meta, rescale=False, clips=1):
        """One speculative window (every frame assumed to keep nms_post proposals) up to its device-side results:
        -> (branches, counts_dev [T] int32, full_count); the caller checks counts == full_count when it reads the results.
        clips = W > 1: W independent clips back to back in x / img_meta, every kernel up to the relation stages on all of them as one
        batch, the relation core per clip in grouped calls -> a list of W such triples."""
        w = self.window_tensors(x, img_meta, None, rescale, speculate=True, clips=clips)
        if clips == 1:
            return self._head_branches(w['roi_feats'], False, w['cur_range'], w['key_rois'], img_meta[0], rescale), w['counts_dev'], w['full_count']
        T = len(img_meta) // clips
        per = self._head_branches(w['roi_feats'], False, w['cur_range'], w['key_rois'], [img_meta[c * T] for c in range(clips)], rescale, clips=clips)
        return [(per[c], w['counts_dev'][c * T:(c + 1) * T], w['full_count']) for c in range(clips)]

    def forward_feat_clips(self, x, img_meta, clips, rescale=False, defer=False):
        """forward_feat for W = `clips` independent clips in ONE call: x [W * T, 1024, h, w] (or a sequence of C4 maps) and img_meta
        hold the clips' frames back to back.  The reference runs one clip at a time (tools/test.py:214-250); here every kernel up to
        the relation stages takes the W clips as one batch and the relation core runs per clip in grouped calls
        (hvr_relation_fwd_grouped).  -> a list of W results (PendingWindow objects with defer=True), clip w's = forward_feat on its own
        frames up to the association of the relation core's f32 sums.  A clip with a short frame (fewer than nms_post proposals) is
        re-run alone through the exact path when its result is read."""
        xc = self._cat_frames(x)
        T = len(img_meta) // clips
        outs = self.window_device_outputs(xc, img_meta, rescale=rescale, clips=clips)
        if clips == 1:
            outs = [outs]
        single = type(self).__name__ == 'SelsaRCNN'

        def exact(w):
            return lambda: self.forward_feat(xc[w * T:(w + 1) * T], img_meta[w * T:(w + 1) * T], None, rescale, speculate=False)
        pend = [PendingWindow(b if not single else b, c, full, self.bbox_head.num_classes, exact(w), single=single)
                for w, (b, c, full) in enumerate(outs)]
        return pend if defer else [p_.result() for p_ in pend]

    def head_device_outputs(self, f1, cur_range, key_rois, counts_dev, meta0, rescale=False):
        """The window part of the per-frame-cache loop on assembled rows: f1 [T * n, 1024] (fc_new_1 rows of the T frames in
        window order), key_rois [n, 5] -> (branches, counts_dev, n)."""
        return self._head_branches(f1, True, cur_range, key_rois, meta0, rescale), counts_dev, int(cur_range['length'])

    def simple_test_bboxes(self, x, img_meta, proposals, rcnn_test_cfg, rescale=False):
        raise NotImplementedError

    def simple_test(self, img, img_meta, proposals=None, rescale=False):
        """Clip-mode test as the reference's simple_test INTENDS it (hnmb_rcnn.py:615-638, selsa_rcnn.py:319-338): img
        [T,3,H,W] = the frames of one window -> extract_feat -> the window path.  As written the reference hands the C4 maps
        (1024 channels) to simple_test_bboxes, whose RoI features then have the wrong width for fc_new_1 when
        feat_from_shared_head=True (both configs), and HNMBRCNN passes its two-branch lists to bbox2result: the method cannot
        run there; tools/test.py never calls it (it drives backbone_feat / forward_feat).  This is forward_feat on the
        extracted features -- the clip mode bench.py times."""
        x = self.extract_feat(img)
        return self.forward_feat(x=x[0], img_meta=img_meta, proposals=proposals, rescale=rescale)


@DETECTORS.register_module
class SelsaRCNN(_WindowDetector):

    def __init__(self, backbone, rpn_head, bbox_roi_extractor, bbox_head, train_cfg, test_cfg, neck=None, shared_head=None,
                 pretrained=None, loss_frames=1):
        super(SelsaRCNN, self).__init__(backbone=backbone, neck=neck, shared_head=shared_head, rpn_head=rpn_head,
                                        bbox_roi_extractor=bbox_roi_extractor, bbox_head=bbox_head, train_cfg=train_cfg,
                                        test_cfg=test_cfg, pretrained=pretrained)
        if self.train_cfg is not None:
            self.key_dim = int(self.train_cfg.rcnn.key_dim)
        else:  # selsa_rcnn.py:39-42
            self.key_dim = int(self.test_cfg.relation_setup.frame_interval)
            self.bbox_head.t_dim = int(test_cfg.bbox_head.t_dim)
            self.bbox_head.sampler_num = int(test_cfg.bbox_head.sampler_num)

    def forward_feat(self, x=None, img_meta=None, proposals=None, rescale=False, defer=False, speculate=True):
        """One window -> 30 per-class [k,5] arrays (selsa_rcnn.py:281-338); defer=True -> PendingWindow."""
        w = self.window_tensors(x, img_meta, proposals, rescale, speculate=speculate and proposals is None)
        cls_score, bbox_pred = self.bbox_head(w['roi_feats'], w['cur_range'], key_dim=self.key_dim, all_res=False)[:2]
        branch, _ = self.bbox_head.get_det_bboxes(w['key_rois'], cls_score, bbox_pred, img_meta[0]['img_shape'],
                                                  img_meta[0]['scale_factor'], rescale=rescale, cfg=self.test_cfg.rcnn,
                                                  defer=True)
        pending = PendingWindow([branch], w['counts_dev'], w['full_count'], self.bbox_head.num_classes,
                                lambda: self.forward_feat(x, img_meta, proposals, rescale, speculate=False), single=True)
        return pending if defer else pending.result()


    def forward_train_sampled(self, img, rois, cur_range, labels, label_weights, bbox_targets, bbox_weights):
        """The RCNN half of forward_train (selsa_rcnn.py:85-279) on FIXED sampled RoIs: backbone on the T frames, res5,
        RoIAlign of every frame's sampled RoIs (rois [K,5] = (frame index, box), frames in order), the SELSA head and
        BBoxHead.loss on the key frame's targets (`bbox_head.get_target` output).  Every forward and backward kernel is
        HIP (f32; enable_training + set_compute_dtype(float32) first).  The assigner / sampler / RPN-loss steps that
        produce rois and targets in the reference are not part of this build yet.  -> dict(loss_cls, loss_bbox, acc, total)."""
        from . import ops
        c4 = self.backbone.forward_train_nhwc(img)
        c5 = self.shared_head.forward_train_nhwc(c4)
        feats = ops.roi_align(c5.permute(0, 3, 1, 2), rois, self.bbox_roi_extractor.roi_layers[0].out_size,
                              self.bbox_roi_extractor.roi_layers[0].spatial_scale, self.bbox_roi_extractor.roi_layers[0].sample_num)
        logits = self.bbox_head.forward_train(feats, cur_range)
        return self.bbox_head.loss_train(logits, labels, label_weights, bbox_targets, bbox_weights)

    def forward_train(self, img, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      keys=None, generator=None):
        """SelsaRCNN.forward_train (selsa_rcnn.py:85-279) with every forward and backward kernel on the HIP path (f32;
        enable_training + set_compute_dtype(float32) first): backbone on the T frames; RPN loss on the key frame only (:127-136);
        proposals of all frames (:138-141, no gradient); per frame MaxIoUAssigner + RandomSampler against the KEY frame's ground
        truth (:151-173); res5 + RoIAlign of the sampled boxes (:177-195); the SELSA head (:201); the key frame's targets
        (:204-206) and, when train_cfg.rcnn.sampler is a list, the loss-ranked second sampler and the loss on its rows (:207-232).
        img [T,3,H,W] with the key frame first (train_cfg.rcnn.key_dim = 0, as the reference's roi bookkeeping assumes, :177-182).
        keys: optional dict(rpn=f32 [H/16 * W/16 * A], rcnn=[f32 [num_gt + nms_post] per frame]) replacing the samplers' random
        draws.  -> dict(loss_rpn_cls [1-list], loss_rpn_bbox [1-list], loss_cls, loss_bbox, acc); summing the 'loss' entries
        and calling backward() trains (the reference's parse_losses does the same sum)."""
        from . import native, ops, targets as T, train_ops as TO
        if self.train_cfg is None:
            raise ValueError('this detector was built without train_cfg (hvrnet_amd.config.selsa_train_config, or the reference config file)')
        key = self.key_dim
        if key != 0:
            raise NotImplementedError('training keeps the key frame first (train_cfg.rcnn.key_dim = 0 in both configs)')
        if gt_bboxes_ignore is not None and any(g is not None and g.numel() for g in gt_bboxes_ignore):
            raise NotImplementedError('gt_bboxes_ignore is not on the HIP training path (ignore_iof_thr = -1 in both configs)')
        keys = keys or {}
        rcnn_cfg = self.train_cfg.rcnn
        c4 = self.backbone.forward_train_nhwc(img)
        losses = dict()
        if self.with_rpn:
            A = self.rpn_head.num_anchors
            o = self.rpn_head.forward_train_fused(c4)
            rl = self.rpn_head.loss_train(o[key], gt_bboxes[key], img_meta[key], self.train_cfg.rpn, keys=keys.get('rpn'),
                                          generator=generator)
            losses.update(loss_rpn_cls=[rl['total'][0]], loss_rpn_bbox=[rl['total'][1]])
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
            with torch.no_grad():
                od = o.detach()
                proposal_list = self.rpn_head.get_bboxes([od[..., :A].permute(0, 3, 1, 2)], [od[..., A:5 * A].permute(0, 3, 1, 2)],
                                                         img_meta, proposal_cfg)
        else:
            proposal_list = proposals
        # assign gts and sample proposals: every frame against the key frame's boxes
        bbox_assigner = T.build_assigner(rcnn_cfg.assigner)
        post_sampler = None
        if isinstance(rcnn_cfg.sampler, (list, tuple)):
            bbox_sampler, post_sampler = T.build_sampler(rcnn_cfg.sampler, context=self)
        else:
            bbox_sampler = T.build_sampler(rcnn_cfg.sampler, context=self)
        gt_b, gt_l = gt_bboxes[key], gt_labels[key]
        sampling_results = []
        for i in range(img.size(0)):
            props = proposal_list[i].contiguous()
            assign_result = bbox_assigner.assign(props, gt_b, None, gt_l)
            k_i = keys['rcnn'][i][:gt_b.shape[0] * int(bbox_sampler.add_gt_as_proposals) + props.shape[0]] if 'rcnn' in keys else None
            sampling_results.append(bbox_sampler.sample(assign_result, props, gt_b, gt_l, keys=k_i, generator=generator))
        T.SamplingResult.resolve(sampling_results)       # every frame's (#pos, #neg) in one host read
        rois, _ = _sampled_rois(sampling_results)
        n_key = sampling_results[key].bboxes.shape[0]
        cur_range = dict(start=key * n_key, length=n_key)
        c5 = self.shared_head.forward_train_nhwc(c4)
        layer = self.bbox_roi_extractor.roi_layers[0]
        feats = ops.roi_align(c5.permute(0, 3, 1, 2), rois, layer.out_size, layer.spatial_scale, layer.sample_num)
        logits = self.bbox_head.forward_train(feats, cur_range)
        labels, label_w, bbox_t, bbox_w = T.bbox_target([sampling_results[key]], [gt_b], [gt_l], rcnn_cfg,
                                                        target_means=self.bbox_head.target_means, target_stds=self.bbox_head.target_stds)
        nc = self.bbox_head.num_classes
        if post_sampler is not None:
            with torch.no_grad():
                row_loss = native.ce_rows(logits.detach(), 0, nc, labels)
                inds, counts = post_sampler.select(labels, row_loss)
                # weights of get_ohem_weights (ohem_hnl_sampler.py:104-111) built on the device: 1 on the picked rows
                slot = torch.arange(inds.numel(), device=inds.device)
                picked = (slot < counts.sum()).float()
                safe = inds.clamp(0, n_key - 1)
                label_w = torch.zeros_like(label_w).scatter_add_(0, safe, picked)
                bbox_w = torch.zeros_like(bbox_w).index_add_(0, safe, ((slot < counts[0]).float())[:, None].expand(-1, 4))
            hl = TO.det_loss_sampled(logits, 0, nc, nc, labels, label_w, bbox_t, bbox_w, counts, beta=1.0)
        else:
            hl = self.bbox_head.loss_train(logits, labels, label_w, bbox_t, bbox_w)
        losses.update(loss_cls=hl['total'][0], loss_bbox=hl['total'][1], acc=hl['acc'])
        return losses

    def forward_feat_frames(self, entries, c4s=None, rescale=False, defer=False):
        """forward_feat from T cached `frame_tensors` entries; c4s (the frames' C4 maps) back the exact re-run that
        replaces the speculative result when some frame kept fewer than nms_post proposals."""
        f1, cur_range, key_rois, counts_dev, mx = self._frames_window(entries)
        meta0 = entries[0]['meta']
        cls_score, bbox_pred = self.bbox_head.forward_from_f1(f1, cur_range, key_dim=self.key_dim, all_res=False)[:2]
        branch, _ = self.bbox_head.get_det_bboxes(key_rois, cls_score, bbox_pred, meta0['img_shape'], meta0['scale_factor'],
                                                  rescale=rescale, cfg=self.test_cfg.rcnn, defer=True)
        metas = [e['meta'] for e in entries]
        pending = PendingWindow([branch], counts_dev, mx, self.bbox_head.num_classes,
                                lambda: self.forward_feat(c4s, metas, None, rescale, speculate=False), single=True)
        return pending if defer else pending.result()


@DETECTORS.register_module
class HNMBRCNN(_WindowDetector):

    def __init__(self, backbone, rpn_head, bbox_roi_extractor, bbox_head, train_cfg, test_cfg, neck=None, shared_head=None,
                 pretrained=None, loss_frames=1):
        super(HNMBRCNN, self).__init__(backbone=backbone, neck=neck, shared_head=shared_head, rpn_head=rpn_head,
                                       bbox_roi_extractor=bbox_roi_extractor, bbox_head=bbox_head, train_cfg=train_cfg,
                                       test_cfg=test_cfg, pretrained=pretrained)
        if self.train_cfg is not None:
            self.key_dim = int(self.train_cfg.rcnn.key_dim)
        else:  # hnmb_rcnn.py:44-48
            self.key_dim = int(self.test_cfg.bbox_head.key_dim)
            self.bbox_head.t_dim = int(test_cfg.bbox_head.t_dim)
            self.bbox_head.sampler_num = int(test_cfg.bbox_head.sampler_num)

    def forward_feat(self, x=None, img_meta=None, proposals=None, rescale=False, defer=False, speculate=True):
        """-> [branch results, final results], each a list of 30 per-class [k,5] arrays (hnmb_rcnn.py:214-218);
        defer=True -> PendingWindow whose result() is that list."""
        w = self.window_tensors(x, img_meta, proposals, rescale, speculate=speculate and proposals is None)
        cls_score, bbox_pred = self.bbox_head.forward_test(w['roi_feats'], [w['cur_range']], key_dim=self.key_dim, all_res=False)
        branches, _ = self.bbox_head.get_det_bboxes(w['key_rois'], cls_score, bbox_pred, img_meta[0]['img_shape'],
                                                    img_meta[0]['scale_factor'], rescale=rescale, cfg=self.test_cfg.rcnn,
                                                    defer=True)
        pending = PendingWindow(branches, w['counts_dev'], w['full_count'], self.bbox_head.num_classes,
                                lambda: self.forward_feat(x, img_meta, proposals, rescale, speculate=False))
        return pending if defer else pending.result()

    # ---- training (hnmb_rcnn.py:54-102, 224-434) -------------------------------------------------------------------
    IMGS_PER_VIDEO, VIDEO_PER_CLS = 3, 3      # hnmb_rcnn.py:265-267 (hard-coded there too)

    def get_triplet_patches(self, c5_feats_all, key_video=0, imgs_per_video=3, extra_cls=2, video_per_cls=3):
        """hnmb_rcnn.py:74-102: the three videos of an iteration -- the key video, the video of the SAME class least similar to
        it and the video of ANOTHER class most similar to those two -- by softmax-normalised dot products of the videos'
        descriptors (global average pool of each frame's res5 map, maximum over the video's frames).  A handful of
        256-vectors: plain tensor arithmetic, one host read of the two chosen indices.
        c5_feats_all: per video a logical [frames, C, h, w] map, or all videos' frames as one tensor.  -> [key_video, same-class id, other-class id]."""
        if isinstance(c5_feats_all, torch.Tensor):     # all videos' frames in one [V * frames, C, h, w] map: three launches for the descriptors
            desc = c5_feats_all.float().mean(dim=(2, 3)).view(-1, imgs_per_video, c5_feats_all.shape[1]).max(dim=1).values
        else:
            desc = torch.stack([f.float().mean(dim=(2, 3)).max(dim=0).values for f in c5_feats_all], 0)   # [V, C]
        key = desc[0:video_per_cls]
        scale = 1.0 / float(key.shape[-1]) ** 0.5
        key_sim = torch.softmax(scale * (desc[0:1] @ key.t()), dim=1)
        same = torch.argmin(key_sim[:, 1:], dim=1) + 1                                            # [1], on the device
        chosen = torch.cat([desc[key_video:key_video + 1], desc[same]], dim=0)
        extra = desc[video_per_cls:]
        extra_sim = torch.softmax(scale * (chosen @ extra.t()), dim=1).sum(dim=0, keepdim=True)
        other = torch.argmax(extra_sim, dim=1) + video_per_cls
        same_id, other_id = (int(v) for v in torch.cat([same, other]).tolist())                   # the iteration's one host read here
        return [key_video, same_id, other_id]

    def forward_train(self, img, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      keys=None, generator=None, c4=None):
        """HNMBRCNN.forward_train (hnmb_rcnn.py:224-434, dynamic=False, single RandomSampler) on the HIP path.
        c4: the C4 maps of `img` when the caller already has them (`self.extract_feat(img)[0]`).  The backbone is frozen in this
        detector's training step (below), so its pass over a batch does not depend on the update in flight: a training loop can run it
        for batch i + 1 on a second stream while batch i trains (dist_train.C4Prefetcher).
        img [V * 3, 3, H, W]: V videos of three frames, key frame first; videos 0..2 share the key video's class, the rest
        are other classes.  As in the reference: the backbone and res5 run WITHOUT a graph over all V videos to pick the
        triplet of videos (:269-277), the chosen videos' C4 maps are reused as constants (:280-283, so the backbone gets no
        gradient in this detector's training step), the RPN only proposes (:318-325, no RPN loss), every frame of a chosen
        video is assigned and sampled against its video's key-frame ground truth (:348-362), res5 (with a graph) + RoIAlign
        feed the head (:333-336, 364-380), and the losses are HRNMPBBoxHead.loss on the three key frames' targets plus the
        head's triplet term (:389-416; the latter through the documented stand-in).
        keys: optional {'rcnn': [[f32 [num_gt + nms_post] per frame] per chosen video]} replacing the sampler's draws.
        -> dict(loss_cls_1, acc_1, loss_bbox_1, loss_cls_2, acc_2, loss_bbox_2, loss_trip)."""
        from . import ops, targets as T
        if self.train_cfg is None:
            raise ValueError('this detector was built without train_cfg (hvrnet_amd.config.hvr_train_config, or the reference config file)')
        if self.key_dim != 0:
            raise NotImplementedError('Key_dim has to be 0 in HNMBRCNN.forward_train (hnmb_rcnn.py:263)')
        if gt_masks is not None or proposals is not None or gt_bboxes_ignore is not None:
            raise NotImplementedError('masks / external proposals / ignore boxes are asserted away by the reference too (:260-261,:292)')
        F_ = self.IMGS_PER_VIDEO
        V = img.shape[0] // F_
        rcnn_cfg = self.train_cfg.rcnn
        if isinstance(rcnn_cfg.sampler, (list, tuple)):
            raise NotImplementedError('the HVR config trains with a single RandomSampler (the head asserts post_sampler is None, :635)')
        # Frames are independent through the backbone, res5 and the RPN, so where the reference loops over videos of three
        # frames (:57-65, :305-336) the whole batch goes through each of them once: same numbers, 5x / 3x the rows per launch.
        with torch.no_grad():                                   # extract_c4_c5_feat (:54-72)
            if c4 is None:
                c4 = self.extract_feat(img)[0]                  # [V * F, 1024, h, w] logical, NHWC in memory
            assert c4.shape[0] == img.shape[0]
            c5_sel = self.shared_head(c4)
        chosen = self.get_triplet_patches(c5_sel, 0, F_, V - self.VIDEO_PER_CLS, self.VIDEO_PER_CLS)
        del c5_sel
        bbox_assigner = T.build_assigner(rcnn_cfg.assigner)
        bbox_sampler = T.build_sampler(rcnn_cfg.sampler, context=self)
        proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
        layer = self.bbox_roi_extractor.roi_layers[0]
        c4k = torch.cat([c4.permute(0, 2, 3, 1)[v * F_:(v + 1) * F_] for v in chosen], 0)          # the chosen videos' frames
        metas_k = [m for v in chosen for m in img_meta[v * F_:(v + 1) * F_]]
        with torch.no_grad():
            rpn_outs = self.rpn_head([c4k.permute(0, 3, 1, 2)])
            if len({tuple(m['img_shape'][:2]) for m in metas_k}) == 1:
                proposal_list = self.rpn_head.get_bboxes(*(rpn_outs + (metas_k, proposal_cfg)))
            else:   # videos of different resolutions in one batch: the proposal pipeline clips per video
                proposal_list = []
                for vi in range(len(chosen)):
                    sl = slice(vi * F_, (vi + 1) * F_)
                    proposal_list += self.rpn_head.get_bboxes([rpn_outs[0][0][sl]], [rpn_outs[1][0][sl]], metas_k[sl], proposal_cfg)
        cur_ranges, key_results, key_gtb, key_gtl, rows = [], [], [], [], []
        per_video = []
        for vi, v in enumerate(chosen):
            gt_b, gt_l = gt_bboxes[v * F_ + self.key_dim], gt_labels[v * F_ + self.key_dim]
            results = []
            for i in range(F_):
                props = proposal_list[vi * F_ + i].contiguous()
                assign_result = bbox_assigner.assign(props, gt_b, None, gt_l)
                k_i = None
                if keys is not None and 'rcnn' in keys:
                    k_i = keys['rcnn'][vi][i][:gt_b.shape[0] * int(bbox_sampler.add_gt_as_proposals) + props.shape[0]]
                results.append(bbox_sampler.sample(assign_result, props, gt_b, gt_l, keys=k_i, generator=generator))
            per_video.append((results, gt_b, gt_l))
        T.SamplingResult.resolve([r for results, _, _ in per_video for r in results])   # every frame's (#pos, #neg) in one host read
        rois, n_rows = _sampled_rois([r for results, _, _ in per_video for r in results])
        for vi, (results, gt_b, gt_l) in enumerate(per_video):
            rows.append(sum(n_rows[vi * F_:(vi + 1) * F_]))
            cur_ranges.append(dict(start=self.key_dim, length=n_rows[vi * F_ + self.key_dim]))
            key_results.append(results[self.key_dim])
            key_gtb.append(gt_b)
            key_gtl.append(gt_l)
        c5 = self.shared_head.forward_train_nhwc(c4k)                                   # res5 with a graph; C4 is a constant
        all_feats = ops.roi_align(c5.permute(0, 3, 1, 2), rois, layer.out_size, layer.spatial_scale, layer.sample_num)
        feats = list(torch.split(all_feats, rows, dim=0))
        targets = T.bbox_target(key_results, key_gtb, key_gtl, rcnn_cfg, target_means=self.bbox_head.target_means,
                                target_stds=self.bbox_head.target_stds)
        # the head through the reference's own entry point and return (hnmb_rcnn.py:431-442, dynamic = False, no post sampler)
        cls_scores, bbox_preds, loss_trip, _similarity = self.bbox_head(feats, cur_range_s=cur_ranges, others=targets[0],
                                                                      all_labels=targets[0], dynamic=False)
        losses = dict(loss_trip) if loss_trip is not None else dict()
        losses.update(self.bbox_head.loss(cls_scores, bbox_preds, *targets))
        return losses

    def forward_feat_frames(self, entries, c4s=None, rescale=False, defer=False):
        """forward_feat from T cached `frame_tensors` entries (see _WindowDetector.frame_tensors)."""
        f1, cur_range, key_rois, counts_dev, mx = self._frames_window(entries)
        meta0 = entries[0]['meta']
        cls_score, bbox_pred = self.bbox_head.forward_from_f1(f1, [cur_range], key_dim=self.key_dim, all_res=False)
        branches, _ = self.bbox_head.get_det_bboxes(key_rois, cls_score, bbox_pred, meta0['img_shape'], meta0['scale_factor'],
                                                    rescale=rescale, cfg=self.test_cfg.rcnn, defer=True)
        metas = [e['meta'] for e in entries]
        pending = PendingWindow(branches, counts_dev, mx, self.bbox_head.num_classes,
                                lambda: self.forward_feat(c4s, metas, None, rescale, speculate=False))
        return pending if defer else pending.result()
