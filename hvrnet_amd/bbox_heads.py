"""SELSA / HVR proposal-relation heads on the MFMA tile engine.

Mirrors of
  BBoxHead        mmdet/models/bbox_heads/bbox_head.py:13-169
  SelsaBBoxHead   mmdet/models/bbox_heads/selsa_bbox_head.py:11-261
  HRNMPBBoxHead   mmdet/models/bbox_heads/hrnmp_bbox_head.py:57-214 (ctor), 800-909 (forward_test),
                  1009-1052 (get_det_bboxes)
with the same class names, constructor kwargs, parameter names/shapes (`fc_new_k`,
`selsa_k.{q_data_fc_k,k_data_fc_k,linear_out_k}`, `fc_cls[_2]`, `fc_reg[_2]`) and call signatures.

One relation stage (selsa_bbox_head.py:108-200) is 4 launches here instead of ~12 ATen ops:
  [Q|K] = X [Wq;Wk]^T + b            one GEMM, N = 2048
  O     = softmax(Q K^T / 32) X      hvr_relation_fwd (scores -> stats -> apply; logits never in HBM)
  H     = relu(Xq + O Wz^T + bz)     GEMM with bias + residual + ReLU epilogue
The intended behaviour is implemented where the reference dump is broken.
Inference only; the training forward of HRNMPBBoxHead (hard-proposal mining + an un-vendored
triplet loss) is not part of this round.
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import native
from .backbone import PackedMixin
from .registry import HEADS


def _pad_rows(w, b, mult=4):
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous(), b.contiguous()
    wp = torch.zeros((npad, w.shape[1]), dtype=w.dtype, device=w.device)
    bp = torch.zeros(npad, dtype=b.dtype, device=b.device)
    wp[:n], bp[:n] = w, b
    return wp, bp


@HEADS.register_module
class BBoxHead(nn.Module, PackedMixin):
    """Simplest RoI head; here it carries what the relation heads inherit (ctor bookkeeping, read-out)."""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256, num_classes=81,
                 target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2], reg_class_agnostic=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)):
        super(BBoxHead, self).__init__()
        assert with_cls and with_reg
        if with_avg_pool:
            raise NotImplementedError('with_avg_pool is outside the HVR hot path')
        self.with_avg_pool, self.with_cls, self.with_reg = with_avg_pool, with_cls, with_reg
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes = in_channels, num_classes
        self.target_means, self.target_stds = target_means, target_stds
        self.reg_class_agnostic = reg_class_agnostic
        if not reg_class_agnostic:
            raise NotImplementedError('class-specific regression is outside the HVR hot path')
        self.fp16_enabled = False
        self.loss_cls_cfg, self.loss_bbox_cfg = loss_cls, loss_bbox
        self.fc_cls = nn.Linear(in_channels * self.roi_feat_area, num_classes)
        self.fc_reg = nn.Linear(in_channels * self.roi_feat_area, 4)
        self._init_packed()

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)
        self._drop_packed()

    # ---- read-out shared by all heads -------------------------------------------------------
    def _decode(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale):
        """(scores [R,ncls], boxes [R,4]): softmax + delta2bbox + clip (+ /scale_factor), bbox_head.py:141-158."""
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        if not isinstance(scale_factor, (int, float)):
            raise NotImplementedError('per-axis scale_factor arrays are outside the HVR hot path')
        ncls = cls_score.shape[1]
        # the heads' own read-out hands two column views of ONE f32 GEMM output (_readout): decode straight from it
        # (row pitch and column offsets are arguments of hvr_det_decode) instead of concatenating a copy first
        if (cls_score.dtype == torch.float32 and bbox_pred.dtype == torch.float32 and cls_score.stride(1) == 1 and bbox_pred.stride(1) == 1
                and cls_score.stride(0) == bbox_pred.stride(0) and cls_score.shape[0] == bbox_pred.shape[0] > 1
                and cls_score.untyped_storage().data_ptr() == bbox_pred.untyped_storage().data_ptr()):
            reg_off = bbox_pred.storage_offset() - cls_score.storage_offset()
            if ncls <= reg_off and reg_off + 4 <= cls_score.stride(0):
                return native.det_decode(cls_score, 0, reg_off, ncls, rois, self.target_means, self.target_stds, img_shape,
                                         float(scale_factor) if rescale else 0.0)
        logits = torch.cat([cls_score.float(), bbox_pred.float()], dim=1).contiguous()
        return native.det_decode(logits, 0, ncls, ncls, rois, self.target_means, self.target_stds, img_shape,
                                 float(scale_factor) if rescale else 0.0)

    def _nms(self, boxes, scores, cfg, defer=False):
        dets, labels, n = native.multiclass_nms(boxes, scores, cfg.score_thr, cfg.nms['iou_thr'], cfg.max_per_img)
        if defer:  # (dets [max,5], labels [max], n [1]) stay on the device; the caller reads them later in one go
            return (dets, labels, n), None
        k = int(n.item())  # the only host read of the read-out
        return dets[:k], labels[:k]

    def get_det_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None, defer=False):
        scores, bboxes = self._decode(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale)
        if cfg is None:
            return bboxes, scores
        if cfg.nms.get('type', 'nms') != 'nms':
            raise NotImplementedError('only greedy nms is on the HVR hot path')
        return self._nms(bboxes, scores, cfg, defer)


class _RelationHead(BBoxHead):
    """fc_new_k + relation stages shared by the SELSA and HVR heads."""
    NUM_STAGES = 0

    def __init__(self, sampler_num, t_dim, fc_feat_dim=1024, non_cur_space=False, dim=(1024, 1024, 1024),
                 output_cur_only=False, conv_z=None, conv_g=None, *args, **kwargs):
        super(_RelationHead, self).__init__(*args, **kwargs)
        n = self.NUM_STAGES
        conv_z = [True] * n if conv_z is None else conv_z
        conv_g = [False] * n if conv_g is None else conv_g
        if non_cur_space or any(conv_g[:n]) or not all(conv_z[:n]) or tuple(dim) != (fc_feat_dim,) * 3:
            raise NotImplementedError('non_cur_space / conv_g / conv_z=False / dim != fc_feat_dim are outside the hot path')
        self.feat_dim = self.in_channels * self.roi_feat_area
        self.sampler_num, self.t_dim, self.fc_feat_dim = sampler_num, t_dim, fc_feat_dim
        self.non_cur_space, self.dim, self.conv_z, self.conv_g = non_cur_space, dim, conv_z, conv_g
        self.nongt_dim = sampler_num * t_dim
        # computes only what reaches an output (identical results); off = every row the reference computes
        self.dead_row_elimination = False
        for k in range(1, n + 1):
            setattr(self, 'fc_new_%d' % k, nn.Linear(self.feat_dim if k == 1 else dim[2], fc_feat_dim))
            setattr(self, 'selsa_%d' % k, nn.ModuleDict({
                'q_data_fc_%d' % k: nn.Linear(fc_feat_dim, dim[0]),
                'k_data_fc_%d' % k: nn.Linear(fc_feat_dim, dim[1]),
                'aff_softmax_%d' % k: nn.Softmax(dim=2),
                'linear_out_%d' % k: nn.Conv2d(dim[2], dim[2], 1)}))
        self.fc_cls = nn.Linear(dim[2], self.num_classes)
        self.fc_reg = nn.Linear(dim[2], 4)

    def init_weights(self):
        super(_RelationHead, self).init_weights()
        for k in range(1, self.NUM_STAGES + 1):
            for m in [getattr(self, 'fc_new_%d' % k)] + list(getattr(self, 'selsa_%d' % k).values()):
                if isinstance(m, nn.Linear):  # linear_out_k keeps the default Conv2d init (selsa_bbox_head.py:96-100)
                    nn.init.normal_(m.weight, 0.0, 0.01)
                    nn.init.constant_(m.bias, 0)
        for name in ('fc_cls', 'fc_reg', 'fc_cls_2', 'fc_reg_2'):
            if hasattr(self, name):
                nn.init.normal_(getattr(self, name).weight, 0, 0.01)
                nn.init.constant_(getattr(self, name).bias, 0)
        self._drop_packed()

    # ---- packing ---------------------------------------------------------------------------
    def _readout_pack(self, cls, reg, dtype):
        w = torch.cat([cls.weight.detach().float(), reg.weight.detach().float()], 0)
        b = torch.cat([cls.bias.detach().float(), reg.bias.detach().float()], 0)
        w, b = _pad_rows(w, b)
        return native.as_operand(w, dtype), b

    def _pack(self, dtype):
        p = {}
        for k in range(1, self.NUM_STAGES + 1):
            fc = getattr(self, 'fc_new_%d' % k)
            sel = getattr(self, 'selsa_%d' % k)
            q, kk, z = sel['q_data_fc_%d' % k], sel['k_data_fc_%d' % k], sel['linear_out_%d' % k]
            w = fc.weight.detach().float()
            if k == 1:
                C, (ph, pw) = self.in_channels, self.roi_feat_size
                p['fc1_chw'] = native.as_operand(w, dtype)  # RoI features flattened (c, ph, pw): the reference order
                p['fc1_hwc'] = native.as_operand(w.view(-1, C, ph, pw).permute(0, 2, 3, 1).reshape(w.shape[0], -1), dtype)
            else:
                p['fc%d' % k] = native.as_operand(w, dtype)
            p['fcb%d' % k] = fc.bias.detach().float().contiguous()
            p['wqk%d' % k] = native.as_operand(torch.cat([q.weight.detach().float(), kk.weight.detach().float()], 0), dtype)
            p['bqk%d' % k] = torch.cat([q.bias.detach().float(), kk.bias.detach().float()], 0).contiguous()
            p['wz%d' % k] = native.as_operand(z.weight.detach().float().view(z.weight.shape[0], -1), dtype)
            p['bz%d' % k] = z.bias.detach().float().contiguous()
        p['out1'] = self._readout_pack(self.fc_cls, self.fc_reg, dtype)
        if hasattr(self, 'fc_cls_2'):
            p['out2'] = self._readout_pack(self.fc_cls_2, self.fc_reg_2, dtype)
        return p

    # ---- device pipeline ---------------------------------------------------------------------
    def _fc1(self, p, bbox_feat):
        """fc_new_1 on RoI features, accepting NCHW-contiguous or channels_last ([K,7,7,C] physical) input."""
        if not bbox_feat.is_cuda:
            raise NotImplementedError('relation heads run on the GPU only (no CPU fallback)')
        if bbox_feat.dim() == 4 and not bbox_feat.is_contiguous() and bbox_feat.permute(0, 2, 3, 1).is_contiguous():
            x, w = bbox_feat.permute(0, 2, 3, 1).reshape(bbox_feat.size(0), -1), p['fc1_hwc']
        else:
            x, w = bbox_feat.contiguous().view(bbox_feat.size(0), -1), p['fc1_chw']
        return native.gemm(native.cast(x, self.compute_dtype), w, p['fcb1'])

    def fc1_rows(self, bbox_feat):
        """fc_new_1 of RoI features [K,256,7,7] -> [K,1024] (selsa_bbox_head.py:222-224); one row per RoI."""
        return self._fc1(self.packed(bbox_feat.device), bbox_feat)

    # ---- training-path building blocks (autograd graphs of HIP ops; f32 master parameters) -------------------
    def _train_rows(self, bbox_feat):
        """RoI features [K,256,7,7] -> [K, 12544] rows in the compute dtype ((c, ph, pw) order, as the reference flattens)."""
        x = bbox_feat.contiguous().view(bbox_feat.size(0), -1)
        if x.dtype != self.compute_dtype:   # f32: parity mode; bf16: operands rounded to bf16, f32 accumulation
            if x.requires_grad:
                raise NotImplementedError('RoI features must arrive in the compute dtype (%s) when they carry a gradient' % self.compute_dtype)
            x = native.cast(x, self.compute_dtype)
        return x

    def _train_qk(self, k, f, q_rows, nongt_dim):
        """q / k projections of relation stage k: queries = rows `q_rows` (slice or None = all) of f, keys = f[:nongt_dim]."""
        from . import train_ops as TO
        sel = getattr(self, 'selsa_%d' % k)
        kv = f if nongt_dim >= f.shape[0] else f[:nongt_dim]
        fq = f if q_rows is None else f[q_rows]
        q = TO.linear(fq, sel['q_data_fc_%d' % k].weight, sel['q_data_fc_%d' % k].bias)
        kk = TO.linear(kv, sel['k_data_fc_%d' % k].weight, sel['k_data_fc_%d' % k].bias)
        return q, kk, kv, fq

    def _train_stage(self, k, f, q_rows=None, nongt_dim=None, qk=None):
        """relu(f[q_rows] + linear_out_k(softmax(q k^T / sqrt(d)) f[:nongt_dim]))  (selsa_bbox_head.py:108-200)."""
        from . import ops, train_ops as TO
        q, kk, kv, fq = qk if qk is not None else self._train_qk(k, f, q_rows, self.nongt_dim if nongt_dim is None else nongt_dim)
        o = ops.relation(q, kk, kv, 1.0 / math.sqrt(float(self.dim[1])))
        z = getattr(self, 'selsa_%d' % k)['linear_out_%d' % k]
        return TO.linear(o, z.weight.view(z.weight.shape[0], -1), z.bias, resid=fq.contiguous(), relu=True)

    def _train_readout(self, h, fc_cls, fc_reg):
        """fused [fc_cls | fc_reg | pad] readout -> f32 logits [rows, num_classes + 4 (+ pad)]."""
        from . import train_ops as TO
        nc = self.num_classes
        w = torch.cat([fc_cls.weight, fc_reg.weight, fc_cls.weight.new_zeros((-(nc + 4) % 4, fc_cls.weight.shape[1]))], 0)
        b = torch.cat([fc_cls.bias, fc_reg.bias, fc_cls.bias.new_zeros(-(nc + 4) % 4)], 0)
        return TO.linear(h.contiguous(), w, b, out_f32=True)

    # bit-for-bit equality of a batched call (clips > 1) with the per-clip calls: native.relation_fwd_grouped(exact=True)
    grouped_exact = False

    def _stage(self, p, k, x, q_range=None, clips=1):
        """relu(Xq + relation_k(X)): rows `q_range` as queries (all rows when None), keys = X[:nongt_dim].
        clips = W > 1: x holds the rows of W independent clips back to back ([W * R, D]); projections and the output layer run on
        all W * R rows as one product each, the relation core per clip in ONE grouped call (hvr_relation_fwd_grouped); q_range
        addresses rows inside every clip -> [W * l, D], clip-major."""
        D = self.fc_feat_dim
        if clips > 1:
            R = x.shape[0] // clips
            assert R * clips == x.shape[0] and self.nongt_dim >= R, 'batched clips: equal row counts, untruncated keys'
            wqk, bqk = p['wqk%d' % k], p['bqk%d' % k]
            scale = 1.0 / math.sqrt(float(self.dim[1]))
            if q_range is None:
                qk = native.gemm(x, wqk, bqk)
                o = native.relation_fwd_grouped(qk[:, :D], qk[:, D:], x, scale, clips, exact=self.grouped_exact)
                return native.gemm(o, p['wz%d' % k], p['bz%d' % k], resid=x, relu=True)
            s, l = q_range
            xq = x.view(clips, R, D)[:, s:s + l].reshape(clips * l, D)
            q = native.gemm(xq, wqk[:D], bqk[:D])
            kk = native.gemm(x, wqk[D:], bqk[D:])
            o = native.relation_fwd_grouped(q, kk, x, scale, clips, exact=self.grouped_exact)
            return native.gemm(o, p['wz%d' % k], p['bz%d' % k], resid=xq, relu=True)
        kv = x if self.nongt_dim >= x.shape[0] else x[:self.nongt_dim]
        wqk, bqk = p['wqk%d' % k], p['bqk%d' % k]
        if q_range is None and kv is x:
            qk = native.gemm(x, wqk, bqk)
            q, kk, xq = qk[:, :D], qk[:, D:], x
        else:
            xq = x if q_range is None else x[q_range[0]:q_range[0] + q_range[1]]
            q = native.gemm(xq, wqk[:D], bqk[:D])
            kk = native.gemm(kv, wqk[D:], bqk[D:])
        o = native.relation_fwd(q, kk, kv, 1.0 / math.sqrt(float(self.dim[1])))
        return native.gemm(o, p['wz%d' % k], p['bz%d' % k], resid=xq, relu=True)

    @staticmethod
    def _key_rows(h, clips, s, l):
        """rows s .. s + l of every clip of h [clips * R, D] -> [clips * l, D] (a view for one clip)."""
        if clips == 1:
            return h[s:s + l]
        R = h.shape[0] // clips
        return h.view(clips, R, h.shape[1])[:, s:s + l].reshape(clips * l, h.shape[1])

    def _readout(self, p, name, h):
        o = native.gemm(h, p[name][0], p[name][1], out_f32=True)
        nc = self.num_classes
        return o[:, :nc], o[:, nc:nc + 4]


@HEADS.register_module
class SelsaBBoxHead(_RelationHead):
    NUM_STAGES = 2

    def forward(self, bbox_feat, cur_range=None, key_dim=0, all_res=False, clips=1):
        """-> (cls_score, bbox_pred, None), selsa_bbox_head.py:203-261 (output_cur_only=False)."""
        return self.forward_from_f1(self.fc1_rows(bbox_feat), cur_range, key_dim, all_res, clips=clips)

    def forward_from_f1(self, f1, cur_range=None, key_dim=0, all_res=False, clips=1):
        """The head from the fc_new_1 rows on (rows are per-RoI: a video runner may cache them per frame).  clips = W > 1: f1 holds
        W independent clips' rows back to back, cur_range addresses the key rows inside every clip; outputs are clip-major."""
        assert cur_range is not None, 'Feature num range along axis need specified'
        self.key_dim = key_dim
        self.nongt_dim = self.sampler_num * self.t_dim
        s, l = int(cur_range['start']), int(cur_range['length'])
        p = self.packed(f1.device)
        h1 = self._stage(p, 1, f1, clips=clips)
        f2 = native.gemm(h1, p['fc2'], p['fcb2'])
        if all_res:
            h2 = self._stage(p, 2, f2, clips=clips)
        elif self.dead_row_elimination:
            h2 = self._stage(p, 2, f2, (s, l), clips=clips)
        else:
            h2 = self._key_rows(self._stage(p, 2, f2, clips=clips), clips, s, l)
        cls, reg = self._readout(p, 'out1', h2)
        return cls, reg, None


    # ---- training step (f32 parameters; selsa_rcnn.py:201,242-243) ------------------------------------------
    def forward_train(self, bbox_feat, cur_range):
        """The head's forward as an autograd graph of HIP ops (train_ops.linear / ops.relation): -> f32 logits [l, 36] of
        the key frame's rows, class logits in columns 0..num_classes-1, box deltas in the next 4 (fused fc_cls | fc_reg)."""
        from . import train_ops as TO
        s, l = int(cur_range['start']), int(cur_range['length'])
        self.nongt_dim = self.sampler_num * self.t_dim      # selsa_bbox_head.py:214; keys / values: the first nongt_dim rows (:130)
        f1 = TO.linear(self._train_rows(bbox_feat), self.fc_new_1.weight, self.fc_new_1.bias)
        h1 = self._train_stage(1, f1)
        f2 = TO.linear(h1, self.fc_new_2.weight, self.fc_new_2.bias)
        h2 = self._train_stage(2, f2)[s:s + l]
        return self._train_readout(h2, self.fc_cls, self.fc_reg)

    def loss_train(self, logits, labels, label_weights, bbox_targets, bbox_weights):
        """BBoxHead.loss on forward_train's fused logits (bbox_head.py:100-130) -> dict(loss_cls, loss_bbox, acc, total)."""
        from . import train_ops as TO
        nc = self.num_classes
        return TO.det_loss(logits, 0, nc, nc, labels, label_weights, bbox_targets, bbox_weights, beta=1.0)


@HEADS.register_module
class HRNMPBBoxHead(_RelationHead):
    NUM_STAGES = 4

    def __init__(self, sampler_num, t_dim, imgs_per_video, *args, **kwargs):
        super(HRNMPBBoxHead, self).__init__(sampler_num, t_dim, *args, **kwargs)
        self.imgs_per_video = imgs_per_video
        self.output_cur_only = False
        self.fc_cls_2 = nn.Linear(self.dim[2], self.num_classes)
        self.fc_reg_2 = nn.Linear(self.dim[2], 4)

    def hardest_proposal_mining(self, labels, all_labels, aff_scale, metric_loss=None):
        """hrnmp_bbox_head.py:357-414: per non-background query row of the scaled affinities aff_scale [1, Mq, Mk] (or
        [Mq, Mk]) the same-label key with the lowest and the different-label key with the highest affinity -- one
        wavefront per row on the device (hvr_mining_argreduce) where the reference builds three masked copies of the matrix
        and calls topk on each.  -> [anchor rows, hardest positive, hardest negative], the reference's return order (:413)."""
        aff = aff_scale.reshape(-1, aff_scale.shape[-1])
        if aff.dtype != torch.float32 or aff.stride(1) != 1:
            aff = aff.float().contiguous()
        picks = native.mining_argreduce(aff, labels, all_labels)
        anchors = torch.nonzero(labels != 0).reshape(-1)
        return [anchors, picks[anchors, 1], picks[anchors, 0]]

    def forward_test(self, bbox_feat_s, cur_range_s=None, key_dim=0, all_res=False, clips=1):
        """-> ([cls_branch, cls], [reg_branch, reg]), hrnmp_bbox_head.py:800-909."""
        return self.forward_from_f1(self.fc1_rows(bbox_feat_s), cur_range_s, key_dim, all_res, clips=clips)

    def forward_from_f1(self, f1, cur_range_s=None, key_dim=0, all_res=False, clips=1):
        """The head from the fc_new_1 rows on (rows are per-RoI: a video runner may cache them per frame).  clips = W > 1: f1 holds
        W independent clips' rows back to back, cur_range_s[0] addresses the key rows inside every clip; outputs are clip-major
        ([W * l, .]: clip w's rows are w * l .. (w + 1) * l)."""
        assert cur_range_s is not None, 'Feature num range along axis need specified'
        self.key_dim = key_dim
        self.nongt_dim = self.sampler_num * self.t_dim
        assert self.nongt_dim >= f1.shape[0] // clips  # hrnmp_bbox_head.py:249
        cur = cur_range_s[0]
        s, l = int(cur['start']), int(cur['length'])
        p = self.packed(f1.device)
        h1 = self._stage(p, 1, f1, clips=clips)
        f2 = native.gemm(h1, p['fc2'], p['fcb2'])
        if self.dead_row_elimination:
            h2_key = self._stage(p, 2, f2, (s, l), clips=clips)
        else:
            h2_key = self._key_rows(self._stage(p, 2, f2, clips=clips), clips, s, l)
        cls_b, reg_b = self._readout(p, 'out1', h2_key)
        x3 = f1.clone()            # non-key rows fall back to the stage-1 pre-attention features (:865-868)
        if clips == 1:
            x3[s:s + l] = h2_key
        else:
            R = f1.shape[0] // clips
            x3.view(clips, R, -1)[:, s:s + l] = h2_key.view(clips, l, -1)
        f3 = native.gemm(x3, p['fc3'], p['fcb3'])
        h3 = self._stage(p, 3, f3, clips=clips)
        f4 = native.gemm(h3, p['fc4'], p['fcb4'])
        h4 = self._stage(p, 4, f4, (s, l), clips=clips)
        cls, reg = self._readout(p, 'out2', h4)
        return [cls_b, cls], [reg_b, reg]

    TRIPLET_MARGIN = 10.0   # TripletNonLocalLoss(margin=10) of the inter-video stage (hrnmp_bbox_head.py:741)

    def forward_train(self, bbox_feat_s, cur_range_s, others, key_dim=0):
        """HRNMPBBoxHead.forward with dynamic=False (hrnmp_bbox_head.py:609-798) as an autograd graph of HIP ops.
        bbox_feat_s: per video the RoI features of its imgs_per_video frames (key frame's rows first); cur_range_s: the key
        rows of each video; others: the key rows' class labels, all videos concatenated (`bbox_targets_key[0]`).
          per video   fc_new_1 -> relation 1 (all rows) -> fc_new_2 -> relation 2 (key rows as queries) -> branch logits;
                      [stage-2 key rows | fc_new_1 rows of the other frames] -> fc_new_3 -> relation 3 (key rows)   (:652-733)
          all videos  key rows concatenated -> fc_new_4 -> relation 4 over all of them, with hard-proposal mining on its
                      affinities and the triplet loss on its q / k projections -> final logits                   (:735-790)
        The triplet term is the documented STAND-IN (train_ops.TripletMarginFunction): the reference's TripletNonLocalLoss
        is not in its tree.  The mined triple is passed in the reference's order, including its acknowledged swap
        ("pos_sm and pos_nsm are in wrong (inversed) positions", :408): positives = the different-label key with the highest
        affinity, negatives = the same-label key with the lowest.
        -> ([branch logits, final logits] f32 fused [rows, num_classes + 4 (+pad)], dict(loss_trip))."""
        from . import ops, train_ops as TO
        assert cur_range_s is not None and len(cur_range_s) == len(bbox_feat_s)
        self.key_dim, self.nongt_dim = key_dim, self.sampler_num * self.t_dim
        per_video = self.imgs_per_video * self.sampler_num
        # The layers' weights are shared by the videos and a linear layer works row by row, so where the reference loops over the videos
        # (:652-733) every fc / projection / output layer runs ONCE on all videos' rows (one forward product, one data gradient and one
        # weight gradient per layer instead of one per video); only the relation core is per video (its softmax is over the video's keys).
        R = [int(f.shape[0]) for f in bbox_feat_s]
        L = []
        for cur in cur_range_s:
            assert int(cur['start']) == 0, 'training keeps the key frame first (hnmb_rcnn.py:263: key_dim has to be 0)'
            L.append(int(cur['length']))
        assert all(r <= per_video for r in R), 'keys of a video: all of its rows (hrnmp_bbox_head.py:249)'
        r0 = [sum(R[:v]) for v in range(len(R))]      # first row of video v among all rows
        l0 = [sum(L[:v]) for v in range(len(L))]      # ... among the key rows
        key_of = lambda t: torch.cat([t[r0[v]:r0[v] + L[v]] for v in range(len(R))], dim=0) if len(R) > 1 else t[:L[0]]
        scale = 1.0 / math.sqrt(float(self.dim[1]))

        def stage(k, f, key_queries):
            """relation stage k on all videos' rows f: queries = every row, or the videos' key rows; keys / values = the video's rows"""
            sel = getattr(self, 'selsa_%d' % k)
            fq = key_of(f) if key_queries else f
            q = TO.linear(fq, sel['q_data_fc_%d' % k].weight, sel['q_data_fc_%d' % k].bias)
            kk = TO.linear(f, sel['k_data_fc_%d' % k].weight, sel['k_data_fc_%d' % k].bias)
            o = []
            for v in range(len(R)):
                qv = q[l0[v]:l0[v] + L[v]] if key_queries else q[r0[v]:r0[v] + R[v]]
                o.append(ops.relation(qv, kk[r0[v]:r0[v] + R[v]], f[r0[v]:r0[v] + R[v]], scale))
            z = sel['linear_out_%d' % k]
            return TO.linear(torch.cat(o, dim=0) if len(o) > 1 else o[0], z.weight.view(z.weight.shape[0], -1), z.bias, resid=fq.contiguous(), relu=True)

        f1 = TO.linear(self._train_rows(torch.cat(list(bbox_feat_s), dim=0) if len(bbox_feat_s) > 1 else bbox_feat_s[0]), self.fc_new_1.weight, self.fc_new_1.bias)
        h1 = stage(1, f1, False)
        f2 = TO.linear(h1, self.fc_new_2.weight, self.fc_new_2.bias)
        h2 = stage(2, f2, True)                                                        # [sum l, 1024]: idx_output_cur_only
        branch = self._train_readout(h2, self.fc_cls, self.fc_reg)
        x3 = torch.cat([t for v in range(len(R)) for t in (h2[l0[v]:l0[v] + L[v]], f1[r0[v] + L[v]:r0[v] + R[v]])], dim=0)   # :700-702
        f3 = TO.linear(x3, self.fc_new_3.weight, self.fc_new_3.bias)
        video_feats = stage(3, f3, True)                                               # the videos' key rows, video-major
        assert self.nongt_dim >= video_feats.shape[0]                                 # :451
        f4 = TO.linear(video_feats.contiguous(), self.fc_new_4.weight, self.fc_new_4.bias)
        qk = self._train_qk(4, f4, None, self.nongt_dim)
        q4, k4 = qk[0], qk[1]
        with torch.no_grad():
            ld = -(-k4.shape[0] // 4) * 4
            kpad = k4.detach() if ld == k4.shape[0] else torch.cat([k4.detach(), k4.new_zeros((ld - k4.shape[0], k4.shape[1]))], 0)
            aff = native.gemm(q4.detach().contiguous(), kpad.contiguous(), out_f32=True)[:, :k4.shape[0]]   # ordering only: unscaled
            anchors, same_min, diff_max = self.hardest_proposal_mining(others, others[:k4.shape[0]], aff)
        losses = dict()
        if anchors.numel() > 0:
            loss_trip, _ = TO.triplet_margin(q4, k4, anchors, diff_max, same_min, self.TRIPLET_MARGIN)
            losses['loss_trip'] = loss_trip
        h4 = self._train_stage(4, f4, qk=qk)                                          # cur_only_for_4 = False: every key row
        final = self._train_readout(h4, self.fc_cls_2, self.fc_reg_2)
        return [branch, final], losses

    def loss_train(self, logits_list, labels, label_weights, bbox_targets, bbox_weights):
        """HRNMPBBoxHead.loss (hrnmp_bbox_head.py:970-1007) on the fused logits of each branch:
        -> dict(loss_cls_1, acc_1, loss_bbox_1, loss_cls_2, acc_2, loss_bbox_2); the loss entries carry the gradient."""
        from . import train_ops as TO
        nc = self.num_classes
        out = dict()
        for i, logits in enumerate(logits_list):
            assert logits.shape[0] == labels.shape[0]
            d = TO.det_loss(logits, 0, nc, nc, labels, label_weights, bbox_targets, bbox_weights, beta=1.0)
            out['loss_cls_%d' % (i + 1)], out['loss_bbox_%d' % (i + 1)], out['acc_%d' % (i + 1)] = d['total'][0], d['total'][1], d['acc']
        return out

    def forward(self, bbox_feat_s, cur_range_s=None, key_dim=0, all_res=False, others=None, dynamic=False, all_labels=None,
                post_sampler=None, bbox_targets_key=None):
        """The reference's entry point, same signature and return (hrnmp_bbox_head.py:609-795; called at hnmb_rcnn.py:438 as
        `self.bbox_head(feats, cur_range_s=cur_ranges, others=bbox_targets_key[0], all_labels=all_labels, dynamic=False)`):
        -> ([cls_score_branch, cls_score], [bbox_pred_branch, bbox_pred], loss_additional, similarity_).
        `others` (the key rows' labels) selects the training graph (forward_train: stages 1-3 per video, the inter-video stage 4
        with hard-proposal mining and the triplet term in `loss_additional['loss_trip']`); without labels there is nothing to mine
        and the call is the inference path (forward_test), with an empty loss dict.  `similarity_` is the reference's debugging
        record (numpy copies of the affinities), None here.  Outside the envelope the reference itself asserts or leaves off:
        `post_sampler` (asserted None at :636), `dynamic=True` (off in its own call, hnmb_rcnn.py:431) -- both raise."""
        if post_sampler is not None:
            raise NotImplementedError('HRNMPBBoxHead.forward: post_sampler is not implemented (the reference asserts it is None, hrnmp_bbox_head.py:636)')
        if dynamic:
            if all_labels is None:
                raise AssertionError('`all_labels` should be specified when `dynamic` is `True`')   # hrnmp_bbox_head.py:639-640
            raise NotImplementedError('HRNMPBBoxHead.forward: dynamic=True (per-video mining, hrnmp_bbox_head.py:416-606) is outside '
                                      'the hot path: the reference calls the head with dynamic=False (hnmb_rcnn.py:431-438)')
        assert cur_range_s is not None, 'Feature num range along axis needs specified'
        nc = self.num_classes
        if others is None:
            cls, reg = self.forward_test(bbox_feat_s, cur_range_s, key_dim, all_res)
            return cls, reg, dict(), None
        logits, loss_additional = self.forward_train(bbox_feat_s, cur_range_s, others, key_dim)
        return [l[:, :nc] for l in logits], [l[:, nc:nc + 4] for l in logits], loss_additional, None

    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, reduction_override=None):
        """HRNMPBBoxHead.loss with the reference's arguments (hrnmp_bbox_head.py:970-1007): the per-branch lists `forward`
        returns -> dict(loss_cls_i, acc_i, loss_bbox_i).  The fused kernel (train_ops.det_loss) takes one [rows, classes + 4]
        matrix per branch: the two column blocks are put side by side again (a copy of rows x 35 floats)."""
        assert reduction_override is None, 'reduction_override is not used on the hot path'
        fused = [torch.cat([c, b], dim=1) for c, b in zip(cls_score, bbox_pred)]
        return self.loss_train(fused, labels, label_weights, bbox_targets, bbox_weights)

    readout_streams = True   # the second branch's read-out on a side stream (class attribute; no environment switch)

    def get_det_bboxes(self, rois, cls_scores, bbox_preds, img_shape, scale_factor, rescale=False, cfg=None, defer=False):
        """Per-branch read-out -> (list of det_bboxes, list of det_labels), hrnmp_bbox_head.py:1009-1052."""
        def branch(cls_score, bbox_pred):
            scores, bboxes = self._decode(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale)
            if cfg is None or not hasattr(cfg, 'nms'):
                return bboxes, scores
            return self._nms(bboxes, scores, cfg, defer)

        pairs = list(zip(cls_scores, bbox_preds))
        # The two branches' read-outs are independent chains of small, latency-bound launches (decode, 30 one-workgroup
        # class sweeps, a one-workgroup merge: ~150 us each on an otherwise idle chip): with deferred results the second
        # runs on a side stream beside the first.  Same kernels on the same inputs: identical results.
        if defer and self.readout_streams and len(pairs) == 2 and rois.is_cuda and cfg is not None and hasattr(cfg, 'nms'):
            main = torch.cuda.current_stream(rois.device)
            pool = self.__dict__.setdefault('_readout_side', {})
            key = (str(rois.device), main.cuda_stream)   # a side stream belongs to one main stream (windows in flight)
            if key not in pool:
                pool[key] = torch.cuda.Stream(device=rois.device)
            side = pool[key]
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(side):
                side.wait_event(fork)
                out1 = branch(*pairs[1])
            out0 = branch(*pairs[0])
            main.wait_stream(side)
            for t in out1[0]:
                t.record_stream(main)
            for t in pairs[1] + (rois,):
                if torch.is_tensor(t):
                    t.record_stream(side)
            outs = [out0, out1]
        else:
            outs = [branch(c, b) for c, b in pairs]
        return [o[0] for o in outs], [o[1] for o in outs]
