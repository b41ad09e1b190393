"""`mmdet.ops` look-alikes backed by libhvr_hip.so.

The reference resolves ops by NAME (`getattr(mmdet.ops, 'RoIAlign')`, roi_extractors/single_level.py:45-52;
`from mmdet.ops import nms`, anchor_heads/rpn_head.py:7; `getattr(nms_wrapper, 'nms')`,
core/post_processing/bbox_nms.py:32-34), so this module exposes the same names with the same call
signatures and error behaviour:
  RoIAlign(out_size, spatial_scale, sample_num=0, use_torchvision=False)   roi_align/roi_align.py:59-87
  roi_align(features, rois, out_size, spatial_scale, sample_num)           roi_align/roi_align.py:56
  nms(dets, iou_thr, device_id=None) -> (dets[inds], inds)                  nms/nms_wrapper.py:8-61
plus `relation(q, k, v, scale)`: the relation core as an autograd Function with a HIP backward (training path).
CPU tensors raise NotImplementedError exactly like the reference's RoIAlign (roi_align.py:27-28);
NMS follows the reference's CPU semantics (`IoU >= thr` suppresses, nms_cpu.cpp:55).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import native


def _channels_last(t):
    """Physical NHWC behind a logical NCHW tensor (ambiguous degenerate shapes count as NCHW)."""
    return t.dim() == 4 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous()


class RoIAlignFunction(Function):

    @staticmethod
    def forward(ctx, features, rois, out_size, spatial_scale, sample_num=0):
        out_h, out_w = _pair(out_size)
        assert isinstance(out_h, int) and isinstance(out_w, int)
        if not features.is_cuda:
            raise NotImplementedError
        if rois.dim() != 2 or rois.size(1) != 5:
            raise ValueError('wrong roi size: expected [K, 5], got %s' % (tuple(rois.shape),))
        ctx.spatial_scale, ctx.sample_num = spatial_scale, sample_num
        ctx.save_for_backward(rois)
        ctx.feature_size, ctx.feature_dtype = features.size(), features.dtype
        ctx.nhwc = _channels_last(features)
        if ctx.nhwc:
            out = native.roi_align_fwd(features.permute(0, 2, 3, 1), rois, out_h, out_w, spatial_scale, sample_num,
                                       native.LAYOUT_NHWC)
            return out.permute(0, 3, 1, 2)
        return native.roi_align_fwd(features.contiguous(), rois, out_h, out_w, spatial_scale, sample_num, native.LAYOUT_NCHW)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois = ctx.saved_tensors[0]
        assert ctx.feature_size is not None and grad_output.is_cuda
        grad_input = None
        if ctx.needs_input_grad[0]:
            B, C, H, W = ctx.feature_size
            if ctx.nhwc:
                g = native.roi_align_bwd(grad_output.permute(0, 2, 3, 1), rois, (B, H, W, C), ctx.spatial_scale,
                                         ctx.sample_num, native.LAYOUT_NHWC)
                grad_input = native.cast(g, ctx.feature_dtype).permute(0, 3, 1, 2)   # accumulated in f32, handed back in the map's dtype
            else:
                grad_input = native.cast(native.roi_align_bwd(grad_output, rois, (B, C, H, W), ctx.spatial_scale, ctx.sample_num,
                                                              native.LAYOUT_NCHW), ctx.feature_dtype)
        return grad_input, None, None, None, None


roi_align = RoIAlignFunction.apply


class RoIAlign(nn.Module):

    def __init__(self, out_size, spatial_scale, sample_num=0, use_torchvision=False):
        super(RoIAlign, self).__init__()
        if use_torchvision:
            raise NotImplementedError('torchvision is not part of this build')
        self.out_size = _pair(out_size)
        self.spatial_scale = float(spatial_scale)
        self.sample_num = int(sample_num)
        self.use_torchvision = use_torchvision

    def forward(self, features, rois):
        return roi_align(features, rois, self.out_size, self.spatial_scale, self.sample_num)

    def __repr__(self):
        return '%s(out_size=%s, spatial_scale=%s, sample_num=%s, use_torchvision=%s)' % (
            self.__class__.__name__, self.out_size, self.spatial_scale, self.sample_num, self.use_torchvision)


class RelationFunction(Function):
    """o = softmax(scale * q k^T) v with a HIP backward (the reference differentiates torch.bmm / nn.Softmax / torch.mm,
    selsa_bbox_head.py:166-182, through autograd).  Backward = one score pass (probabilities recomputed, not stored),
    one fused softmax-backward kernel and four tile-engine GEMMs:
        dV = P^T dO      dP = dO V^T      dS = scale * P * (dP - rowsum(dO * O))      dQ = dS K      dK = dS^T Q
    K-contiguous operands for the transposed products come from hvr_transpose_pad (zero-padded to the K-step)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        if not q.is_cuda:
            raise NotImplementedError('the relation core runs on the GPU only (no CPU fallback)')
        o = native.relation_fwd(q.contiguous(), k.contiguous(), v.contiguous(), scale)
        ctx.save_for_backward(q, k, v, o)
        ctx.scale = float(scale)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_o):
        q, k, v, o = ctx.saved_tensors
        q, k, v, go = q.contiguous(), k.contiguous(), v.contiguous(), grad_o.contiguous().to(q.dtype)
        Mq, D = q.shape
        Mk = k.shape[0]
        step = 64 if q.dtype == torch.bfloat16 else 32              # K-step of the tile engine, in elements
        ldq = (Mq + step - 1) // step * step
        P = native.relation_probs(q, k, ctx.scale)                   # [Mq, ldp], padding columns zero
        ldp = P.shape[1]
        if ldp == Mk:
            vp = v
        else:
            vp = v.new_zeros((ldp, D))                               # V with zero rows for the padded keys
            vp[:Mk] = v
        dP = native.gemm(go, vp)                                     # [Mq, ldp]
        dS = native.relation_dscore(P, dP, go, o, ctx.scale)         # [Mq, ldp]
        go_t = native.transpose_pad(go, ldq)                         # [D, ldq]
        dv = native.gemm(native.transpose_pad(P, ldq)[:Mk], go_t)    # P^T dO        -> [Mk, D]
        dq = native.gemm(dS, native.transpose_pad(k, ldp))           # dS K          -> [Mq, D]
        dk = native.gemm(native.transpose_pad(dS, ldq)[:Mk], native.transpose_pad(q, ldq))  # dS^T Q -> [Mk, D]
        return dq, dk, dv, None


relation = RelationFunction.apply


def nms(dets, iou_thr, device_id=None):
    """Same dispatch contract as nms_wrapper.nms: tensor or ndarray in, same type out."""
    if isinstance(dets, torch.Tensor):
        is_numpy = False
        dets_th = dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        dets_th = torch.from_numpy(dets).to('cuda:%d' % (device_id or 0))
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        if not dets_th.is_cuda:
            raise NotImplementedError('hvr nms runs on the GPU only (no CPU fallback)')
        inds = native.nms(dets_th, iou_thr, ge_semantics=True)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds
