"""Training-path ops of the relation heads with HIP backwards (f32 parameters, exact-f32 MFMA path).

The reference trains through plain autograd over nn.Linear / nn.Conv2d / torch.bmm / nn.Softmax and its loss modules
(selsa_bbox_head.py:108-261, bbox_head.py:100-130).  Here every forward AND backward product is a tile-engine GEMM:

    y  = act(x W^T + b (+ resid))        hvr_gemm with the fused epilogue
    dz = dy * (y > 0)                    hvr_relu_bwd          (only when the epilogue had the ReLU)
    dx = dz W                            hvr_gemm   (W^T made K-contiguous by hvr_transpose_pad)
    dW = dz^T x                          hvr_gemm   (both operands transposed / zero-padded to the K-step)
    db = column sums of dz               hvr_colsum
    d resid = dz

plus `ops.relation` (relation core) and `det_loss` (BBoxHead.loss: cross entropy + smooth-L1 + accuracy in one kernel).
torch only moves memory here (views, cat, zero-padding) and drives the autograd graph.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import native


def _pad_cols(t, n):
    """[M, N] -> [M, n] with zero columns appended (K-step padding of a GEMM operand)."""
    if t.shape[1] == n and t.is_contiguous():
        return t
    out = t.new_zeros((t.shape[0], n))
    out[:, :t.shape[1]] = t
    return out


class LinearFunction(Function):
    """y = act(x @ w^T + b (+ resid)); x [M, K], w [N, K], b [N]; N a multiple of 4, K a multiple of the K-step."""

    @staticmethod
    def forward(ctx, x, w, b, resid, relu):
        if not x.is_cuda:
            raise NotImplementedError('the head runs on the GPU only (no CPU fallback)')
        x, w = x.contiguous(), w.contiguous()
        y = native.gemm(x, w, b, resid=resid.contiguous() if resid is not None else None, relu=bool(relu))
        ctx.relu, ctx.has_resid, ctx.has_bias = bool(relu), resid is not None, b is not None
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dz = native.relu_bwd(dy.contiguous(), y) if ctx.relu else dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        step = native.kstep(x.dtype)
        ldn, ldm = (N + step - 1) // step * step, (M + step - 1) // step * step
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = native.gemm(_pad_cols(dz, ldn), native.transpose_pad(w, ldn))           # [M, K] = dz [M, N] W [N, K]
        if ctx.needs_input_grad[1]:
            dw = native.gemm(native.transpose_pad(dz, ldm), native.transpose_pad(x, ldm))  # [N, K] = dz^T [N, M] x [M, K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = native.colsum(dz)
        dr = dz if (ctx.has_resid and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dr, None


def linear(x, w, b=None, resid=None, relu=False):
    return LinearFunction.apply(x, w, b, resid, relu)


class DetLossFunction(Function):
    """(total, loss_cls, loss_bbox, acc) of BBoxHead.loss on a fused f32 logit matrix [R, ld] (class logits at cls_off,
    box deltas at reg_off); only `total` = w_cls * loss_cls + w_bbox * loss_bbox carries a gradient."""

    @staticmethod
    def forward(ctx, logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, beta, w_cls, w_bbox):
        out3, dlogits = native.det_loss(logits.contiguous(), cls_off, reg_off, ncls, labels, label_weights, bbox_targets,
                                        bbox_weights, beta, w_cls, w_bbox)
        ctx.save_for_backward(dlogits)
        if (w_cls, w_bbox) != (1.0, 1.0):
            raise NotImplementedError('loss weights other than 1 (the two configs use loss_weight=1.0) are not wired up')
        # shapes as the reference returns them: scalar losses, accuracy of shape [1] (losses/accuracy.py:19-21)
        loss_cls, loss_bbox, acc = out3[0], out3[1], out3[2:3]
        total = out3[0:2].clone()  # the two loss terms; det_loss's gradient is d(sum(total)) / d logits
        ctx.mark_non_differentiable(loss_cls, loss_bbox, acc)
        return total, loss_cls, loss_bbox, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_cls, g_bbox, g_acc):
        # g_total: gradient w.r.t. the two weighted loss terms; the usual (loss_cls + loss_bbox).backward() gives ones
        dlogits, = ctx.saved_tensors
        if not bool((g_total == 1).all()):
            raise NotImplementedError('DetLossFunction supports the unit upstream gradient of sum(total) only')
        return (dlogits,) + (None,) * 10


def det_loss(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, beta=1.0, w_cls=1.0, w_bbox=1.0):
    """-> dict(loss_cls, loss_bbox (scalars), acc [1], total [2] = the two loss terms; `total.sum().backward()` trains)."""
    total, lc, lb, acc = DetLossFunction.apply(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights,
                                               float(beta), float(w_cls), float(w_bbox))
    return dict(total=total, loss_cls=lc, loss_bbox=lb, acc=acc)
