"""RPN head: 3x3 conv + ReLU, 1x1 objectness / delta convs and the fused proposal kernel.

Mirror of mmdet/models/anchor_heads/rpn_head.py:12-104 (+ AnchorHead ctor / get_bboxes,
anchor_head.py:33-81,208-278).  `forward` keeps the reference's per-level list interface;
`get_bboxes` runs sigmoid -> top-6000 -> delta2bbox -> NMS(0.7) -> first 300 -> top-k for ALL
frames of the window in one device pipeline (the reference loops over frames and synchronises
with the host inside every NMS, nms_kernel.cu:104-108).
"""
import torch
import torch.nn as nn

from . import native
from .backbone import PackedMixin, as_logical, as_nhwc, fold_conv_bn
from .box_ops import AnchorGenerator
from .registry import HEADS


@HEADS.register_module
class RPNHead(nn.Module, PackedMixin):

    def __init__(self, in_channels, feat_channels=256, anchor_scales=[8, 16, 32], anchor_ratios=[0.5, 1.0, 2.0],
                 anchor_strides=[4, 8, 16, 32, 64], anchor_base_sizes=None, target_means=(.0, .0, .0, .0),
                 target_stds=(1.0, 1.0, 1.0, 1.0),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)):
        super(RPNHead, self).__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, 2, feat_channels
        self.anchor_scales, self.anchor_ratios, self.anchor_strides = anchor_scales, anchor_ratios, anchor_strides
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else anchor_base_sizes
        self.target_means, self.target_stds = target_means, target_stds
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        if not self.use_sigmoid_cls:
            raise NotImplementedError('softmax RPN scores are outside the HVR hot path (configs use sigmoid)')
        self.cls_out_channels = self.num_classes - 1
        self.loss_cls_cfg, self.loss_bbox_cfg = loss_cls, loss_bbox
        self.fp16_enabled = False
        self.anchor_generators = [AnchorGenerator(b, anchor_scales, anchor_ratios) for b in self.anchor_base_sizes]
        self.num_anchors = len(self.anchor_ratios) * len(self.anchor_scales)
        self.rpn_conv = nn.Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 1)
        self._init_packed()

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)
        self._drop_packed()

    def _pack(self, dtype):
        A = self.num_anchors
        wc, bc = fold_conv_bn(self.rpn_cls, None, dtype)
        wr, br = fold_conv_bn(self.rpn_reg, None, dtype)
        # objectness and deltas share one GEMM: rows [0, A) scores, [A, 5A) deltas, padded to a multiple of 4
        n = 5 * A
        npad = (n + 3) // 4 * 4
        w = torch.zeros((npad, 1, 1, self.feat_channels), dtype=dtype, device=wc.device)
        b = torch.zeros(npad, dtype=torch.float32, device=wc.device)
        w[:A], w[A:n], b[:A], b[A:n] = wc, wr, bc, br
        return dict(conv=fold_conv_bn(self.rpn_conv, None, dtype), heads=(w, b))

    two_level_dtypes = (torch.float32,)   # class attribute: the compute modes whose RPN conv sums in two levels (tools add native.SPLIT); no environment switch

    def forward_single(self, x):
        """x logical [T,C,H,W] -> (cls [T,A,H,W], reg [T,4A,H,W]) f32, physically NHWC."""
        if not x.is_cuda:
            raise NotImplementedError('RPNHead runs on the GPU only (no CPU fallback)')
        p = self.packed(x.device)
        A = self.num_anchors
        # (K = 9 x 1024: in the exact-f32 mode the sum runs in two levels -- the conv's rounding noise reaches the final boxes through the proposal
        # coordinates, and with it that mode's distance to the CPU reference stays inside 1e-3 px on 24 of 24 clips, profiles/r06_noise_two_level.txt.
        # Split half takes the same kernel when `two_level_dtypes` names it: measured (RPN share 7.1 -> 2.6e-4 px, +1.5 % of its window), its claim over
        # 24 clips is 23 either way, so it stays on the big tiles; bf16 / half always run the usual kernels)
        two = native.TWO_LEVEL_HINT if self.compute_dtype in self.two_level_dtypes else None
        y = native.conv2d_nhwc(as_nhwc(x, self.compute_dtype), p['conv'][0], p['conv'][1], relu=True, pad=1, tile=two)
        o = native.conv2d_nhwc(y, p['heads'][0], p['heads'][1], relu=False, out_f32=True)  # [T,H,W,5A(+pad)] f32
        return as_logical(o[..., :A]), as_logical(o[..., A:5 * A])

    def forward_train_fused(self, x):
        """rpn_head.py:30-35 as an autograd graph of HIP convs: x [T,H,W,C] f32 NHWC -> o [T,H,W,ld] f32 whose columns
        0..A are the objectness logits and A..5A the deltas (anchor a's four at A + 4a); ld = 5A padded to a K-step multiple.
        The two 1x1 heads run as one conv over the concatenated weights."""
        from . import train_ops as TO
        y = TO.conv_bias(x, self.rpn_conv, relu=True)
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight], 0)
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias], 0)
        pad = -w.shape[0] % 32
        w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
        b = torch.cat([b, b.new_zeros(pad)], 0)
        return TO.ConvFunction.apply(y, w, TO.ones(w.shape[0], x.device), b, None, False, 1, 0, 1, True)      # f32 out: feeds the loss / proposal kernels

    def forward_train_nhwc(self, x):
        """-> (cls [T,H,W,A], reg [T,H,W,4A]) views of forward_train_fused's output."""
        A = self.num_anchors
        o = self.forward_train_fused(x)
        return o[..., :A], o[..., A:5 * A]

    def _train_anchors(self, featmap_size, img_meta, device):
        """get_anchors (anchor_head.py:100-139) for the single level: grid anchors and their valid flags, cached per shape."""
        key = (tuple(featmap_size), tuple(img_meta['pad_shape'][:2]), str(device))
        cache = self.__dict__.setdefault('_anchor_cache', {})
        if key not in cache:
            stride, gen = self.anchor_strides[0], self.anchor_generators[0]
            feat_h, feat_w = featmap_size
            anchors = gen.grid_anchors(featmap_size, stride, device=device).contiguous()
            h, w = img_meta['pad_shape'][:2]
            vh, vw = min(-(-h // stride), feat_h), min(-(-w // stride), feat_w)
            vy = torch.arange(feat_h, device=device) < vh
            vx = torch.arange(feat_w, device=device) < vw
            valid = (vy[:, None] & vx[None, :]).reshape(-1, 1).expand(-1, self.num_anchors).reshape(-1).to(torch.uint8)
            cache[key] = (anchors, valid.contiguous())
        return cache[key]

    def loss_train(self, o_key, gt_bboxes, img_meta, cfg, keys=None, generator=None):
        """AnchorHead.loss / RPNHead.loss (anchor_head.py:162-206, rpn_head.py:36-53) for ONE image -- SelsaRCNN trains the
        RPN on the key frame only (selsa_rcnn.py:127-136) -- on forward_train_fused's output of that frame, o_key [H,W,ld].
        Targets (assign, sample, deltas) and the loss run on the device without a host copy; `keys` [H*W*A] replaces the
        sampler's random draw.  -> dict(loss_rpn_cls, loss_rpn_bbox, total [2])."""
        from . import targets as T, train_ops as TO
        if len(self.anchor_strides) != 1:
            raise NotImplementedError('single-level RPN only (anchor_strides=[16])')
        H, W, ld = o_key.shape
        anchors, valid = self._train_anchors((H, W), img_meta, o_key.device)
        labels, label_w, bbox_t, bbox_w, counts = T.anchor_target_single(anchors, valid, gt_bboxes, img_meta, self.target_means,
                                                                         self.target_stds, cfg, keys=keys, generator=generator)
        if self.loss_cls_cfg.get('loss_weight', 1.0) != 1.0 or self.loss_bbox_cfg.get('loss_weight', 1.0) != 1.0:
            raise NotImplementedError('RPN loss weights other than 1.0')
        return TO.rpn_loss(o_key.reshape(H * W, ld), self.num_anchors, labels, label_w, bbox_t, bbox_w, counts,
                           beta=self.loss_bbox_cfg.get('beta', 1.0))

    def forward(self, feats):
        outs = [self.forward_single(f) for f in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    def get_bboxes_batched(self, cls_scores, bbox_preds, img_metas, cfg):
        """(proposals [T,max_num,5], counts [T] int32) on the device, no host synchronisation."""
        if len(cls_scores) != 1:
            raise NotImplementedError('single-level RPN only (anchor_strides=[16])')
        if cfg.get('nms_across_levels', False) or cfg.get('min_bbox_size', 0) > 0:
            raise NotImplementedError('nms_across_levels / min_bbox_size > 0 are outside the HVR hot path')
        cls, reg = cls_scores[0], bbox_preds[0]
        shape0 = tuple(img_metas[0]['img_shape'][:2])
        if any(tuple(m['img_shape'][:2]) != shape0 for m in img_metas):
            raise NotImplementedError('frames of one window share img_shape (one video)')
        cls_n, reg_n = cls.permute(0, 2, 3, 1), reg.permute(0, 2, 3, 1)  # physical NHWC views, no copy
        if cls_n.dtype != torch.float32 or cls_n.stride(3) != 1 or reg_n.stride(3) != 1:
            cls_n, reg_n = cls_n.float().contiguous(), reg_n.float().contiguous()
        gen = self.anchor_generators[0]
        return native.rpn_proposals(cls_n, reg_n, gen.base_anchors, self.anchor_strides[0], self.target_means,
                                    self.target_stds, shape0, cfg['nms_pre'], cfg['nms_post'], cfg['max_num'], cfg['nms_thr'])

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, rescale=False):
        """list of [n_i, 5] proposals per frame (anchor_head.py:208-278); one host read of the counts."""
        props, counts = self.get_bboxes_batched(cls_scores, bbox_preds, img_metas, cfg)
        return [props[i, :c] for i, c in enumerate(counts.tolist())]
