"""ResNet-C4 backbone and the res5 shared head, executed by the gfx950 implicit-GEMM kernels.

Host-side mirror of mmdet/models/backbones/resnet.py (Bottleneck 86-266, make_res_layer 269-329,
ResNet 332-543) and mmdet/models/shared_heads/res_layer.py (ResLayer 13-81): same class names,
constructor kwargs, sub-module names and parameter shapes, so `build_from_cfg` and reference
checkpoints work unchanged.  The nn.Conv2d / nn.BatchNorm2d children only HOLD parameters; the
forward pass runs on packed device buffers:
  * activations are physical NHWC (exposed as logical NCHW `channels_last` tensors),
  * frozen BatchNorm (eval mode, eps 1e-5: mmdet/models/utils/norm.py:43) is folded into the
    conv weight / an f32 bias at pack time,
  * bias + residual add + ReLU run in the GEMM epilogue (one launch per conv, no BN/ReLU/add
    kernels; the reference issues ~4 launches per conv).
Inference only: BN layers are always treated as frozen (both configs set norm_eval=True,
requires_grad=False).  There is no CPU path (native.py raises on CPU tensors).
"""
import torch
import torch.nn as nn

from . import native
from .registry import BACKBONES, SHARED_HEADS

ARCH_SETTINGS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
STEM_KP = 192  # 7*7*3 = 147 patch elements padded to a K-step multiple


def as_nhwc(x, dtype):
    """Logical [B,C,H,W] tensor -> physical [B,H,W,C] contiguous tensor of `dtype` (no copy when it already is)."""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v if x.dtype == dtype else native.cast(v, dtype)
    return native.nchw_to_nhwc(x.contiguous(), dtype)


def as_logical(y):
    """Physical [B,H,W,C] -> logical [B,C,H,W] (channels_last strides, no copy)."""
    return y.permute(0, 3, 1, 2)


def fold_conv_bn(conv, bn, dtype):
    """(w [Cout,KH,KW,Cin] in `dtype`, bias [Cout] f32) with eval-mode BN folded in."""
    if bn is not None and conv.bias is None and dtype == torch.bfloat16 and conv.weight.is_cuda \
            and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous() and not bn.training:
        # the same numbers in one launch (or none; bf16 only -- the half instance of the pack kernel rounds product and conversion in ONE
        # step, v_fma_mixlo_f16, where torch rounds twice: tools/probe_fold.py): the frozen BatchNorm's affine is cached until one of its tensors changes, and the
        # permute + scale + rounding is the pack kernel the training path uses -- inside a training iteration the operand this
        # iteration's weight table already holds (the no-grad res5 pass of HNMBRCNN.forward_train repacks after every update)
        from . import train_ops as TO
        s_, t_ = TO._frozen_bn_affine(bn)
        w_ = conv.weight.detach()
        eff = TO.table_operand(conv.weight, s_, tuple(w_.shape), dtype)
        return (eff if eff is not None else native.pack_conv_weight(w_, s_, dtype)), t_
    w = conv.weight.detach().float()
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w = w * scale[:, None, None, None]
    else:
        bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None and conv.bias is not None:
        bias = bias + conv.bias.detach().float() * scale
    return native.as_operand(w.permute(0, 2, 3, 1), dtype), bias.contiguous()


class PackedMixin(object):
    """Lazily packed device weights, rebuilt when the dtype/device changes or a state dict is loaded."""

    def _init_packed(self):
        self._packed = None
        self._packed_key = None
        self.compute_dtype = torch.bfloat16
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_packed())

    def _drop_packed(self):
        if self._packed is not None:
            from . import graphs   # captured hipGraphs read the packed buffers by address: they must not replay after this
            graphs.invalidate_all('packed weights of %s were rebuilt' % type(self).__name__)
        self._packed = None
        self._packed_ready = None

    _packed_ready = None

    def packed(self, device):
        """The packed weights, built on first use.  The pack kernels run on the stream that first asks (with frame groups that
        is a SIDE stream: the host enqueues the side groups first); any other stream that uses the same tensors afterwards
        -- the main stream's group, a second window lane -- must be ordered behind them, or it reads weights that are still
        being written: an event recorded behind the pack, awaited once per consuming stream."""
        key = (self.compute_dtype, str(device))
        if self._packed is None or self._packed_key != key:
            with torch.no_grad():
                self._packed = self._pack(self.compute_dtype)
            self._packed_key = key
            self._packed_ready = None
            if getattr(device, 'type', None) == 'cuda':
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                self._packed_ready = (ev, {native._raw_stream(device)})
        elif self._packed_ready is not None:
            ev, seen = self._packed_ready
            rs = native._raw_stream(device)
            if rs not in seen:
                torch.cuda.current_stream(device).wait_event(ev)
                seen.add(rs)
        return self._packed


def set_compute_dtype(module, dtype):
    """The operand format of every packed module below `module` -- the precision ladder of include/hvr_hip.h:
    torch.bfloat16 (benchmark dtype, every dedicated kernel), torch.float16 (half operands on the tile engine),
    native.SPLIT (split half: three half MFMAs per product, f32-grade results) or torch.float32 (exact-f32 MFMA).
    Modules convert their inputs on entry, so sub-trees may differ: e.g. set_compute_dtype(model.backbone, native.SPLIT)
    after set_compute_dtype(model, torch.float16)."""
    assert dtype in native.COMPUTE_DTYPES, dtype
    for m in module.modules():
        if isinstance(m, PackedMixin):
            m.compute_dtype = dtype
            m._drop_packed()
    return module


def enable_training(module):
    """Marks the parameters the reference trains as requires_grad (the builders freeze everything for inference):
    all conv / linear weights and biases EXCEPT BatchNorm parameters (norm_cfg requires_grad=False, norm_eval=True in both
    configs) and a ResNet's stem + first `frozen_stages` stages (resnet.py:484-494).  Use with
    set_compute_dtype(module, torch.float32); the forward_train_* methods then build autograd graphs of HIP ops."""
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            continue
        for p in m.parameters(recurse=False):
            p.requires_grad = True
    for m in module.modules():
        if isinstance(m, ResNet):
            frozen = [m.conv1, m.bn1] + [getattr(m, 'layer%d' % i) for i in range(1, m.frozen_stages + 1)]
            for f in frozen:
                for p in f.parameters():
                    p.requires_grad = False
    for m in module.modules():
        if type(m).__name__ == 'HNMBRCNN':
            # this detector's forward_train computes C4 under no_grad and never adds an RPN loss (hnmb_rcnn.py:269-283,318-325):
            # backbone and RPN parameters never receive a gradient there, and torch.optim.SGD skips gradient-less parameters
            # (no weight decay either) -- keep them out of the flat parameter set so the fused update leaves them alone too
            for part in (m.backbone, getattr(m, 'rpn_head', None)):
                if part is not None:
                    for p in part.parameters():
                        p.requires_grad = False
    return module


class Bottleneck(nn.Module, PackedMixin):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', norm_cfg=None):
        super(Bottleneck, self).__init__()
        assert style in ['pytorch', 'caffe']
        self.inplanes, self.planes, self.stride, self.dilation, self.style = inplanes, planes, stride, dilation, style
        # caffe: stride on the first 1x1 (resnet.py:127-132)
        self.conv1_stride, self.conv2_stride = (1, stride) if style == 'pytorch' else (stride, 1)
        eps = (norm_cfg or {}).get('eps', 1e-5)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=self.conv1_stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, eps=eps)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=self.conv2_stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=eps)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, eps=eps)
        self.downsample = downsample
        self._init_packed()

    def _pack(self, dtype):
        p = dict(c1=fold_conv_bn(self.conv1, self.bn1, dtype), c2=fold_conv_bn(self.conv2, self.bn2, dtype),
                 c3=fold_conv_bn(self.conv3, self.bn3, dtype))
        if self.downsample is not None:
            p['ds'] = fold_conv_bn(self.downsample[0], self.downsample[1], dtype)
            # the projection shortcut as a second K segment of the closing 1x1 (hvr_bottleneck_tail): rows [W3 | Wd]
            p['tail'] = (torch.cat([p['c3'][0].reshape(p['c3'][0].shape[0], -1), p['ds'][0].reshape(p['ds'][0].shape[0], -1)], 1).contiguous(),
                         (p['c3'][1] + p['ds'][1]).contiguous())
        return p

    def forward_nhwc(self, x, out=None, h1=None, nxt=None):
        """out: where the block's output goes (a contiguous [B,OH,OW,4*planes] tensor), e.g. a frame group's slice of a map.
        h1: this block's conv1 output, when the previous block's tail already computed it.  nxt: the block that consumes
        this one's output; the return value is then (y, h1 of nxt or None) -- nxt's conv1 rides on this block's tail when
        hvr_bottleneck_tail_next has a kernel for the shapes."""
        p = self.packed(x.device)
        dst = out
        out = h1 if h1 is not None else native.conv2d_nhwc(x, p['c1'][0], p['c1'][1], relu=True, stride=self.conv1_stride)
        out = native.conv2d_nhwc(out, p['c2'][0], p['c2'][1], relu=True, stride=self.conv2_stride, pad=self.dilation,
                                 dil=self.dilation)
        if nxt is not None and self.fuse_next and nxt.conv1_stride == 1:
            pn = nxt.packed(x.device)
            wn, bn = pn['c1'][0].reshape(pn['c1'][0].shape[0], -1), pn['c1'][1]
            if self.downsample is not None:
                args = (out, x, None, p['tail'][0], p['tail'][1], self.stride) if self.fuse_tail else None
                if x.dtype == native.SPLIT:
                    # split half has no second-K-segment tail: the projection is its own conv and enters the fused closing 1x1 +
                    # next conv1 as the residual (the identity form of hvr_bottleneck_tail_next)
                    ident = native.conv2d_nhwc(x, p['ds'][0], p['ds'][1], relu=False, stride=self.stride)
                    w3, b3 = p['c3'][0].reshape(p['c3'][0].shape[0], -1), p['c3'][1]
                    if native.bottleneck_tail_next_supported(out, None, ident, w3, b3, 1, wn, bn):
                        return native.bottleneck_tail_next(out, None, ident, w3, b3, wn, bn, out=dst)
                    y = native.conv2d_nhwc(out, p['c3'][0], p['c3'][1], resid=ident, relu=True, out=dst)
                    return y, None
            else:
                args = (out, None, x, p['c3'][0].reshape(p['c3'][0].shape[0], -1), p['c3'][1], 1)
            if args is not None and native.bottleneck_tail_next_supported(*args, wn, bn):
                return native.bottleneck_tail_next(args[0], args[1], args[2], args[3], args[4], wn, bn, stride2=args[5], out=dst)
        y = self._tail(x, out, p, dst)
        return y if nxt is None else (y, None)

    def _tail(self, x, out, p, dst):
        identity = x
        if self.downsample is not None:
            if self.fuse_tail and native.bottleneck_tail_supported(out, x, p['tail'][0], p['tail'][1], self.stride):
                # relu(conv3(out) + downsample(x)) in one pass: the identity map is never written (resnet.py:248-264)
                return native.bottleneck_tail(out, x, p['tail'][0], p['tail'][1], stride2=self.stride, relu=True, out=dst)
            identity = native.conv2d_nhwc(x, p['ds'][0], p['ds'][1], relu=False, stride=self.stride)
        return native.conv2d_nhwc(out, p['c3'][0], p['c3'][1], resid=identity, relu=True, out=dst)

    fuse_next = True   # class attributes (tests flip them to compare the fused kernels with the per-conv path); no environment switch
    fuse_tail = True

    def forward(self, x):
        return as_logical(self.forward_nhwc(as_nhwc(x, self.compute_dtype)))

    def forward_train_nhwc(self, x):
        """The block as an autograd graph of HIP convs (train_ops.conv_bn): frozen BatchNorm statistics and affine
        (norm_eval=True, requires_grad=False in both configs), trainable conv weights; f32, physical NHWC."""
        from . import train_ops as TO
        out = TO.conv_bn(x, self.conv1, self.bn1, relu=True)
        out = TO.conv_bn(out, self.conv2, self.bn2, relu=True)
        identity = x if self.downsample is None else TO.conv_bn(x, self.downsample[0], self.downsample[1])
        return TO.conv_bn(out, self.conv3, self.bn3, resid=identity, relu=True)


def make_res_layer(block, inplanes, planes, blocks, stride=1, dilation=1, style='pytorch', norm_cfg=None, **_unused):
    downsample = None
    eps = (norm_cfg or {}).get('eps', 1e-5)
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * block.expansion, eps=eps))
    layers = [block(inplanes, planes, stride, dilation, downsample, style=style, norm_cfg=norm_cfg)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(inplanes, planes, 1, dilation, style=style, norm_cfg=norm_cfg))
    return nn.Sequential(*layers)


def _freeze(module):
    module.eval()
    for p in module.parameters():
        p.requires_grad = False


@BACKBONES.register_module
class ResNet(nn.Module, PackedMixin):
    """Same kwargs as the reference ResNet (resnet.py:372-395); dcn / gcb / gen_attention must be None
    (both hot-path configs leave them unset)."""
    arch_settings = {d: (Bottleneck, s) for d, s in ARCH_SETTINGS.items()}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), gcb=None, stage_with_gcb=(False, False, False, False),
                 gen_attention=None, stage_with_gen_attention=((), (), (), ()), with_cp=False, zero_init_residual=True):
        super(ResNet, self).__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        if dcn is not None or gcb is not None or gen_attention is not None or conv_cfg is not None:
            raise NotImplementedError('dcn / gcb / gen_attention / conv_cfg are outside the HVR hot path')
        if norm_cfg.get('type', 'BN') != 'BN':
            raise NotImplementedError('only frozen BatchNorm is supported')
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages and max(out_indices) < num_stages
        assert in_channels == 3
        self.depth, self.num_stages, self.strides, self.dilations = depth, num_stages, strides, dilations
        self.out_indices, self.style, self.frozen_stages, self.norm_eval = out_indices, style, frozen_stages, norm_eval
        self.zero_init_residual = zero_init_residual
        block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=norm_cfg.get('eps', 1e-5))
        self.res_layers = []
        inplanes = 64
        for i, nb in enumerate(self.stage_blocks):
            planes = 64 * 2 ** i
            name = 'layer{}'.format(i + 1)
            self.add_module(name, make_res_layer(block, inplanes, planes, nb, stride=strides[i], dilation=dilations[i],
                                                 style=style, norm_cfg=norm_cfg))
            inplanes = planes * block.expansion
            self.res_layers.append(name)
        self.feat_dim = block.expansion * 64 * 2 ** (len(self.stage_blocks) - 1)
        self.fused_stem = True  # bf16 only: conv1 + bn1 + relu + maxpool in one kernel
        _freeze(self)
        self._init_packed()

    def init_weights(self, pretrained=None):
        """resnet.py:496-520 (pretrained checkpoints are loaded by the caller via load_state_dict)."""
        if pretrained is not None:
            raise NotImplementedError('load checkpoints with load_state_dict (keys match the reference)')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
        set_compute_dtype(self, self.compute_dtype)

    def _pack(self, dtype):
        w, b = fold_conv_bn(self.conv1, self.bn1, torch.float32)  # [64,7,7,3]
        wp = torch.zeros((64, STEM_KP), dtype=torch.float32, device=w.device)
        wp[:, :147] = w.reshape(64, 147)
        # fused bf16 stem: [n][ky][kx*4 + c], zero weight for the pad channel (c = 3) and the pad tap (kx = 7)
        wf = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=w.device)
        wf[:, :, :7, :3] = w
        if dtype == native.SPLIT:
            fused, fused_bias = native.stem_split_weights(wf.view(64, 7, 32)), native.stem_split_bias(b)
        else:
            fused, fused_bias = wf.view(64, 7, 32).to(dtype if dtype in (torch.bfloat16, torch.float16) else torch.bfloat16).contiguous(), b
        return dict(stem=(native.as_operand(wp, dtype), b), fused=fused, fused_bias=fused_bias)

    def out_shape_nhwc(self, B, H, W):
        """Physical [B,h,w,C] shape of the LAST returned map for a [B,3,H,W] input (stem 7x7/2 + pool 3x3/2, then one
        stride per stage), or None when the net returns several maps."""
        if len(self.out_indices) != 1:
            return None
        last = self.out_indices[0]
        h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        for i in range(last + 1):
            h, w = (h - 1) // self.strides[i] + 1, (w - 1) // self.strides[i] + 1
        return (B, h, w, 64 * 2 ** last * 4)

    def forward(self, x, out=None):
        """x [B,3,H,W] f32 -> tuple of logical-NCHW feature maps (resnet.py:522-533).  out: a contiguous NHWC tensor of
        out_shape_nhwc(...) in the compute dtype that receives the (single) returned map, e.g. a slice of a larger batch."""
        if not x.is_cuda:
            raise NotImplementedError('ResNet runs on the GPU only (no CPU fallback)')
        p = self.packed(x.device)
        dt = self.compute_dtype
        if dt in (torch.bfloat16, torch.float16, native.SPLIT) and self.fused_stem:
            y = native.stem_fused(x.contiguous().float(), p['fused'], p['fused_bias'])
        else:  # generic route (f32 / split-half modes): patch matrix + GEMM + pooling
            cols, OH, OW = native.im2col_stem(x.contiguous().float(), dt, STEM_KP)
            if dt == native.SPLIT:   # pooled in f32 (max does not act per half plane): the GEMM hands its f32 tile over directly
                y = native.gemm(cols, p['stem'][0], p['stem'][1], relu=True, out_f32=True).view(x.shape[0], OH, OW, 64)
                y = native.cast(native.maxpool3x3s2_nhwc(y), dt)
            else:
                y = native.gemm(cols, p['stem'][0], p['stem'][1], relu=True).view(x.shape[0], OH, OW, 64)
                y = native.maxpool3x3s2_nhwc(y)
        outs = []
        if out is not None:
            assert len(self.out_indices) == 1, 'out= needs a single returned map'
        for i, name in enumerate(self.res_layers):
            blocks = list(getattr(self, name))
            h1 = None
            for j, blk in enumerate(blocks):
                last = out is not None and i == self.out_indices[0] and j == len(blocks) - 1
                if j + 1 < len(blocks):  # the next block's conv1 rides on this block's tail where a kernel exists
                    y, h1 = blk.forward_nhwc(y, h1=h1, nxt=blocks[j + 1])
                else:
                    y = blk.forward_nhwc(y, out=out if last else None, h1=h1)
            if i in self.out_indices:
                outs.append(as_logical(y))
            if out is not None and i == self.out_indices[0]:
                break
        return tuple(outs)

    def forward_train_nhwc(self, x):
        """Training forward in the compute dtype (f32 = parity mode, bf16 = throughput mode with f32 master weights): the
        frozen stem and stages (`frozen_stages`, BatchNorm everywhere) run the inference kernels without a graph, the remaining stages are autograd graphs of HIP convs (Bottleneck.forward_train_nhwc).
        -> the last out_indices map, physical NHWC."""
        p = self.packed(x.device)
        dt = self.compute_dtype
        with torch.no_grad():
            if dt in (torch.bfloat16, torch.float16, native.SPLIT) and self.fused_stem:
                y = native.stem_fused(x.contiguous().float(), p['fused'], p['fused_bias'])
            else:
                cols, OH, OW = native.im2col_stem(x.contiguous().float(), dt, STEM_KP)
                y = native.gemm(cols, p['stem'][0], p['stem'][1], relu=True).view(x.shape[0], OH, OW, 64)
                y = native.maxpool3x3s2_nhwc(y)
            for i, name in enumerate(self.res_layers):
                if i + 1 <= self.frozen_stages:
                    for blk in getattr(self, name):
                        y = blk.forward_nhwc(y)
        for i, name in enumerate(self.res_layers):
            if i + 1 > self.frozen_stages:
                for blk in getattr(self, name):
                    y = blk.forward_train_nhwc(y)
        return y

    def train(self, mode=True):
        super(ResNet, self).train(False)  # BatchNorm stays in eval mode (norm_eval=True); forward_train_* build the graphs
        return self


@SHARED_HEADS.register_module
class ResLayer(nn.Module, PackedMixin):
    """res5 applied to the whole stride-16 map + the 1x1 2048->256 `new_layer_1` (res_layer.py:16-52,67-74)."""

    def __init__(self, depth, stage=3, stride=2, dilation=1, style='pytorch', norm_cfg=dict(type='BN', requires_grad=True),
                 norm_eval=True, with_cp=False, external_conv=False, dcn=None):
        super(ResLayer, self).__init__()
        if dcn is not None:
            raise NotImplementedError('dcn is outside the HVR hot path')
        self.norm_eval, self.norm_cfg, self.stage, self.external_conv = norm_eval, norm_cfg, stage, external_conv
        self.fp16_enabled = False
        stage_block = ARCH_SETTINGS[depth][stage]
        planes = 64 * 2 ** stage
        inplanes = 64 * 2 ** (stage - 1) * Bottleneck.expansion
        self.add_module('layer{}'.format(stage + 1),
                        make_res_layer(Bottleneck, inplanes, planes, stage_block, stride=stride, dilation=dilation, style=style,
                                       norm_cfg=norm_cfg))
        if external_conv:
            # ConvModule(2048, 256, 1): conv + bias + ReLU, parameters under new_layer_1.conv.*
            self.new_layer_1 = nn.Module()
            self.new_layer_1.conv = nn.Conv2d(2048, 256, 1)
        _freeze(self)
        self._init_packed()

    def init_weights(self, pretrained=None):
        if pretrained is not None:
            raise NotImplementedError('load checkpoints with load_state_dict (keys match the reference)')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        set_compute_dtype(self, self.compute_dtype)

    def _pack(self, dtype):
        if not self.external_conv:
            return {}
        return dict(ext=fold_conv_bn(self.new_layer_1.conv, None, dtype))

    def forward(self, x):
        if not x.is_cuda:
            raise NotImplementedError('ResLayer runs on the GPU only (no CPU fallback)')
        y = as_nhwc(x, self.compute_dtype)
        for blk in getattr(self, 'layer{}'.format(self.stage + 1)):
            y = blk.forward_nhwc(y)
        if self.external_conv:
            w, b = self.packed(x.device)['ext']
            y = native.conv2d_nhwc(y, w, b, relu=True)
        return as_logical(y)

    def forward_train_nhwc(self, y):
        """res5 + the external 1x1 conv as an autograd graph of HIP convs (f32, physical NHWC in and out)."""
        from . import train_ops as TO
        for blk in getattr(self, 'layer{}'.format(self.stage + 1)):
            y = blk.forward_train_nhwc(y)
        if self.external_conv:
            y = TO.conv_bias(y, self.new_layer_1.conv, relu=True)
        return y

    def train(self, mode=True):
        super(ResLayer, self).train(False)
        return self
