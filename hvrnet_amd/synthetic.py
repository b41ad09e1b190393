"""Seeded synthetic weights and frames for benchmarks, smoke and parity runs.

There is no network for datasets or checkpoints, so the benchmark uses random-init weights
of the reference architecture and synthetic frames of the reference's input shape
(SURVEY.md 8(d)).  Everything is generated on the CPU torch generator so that the GPU path,
the CPU oracle and the golden-vector script (which feeds the same tensors to the reference's
own modules) see bit-identical tensors.

State-dict keys and shapes are the reference's checkpoint contract:
  backbone.*      mmdet/models/backbones/resnet.py:456-466,269-329 (R101, 3 stages, caffe)
  shared_head.*   mmdet/models/shared_heads/res_layer.py:37-52
  rpn_head.*      mmdet/models/anchor_heads/rpn_head.py:18-23
  bbox_head.*     mmdet/models/bbox_heads/selsa_bbox_head.py:45-89, hrnmp_bbox_head.py:121-189

The reference's own init (`zero_init_residual=True`, resnet.py:513-518; N(0, 0.01) head linears)
makes random-init parity degenerate (dead residual branches, uniform attention rows), so the BN
statistics are randomised and the head weights are scaled up; the scales below were chosen so
that RPN scores, attention rows and class scores are all well away from uniform/saturated.
"""
import math

import torch

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

# multipliers on the reference's N(0, 0.01) init
NEW_LAYER_GAIN = 0.0125  # shared_head.new_layer_1: brings RoI features to unit scale
HEAD_FC1_GAIN = 1.0      # fc_new_1 (12544 inputs)
HEAD_FCK_GAIN = 3.0      # fc_new_2..4 (1024 inputs)
HEAD_QK_GAIN = 8.0      # q / k projections: attention logit std ~3 (peaky rows, not one-hot)
HEAD_OUT_GAIN = 3.0      # linear_out_k (1x1 conv): relation update comparable to the residual
HEAD_CLS_GAIN = 7.0      # fc_cls: spreads class scores
HEAD_REG_GAIN = 3.5
RPN_CONV_GAIN = 1.0
RPN_CLS_GAIN = 0.3
RPN_REG_GAIN = 0.07


def _conv(g, cout, cin, k):
    std = math.sqrt(2.0 / (cout * k * k))  # kaiming_normal_, mode='fan_out', relu (mmcv kaiming_init)
    return torch.randn((cout, cin, k, k), generator=g) * std


def _bn(g, sd, prefix, c, gamma_scale=1.0):
    sd[prefix + '.weight'] = (torch.rand(c, generator=g) + 0.5) * gamma_scale
    sd[prefix + '.bias'] = torch.randn(c, generator=g) * 0.1
    sd[prefix + '.running_mean'] = torch.randn(c, generator=g) * 0.1
    sd[prefix + '.running_var'] = torch.rand(c, generator=g) + 0.5
    sd[prefix + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)


def _res_layer(g, sd, prefix, inplanes, planes, blocks):
    for i in range(blocks):
        p = '%s.%d' % (prefix, i)
        cin = inplanes if i == 0 else planes * 4
        sd[p + '.conv1.weight'] = _conv(g, planes, cin, 1)
        _bn(g, sd, p + '.bn1', planes)
        sd[p + '.conv2.weight'] = _conv(g, planes, planes, 3)
        _bn(g, sd, p + '.bn2', planes)
        sd[p + '.conv3.weight'] = _conv(g, planes * 4, planes, 1)
        # a live but damped residual branch keeps 33 stacked blocks from blowing up
        _bn(g, sd, p + '.bn3', planes * 4, gamma_scale=0.25)
        if i == 0:
            sd[p + '.downsample.0.weight'] = _conv(g, planes * 4, cin, 1)
            _bn(g, sd, p + '.downsample.1', planes * 4)


def _linear(g, sd, prefix, cout, cin, gain, conv=False):
    w = torch.randn((cout, cin), generator=g) * (0.01 * gain)
    sd[prefix + '.weight'] = w.view(cout, cin, 1, 1) if conv else w
    sd[prefix + '.bias'] = torch.randn(cout, generator=g) * 0.01


def synth_state_dict(head='hvr', seed=0, depth=101, num_classes=31, fc_dim=1024, roi_feat=256 * 49):
    """Full detector state dict (f32, CPU) with the reference's key names."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    blocks = STAGE_BLOCKS[depth]
    sd['backbone.conv1.weight'] = _conv(g, 64, 3, 7)
    _bn(g, sd, 'backbone.bn1', 64)
    inplanes = 64
    for i in range(3):
        _res_layer(g, sd, 'backbone.layer%d' % (i + 1), inplanes, 64 * 2 ** i, blocks[i])
        inplanes = 64 * 2 ** i * 4
    _res_layer(g, sd, 'shared_head.layer4', 1024, 512, blocks[3])
    sd['shared_head.new_layer_1.conv.weight'] = _conv(g, 256, 2048, 1) * NEW_LAYER_GAIN
    sd['shared_head.new_layer_1.conv.bias'] = torch.randn(256, generator=g) * 0.01
    # RPN (12 anchors)
    w = torch.randn((512, 1024, 3, 3), generator=g) * (0.01 * RPN_CONV_GAIN)
    sd['rpn_head.rpn_conv.weight'] = w
    sd['rpn_head.rpn_conv.bias'] = torch.zeros(512)
    _linear(g, sd, 'rpn_head.rpn_cls', 12, 512, RPN_CLS_GAIN, conv=True)
    _linear(g, sd, 'rpn_head.rpn_reg', 48, 512, RPN_REG_GAIN, conv=True)
    # relation head
    stages = 4 if head == 'hvr' else 2
    for k in range(1, stages + 1):
        _linear(g, sd, 'bbox_head.fc_new_%d' % k, fc_dim, roi_feat if k == 1 else fc_dim,
                HEAD_FC1_GAIN if k == 1 else HEAD_FCK_GAIN)
        p = 'bbox_head.selsa_%d.' % k
        _linear(g, sd, p + 'q_data_fc_%d' % k, fc_dim, fc_dim, HEAD_QK_GAIN)
        _linear(g, sd, p + 'k_data_fc_%d' % k, fc_dim, fc_dim, HEAD_QK_GAIN)
        _linear(g, sd, p + 'linear_out_%d' % k, fc_dim, fc_dim, HEAD_OUT_GAIN, conv=True)
    _linear(g, sd, 'bbox_head.fc_cls', num_classes, fc_dim, HEAD_CLS_GAIN)
    _linear(g, sd, 'bbox_head.fc_reg', 4, fc_dim, HEAD_REG_GAIN)
    if head == 'hvr':
        _linear(g, sd, 'bbox_head.fc_cls_2', num_classes, fc_dim, HEAD_CLS_GAIN)
        _linear(g, sd, 'bbox_head.fc_reg_2', 4, fc_dim, HEAD_REG_GAIN)
    return sd


def synth_frame(index, seed=0, img_hw=(600, 1000), pad_hw=(608, 1008)):
    """One mean-subtracted BGR-scale frame [1,3,pad_h,pad_w] f32 (zero padded, Pad(size_divisor=16))."""
    g = torch.Generator().manual_seed(seed * 100003 + index)
    h, w = img_hw
    # low-frequency content + noise, so RPN / RoI features are not white noise
    coarse = torch.randn((1, 3, h // 40 + 1, w // 40 + 1), generator=g)
    img = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=False) * 40.0
    img = img + torch.randn((1, 3, h, w), generator=g) * 15.0
    out = torch.zeros((1, 3, pad_hw[0], pad_hw[1]))
    out[:, :, :h, :w] = img
    return out


def synth_meta(img_hw=(600, 1000), pad_hw=(608, 1008)):
    return dict(img_shape=(img_hw[0], img_hw[1], 3), ori_shape=(img_hw[0], img_hw[1], 3), pad_shape=(pad_hw[0], pad_hw[1], 3),
                scale_factor=1.0, flip=False)
