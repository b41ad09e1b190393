"""SingleRoIExtractor (mmdet/models/roi_extractors/single_level.py:11-107), single level only.

`roi_layer` is resolved by name from `hvrnet_amd.ops`, exactly as the reference resolves it from
`mmdet.ops` (single_level.py:45-52).  With one level the reference returns
`self.roi_layers[0](feats[0], rois)` (single_level.py:91-92); multi-level mapping is outside the
hot path (`featmap_strides=[16]` in both configs).
"""
import torch.nn as nn

from . import ops
from .registry import ROI_EXTRACTORS


@ROI_EXTRACTORS.register_module
class SingleRoIExtractor(nn.Module):

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super(SingleRoIExtractor, self).__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides):
        cfg = layer_cfg.copy()
        layer_type = cfg.pop('type')
        assert hasattr(ops, layer_type)
        layer_cls = getattr(ops, layer_type)
        return nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def forward(self, feats, rois, roi_scale_factor=None):
        if len(feats) != 1:
            raise NotImplementedError('multi-level RoI extraction is outside the HVR hot path')
        return self.roi_layers[0](feats[0], rois)
