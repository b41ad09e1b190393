"""hvrnet_amd -- MI355X-native HVRNet video-detection forward path.

Importing the package registers the hot-path components under the reference's names
(ResNet, ResLayer, RPNHead, SingleRoIExtractor, SelsaBBoxHead, HRNMPBBoxHead, SelsaRCNN, HNMBRCNN).
"""
__version__ = '0.1.0'

from . import registry  # noqa: F401
from .backbone import ResLayer, ResNet, enable_training, set_compute_dtype  # noqa: F401
from .bbox_heads import BBoxHead, HRNMPBBoxHead, SelsaBBoxHead  # noqa: F401
from .config import Config, hvr_config, selsa_config  # noqa: F401
from .detectors import HNMBRCNN, SelsaRCNN  # noqa: F401
from .registry import build_detector  # noqa: F401
from .roi_extractor import SingleRoIExtractor  # noqa: F401
from .rpn_head import RPNHead  # noqa: F401


def build_model(cfg, state_dict=None, dtype=None, device='cuda:0'):
    """Detector from a Config (reference file or built-in), optionally loading a reference-keyed state dict."""
    import torch
    model = build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    model.eval()
    if dtype is not None:
        set_compute_dtype(model, dtype)
    if device is not None and torch.cuda.is_available():
        model.to(device)
    return model
