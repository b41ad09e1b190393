"""Frame ingest on the device.

The reference's test pipeline (configs/faster_rcnn_r101_hrnmp_c5.py:193-201) is, per frame, on a DataLoader worker's CPU:
LoadImageFromFile (uint8 BGR) -> Resize(img_scale=(1000, 600), keep_ratio=True) = mmcv.imrescale -> cv2.resize(INTER_LINEAR)
-> RandomFlip(0) -> Normalize(mean, std, to_rgb) -> Pad(size_divisor=16) -> ImageToTensor -> Collect, then a 7 MB f32
host-to-device copy.  `FrameIngest` keeps the decode on the host and moves everything after it into one HIP kernel
(`hvr_ingest_frame`): the uint8 frame (1/4 of the bytes, before the upscale at that) is copied to the device and the
resized, mean-subtracted, zero-padded [1, 3, H, W] f32 tensor is written once, together with the img_meta the detector
needs (transforms.py:118-124,273-276: img_shape, pad_shape, scale_factor, flip).
"""
import numpy as np
import torch

from . import native


def rescale_size(h, w, scale):
    """mmcv.imrescale's target size for scale = (long edge, short edge): factor = min(long / max(h, w), short / min(h, w)),
    new (w, h) = int(w * factor + 0.5), int(h * factor + 0.5).  -> (new_h, new_w, factor)."""
    max_long, max_short = max(scale), min(scale)
    factor = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5), factor


class FrameIngest(object):
    """Resize + RandomFlip(0) + Normalize + Pad + ImageToTensor + Collect of the reference's test pipeline as one call.

    frame: uint8 [H, W, 3] BGR -- a numpy array / CPU tensor (copied to `device`, asynchronously when pinned) or a tensor
    already on the device.  -> dict(img=[1, 3, pad_h, pad_w] f32 on the device, img_meta=dict(...))."""

    def __init__(self, img_scale=(1000, 600), mean=(103.06, 115.90, 123.15), std=(1.0, 1.0, 1.0), to_rgb=False, size_divisor=16,
                 keep_ratio=True, device='cuda:0'):
        if not keep_ratio:
            raise NotImplementedError('keep_ratio=False is not used by the HVRNet configs')
        self.img_scale, self.mean, self.std, self.to_rgb = tuple(img_scale), tuple(mean), tuple(std), bool(to_rgb)
        self.size_divisor, self.device = int(size_divisor), device

    def __call__(self, frame):
        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise ValueError('expected a uint8 [H, W, 3] frame, got %s %s' % (frame.dtype, tuple(frame.shape)))
        if not frame.is_cuda:
            frame = frame.contiguous().to(self.device, non_blocking=True)
        h, w = int(frame.shape[0]), int(frame.shape[1])
        nh, nw, factor = rescale_size(h, w, self.img_scale)
        d = self.size_divisor
        ph, pw = -(-nh // d) * d, -(-nw // d) * d
        img = native.ingest_frame(frame.contiguous(), (nh, nw), (ph, pw), self.mean, self.std, self.to_rgb)
        meta = dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3), scale_factor=factor, flip=False,
                    img_norm_cfg=dict(mean=np.array(self.mean, dtype=np.float32), std=np.array(self.std, dtype=np.float32),
                                      to_rgb=self.to_rgb))
        return dict(img=img, img_meta=meta)
