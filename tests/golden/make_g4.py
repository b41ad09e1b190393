"""G4: RoIAlign fixtures.  UNLIKE G1-G3 / G5-G14 these are NOT outputs of the reference: its RoIAlign has no CPU
path (mmdet/ops/roi_align/roi_align.py:27-28) and roi_align_kernel.cu cannot be built here (CUDA + removed THC headers), so
the expected outputs come from this build's C restatement of ROIAlignForward (oracle/oracle_ref.c, roi_align_kernel.cu:16-118)
and the file says so: `roi_align_source = "oracle"`.  What G4 pins is (a) that restatement against drift and (b) the HIP kernels
against it on SURVEY's cases (C = 8, 15x15 and 38x63 maps, 64 RoIs incl. degenerate / out-of-bounds / full-image boxes,
sample_num 2 and 0, out 7) plus the shapes of the reference's own gradcheck recipe (roi_align/gradcheck.py:11-30).

    python tests/golden/make_g4.py        # runs on the CPU, needs only oracle/libhvr_oracle.so
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hvr_oracle as O  # noqa: E402


def rois_case(B, H, W, n, seed, stride=16.0):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand((n, 2), generator=g) * torch.tensor([W * stride, H * stride])
    wh = torch.rand((n, 2), generator=g) * torch.tensor([W * stride / 2, H * stride / 2]) + 1
    rois = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), xy, xy + wh], 1)
    iw, ih = W * stride, H * stride
    rois[0, 1:] = torch.tensor([0., 0., iw - 1, ih - 1])                    # full image
    rois[1, 1:] = torch.tensor([50., 60., 50., 60.])                         # single pixel
    rois[2, 1:] = torch.tensor([120., 90., 40., 30.])                        # malformed (x2 < x1): size clamps to 0
    rois[3, 1:] = torch.tensor([-200., -150., 80., 60.])                     # sticks out top-left
    rois[4, 1:] = torch.tensor([iw - 40, ih - 30, iw + 300, ih + 200])       # sticks out bottom-right
    rois[5, 1:] = torch.tensor([iw + 50, ih + 50, iw + 90, ih + 90])         # fully outside: all samples out of bounds -> 0
    rois[6, 1:] = torch.tensor([-15., -15., -2., -2.])                       # every sample in (-1, 0): clamped to pixel (0, 0), not zeroed
    rois[7, 1:] = torch.tensor([0.25, 0.25, 0.5, 0.5])                       # sub-pixel box
    return rois


def main():
    out = dict(roi_align_source=np.array('oracle'))
    for name, (B, C, H, W) in dict(m15=(2, 8, 15, 15), m38=(3, 4, 38, 63)).items():
        g = torch.Generator().manual_seed(400 + H)
        feat = torch.randn((B, C, H, W), generator=g)
        rois = rois_case(B, H, W, 64, 410 + H)
        out[name + '_feat'], out[name + '_rois'] = feat.numpy(), rois.numpy()
        for sn in (2, 0):
            out['%s_out_s%d' % (name, sn)] = O.roi_align(feat, rois, 7, 1.0 / 16, sn).numpy()
    # the reference's gradcheck recipe (gradcheck.py:11-24): 15x15 map, scale 1/8, 2 images, 20 RoIs in the lower-right half, out 3
    rs = np.random.RandomState(42)
    feat_size, scale, num_imgs, num_rois = 15, 1.0 / 8, 2, 20
    img_size = feat_size / scale
    batch_ind = rs.randint(num_imgs, size=(num_rois, 1))
    rois = rs.rand(num_rois, 4) * img_size * 0.5
    rois[:, 2:] += img_size * 0.5
    rois = torch.from_numpy(np.hstack((batch_ind, rois))).float()
    feat = torch.randn((num_imgs, 16, feat_size, feat_size), generator=torch.Generator().manual_seed(43))
    out['gc_feat'], out['gc_rois'] = feat.numpy(), rois.numpy()
    for sn in (0, 2):
        out['gc_out_s%d' % sn] = O.roi_align(feat, rois, 3, scale, sn).numpy()
    np.savez(os.path.join(ROOT, 'tests', 'golden', 'g4_roi_align.npz'), **out)
    print('wrote g4_roi_align.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
