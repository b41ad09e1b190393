"""Does v_mfma_f32_16x16x32_f16 honour subnormal half inputs?  a = 2^-20 (subnormal in half), w = 1 -> K * 2^-20 if so."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hvrnet_amd import native
a = torch.full((128, 64), 2.0 ** -20, dtype=torch.float16, device='cuda')
w = torch.ones((128, 64), dtype=torch.float16, device='cuda')
y = native.gemm(a, w, out_f32=True)
print('half subnormal inputs through the MFMA: got %.6g per output, expected %.6g' % (y[0, 0].item(), 64 * 2.0 ** -20))
