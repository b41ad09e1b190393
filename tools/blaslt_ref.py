"""Calibration only (not product): what rocBLAS/hipBLASLt reach on this path's GEMM shapes."""
import torch
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
shapes = [('scores', 4500, 4500, 1024), ('apply', 4500, 1024, 4608), ('fc_new_1', 4500, 1024, 12544), ('qk', 4500, 2048, 1024),
          ('l3.conv1', 35910, 256, 1024), ('l3.conv3', 35910, 1024, 256), ('l3.conv2', 35910, 256, 2304), ('sq4k', 4096, 4096, 4096),
          ('sq8k', 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = torch.randn(N, K, device='cuda').bfloat16()
    ms = timed(lambda: torch.mm(a, b.t()))
    print('%-10s %6dx%5dx%5d  NT %.4f ms %.0f TF/s' % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    bt = b.t().contiguous()
    ms = timed(lambda: torch.mm(a, bt))
    print('%-10s %6dx%5dx%5d  NN %.4f ms %.0f TF/s' % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
