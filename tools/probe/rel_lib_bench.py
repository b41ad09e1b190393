"""A/B of privately built libraries on the full relation stage (Mq = Mk = 4500, D = 1024, bf16): raw ctypes, only the two entry
points every build since round 1 exports.   python tools/probe/rel_lib_bench.py lib_a.so [lib_b.so ...]"""
import ctypes
import sys

import torch

Mq = Mk = 4500
D = 1024
torch.manual_seed(0)
q = torch.randn(Mq, D, device='cuda').bfloat16()
k = torch.randn(Mk, D, device='cuda').bfloat16()
v = torch.randn(Mk, D, device='cuda').bfloat16()
o = torch.empty_like(q)
vp, i64, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.hvr_relation_workspace_bytes.restype = sz
    lib.hvr_relation_workspace_bytes.argtypes = [ctypes.c_int] * 4
    lib.hvr_relation_fwd.restype = ctypes.c_int
    lib.hvr_relation_fwd.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                     ctypes.c_int, vp, sz, vp]
    n = lib.hvr_relation_workspace_bytes(Mq, Mk, D, 1)
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.hvr_relation_fwd(q.data_ptr(), D, k.data_ptr(), D, v.data_ptr(), D, o.data_ptr(), D, Mq, Mk, D, 1 / 32.0, 1, 1, ws.data_ptr(), n, st)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    print('%-48s %.4f ms  %.1f TF/s  checksum %.6f' % (path.split('/')[-1], best, 4.0 * Mq * Mk * D / best / 1e9, o.float().abs().mean().item()))
