"""Per-workgroup wall-clock stamps of the scores kernel (debug build -DHVR_DBG_BT_CLK): start skew, prologue, loop, end."""
import os, sys, subprocess, re
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    from hvrnet_amd import native
    native.LIB_PATH = os.path.abspath('dbg/libhvr_btclk.so')
    torch.manual_seed(0)
    q = torch.randn(4500, 1024, device='cuda').bfloat16(); k = torch.randn(4500, 1024, device='cuda').bfloat16(); v = torch.randn(4500, 1024, device='cuda').bfloat16()
    for _ in range(3):
        native.relation_fwd(q, k, v, 1 / 32, staging=1)
    torch.cuda.synchronize()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, 'child'], capture_output=True, text=True).stdout
rows = [list(map(int, l.split()[1:])) for l in out.splitlines() if l.startswith('BTCLK')]
a = np.array(rows[-234:], dtype=np.int64)   # last launch
t0 = a[:, 2] - a[:, 2].min()
print('workgroups', len(a), ' (100 MHz ticks = 10 ns)')
for name, col in (('start skew', t0), ('prologue', a[:, 3]), ('loop end', a[:, 4]), ('own total', a[:, 5]), ('end since first start', t0 + a[:, 5])):
    print('%-22s min %6d  p50 %6d  p90 %6d  max %6d' % (name, col.min(), np.median(col), np.percentile(col, 90), col.max()))
for x in range(8):
    m = a[:, 1] == x
    if m.any(): print('xcc', x, 'n', m.sum(), 'start p50', int(np.median(t0[m])), 'own total p50', int(np.median(a[m, 5])), 'max', int(a[m, 5].max()))
