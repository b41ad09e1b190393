import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from hvrnet_amd.pipelines import FrameIngest
from hvrnet_amd import native
frame = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (720, 1280, 3)).astype(np.uint8)).to('cuda:0')
ing = FrameIngest(device='cuda:0')
ing(frame); torch.cuda.synchronize()
ts = []
for _ in range(200):
    t = time.perf_counter(); ing(frame); ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
ts = np.array(ts) * 1e6
print('call us: median %.1f mean %.1f max %.1f p99 %.1f' % (np.median(ts), ts.mean(), ts.max(), np.percentile(ts, 99)))
t = time.perf_counter()
for _ in range(1000): native._stream()
print('_stream us', (time.perf_counter() - t) * 1e3)
t = time.perf_counter()
for _ in range(1000): torch.empty((1, 3, 576, 1008), device='cuda:0')
print('empty us', (time.perf_counter() - t) * 1e3)
