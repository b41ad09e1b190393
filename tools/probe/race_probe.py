"""Does a tile-engine launch give the same output when another stream keeps the chip busy?  Two streams run the same op on their
own buffers concurrently; every output is compared with the op's output computed alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native
dtn = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x2': native.SPLIT, 'f32': torch.float32}[dtn]
g = torch.Generator(device='cuda').manual_seed(0)
act = lambda *s: native.cast(torch.randn(s, device='cuda', generator=g), DT)
wgt = lambda *s: native.as_operand(torch.randn(s, device='cuda', generator=g) * 0.03, DT)
B, H, W = 15, 38, 63
cases = {
    'l3.conv2 3x3 256': (act(B, H, W, 256), wgt(256, 3, 3, 256), dict(pad=1)),
    'l3.conv1 1x1 1024->256': (act(B, H, W, 1024), wgt(256, 1, 1, 1024), dict()),
    'l2.conv2 3x3 128': (act(B, 76, 126, 128), wgt(128, 3, 3, 128), dict(pad=1)),
}
streams = [torch.cuda.Stream() for _ in range(2)]
for name, (x, w, kw) in cases.items():
    bias = torch.randn(w.shape[0], device='cuda', generator=g)
    for hint in (0, 1, 4, 11, 12):
        ref = native.conv2d_nhwc(x, w, bias, relu=True, tile=hint, **kw)
        torch.cuda.synchronize()
        bad = 0
        outs = []
        for it in range(8):
            for st in streams:
                with torch.cuda.stream(st):
                    outs.append(native.conv2d_nhwc(x, w, bias, relu=True, tile=hint, **kw))
        torch.cuda.synchronize()
        bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print('%-24s %s hint %2d: %d of %d concurrent outputs differ' % (name, dtn, hint, bad, len(outs)), flush=True)
