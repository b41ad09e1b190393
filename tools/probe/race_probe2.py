"""A split-half conv under heavy contention: three background streams run other kernels (big casts, other convs) while the conv
under test runs repeatedly on its own stream; every output is compared with the conv's output computed alone."""
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
big = torch.randn((15, 152, 252, 256), device='cuda', generator=g)
xb, wb = act(B, 76, 126, 128), wgt(128, 3, 3, 128)
xc, wc = act(B, H, W, 1024), wgt(512, 3, 3, 1024)
cases = {'l3.conv2 3x3 256': (act(B, H, W, 256), wgt(256, 3, 3, 256), dict(pad=1)), 'l3.conv1 1x1 1024->256': (act(B, H, W, 1024), wgt(256, 1, 1, 1024), dict()),
         'l1.conv1 1x1 256->64': (act(B, 152, 252, 256), wgt(64, 1, 1, 256), dict())}
main = torch.cuda.Stream()
bg = [torch.cuda.Stream() for _ in range(3)]
for name, (x, w, kw) in cases.items():
    bias = torch.randn(w.shape[0], device='cuda', generator=g)
    for hint in (0, 1, 11, 12):
        ref = native.conv2d_nhwc(x, w, bias, relu=True, tile=hint, **kw)
        torch.cuda.synchronize()
        outs = []
        for it in range(12):
            with torch.cuda.stream(bg[0]):
                native.cast(big, DT)
            with torch.cuda.stream(bg[1]):
                native.conv2d_nhwc(xb, wb, None, relu=True, pad=1)
            with torch.cuda.stream(bg[2]):
                native.conv2d_nhwc(xc, wc, None, relu=True, pad=1)
            with torch.cuda.stream(main):
                outs.append(native.conv2d_nhwc(x, w, bias, relu=True, tile=hint, **kw))
        torch.cuda.synchronize()
        print('%-24s %s hint %2d: %d of %d outputs differ under contention' % (name, dtn, hint, sum(0 if torch.equal(o, ref) else 1 for o in outs), len(outs)), flush=True)
