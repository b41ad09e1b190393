"""Big-tile (tile 17 = forced) against the library's own choice on the stride-16 conv shapes of a 15-frame batch."""
import sys, torch
sys.path.insert(0, '/root/repo')
from hvrnet_amd import native
dev = 'cuda:0'
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
def run(name, B, H, W, Cin, Cout, k, stride, pad, dil, res):
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.03).bfloat16()
    b = torch.randn(Cout, device=dev)
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn(B, OH, OW, Cout, device=dev).bfloat16() if res else None
    t0 = timed(lambda: native.conv2d_nhwc(x, w, b, r, relu=True, stride=stride, pad=pad, dil=dil))
    t1 = timed(lambda: native.conv2d_nhwc(x, w, b, r, relu=True, stride=stride, pad=pad, dil=dil, tile=17))
    tiles = ((B * OH * OW + 287) // 288) * (Cout // 256)
    print('%-28s library %6.1f us   big tile %6.1f us  (%d workgroups)' % (name, t0, t1, tiles), flush=True)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 15
run('l3.conv1 1x1 1024->256', T, 38, 63, 1024, 256, 1, 1, 0, 1, False)
run('l3.conv2 3x3 256', T, 38, 63, 256, 256, 3, 1, 1, 1, False)
run('l3.conv3 1x1 256->1024+res', T, 38, 63, 256, 1024, 1, 1, 0, 1, True)
run('r5.conv1 1x1 2048->512', T, 38, 63, 2048, 512, 1, 1, 0, 1, False)
run('r5.conv2 3x3 d2 512', T, 38, 63, 512, 512, 3, 1, 2, 2, False)
run('r5.conv3 1x1 512->2048+res', T, 38, 63, 512, 2048, 1, 1, 0, 1, True)
run('rpn 3x3 1024->512', T, 38, 63, 1024, 512, 3, 1, 1, 1, False)
run('l2.conv3 1x1 128->512+res', T, 76, 126, 128, 512, 1, 1, 0, 1, True)
run('l1.conv3 1x1 64->256+res', T, 152, 252, 64, 256, 1, 1, 0, 1, True)
run('l3.0 1x1/2 512->256', T, 76, 126, 512, 256, 1, 2, 0, 1, False)
