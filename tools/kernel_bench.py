"""Per-kernel timings of the hot-path shapes (HIP events on the launch stream).

    python tools/kernel_bench.py [--dtype bf16|f32] [--iters 20]

Prints one line per (kernel, shape): ms, achieved TFLOP/s or GB/s.  Used to pick tile
shapes / staging and to fill DESIGN.md's roofline table; bench.py is the contract bench.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvrnet_amd import native  # noqa: E402

if os.environ.get('HVR_BENCH_LIB'):  # A/B a privately built library (tuning experiments only)
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--frames', type=int, default=15)
    ap.add_argument('--only', default='all', help='gemm | relation | conv | all')
    ap.add_argument('--tiles', default='0,1,2,3,4,5,6,7,8,9,10', help='tile hints to sweep (0 = cost model)')
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    dev = 'cuda:0'
    T = args.frames
    tiles = [int(t) for t in args.tiles.split(',')]
    want = lambda k: args.only in ('all', k)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dt)

    print('# dtype=%s frames=%d   columns: tile hint -> ms (0 = cost model)' % (args.dtype, T))

    def sweep(label, fn, flops):
        cells = []
        for tile in tiles:
            ms = timed(lambda: fn(tile), args.iters)
            cells.append('%d:%.3f' % (tile, ms))
        best = min(float(c.split(':')[1]) for c in cells)
        print('%-34s %s   best %.1f TF/s' % (label, ' '.join(cells), flops / best / 1e9), flush=True)

    # ---- plain GEMMs of the head ----
    if want('gemm'):
        for name, M, N, K in [('fc_new_1', 4500, 1024, 12544), ('qk_proj', 4500, 2048, 1024), ('out_proj', 4500, 1024, 1024),
                              ('fc_key', 300, 1024, 1024), ('square4k', 4096, 4096, 4096)]:
            a, w = rnd(M, K), rnd(N, K, scale=0.05)
            sweep('gemm %s %dx%dx%d' % (name, M, N, K), lambda t: native.gemm(a, w, staging=1, tile=t), 2.0 * M * N * K)
    # ---- relation core ----
    if want('relation'):
        for Mq, Mk in [(4500, 4500), (300, 4500)]:
            q, k, v = rnd(Mq, 1024), rnd(Mk, 1024), rnd(Mk, 1024)
            ms = timed(lambda: native.relation_fwd(q, k, v, 1 / 32, staging=1), args.iters)
            print('relation Mq=%d Mk=%d  %.3f ms  %.1f TF/s' % (Mq, Mk, ms, 4.0 * Mq * Mk * 1024 / ms / 1e9), flush=True)
    if not want('conv'):
        return
    # ---- backbone conv classes (T frames) ----
    convs = [
        ('l1.conv2 3x3 64', 152, 252, 64, 64, 3, 1, 1, 1),
        ('l1.conv3 1x1 64->256', 152, 252, 64, 256, 1, 1, 0, 1),
        ('l1.conv1 1x1 256->64', 152, 252, 256, 64, 1, 1, 0, 1),
        ('l2.conv2 3x3 128', 76, 126, 128, 128, 3, 1, 1, 1),
        ('l2.conv3 1x1 128->512', 76, 126, 128, 512, 1, 1, 0, 1),
        ('l3.conv1 1x1 1024->256', 38, 63, 1024, 256, 1, 1, 0, 1),
        ('l3.conv2 3x3 256', 38, 63, 256, 256, 3, 1, 1, 1),
        ('l3.conv3 1x1 256->1024', 38, 63, 256, 1024, 1, 1, 0, 1),
        ('l3.0.conv1 1x1/2 512->256', 76, 126, 512, 256, 1, 2, 0, 1),
        ('r5.conv2 3x3 d2 512', 38, 63, 512, 512, 3, 1, 2, 2),
        ('r5.conv3 1x1 512->2048', 38, 63, 512, 2048, 1, 1, 0, 1),
        ('r5.conv1 1x1 2048->512', 38, 63, 2048, 512, 1, 1, 0, 1),
        ('rpn 3x3 1024->512', 38, 63, 1024, 512, 3, 1, 1, 1),
    ]
    for name, H, W, Cin, Cout, k, s, p, d in convs:
        x, w = rnd(T, H, W, Cin), rnd(Cout, k, k, Cin, scale=0.05)
        b = torch.zeros(Cout, device=dev)
        OH = (H + 2 * p - d * (k - 1) - 1) // s + 1
        OW = (W + 2 * p - d * (k - 1) - 1) // s + 1
        fl = 2.0 * T * OH * OW * Cout * k * k * Cin
        by = (x.numel() + T * OH * OW * Cout + w.numel()) * x.element_size()
        # the block-closing 1x1 convs carry the residual add, as in the network
        res = rnd(T, OH, OW, Cout) if 'conv3' in name else None
        sweep('conv ' + name + ('+res' if res is not None else ''),
              lambda t: native.conv2d_nhwc(x, w, b, res, relu=True, stride=s, pad=p, dil=d, staging=1, tile=t), fl)
    # ---- RoIAlign, all frames in one launch ----
    feat = rnd(T, 38, 63, 256)
    g = torch.Generator(device='cpu').manual_seed(0)
    xy = torch.rand((T * 300, 2), generator=g) * torch.tensor([800.0, 450.0])
    wh = torch.rand((T * 300, 2), generator=g) * 200 + 16
    rois = torch.cat([torch.arange(T).repeat_interleave(300)[:, None].float(), xy, xy + wh], 1).to(dev)
    ms = timed(lambda: native.roi_align_fwd(feat, rois, 7, 7, 1 / 16, 2, native.LAYOUT_NHWC), args.iters)
    by = (feat.numel() + T * 300 * 49 * 256) * feat.element_size() + rois.numel() * 4
    print('roi_align nhwc K=%d  %.3f ms  %.0f GB/s(algorithmic)' % (T * 300, ms, by / ms / 1e6))


if __name__ == '__main__':
    main()
