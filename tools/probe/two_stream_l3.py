"""Do two frame groups on two HIP streams overlap the layer-3 bottleneck's MFMA-bound (3x3) and HBM-bound (1x1 + residual)
convolutions?  Each group's chain of NB identity blocks is captured as its own (linear) hipGraph and the graphs are
replayed on two streams; compared with one stream running all frames.  python tools/probe/two_stream_l3.py [split ...]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hvrnet_amd import native  # noqa: E402

dev = 'cuda:0'
NB = 22
H, W = 38, 63


def make(frames, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
    x = rn(frames, H, W, 1024)
    w1, w2, w3 = rn(256, 1, 1, 1024, sc=0.03), rn(256, 3, 3, 256, sc=0.02), rn(1024, 1, 1, 256, sc=0.03)
    b1, b2, b3 = torch.zeros(256, device=dev), torch.zeros(256, device=dev), torch.zeros(1024, device=dev)

    def chain():
        y = x
        for _ in range(NB):
            h = native.conv2d_nhwc(y, w1, b1, relu=True)
            h = native.conv2d_nhwc(h, w2, b2, relu=True, pad=1)
            y = native.conv2d_nhwc(h, w3, b3, resid=y, relu=True)
        return y
    return chain


def graph_of(chain, stream):
    with torch.cuda.stream(stream):
        for _ in range(2):
            chain()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            chain()
    return g


def run(splits, iters=20, delay_us=0):
    streams = [torch.cuda.Stream() for _ in splits]
    graphs = [graph_of(make(f, 7 + i), s) for i, (f, s) in enumerate(zip(splits, streams))]
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()

    def once():
        ev = torch.cuda.Event(); ev.record(main)
        dones = []
        for i, (g, s) in enumerate(zip(graphs, streams)):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                if i and delay_us:
                    torch.cuda._sleep(int(delay_us * 2400))
                g.replay()
                d = torch.cuda.Event(); d.record(s); dones.append(d)
        for d in dones:
            main.wait_event(d)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(main)
    for _ in range(iters):
        once()
    t1.record(main)
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    print('split %-10s delay %3d us: %.3f ms per %d blocks = %.1f us per block (all frames)' % (splits, delay_us, ms, NB, ms * 1000 / NB), flush=True)


if __name__ == '__main__':
    run((15,))
    for sp in [(8, 7), (9, 6), (10, 5), (5, 5, 5)]:
        run(sp)
    run((8, 7), delay_us=40)
    run((8, 7), delay_us=70)
