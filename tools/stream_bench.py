"""Stream mode (the reference's steady-state loop, tools/test.py:214-250, with the per-frame cache) as hipGraphs, pipelined: the next
`--frame-lanes` (2) frames go through backbone / res5 / RPN / RoIAlign / fc_new_1 (graphs FC, one stream each) while window i runs its
relation stages and read-out (graph W) -- with graph W on the caller's stream and one frame in flight, and with graph W on a stream of
its own over `--window-cus` CUs (GraphedStream(window_cus=..., frame_lanes=...), native.cu_masked_stream; 256 = the whole chip).  One
JSON line.  Run in a process of its own (bench.py does): which hardware queue a HIP stream lands on depends on how many streams the
process has used before, and the loop loses its gain in a process that has built dozens of graphs on dozens of streams first.

What round 5 measured on one MI355X (profiles/r05_stream_lanes.txt): one frame's chain is ~150 launches of at most 152 small
workgroups, 1.40 ms by itself; the window graph is 0.92 ms by itself on the whole chip, 1.54 ms on 96 CUs.  One frame in flight
beside a 96-CU window: 535 frames/s.  TWO frames in flight overlap almost completely -- if the streams get hardware queues of their
own: the HIP runtime multiplexes its streams onto GPU_MAX_HW_QUEUES = 4 queues by default and a queue is served in order; with 8
(set below, before the runtime starts) two frame lanes + the window on its own full-width stream run at 675 frames/s (4 queues: 582;
three lanes lose, 380-480, whatever the queues).

    python tools/stream_bench.py [--head hvr] [--window-cus 256] [--frame-lanes 2] [--steps 40]
"""
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # (read by the HIP runtime when it initialises: before torch is imported)
import argparse
import json
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hvrnet_amd  # noqa: E402
from hvrnet_amd import synthetic as S  # noqa: E402
from hvrnet_amd.config import hvr_config, selsa_config  # noqa: E402
from hvrnet_amd.graphs import GraphedStream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--head', default='hvr', choices=['hvr', 'selsa'])
    ap.add_argument('--window-cus', type=int, default=256)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--frames', type=int, default=15)
    ap.add_argument('--frame-lanes', type=int, default=2, help='frames in flight beside the window graph (GraphedStream(frame_lanes=L))')
    args = ap.parse_args()
    T, N, dev = args.frames, 300, torch.device('cuda:0')
    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=T // 2, nms_post=N),
                                   S.synth_state_dict(args.head), torch.bfloat16, 'cuda:0')
    frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
    meta = S.synth_meta()

    def loop(gs, nsg):
        drive = torch.cuda.Stream(device=dev)   # not the legacy default stream: the CU-masked stream is a blocking stream
        drive.wait_stream(torch.cuda.current_stream(dev))
        last, L, fed = None, len(gs._lanes), [0]

        def feed():   # frames enter in the order 0, 1, 2, ... (mod T) whatever the number of frame lanes
            gs.push_async(frames[fed[0] % T:fed[0] % T + 1])
            fed[0] += 1
        with torch.cuda.stream(drive):
            for _ in range(L):
                feed()
            for i in range(T):
                gs.commit()
                feed()
                gs.emit().result()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = None
            for i in range(nsg):
                gs.commit()
                feed()
                nxt = gs.emit()
                if pend is not None:
                    last = pend.result()
                pend = nxt
            last = pend.result()
            for _ in range(L):
                gs.commit()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        torch.cuda.current_stream(dev).wait_stream(drive)
        return nsg / el, last

    import numpy as np

    def flat(r):
        return [np.asarray(x) for br in (r if isinstance(r[0], (list, tuple)) else [r]) for x in br]

    # (the confined loop first: see the header -- by the time a second GraphedStream has been built the process has used enough
    # streams for the gain to shrink: 377 instead of 425 frames/s with the two loops in the other order)
    gs_conf = GraphedStream(model, frames[0:1], meta, rescale=True, window_cus=args.window_cus, frame_lanes=args.frame_lanes)
    conf, res_c = loop(gs_conf, args.steps)

    def alone(graph, stream, n=20):
        """ms per replay of one graph with nothing else on the chip: the two sides of the pipelined loop, each by itself"""
        with torch.cuda.stream(stream):
            graph.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(n):
                graph.replay()
            b.record(stream)
            torch.cuda.synchronize()
        return round(a.elapsed_time(b) / n, 4)
    with gs_conf._window_stream() as wst:
        pass
    parts = dict(frame_graph_alone_ms=alone(gs_conf.graph_fc, gs_conf._fstream),
                 window_graph_alone_on_confined_stream_ms=alone(gs_conf._graphs_w[0], wst),
                 window_graph_alone_on_frame_stream_ms=alone(gs_conf._graphs_w[0], gs_conf._fstream))
    plain, res_p = loop(GraphedStream(model, frames[0:1], meta, rescale=True), args.steps)
    same = all(np.array_equal(a, b) for a, b in zip(flat(res_p), flat(res_c)))

    def proposals_one_frame():
        """hvr_rpn_proposals on the RPN outputs of one frame (what the frame graph holds on its critical path behind C4): us per
        call, HIP events around 50 calls, in the chip-wide form a one-frame call takes by default and in the one-workgroup-per-frame
        form (hvr_rpn_wide_frames(0)); the two outputs compared bit for bit."""
        from hvrnet_amd import native
        import hvrnet_amd.rpn_head as RH
        cap, orig = {}, native.rpn_proposals

        def spy(*a, **k):
            cap['a'], cap['k'] = a, k
            return orig(*a, **k)
        RH.native.rpn_proposals = spy
        try:
            with torch.no_grad():
                c4 = model(img=frames[0:1], img_meta=[meta], backbone_feat=True)[0]
                model.frame_tensors(c4, meta)
        finally:
            RH.native.rpn_proposals = orig
        a, k = cap['a'], cap['k']
        res, outs = {}, {}
        prev = native.rpn_wide_frames(-1)
        try:
            for name, frames_limit in (('chip_wide', max(prev, 1)), ('one_workgroup_per_frame', 0)):
                native.rpn_wide_frames(frames_limit)
                for _ in range(3):
                    outs[name] = orig(*a, **k)
                torch.cuda.synchronize()
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(50):
                    orig(*a, **k)
                s1.record()
                torch.cuda.synchronize()
                res[name + '_us'] = round(s0.elapsed_time(s1) / 50 * 1e3, 1)
        finally:
            native.rpn_wide_frames(prev)
        n = int(outs['chip_wide'][1][0])
        res['same_proposals'] = bool(torch.equal(outs['chip_wide'][1], outs['one_workgroup_per_frame'][1]) and
                                     torch.equal(outs['chip_wide'][0][0, :n], outs['one_workgroup_per_frame'][0][0, :n]))
        res['proposals'] = n
        res['what'] = 'hvr_rpn_proposals, T = 1 (eager launches, HIP events around 50 calls): default form for one frame vs hvr_rpn_wide_frames(0)'
        return res

    try:
        rpn1 = proposals_one_frame()
    except Exception as exc:   # noqa: BLE001 -- a side measurement must not take the stream numbers down
        rpn1 = dict(error=repr(exc))
    gf = 650.0 if args.head == 'hvr' else 504.0
    print(json.dumps(dict(metric='stream mode, pipelined hipGraphs: output frames/s (one new frame per output frame, T = %d, %d proposals)' % (T, N),
                          window_on_the_callers_stream=round(plain, 2), window_cus=args.window_cus, frame_lanes=args.frame_lanes, hw_queues=int(os.environ['GPU_MAX_HW_QUEUES']), window_on_confined_stream=round(conf, 2),
                          ms_per_frame=round(1e3 / conf, 3), tflops=round(conf * gf / 1e3, 1), frac_mfma_peak=round(conf * gf / 1e3 / 2500.0, 4),
                          same_detections=bool(same), steps=args.steps, head=args.head, graphs_alone=parts, rpn_proposals_one_frame=rpn1)))


if __name__ == '__main__':
    main()
