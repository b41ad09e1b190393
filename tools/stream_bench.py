"""Stream mode (the reference's steady-state loop, tools/test.py:214-250, with the per-frame cache) as hipGraphs, pipelined: frame
i + 1 goes through backbone / res5 / RPN / RoIAlign / fc_new_1 (graph FC) while window i runs its relation stages and read-out
(graph W) -- with graph W on the caller's stream, and with graph W on a stream confined to `--window-cus` CUs
(GraphedStream(window_cus=...), native.cu_masked_stream).  One JSON line.  Run in a process of its own (bench.py does): which
hardware queue a HIP stream lands on depends on how many streams the process has used before, and the confined-window loop loses
its gain in a process that has built dozens of graphs on dozens of streams first (bench.py after its precision ladder: 140-160
frames/s instead of 420).

    python tools/stream_bench.py [--head hvr] [--window-cus 96] [--steps 40]
"""
import argparse
import json
import os
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
    ap.add_argument('--window-cus', type=int, default=96)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--frames', type=int, default=15)
    args = ap.parse_args()
    T, N, dev = args.frames, 300, torch.device('cuda:0')
    model = hvrnet_amd.build_model((hvr_config if args.head == 'hvr' else selsa_config)(frame_interval=T // 2, nms_post=N),
                                   S.synth_state_dict(args.head), torch.bfloat16, 'cuda:0')
    frames = torch.cat([S.synth_frame(i) for i in range(T)], 0).to(dev)
    meta = S.synth_meta()

    def loop(gs, nsg):
        drive = torch.cuda.Stream(device=dev)   # not the legacy default stream: the CU-masked stream is a blocking stream
        drive.wait_stream(torch.cuda.current_stream(dev))
        last = None
        with torch.cuda.stream(drive):
            gs.push_async(frames[0:1])
            for i in range(T):
                gs.commit()
                gs.push_async(frames[(i + 1) % T:(i + 1) % T + 1])
                gs.emit().result()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = None
            for i in range(nsg):
                gs.commit()
                gs.push_async(frames[(i + 2) % T:(i + 2) % T + 1])
                nxt = gs.emit()
                if pend is not None:
                    last = pend.result()
                pend = nxt
            last = pend.result()
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
    conf, res_c = loop(GraphedStream(model, frames[0:1], meta, rescale=True, window_cus=args.window_cus), args.steps)
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
                          window_on_the_callers_stream=round(plain, 2), window_cus=args.window_cus, window_on_confined_stream=round(conf, 2),
                          ms_per_frame=round(1e3 / conf, 3), tflops=round(conf * gf / 1e3, 1), frac_mfma_peak=round(conf * gf / 1e3 / 2500.0, 4),
                          same_detections=bool(same), steps=args.steps, head=args.head, rpn_proposals_one_frame=rpn1)))


if __name__ == '__main__':
    main()
