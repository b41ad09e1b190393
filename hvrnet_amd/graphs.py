"""hipGraph capture of the window (clip mode) and of the per-frame / per-window chains (stream mode).

The reference's loop is launch-bound by construction: ~250 kernel launches per window from Python (tools/test.py:214-250 ->
hnmb_rcnn.py:195-222), each a few microseconds of host time; the eager HIP path here enqueues a clip-mode window in ~2.8 ms
of host time against 7.5 ms of GPU time, and a stream-mode output frame (one new frame through the backbone + the head on
cached rows) is host-bound outright.  The kernels of a window are shape-static once every frame keeps its `nms_post`
proposals (the speculative path of detectors.py; a short frame is detected from the counts read with the results and re-run
through the exact eager path), so the whole chain -- frame groups on their side streams, the RPN side stream, the second
branch's read-out stream included -- is captured ONCE into a hipGraph and replayed: one host call per window, dependencies
resolved on the device.

  GraphedClip(model, frames, metas)          clip mode: T frames -> key-frame detections, one graph
  GraphedStream(model, frame_shape, meta)    stream mode (the reference's loop with the per-frame cache): one
                                             graph for "a frame arrives" (backbone, res5, RPN, proposals, RoIAlign, fc_new_1
                                             -> the frame's cached rows), one for "a window is emitted" (relation stages and
                                             read-out on the T cached entries)

torch.cuda.CUDAGraph is the capture mechanism (hipStreamBeginCapture / hipGraphLaunch underneath); every kernel inside is this
build's own, launched through the C ABI on the capturing streams.  Results are identical to the eager path (same kernels,
same order per stream): tests/test_graphs_gpu.py.
"""
import contextlib
import weakref

import torch

from .box_ops import bbox2result

# Captured graphs hold raw device pointers: of the packed weights (PackedMixin.packed) and of the per-stream scratch buffers
# (native._workspace).  native._workspace keeps every buffer it handed out during a capture alive for the life of the process;
# packed weights are rebuilt by load_state_dict / set_compute_dtype / an optimizer step -- when that happens while graphs are
# alive, the graphs are marked stale here and refuse to replay (PackedMixin._drop_packed -> invalidate_all).
_LIVE = weakref.WeakSet()


def invalidate_all(reason):
    for g in list(_LIVE):
        g._stale = reason


@contextlib.contextmanager
def _capture(graph, **kw):
    """torch.cuda.graph(graph, **kw) with Python's cyclic garbage collector held off for the duration.  A Graphed* object and its
    pending windows reference each other, so a dropped one is freed by the collector, whenever that runs -- and when it ran in the
    middle of ANOTHER object's capture (it is triggered by allocation counts: any ctypes launch can be the one), the dead
    CUDAGraph's destructor called hipGraphExecDestroy while a stream was capturing in global mode: 'operation not permitted when
    stream is capturing', from a destructor, i.e. std::terminate.  Collect first, then capture with the collector disabled."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            gc.enable()



def _check_live(g):
    if g._stale is not None:
        raise RuntimeError('this captured graph reads buffers that were freed after its capture (%s): build a new %s'
                           % (g._stale, type(g).__name__))


class _HostOut(object):
    """Pinned host mirrors of a window's device outputs; the D2H copies are graph nodes."""

    def __init__(self, branches, counts_dev):
        self.dev = [tuple(b) for b in branches]
        self.counts_dev = counts_dev
        self.host = [tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in b) for b in self.dev]
        self.counts_host = torch.empty(counts_dev.shape, dtype=counts_dev.dtype, pin_memory=True) if counts_dev is not None else None

    def enqueue_copies(self):
        for hb, db in zip(self.host, self.dev):
            for h, d in zip(hb, db):
                h.copy_(d, non_blocking=True)
        if self.counts_host is not None:
            self.counts_host.copy_(self.counts_dev, non_blocking=True)


class PendingGraphWindow(object):
    """A replayed window whose results have not been read; result() is its single host synchronisation."""

    def __init__(self, owner, out, event, exact):
        self._owner, self._out, self._event, self._exact, self._result = owner, out, event, exact, None
        self.respeculated = False

    @property
    def read(self):
        return self._result is not None

    def result(self):
        if self._result is None:
            self._event.synchronize()
            o = self._owner
            out = self._out
            if out.counts_host is not None and any(int(c) != o._full for c in out.counts_host.tolist()):
                self.respeculated = True
                self._result = self._exact()
            else:
                res = []
                for dets, labels, n in out.host:
                    k = int(n[0])
                    res.append(bbox2result(dets[:k].clone(), labels[:k].clone(), o._num_classes))
                self._result = res[0] if o._single else res
        return self._result


class GraphedClip(object):
    """One clip-mode window (all T frames through backbone, res5, RPN, proposals, RoIAlign, head, read-out) as a hipGraph.

    frames: [T,3,H,W] f32 on the device -- the graph reads THIS buffer; write the next clip's frames into `self.frames`
    (or pass them to `run`) before replaying.  A replay must have been read (`result()`) before the next one is launched:
    the outputs are static buffers."""

    def __init__(self, model, frames, metas, rescale=True, warmup=2, n_out=2, throughput=False, windows=1, batch_head=True):
        assert frames.is_cuda and frames.dim() == 4 and frames.shape[0] == len(metas) and len(metas) % windows == 0
        self.model, self.metas, self.rescale = model, list(metas), rescale
        # windows = W > 1: `frames` holds W independent clips back to back; ONE graph takes all W * T frames through the backbone
        # in one batch -- 30 frames give layer 3's convs 250 of the 288 x 256 tiles (bigtile.hip) where 15 give 125 -- and then
        # runs res5 / RPN / RoIAlign / head / read-out per clip.  run() returns W pending windows.  Frames are independent through
        # the backbone, so every clip's detections are the single-clip graph's.  batch_head (round 5): res5, the RPN, its proposals,
        # RoIAlign and every product of the head take the W clips as one batch too, and the relation core runs per clip in GROUPED
        # calls (hvr_relation_fwd_grouped: persistent score tiles over the W windows, the 288 x 256 apply launch from W = 3 on in
        # bf16) -- per clip the single-clip result up to the association of the relation core's f32 sums (bit for bit with
        # bbox_head.grouped_exact = True)
        self.windows = int(windows)
        self.batch_head = bool(batch_head) and self.windows > 1
        self.T = len(metas) // self.windows
        # throughput: the graph is one of several replayed side by side (bench.py's lanes) -- its launches prefer CU-time to
        # latency (native.throughput_mode: the 288 x 256 tiles on half the grid for layer 3's convs); same detections
        self.throughput = bool(throughput)
        self.frames = frames.clone()
        self._num_classes = model.bbox_head.num_classes
        self._single = type(model).__name__ == 'SelsaRCNN'
        self._stale, self._unread = None, {}
        dev = frames.device
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        self._graphs, self._outs, self._turn, self._generation = [], [], 0, 0
        with torch.no_grad(), torch.cuda.stream(self._stream):
            for _ in range(max(1, warmup)):   # builds every lazily created object: packed weights, workspaces, side streams
                per_window = self._enqueue()
            self._stream.synchronize()
            self._full = per_window[0][2]
            # n_out graphs of the same window(s), each with its own output buffers (replayed in turn: window i + 1 can be
            # enqueued before window i's results are read); they share one memory pool -- same stream, never concurrent
            for k in range(max(1, n_out)):
                outs = [_HostOut(b, c) for b, c, _ in per_window]   # pinned buffers exist before the capture (no host allocation inside it)
                graph = torch.cuda.CUDAGraph()
                with _capture(graph, stream=self._stream, **(dict(pool=self._graphs[0].pool()) if self._graphs else {})):
                    for out, (b2, c2, _) in zip(outs, self._enqueue()):
                        out.dev, out.counts_dev = [tuple(b) for b in b2], c2
                        out.enqueue_copies()
                self._graphs.append(graph)
                self._outs.append(outs)
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        _LIVE.add(self)

    def _enqueue(self):
        from . import native
        m = self.model
        T = self.T
        with native.throughput_mode(self.throughput):
            c4 = m(img=self.frames, img_meta=self.metas, backbone_feat=True)[0]
            if self.batch_head:
                return m.window_device_outputs(c4, self.metas, rescale=self.rescale, clips=self.windows)
            return [m.window_device_outputs(c4[w * T:(w + 1) * T], self.metas[w * T:(w + 1) * T], rescale=self.rescale)
                    for w in range(self.windows)]

    def _exact(self, w=0):
        T = self.T
        with torch.no_grad():
            c4 = self.model(img=self.frames[w * T:(w + 1) * T], img_meta=self.metas[w * T:(w + 1) * T], backbone_feat=True)[0]
            return self.model.forward_feat(x=c4, img_meta=self.metas[w * T:(w + 1) * T], rescale=self.rescale, speculate=False)

    def run(self, frames=None):
        """Replays the window on the current stream; -> PendingGraphWindow (a list of them, one per clip, when windows > 1).  With
        n_out graphs at most n_out - 1 earlier replays may still be unread."""
        if frames is not None:
            self.frames.copy_(frames, non_blocking=True)
        if frames is not None:
            self._generation += 1
        gen = self._generation
        k = self._turn % len(self._graphs)
        _check_live(self)
        # graph k's pinned output buffers are about to be overwritten: its previous replay must have been read
        if any(not p_.read for p_ in self._unread.get(k, ())):
            raise RuntimeError('replaying output slot %d of %d before result() of its previous window was called (n_out = %d graphs allow '
                               '%d unread windows)' % (k, len(self._graphs), len(self._graphs), len(self._graphs) - 1))
        self._turn += 1
        self._graphs[k].replay()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.frames.device))

        def exact(w):
            def go():
                if gen != self._generation:
                    raise RuntimeError('a frame of this window kept fewer than nms_post proposals, but its input buffer has been overwritten by a '
                                       'later run(frames=...): read result() before handing over the next clip')
                return self._exact(w)
            return go
        pend = [PendingGraphWindow(self, self._outs[k][w], ev, exact(w)) for w in range(self.windows)]
        self._unread[k] = pend
        return pend[0] if self.windows == 1 else pend


class GraphedStream(object):
    """The reference's steady-state loop (tools/test.py:214-250) with the per-frame cache, as two hipGraphs.

      push(frame)   graph F: backbone -> res5 / RPN / proposals / RoIAlign / fc_new_1 of ONE frame; its rows are appended to
                    the window buffers (the oldest frame's rows drop out: a device-side shift, part of the graph)
      emit()        graph W: the relation stages and the read-out on the T frames in the window buffers -> PendingGraphWindow
      push_async(frame) / commit()   the same as push(), split in two: graph FC (the frame's per-frame part, on a second
                    stream) runs beside the previous window's graph W; commit() appends its rows (graph C)

    Window buffers hold T entries in arrival order: fc_new_1 rows [T * n, 1024], proposals [T, n, 5], counts [T]; n = nms_post.
    A frame may be pushed several times without recomputing it (`repeat_last()`): the reference pads the first and last
    windows of a video with copies of a frame (test.py:201-212, 257-300)."""

    def __init__(self, model, frame, meta, rescale=True, warmup=2, n_out=2, lookahead=1, fewrow_split=True, window_cus=None, frame_lanes=1):
        # window_cus: every graph of this object is captured as ONE chain (the RPN branch behind res5, the second read-out branch
        # behind the first).  A graph with parallel branches replayed on a CU-masked stream makes the runtime set its branch
        # streams up from that stream -- hipGraphLaunch segfaulted (ROCm 7.2) when that was the process's first forked replay --
        # and where the branch streams of the two graphs land among the hardware queues decides whether the window still runs
        # beside the frame (frame graph forked, window graph not: 331 frames/s, no gain; both forked 425; neither 420).
        undo = []
        if window_cus:
            for obj, attr in ((model, 'rpn_side_stream'), (model.bbox_head, 'readout_streams')):
                if getattr(obj, attr, False):
                    setattr(obj, attr, False)
                    undo.append((obj, attr))
        try:
            self._build(model, frame, meta, rescale, warmup, n_out, lookahead, fewrow_split, window_cus, max(1, int(frame_lanes)))
        finally:
            for obj, attr in undo:
                delattr(obj, attr)   # back to the class attribute

    def _build(self, model, frame, meta, rescale, warmup, n_out, lookahead, fewrow_split, window_cus, frame_lanes=1):
        assert frame.is_cuda and frame.dim() == 4 and frame.shape[0] == 1
        # window_cus = n (meant for the pipelined loop, push_async / commit / emit): everything that touches the window buffers --
        # graph C, graph W, the padding / staging graphs -- is replayed on ONE stream confined to n of the chip's CUs
        # (native.cu_masked_stream); the next frame's graph FC keeps its own unconfined stream.  That loop's critical path is the
        # frame chain, ~150 small launches; the window's relation kernels are chip-filling launches of 150 KB-LDS workgroups, and
        # every one of them that sits on a CU in front of a frame launch is 50 us of waiting.  With the window on 96 CUs (12 per
        # XCD) the frame chain always finds free CUs: 332 -> 420 frames/s on one box (tools/stream_bench.py; 64 / 128 / 176
        # CUs: 411 / 410 / 409 with forked graphs).  Same graphs in the same order: same results.
        self._wstream = None
        self.window_cus = window_cus
        self.model, self.meta, self.rescale = model, meta, rescale
        self.lookahead = int(lookahead)
        # one frame gives the stride-16 stages 2 394 rows: its convs / fc_new_1 run split over K (native.fewrow_split) --
        # same products, f32 sums in slice order, so a frame's rows equal the batched computation to an output ulp, not bit
        # for bit; fewrow_split=False keeps the unsplit kernels (the bit-identity tests)
        self.fewrow_split = bool(fewrow_split)
        dev = frame.device
        self.T = int(model.bbox_head.t_dim)
        self.key = int(model.key_dim)
        self.frame = frame.clone()
        self._num_classes = model.bbox_head.num_classes
        self._single = type(model).__name__ == 'SelsaRCNN'
        self._stale, self._unread = None, {}
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        T = self.T
        with torch.no_grad(), torch.cuda.stream(self._stream):
            if self.lookahead > 1:
                # the LARGEST shapes first: the look-ahead batch sizes this stream's scratch buffers (native._workspace regrows a
                # buffer by replacing it) before any graph captures their addresses
                self.batch = self.frame.new_zeros((self.lookahead,) + tuple(self.frame.shape[1:]))
                for _ in range(max(1, warmup)):
                    c4 = self.model(img=self.batch, img_meta=[self.meta] * self.lookahead, backbone_feat=True)[0]
                    self.model.frames_tensors(c4, [self.meta] * self.lookahead)
            e = None
            for _ in range(max(1, warmup)):
                e = self._frame_entry()
            n, D = e['f1'].shape
            self.n = n
            self.f1 = torch.zeros((T * n, D), dtype=e['f1'].dtype, device=dev)
            self.f1_tmp = torch.empty(((T - 1) * n, D), dtype=e['f1'].dtype, device=dev)
            self.props = torch.zeros((T, n, 5), dtype=e['props'].dtype, device=dev)
            self.props_tmp = torch.empty((T - 1, n, 5), dtype=e['props'].dtype, device=dev)
            self.counts = torch.full((T,), n, dtype=torch.int32, device=dev)
            self.counts_tmp = torch.empty((T - 1,), dtype=torch.int32, device=dev)
            self.last = dict(f1=torch.zeros_like(e['f1']), props=torch.zeros_like(e['props']), count=torch.zeros_like(e['count']))
            for _ in range(max(1, warmup)):
                self._push_from(e)
                branches, counts, full = self._window()
            self._stream.synchronize()
            self._full = n
            self.graph_f = torch.cuda.CUDAGraph()
            with _capture(self.graph_f, stream=self._stream):
                e = self._frame_entry()
                self.last['f1'].copy_(e['f1'])
                self.last['props'].copy_(e['props'])
                self.last['count'].copy_(e['count'])
                self._push_from(self.last)
            self.graph_p = torch.cuda.CUDAGraph()          # push the last computed frame again (padding)
            with _capture(self.graph_p, stream=self._stream, pool=self.graph_f.pool()):
                self._push_from(self.last)
            # Look-ahead (offline video: the frames of a clip are all there): `lookahead` frames go through backbone / res5 / RPN /
            # RoIAlign / fc_new_1 in ONE batch -- a single 600x1000 frame gives the stride-16 stages 2 394 rows, 17-19 row tiles
            # for 256 CUs -- into a staging area; frame i of the batch then enters the window buffers (graph S_i + graph P)
            # when its turn comes.  Same rows as one frame at a time (every kernel of the per-frame part is row-independent).
            self.graph_fb, self._stage_graphs = None, []
            if self.lookahead > 1:
                B = self.lookahead
                metas = [self.meta] * B
                for _ in range(max(1, warmup)):
                    c4 = self.model(img=self.batch, img_meta=metas, backbone_feat=True)[0]
                    eb = self.model.frames_tensors(c4, metas)
                self._stream.synchronize()
                self.graph_fb = torch.cuda.CUDAGraph()
                with _capture(self.graph_fb, stream=self._stream, pool=self.graph_f.pool()):
                    c4 = self.model(img=self.batch, img_meta=metas, backbone_feat=True)[0]
                    eb = self.model.frames_tensors(c4, metas)
                self._staged = eb   # static buffers of the batch graph: props [B,n,5], count [B], f1 [B * n, D]
                for i in range(B):
                    gi = torch.cuda.CUDAGraph()
                    with _capture(gi, stream=self._stream, pool=self.graph_f.pool()):
                        self.last['f1'].copy_(eb['f1'][i * n:(i + 1) * n])
                        self.last['props'].copy_(eb['props'][i])
                        self.last['count'].copy_(eb['count'][i:i + 1])
                        self._push_from(self.last)
                    self._stage_graphs.append(gi)
            # n_out window graphs with their own output buffers, replayed in turn (see GraphedClip)
            self._graphs_w, self._outs, self._turn = [], [], 0
            for k in range(max(1, n_out)):
                out = _HostOut(branches, counts)
                graph = torch.cuda.CUDAGraph()
                with _capture(graph, stream=self._stream, pool=self.graph_f.pool()):
                    b2, c2, _ = self._window()
                    out.dev, out.counts_dev = [tuple(b) for b in b2], c2
                    out.enqueue_copies()
                self._graphs_w.append(graph)
                self._outs.append(out)
        # Pipelined form (push_async / commit): the NEXT frame's per-frame part runs on its own stream beside the current
        # window's relation stages and read-out -- the two are independent until the frame's rows enter the window buffers.
        # Graph FC (frame -> staging rows `nxt`) is captured on that stream, with its own memory pool and its own per-stream
        # scratch and side streams (it runs concurrently with the window graphs); graph C (staging rows -> window buffers)
        # belongs to the main stream's family.
        # frame_lanes = L > 1: L frames in flight, each with a graph FC, a stream, staging rows and a graph C of its own, taken in turn
        # by push_async() and committed in arrival order.  One frame's chain is ~150 launches of at most 152 small workgroups -- a
        # latency chain that leaves most of the chip idle; two chains side by side overlap almost completely, which moves the loop's
        # bound from the frame chain to the window graph (tools/stream_bench.py --frame-lanes 2).
        self._lanes = []
        for _ in range(frame_lanes):
            lane = dict(stream=torch.cuda.Stream(device=dev), ev_fc=torch.cuda.Event(), ev_commit=None, pending=None)
            lane['stream'].wait_stream(self._stream)
            lane['frame'] = self.frame.clone()
            lane['nxt'] = dict(f1=torch.zeros_like(self.last['f1']), props=torch.zeros_like(self.last['props']), count=torch.zeros_like(self.last['count']))
            with torch.no_grad(), torch.cuda.stream(lane['stream']):
                for _ in range(max(1, warmup)):
                    self._frame_entry(lane['frame'])
                lane['stream'].synchronize()
                lane['graph_fc'] = torch.cuda.CUDAGraph()
                with _capture(lane['graph_fc'], stream=lane['stream']):
                    e = self._frame_entry(lane['frame'])
                    lane['nxt']['f1'].copy_(e['f1'])
                    lane['nxt']['props'].copy_(e['props'])
                    lane['nxt']['count'].copy_(e['count'])
            self._stream.wait_stream(lane['stream'])
            with torch.no_grad(), torch.cuda.stream(self._stream):
                lane['graph_c'] = torch.cuda.CUDAGraph()
                with _capture(lane['graph_c'], stream=self._stream, pool=self.graph_f.pool()):
                    self.last['f1'].copy_(lane['nxt']['f1'])
                    self.last['props'].copy_(lane['nxt']['props'])
                    self.last['count'].copy_(lane['nxt']['count'])
                    self._push_from(self.last)
            self._lanes.append(lane)
        self._push_turn = self._commit_turn = 0
        # (lane 0 under the names the one-lane form had)
        self._fstream, self.frame_nxt, self.nxt = self._lanes[0]['stream'], self._lanes[0]['frame'], self._lanes[0]['nxt']
        self.graph_fc, self.graph_c = self._lanes[0]['graph_fc'], self._lanes[0]['graph_c']
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        self._hist = []  # the window's input frames (copies), for the exact re-run of a window that holds a short frame
        _LIVE.add(self)

    # ---- pieces (each runs eagerly during warm-up and inside a capture afterwards) ----
    def _frame_entry(self, frame=None):
        from . import native
        m = self.model
        with native.fewrow_split(self.fewrow_split):
            c4 = m(img=self.frame if frame is None else frame, img_meta=[self.meta], backbone_feat=True)[0]
            return m.frame_tensors(c4, self.meta)

    def _push_from(self, e):
        n, T = self.n, self.T
        # shift by one frame through a scratch copy (source and destination overlap), then append
        self.f1_tmp.copy_(self.f1[n:])
        self.f1[:(T - 1) * n].copy_(self.f1_tmp)
        self.f1[(T - 1) * n:].copy_(e['f1'])
        self.props_tmp.copy_(self.props[1:])
        self.props[:T - 1].copy_(self.props_tmp)
        self.props[T - 1].copy_(e['props'])
        self.counts_tmp.copy_(self.counts[1:])
        self.counts[:T - 1].copy_(self.counts_tmp)
        self.counts[T - 1:].copy_(e['count'])

    def _window(self):
        m, n = self.model, self.n
        cur_range = dict(start=self.key * n, length=n)
        key_rois = torch.cat([self.props.new_zeros((n, 1)), self.props[self.key, :, :4]], dim=1)
        return m.head_device_outputs(self.f1, cur_range, key_rois, self.counts, self.meta, rescale=self.rescale)

    def _exact(self, hist):
        if len(hist) != self.T:
            raise RuntimeError('a window that holds a short frame is re-run from its %d input frames, but only %d have been pushed since '
                               'this object was built' % (self.T, len(hist)))
        with torch.no_grad():
            frames = torch.cat(hist, 0)
            metas = [self.meta] * frames.shape[0]
            c4 = self.model(img=frames, img_meta=metas, backbone_feat=True)[0]
            return self.model.forward_feat(x=c4, img_meta=metas, rescale=self.rescale, speculate=False)

    # ---- the loop ----
    @contextlib.contextmanager
    def _window_stream(self, behind_caller=True):
        """The stream the window buffers are worked on: the caller's current stream, or -- window_cus -- the CU-masked stream,
        made current for the block and (behind_caller) ordered behind what the caller has enqueued so far.  emit() does not ask
        for that order: what graph W reads was written by graphs on this very stream, and a wait on the caller's stream is a wait
        on whatever shares its hardware queue -- HIP streams outnumber hardware queues (GPU_MAX_HW_QUEUES = 4), a queue is served
        in order, and with the caller's stream on the frame stream's queue the event sat behind the ~150 launches of graph FC:
        the window ran AFTER the frame it was meant to run beside (266 instead of 425 frames/s)."""
        cur = torch.cuda.current_stream(self.frame.device)
        if not self.window_cus:
            yield cur
            return
        if self._wstream is None:
            from . import native
            self._wstream = native.cu_masked_stream(self.frame.device, 0, int(self.window_cus))
        if behind_caller and cur != self._wstream:
            self._wstream.wait_stream(cur)
        with torch.cuda.stream(self._wstream):
            yield self._wstream

    def push(self, frame=None):
        """A new frame arrives: graph F (its rows enter the window buffers)."""
        _check_live(self)
        with self._window_stream() as st:
            if frame is not None:
                self.frame.copy_(frame, non_blocking=True)
                if frame.is_cuda and self.window_cus:
                    # the copy runs on the confined window stream, not on the stream `frame` was made on: a temporary handed in
                    # (push(decode())) must outlive it.  (A caller that REWRITES the same tensor must order that write itself.)
                    frame.record_stream(st)
            self.graph_f.replay()
            self._hist = (self._hist + [self.frame.clone()])[-self.T:]

    def push_async(self, frame):
        """A new frame arrives: its per-frame part (graph FC) starts on a frame stream -- the next lane's in turn -- and runs beside
        whatever the caller's stream does next (normally `emit()` of the current window) and beside the other lanes' frames.
        `commit()` moves the rows of the OLDEST frame in flight into the window buffers."""
        lane = self._lanes[self._push_turn % len(self._lanes)]
        assert lane['pending'] is None, 'commit() the frame in flight first (%d frame lane(s))' % len(self._lanes)
        cur = torch.cuda.current_stream(self.frame.device)
        lane['stream'].wait_stream(cur)                 # `frame` is produced on the caller's stream
        if lane['ev_commit'] is not None:
            lane['stream'].wait_event(lane['ev_commit'])  # the staging rows of this lane's previous frame have been taken
        _check_live(self)
        with torch.cuda.stream(lane['stream']):
            lane['frame'].copy_(frame, non_blocking=True)
            if frame.is_cuda:
                frame.record_stream(lane['stream'])   # the caller may drop or recycle `frame` before the copy has run on the frame stream
            lane['pending'] = lane['frame'].clone()   # (on the frame stream: nothing of the loop is enqueued on the caller's)
            lane['graph_fc'].replay()
            lane['ev_fc'].record(lane['stream'])
        self._push_turn += 1

    @property
    def _pending_frame(self):
        return self._lanes[self._commit_turn % len(self._lanes)]['pending']

    def commit(self):
        """The oldest frame started by push_async() enters the window buffers (graph C on the caller's stream, behind its graph FC)."""
        lane = self._lanes[self._commit_turn % len(self._lanes)]
        assert lane['pending'] is not None, 'push_async() first'
        with self._window_stream() as st:
            st.wait_event(lane['ev_fc'])
            lane['graph_c'].replay()
            lane['ev_commit'] = torch.cuda.Event()
            lane['ev_commit'].record(st)
        self._hist = (self._hist + [lane['pending']])[-self.T:]
        lane['pending'] = None
        self._commit_turn += 1

    def push_batch(self, frames):
        """`lookahead` frames arrive together: their per-frame rows are computed in one batch (graph FB) and staged; call
        `advance(i)` to move frame i of the batch into the window buffers."""
        assert self.graph_fb is not None and frames.shape[0] == self.lookahead
        _check_live(self)
        if self._wstream is not None:   # (the staging rows may still be read by advance() graphs on the window stream)
            torch.cuda.current_stream(self.frame.device).wait_stream(self._wstream)
        self.batch.copy_(frames, non_blocking=True)
        self.graph_fb.replay()
        self._batch_frames = frames.clone()   # (the caller may reuse its decode buffer: a re-run of a short-frame window reads these)

    def advance(self, i):
        with self._window_stream():
            self._stage_graphs[i].replay()
        self._hist = (self._hist + [self._batch_frames[i:i + 1]])[-self.T:]

    def repeat_last(self):
        with self._window_stream():
            self.graph_p.replay()
        self._hist = (self._hist + [self._hist[-1]])[-self.T:]

    def emit(self):
        k = self._turn % len(self._graphs_w)
        _check_live(self)
        prev = self._unread.get(k)
        if prev is not None and not prev.read:
            raise RuntimeError('emitting into output slot %d of %d before result() of its previous window was called' % (k, len(self._graphs_w)))
        self._turn += 1
        with self._window_stream(behind_caller=False) as st:
            self._graphs_w[k].replay()
            ev = torch.cuda.Event()
            ev.record(st)
        hist = list(self._hist[-self.T:])   # this window's frames: later pushes must not change what a re-run sees
        pend = PendingGraphWindow(self, self._outs[k], ev, lambda: self._exact(hist))
        self._unread[k] = pend
        return pend
