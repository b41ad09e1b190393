"""Training-path ops of the relation heads with HIP backwards.  Parameters are f32 masters; the compute
dtype is the activations': f32 (exact-f32 MFMA path, the parity mode) or bf16 (operands rounded to bf16, f32 accumulation,
f32 weight gradients -- the throughput mode, `set_compute_dtype(model, torch.bfloat16)`).

The reference trains through plain autograd over nn.Linear / nn.Conv2d / torch.bmm / nn.Softmax and its loss modules
(selsa_bbox_head.py:108-261, bbox_head.py:100-130).  Here every forward AND backward product is a tile-engine GEMM:

    y  = act(x W^T + b (+ resid))        hvr_gemm with the fused epilogue
    dz = dy * (y > 0)                    hvr_relu_bwd          (only when the epilogue had the ReLU)
    dx = dz W                            hvr_gemm   (W^T made K-contiguous by hvr_transpose_pad)
    dW = dz^T x                          hvr_gemm   (both operands transposed / zero-padded to the K-step)
    db = column sums of dz               hvr_colsum
    d resid = dz

plus `ops.relation` (relation core) and `det_loss` (BBoxHead.loss: cross entropy + smooth-L1 + accuracy in one kernel).
torch only moves memory here (views, cat, zero-padding) and drives the autograd graph.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import native


# ---- weight gradients off the critical path --------------------------------------------------------------------------
# In a backward pass only the data gradients form a chain; a conv's weight gradient (two transposes, a split-K GEMM, the
# unpack) is a leaf that nothing waits for until the all-reduce.  With `wgrad_overlap(True)` ConvFunction.backward enqueues
# that leaf on a second HIP stream and adds the result straight into the parameter's .grad (which FlatParams pre-allocates
# as a view of the flat gradient buffer), returning None to autograd for it; the main stream goes on with the previous
# layer's data gradient, whose kernels are small enough (57 row tiles for 256 CUs at three frames) to leave room.
# `join_wgrad()` makes the main stream wait for the side stream: call it before the gradients are read (dist_train does).
_overlap = dict(on=False, streams={})


def wgrad_overlap(flag):
    prev = _overlap['on']
    _overlap['on'] = bool(flag)
    return prev


def _side_stream(device):
    key = str(device)
    if key not in _overlap['streams']:
        _overlap['streams'][key] = torch.cuda.Stream(device=device)
    return _overlap['streams'][key]


def join_wgrad():
    for st in _overlap['streams'].values():
        torch.cuda.current_stream(st.device).wait_stream(st)


def _pad_cols(t, n):
    """[M, N] -> [M, n] with zero columns appended (K-step padding of a GEMM operand)."""
    if t.shape[1] == n and t.is_contiguous():
        return t
    out = t.new_zeros((t.shape[0], n))
    out[:, :t.shape[1]] = t
    return out


def _two_byte(dtype):
    return dtype in (torch.bfloat16, torch.float16)


# ---- the weight side of an iteration in two launches (hvr_pack_conv_weights_multi + hvr_transpose_multi) ------------------------------
# Every trainable conv / linear layer needs its f32 master weight (x the frozen BatchNorm scale) in the kernels' operand layout for the
# forward, and that operand transposed (1x1 / linear) or rotated (KxK) for the input gradient.  The first iteration of a
# dist_train.train_iteration loop RECORDS the layers as they run (weights that are parameters, or views of one: stable addresses in
# FlatParams' buffer); from the second on prep_begin() rebuilds all of them with two launches before the forward and the Functions
# pick their operands out of the table.  Outside such a loop (tests, one-off calls) nothing is cached: every layer prepares its own.
_prep = dict(recording=False, valid=False, entries={}, tables=None, owner=None, enabled=True, stamp=0)


def _prep_key(w):
    base = w if w.is_leaf else w._base
    if base is None or not isinstance(base, torch.nn.Parameter) or not w.is_contiguous():
        return None
    return (w.data_ptr(), tuple(w.shape))


def _prep_lookup(w, s, shape4, dtype):
    """-> (packed operand or None, key): the table's operand when this iteration's table holds the layer; records it while recording."""
    if not (_prep['valid'] or _prep['recording']) or not _two_byte(dtype) or shape4[1] % 8:
        return None, None
    key = _prep_key(w)
    if key is None:
        return None, None
    e = _prep['entries'].get(key)
    if _prep['valid'] and e is not None and e['dtype'] == dtype and e['s'] is s and e['eff'] is not None:
        return e['eff'], (key, _prep['stamp'])   # (the stamp: the table's buffers are rewritten by the next prep_begin -- see _prep_check)
    if _prep['recording'] and e is None:
        _prep['entries'][key] = dict(w=w.detach(), s=s, shape=tuple(shape4), dtype=dtype, eff=None, aux=None)
    return None, None


def table_operand(w, s, shape4, dtype):
    """This iteration's packed operand [Cout][KH][KW][Cin] of conv weight w x scale s when the weight table holds it (a no-grad forward
    inside train_iteration that packs for itself: same numbers, no launch), else None.  The buffer is rewritten by the next prep_begin:
    the caller's cache must not outlive the iteration (PackedMixin caches are dropped by FlatParams.sgd_step)."""
    if not _prep['valid'] or not _two_byte(dtype):
        return None
    key = _prep_key(w)
    e = _prep['entries'].get(key) if key is not None else None
    if e is not None and e['dtype'] == dtype and e['eff'] is not None and e['shape'] == tuple(shape4) and \
            (e['s'] is s or (e['s'].data_ptr() == s.data_ptr() and e['s']._version == s._version)):
        return e['eff']
    return None


def _prep_check(key):
    """A Function whose forward took its operand out of the iteration's table saved a TABLE BUFFER for its backward: the next
    prep_begin() rewrites it (and the update in between made it stale).  A backward that runs after that -- retain_graph across
    iterations, gradient accumulation over several train_iteration calls -- would silently use the wrong weights: refuse it."""
    if key is not None and (key[1] != _prep['stamp'] or not _prep['valid']):
        raise RuntimeError('backward of a layer whose operands came from the weight table of an earlier iteration (train_ops.prep_begin has '
                           'run since, or prep_end has closed the iteration): run forward and backward inside one train_iteration, or '
                           'train_ops.prep_enable(False)')


def _prep_aux(key):
    _prep_check(key)
    e = _prep['entries'].get(key[0]) if (key is not None and _prep['valid']) else None
    return None if e is None else e['aux']


def _prep_build():
    """One pair of multi-tensor launches per (operand dtype, device) among the recorded layers (a model that mixes bf16 and half layers
    gets one table per format: every entry is packed in ITS recorded dtype, which is what _prep_lookup matches on)."""
    groups = {}
    for e in _prep['entries'].values():
        groups.setdefault((e['dtype'], e['w'].device), []).append(e)
    tables = []
    for (dtype, dev), ents in groups.items():
        step = native.kstep(dtype)
        packs, trans, first, tile0 = [], [], 0, 0
        for e in ents:
            Cout, Cin, KH, KW = e['shape']
            KK = KH * KW
            e['eff'] = torch.empty((Cout, KH, KW, Cin), dtype=dtype, device=dev)
            packs.append(native.PackItem(w=e['w'].data_ptr(), scale=e['s'].data_ptr(), out=e['eff'].data_ptr(), first=first, Cout=Cout, Cin=Cin, KK=KK))
            first += Cout * KK * Cin
            es = e['eff'].element_size()
            tiles_r, tiles_c = (Cout + 63) // 64, (Cin + 63) // 64
            if KK == 1:
                ldn = (Cout + step - 1) // step * step
                e['aux'] = torch.zeros((Cin, ldn), dtype=dtype, device=dev)
                trans.append(native.TransposeItem(src=e['eff'].data_ptr(), dst=e['aux'].data_ptr(), lds=Cin, ldd=ldn, R=Cout, C=Cin, first_tile=tile0,
                                                  tiles_c=tiles_c, dcols=ldn))
                tile0 += ((ldn + 63) // 64) * tiles_c
            else:
                e['aux'] = torch.zeros((Cin, KH, KW, Cout), dtype=dtype, device=dev)   # [Cin][KH][KW][Cout], taps reversed (the dX conv's operand)
                if Cout % 8:
                    e['aux'] = None
                    continue
                for t in range(KK):
                    trans.append(native.TransposeItem(src=e['eff'].data_ptr() + t * Cin * es, dst=e['aux'].data_ptr() + (KK - 1 - t) * Cout * es,
                                                      lds=KK * Cin, ldd=KK * Cout, R=Cout, C=Cin, first_tile=tile0, tiles_c=tiles_c, dcols=Cout))
                    tile0 += tiles_r * tiles_c
        tables.append(dict(packs=native.items_to_device(packs, dev), n_packs=len(packs), total=first, dtype=dtype,
                           trans=native.items_to_device(trans, dev) if trans else None, n_trans=len(trans), tiles=tile0))
    _prep['tables'] = tables or None


def prep_begin(owner=None):
    """Start of an iteration (dist_train.train_iteration; owner = its FlatParams: another owner starts a new table): rebuild every
    recorded layer's operands, or start recording."""
    if not _prep['enabled']:
        return
    if owner is not _prep['owner']:
        prep_reset()
        _prep['owner'] = owner
    _prep['stamp'] += 1
    if _prep['tables'] is None:
        _prep['recording'], _prep['valid'] = True, False
        return
    for t in _prep['tables']:
        native.pack_conv_weights_multi(t['packs'], t['n_packs'], t['total'], t['dtype'])
        if t['trans'] is not None:
            native.transpose_multi(t['trans'], t['n_trans'], t['tiles'])
    _prep['valid'] = True


def prep_end():
    """End of an iteration (after the update: the table's operands are stale from here on)."""
    _prep['valid'] = False
    if _prep['recording']:
        _prep['recording'] = False
        _prep_build()


def prep_reset():
    wgrad_reset()      # (the parked-gradient slabs and tables belong to the loop that is ending too)
    _prep.update(recording=False, valid=False, entries={}, tables=None, owner=None, stamp=_prep['stamp'] + 1)


def prep_enable(flag):
    """The per-iteration weight table on / off (off: every layer prepares its own operands, as outside a training loop); -> previous."""
    prev = _prep['enabled']
    _prep['enabled'] = bool(flag)
    prep_reset()
    return prev


# Weight gradients of convs go STRAIGHT into the parameter's gradient buffer when it has one (dist_train.FlatParams gives every
# parameter a view of the flat gradient buffer, zeroed at the start of the step): the f32 product is scaled, permuted and ADDED in
# place by hvr_unpack_conv_wgrad and autograd is handed no gradient for the weight -- no temporary, no AccumulateGrad add per layer
# (151 torch launches per iteration in round 4).  Same values: one f32 addition to a zeroed buffer.
# OFF by default: handing autograd no gradient bypasses tensor / post-accumulate hooks and mutates .grad under
# torch.autograd.grad(); dist_train.train_iteration -- which owns the gradient buffer and zeroes it first -- switches it on for its own
# backward (`wgrad_direct(True)` ... restore).  Outside that scope the Functions return dW to autograd like any other.
# Linear layers (round 6): the same idea without a kernel of their own.  dW = dz^T x already has the parameter's layout, so the product
# (and the bias gradient's column sums) is WRITTEN into the gradient view when this is the first contribution of the iteration -- the
# buffer holds the zeros zero_grad left -- and added to it otherwise; autograd gets None and runs no AccumulateGrad add for the pair
# (two launches and, for fc_new_1, 150 MB of f32 traffic per layer).  `pristine`: ids of the parameters whose gradient still is
# zero_grad's zero; train_iteration hands the set in, every Function that adds to a parameter's gradient takes the parameter out.
# Contract of the scope: a parameter consumed by train_ops.linear / conv_* inside train_iteration receives its gradient through those
# Functions only (true of every module here: the two read-out layers, whose weights reach the product through torch.cat, are not leaves
# of the Function and take autograd's own path).
_direct = dict(on=False, pristine=None)


def wgrad_direct(flag, pristine=None):
    prev = (_direct['on'], _direct['pristine'])
    if isinstance(flag, tuple):
        flag, pristine = flag
    _direct['on'], _direct['pristine'] = bool(flag), (pristine if flag else None)
    return prev


def _claim(param):
    """True when `param`'s gradient buffer still holds zero_grad's zeros (and from now on it does not)."""
    pr = _direct['pristine']
    if pr is not None and id(param) in pr:
        pr.discard(id(param))
        return True
    return False


def _direct_target(t):
    """The parameter behind t (t itself, or the parameter t is a whole, contiguous view of) when a gradient may be written straight into
    its f32 gradient buffer, else None."""
    base = t if t.is_leaf else t._base
    if base is None or not isinstance(base, torch.nn.Parameter) or not base.requires_grad or not t.is_contiguous():
        return None
    if base is not t and (t.data_ptr() != base.data_ptr() or t.numel() != base.numel() or not base.is_contiguous()):
        return None
    return base


# Weight gradients of a stage's identical blocks in ONE product (round 6).  At three frames per rank a conv's dW = dZ^T X is a few-tile,
# long-K product: K slices + a reduce + the unpack = three launches per conv that leave most of the chip idle, ~100 convs per iteration.
# Inside train_iteration (direct mode) ConvFunction.backward only PARKS its two K-contiguous operands -- dZ^T and the transposed patch
# matrix, which its one-pass kernels write straight into the next free slot of a per-shape slab -- and `wgrad_flush()` after the backward
# pass runs one batched product (hvr_gemm_splitk_batched: the tile engine's batch dimension) and one table-driven unpack per shape
# class: layer 3's 69 convs become 3 + 3 launches.  Slabs are sized from the previous iteration's counts (the first iteration of a loop
# computes every gradient on the spot and only counts); a layer that finds its class full computes on the spot.  Values: the same
# products, accumulated over K in slices whose count follows the batched grid (the f32 summation order of a gradient may differ from
# the single call's, as between any two slice counts).
_wq = dict(on=False, classes={}, seen={})


def wgrad_defer(flag):
    """Parking of conv weight gradients on / off (dist_train.train_iteration: on for its backward); -> previous."""
    prev = _wq['on']
    _wq['on'] = bool(flag)
    if flag:
        for c in _wq['classes'].values():
            c['n'], c['items'] = 0, []
        _wq['seen'] = {}
    return prev


def wgrad_reset():
    _wq.update(on=False, classes={}, seen={})


def _wq_slot(key, Cout, KC, ldp, dtype, device):
    """-> (class, slot index) with room for one more layer of this shape, or None (compute on the spot)."""
    if not _wq['on']:
        return None
    _wq['seen'][key] = _wq['seen'].get(key, 0) + 1
    c = _wq['classes'].get(key)
    if c is None or c['n'] >= c['cap']:
        return None
    c['n'] += 1
    return c, c['n'] - 1


def wgrad_flush():
    """The parked weight gradients: one batched product + one unpack per shape class, added into the parameters' gradient buffers.
    Then the slabs are (re)sized for the next iteration from this one's counts."""
    for key, c in _wq['classes'].items():
        n = c['n']
        if n == 0:
            continue
        native.gemm_splitk_batched(c['dzt'][:n], c['cols'][:n], out=c['dw'][:n])     # [n, Cout, KK * Cin] f32
        sig = tuple((wp.grad.data_ptr(), s_.data_ptr()) for wp, s_ in c['items'])
        if c.get('table_sig') != sig:            # (the same layers in the same order every iteration: built once)
            Cout, Cin, KH, KW = c['shape']
            per = Cout * Cin * KH * KW
            items = [native.PackItem(w=wp.grad.data_ptr(), scale=s_.data_ptr(), out=c['dw'].data_ptr() + i * per * 4, first=i * per, Cout=Cout, Cin=Cin, KK=KH * KW)
                     for i, (wp, s_) in enumerate(c['items'])]
            c['table'], c['table_sig'], c['total'] = native.items_to_device(items, c['dw'].device), sig, n * per
        native.unpack_conv_wgrads_multi(c['table'], n, c['total'], accumulate=True)
        c['n'], c['items'] = 0, []
    # classes for the next iteration: every shape that came up at least twice gets a slab of that many slots
    for key, cnt in _wq['seen'].items():
        c = _wq['classes'].get(key)
        if cnt >= 2 and (c is None or c['cap'] < cnt):
            Cout, KC, ldp, dtype, device, shape = key[0], key[1], key[2], key[3], key[4], key[5]
            _wq['classes'][key] = dict(cap=cnt, n=0, items=[], shape=shape,
                                       dzt=torch.empty((cnt, Cout, ldp), dtype=dtype, device=device),
                                       cols=torch.empty((cnt, KC, ldp), dtype=dtype, device=device),
                                       dw=torch.empty((cnt, Cout, KC), dtype=torch.float32, device=device))
    _wq['seen'] = {}


class LinearFunction(Function):
    """y = act(x @ w^T + b (+ resid)); x [M, K], w [N, K], b [N]; N a multiple of 4, K a multiple of the K-step.
    x's dtype is the compute dtype: with bf16 activations the f32 master weight is rounded to bf16 on the way in, products
    accumulate in f32, dx comes back in bf16 and dW / db in f32 (the master's dtype).  out_f32: f32 output from bf16 operands
    (the layer that feeds a loss kernel)."""

    @staticmethod
    def forward(ctx, x, w, b, resid, relu, out_f32=False):
        if not x.is_cuda:
            raise NotImplementedError('the head runs on the GPU only (no CPU fallback)')
        x = x.contiguous()
        wc, key = _prep_lookup(w, ones(w.shape[0], x.device), (w.shape[0], w.shape[1], 1, 1), x.dtype) if w.dim() == 2 else (None, None)
        wc = native.cast(w.contiguous(), x.dtype) if wc is None else wc.view(w.shape[0], w.shape[1])
        ctx.prep_key = key
        assert not (relu and out_f32 and x.dtype != torch.float32), 'the ReLU mask is kept in the compute dtype'
        y = native.gemm(x, wc, b, resid=resid.contiguous() if resid is not None else None, relu=bool(relu), out_f32=bool(out_f32))
        ctx.relu, ctx.has_resid, ctx.has_bias = bool(relu), resid is not None, b is not None
        ctx.w_param = _direct_target(w) if w.requires_grad else None     # where the weight / bias gradients may land directly
        ctx.b_param = _direct_target(b) if (b is not None and b.requires_grad) else None
        ctx.save_for_backward(x, wc, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        _prep_check(ctx.prep_key)
        dy = native.cast(dy.contiguous(), x.dtype)
        M, K = x.shape
        N = w.shape[0]
        step = native.kstep(x.dtype)
        ldn, ldm = (N + step - 1) // step * step, (M + step - 1) // step * step
        dzt = None
        if ctx.relu and _two_byte(x.dtype) and N % 8 == 0 and ctx.needs_input_grad[1]:
            dz, dzt = native.relu_bwd_t(dy, y, ldm)      # the ReLU mask and dz^T (the weight gradient's K-contiguous operand) in one pass
        else:
            dz = native.relu_bwd(dy, y) if ctx.relu else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _prep_aux(ctx.prep_key)
            dx = native.gemm(_pad_cols(dz, ldn), wt if wt is not None else native.transpose_pad(w, ldn))   # [M, K] = dz [M, N] W [N, K]
        def grad_buf(p_, shape):
            g = p_.grad if (_direct['on'] and p_ is not None) else None
            return g.view(shape) if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == N * (K if len(shape) == 2 else 1)) else None

        if ctx.needs_input_grad[1]:
            ops_ = (dzt if dzt is not None else native.transpose_pad(dz, ldm), native.transpose_pad(x, ldm))   # [N, K] = dz^T x, f32
            gw = grad_buf(ctx.w_param, (N, K))
            if gw is None:
                dw = native.gemm_splitk(*ops_)
            elif _claim(ctx.w_param):
                native.gemm_splitk(*ops_, out=gw)      # first contribution of the iteration: written over zero_grad's zeros
            else:
                gw.add_(native.gemm_splitk(*ops_))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = grad_buf(ctx.b_param, (N,))
            if gb is None:
                db = native.colsum(dz)
            elif _claim(ctx.b_param):
                native.colsum(dz, out=gb)
            else:
                gb.add_(native.colsum(dz))
        dr = dz if (ctx.has_resid and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dr, None, None


def linear(x, w, b=None, resid=None, relu=False, out_f32=False):
    return LinearFunction.apply(x, w, b, resid, relu, out_f32)


_colsel = {}


def _scale_columns(d, g, first, n_first):
    """d [R, ld] = the gradient of sum(total) w.r.t. a fused logit matrix whose columns first .. first + n_first - 1 belong to total[0]
    (the classification term) and all others to total[1]: -> the gradient under the upstream g [2], d * g[term of the column].  Two small
    launches and no host read (the unit-gradient check this replaces read g back in the middle of the backward pass)."""
    key = (int(d.shape[1]), int(first), int(n_first), str(d.device))
    idx = _colsel.get(key)
    if idx is None:
        col = torch.arange(d.shape[1], device=d.device)
        idx = _colsel[key] = ((col < first) | (col >= first + n_first)).long()
    return d * g.to(d.dtype)[idx][None, :]


class DetLossFunction(Function):
    """(total, loss_cls, loss_bbox, acc) of BBoxHead.loss on a fused f32 logit matrix [R, ld] (class logits at cls_off,
    box deltas at reg_off); only `total` = w_cls * loss_cls + w_bbox * loss_bbox carries a gradient."""

    @staticmethod
    def forward(ctx, logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, beta, w_cls, w_bbox):
        out3, dlogits = native.det_loss(logits.contiguous(), cls_off, reg_off, ncls, labels, label_weights, bbox_targets,
                                        bbox_weights, beta, w_cls, w_bbox)
        ctx.save_for_backward(dlogits)
        ctx.cols = (int(cls_off), int(ncls))
        if (w_cls, w_bbox) != (1.0, 1.0):
            raise NotImplementedError('loss weights other than 1 (the two configs use loss_weight=1.0) are not wired up')
        # shapes as the reference returns them: scalar losses, accuracy of shape [1] (losses/accuracy.py:19-21)
        loss_cls, loss_bbox, acc = out3[0], out3[1], out3[2:3]
        total = out3[0:2].clone()  # the two loss terms; det_loss's gradient is d(sum(total)) / d logits
        ctx.mark_non_differentiable(loss_cls, loss_bbox, acc)
        return total, loss_cls, loss_bbox, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_cls, g_bbox, g_acc):
        # g_total: gradient w.r.t. the two weighted loss terms; the usual (loss_cls + loss_bbox).backward() gives ones
        dlogits, = ctx.saved_tensors
        return (_scale_columns(dlogits, g_total, *ctx.cols),) + (None,) * 10


def det_loss(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, beta=1.0, w_cls=1.0, w_bbox=1.0):
    """-> dict(loss_cls, loss_bbox (scalars), acc [1], total [2] = the two loss terms; `total.sum().backward()` trains)."""
    total, lc, lb, acc = DetLossFunction.apply(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights,
                                               float(beta), float(w_cls), float(w_bbox))
    return dict(total=total, loss_cls=lc, loss_bbox=lb, acc=acc)


class DetLossSampledFunction(Function):
    """DetLossFunction in the OHEM form (selsa_rcnn.py:224-232): the rows cat(pos_inds, neg_inds) picked by the loss-ranked
    sampler carry weight 1, all others 0; `sel_counts` int32 [2] (device) = how many were picked of each kind."""

    @staticmethod
    def forward(ctx, logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, sel_counts, beta):
        out3, dlogits = native.det_loss_sampled(logits.contiguous(), cls_off, reg_off, ncls, labels, label_weights, bbox_targets,
                                                bbox_weights, sel_counts, beta)
        ctx.save_for_backward(dlogits)
        ctx.cols = (int(cls_off), int(ncls))
        loss_cls, loss_bbox, acc = out3[0], out3[1], out3[2:3]
        total = out3[0:2].clone()
        ctx.mark_non_differentiable(loss_cls, loss_bbox, acc)
        return total, loss_cls, loss_bbox, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_cls, g_bbox, g_acc):
        dlogits, = ctx.saved_tensors
        return (_scale_columns(dlogits, g_total, *ctx.cols),) + (None,) * 9


def det_loss_sampled(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, sel_counts, beta=1.0):
    total, lc, lb, acc = DetLossSampledFunction.apply(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights,
                                                      sel_counts, float(beta))
    return dict(total=total, loss_cls=lc, loss_bbox=lb, acc=acc)


class RpnLossFunction(Function):
    """AnchorHead.loss_single (anchor_head.py:141-160) on the fused RPN head output o [rows, ld] of ONE frame (A objectness
    logits, then 4A deltas per position): -> (total [2], loss_rpn_cls, loss_rpn_bbox); only `total` carries a gradient.
    counts int32 [2] (device): sampled positives / negatives; avg_factor = max(c0,1) + max(c1,1)."""

    @staticmethod
    def forward(ctx, o, A, labels, label_weights, bbox_targets, bbox_weights, counts, beta):
        out2, d_o = native.rpn_loss(o.contiguous(), A, labels, label_weights, bbox_targets, bbox_weights, counts, beta)
        ctx.save_for_backward(d_o)
        ctx.cols = (0, int(A))
        loss_cls, loss_bbox = out2[0], out2[1]
        total = out2.clone()
        ctx.mark_non_differentiable(loss_cls, loss_bbox)
        return total, loss_cls, loss_bbox

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_cls, g_bbox):
        d_o, = ctx.saved_tensors
        return (_scale_columns(d_o, g_total, *ctx.cols),) + (None,) * 7


def rpn_loss(o, A, labels, label_weights, bbox_targets, bbox_weights, counts, beta=1.0 / 9.0):
    total, lc, lb = RpnLossFunction.apply(o, int(A), labels, label_weights, bbox_targets, bbox_weights, counts, float(beta))
    return dict(total=total, loss_rpn_cls=lc, loss_rpn_bbox=lb)


class TripletMarginFunction(Function):
    """STAND-IN for `TripletNonLocalLoss(margin).compute_loss(q, k, labels, [anchors, pos, neg])` (hrnmp_bbox_head.py:555-561): the
    fork of pytorch_metric_learning that defines it is not in the reference tree, so this is the library's published
    TripletMarginLoss over the mined triples (anchors from q, positives / negatives from k; include/hvr_hip.h has the formula).
    -> (loss 0-dim, active triples 0-dim); parity with the reference is unpinned by construction."""

    @staticmethod
    def forward(ctx, q, k, anchor_idx, pos_idx, neg_idx, margin):
        out2, dq, dk = native.triplet_margin(q.contiguous(), k.contiguous(), anchor_idx, pos_idx, neg_idx, margin)
        ctx.save_for_backward(dq, dk)
        ctx.dtypes = (q.dtype, k.dtype)
        active = out2[1]
        ctx.mark_non_differentiable(active)
        return out2[0].clone(), active

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, g_active):
        dq, dk = ctx.saved_tensors
        return native.cast(dq * g_loss, ctx.dtypes[0]), native.cast(dk * g_loss, ctx.dtypes[1]), None, None, None, None


def triplet_margin(q, k, anchor_idx, pos_idx, neg_idx, margin):
    return TripletMarginFunction.apply(q, k, anchor_idx, pos_idx, neg_idx, float(margin))


class ConvFunction(Function):
    """y = act(conv(x, w * s) + t (+ resid)) on physical NHWC tensors: an nn.Conv2d (no bias) followed by a frozen
    BatchNorm (scale s, shift t per output channel; pass s = ones / t = bias for a plain conv with bias) and optionally the
    residual add and ReLU of a Bottleneck (resnet.py:220-266).  x [B,H,W,Cin], w [Cout,Cin,KH,KW] (the nn.Conv2d
    parameter), stride 1 for KxK kernels; a strided 1x1 conv is a row subset followed by a linear layer.
    Backward: ReLU mask, dX by the same implicit-GEMM conv kernel on the rotated weights (or a GEMM for 1x1), dW^T as a
    GEMM of dZ^T with the patch matrix, both scaled by s; s and t are frozen and get no gradient.
    x's dtype is the compute dtype (bf16 activations: the scaled f32 master weight is rounded to bf16 on the way in, dX is
    bf16, dW accumulates and is returned in f32); out_f32 asks for an f32 output from bf16 operands (the RPN's 1x1 heads)."""

    @staticmethod
    def forward(ctx, x, w, s, t, resid, relu, stride, pad, dil, out_f32=False):
        if not x.is_cuda:
            raise NotImplementedError('convs run on the GPU only (no CPU fallback)')
        Cout, Cin, KH, KW = w.shape
        assert (KH, KW) == (1, 1) or stride == 1, 'strided KxK convs are not on this path (caffe-style ResNet strides its 1x1s)'
        xs = x[:, ::stride, ::stride, :].contiguous() if stride > 1 else x.contiguous()
        s = s.float().contiguous()
        w_eff, key = _prep_lookup(w, s, (Cout, Cin, KH, KW), x.dtype)                    # this iteration's table (prep_begin), or:
        if w_eff is None:
            w_eff = native.pack_conv_weight(w.contiguous(), s, x.dtype)                  # [Cout][KH][KW][Cin] * s, compute dtype
        ctx.prep_key = key
        y = native.conv2d_nhwc(xs, w_eff, t, resid.contiguous() if resid is not None else None, relu=bool(relu), pad=pad, dil=dil,
                               out_f32=bool(out_f32))
        ctx.cfg = (bool(relu), int(stride), int(pad), int(dil), resid is not None, tuple(x.shape))
        ctx.w_param = w if (w.is_leaf and w.requires_grad) else None       # where an off-stream weight gradient may land
        ctx.save_for_backward(xs, w_eff, s, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xs, w_eff, s, y = ctx.saved_tensors
        _prep_check(ctx.prep_key)     # (w_eff may be a buffer of the iteration's weight table: it must still be this iteration's)
        relu, stride, pad, dil, has_resid, x_shape = ctx.cfg
        dy = native.cast(dy.contiguous(), xs.dtype)
        Cout, KH, KW, Cin = w_eff.shape
        B, OH, OW, _ = dy.shape
        P = B * OH * OW
        step = native.kstep(dy.dtype)
        ldp = (P + step - 1) // step * step
        fast = _two_byte(dy.dtype) and Cout % 8 == 0 and Cin % 8 == 0     # the one-pass K-contiguous operands (hvr_relu_bwd_t / hvr_im2col_t)
        wp = ctx.w_param
        has_grad_buf = wp is not None and wp.grad is not None and wp.grad.is_contiguous() and wp.grad.dtype == torch.float32
        # parked weight gradient (see _wq): this layer's slot in its shape class's slab, when the class has room
        slot = None
        if ctx.needs_input_grad[1] and fast and _direct['on'] and not _overlap['on'] and has_grad_buf:
            slot = _wq_slot((Cout, KH * KW * Cin, ldp, dy.dtype, dy.device, (Cout, Cin, KH, KW)), Cout, KH * KW * Cin, ldp, dy.dtype, dy.device)
        dzt = None
        if relu and fast and ctx.needs_input_grad[1]:
            dz2, dzt = native.relu_bwd_t(dy.view(P, Cout), y.view(P, Cout), ldp, dzt_out=slot[0]['dzt'][slot[1]] if slot else None)
            dz = dz2.view(B, OH, OW, Cout)
        else:
            dz = native.relu_bwd(dy, y) if relu else dy
            dz2 = dz.view(P, Cout)
        dx = dw = None
        side_reads_dz = False
        if ctx.needs_input_grad[0]:
            aux = _prep_aux(ctx.prep_key)
            if (KH, KW) == (1, 1):
                ldn = (Cout + step - 1) // step * step
                wt = aux if aux is not None else native.transpose_pad(w_eff.view(Cout, Cin), ldn)
                dxs = native.gemm(_pad_cols(dz2, ldn), wt).view(B, OH, OW, Cin)  # dz W
            else:
                w_rot = aux if aux is not None else w_eff.flip(1, 2).permute(3, 1, 2, 0).contiguous()   # [Cin][KH][KW][Cout], taps reversed
                dxs = native.conv2d_nhwc(dz, w_rot, None, None, relu=False, pad=dil * (KH - 1) - pad, dil=dil)
            if stride > 1:
                dx = dxs.new_zeros(x_shape)
                dx[:, ::stride, ::stride, :] = dxs
            else:
                dx = dxs
        if ctx.needs_input_grad[1]:
            def cols_t(out=None):
                if (KH, KW) == (1, 1):
                    return native.transpose_pad(xs.view(P, Cin), ldp, out=out)
                if fast:
                    return native.im2col_t(xs, KH, KW, pad, dil, ldp, out=out)      # [KH*KW*Cin, ldp]: no row-major patch matrix in between
                return native.transpose_pad(native.im2col_nhwc(xs, KH, KW, pad, dil), ldp, out=out)

            def weight_gradient(into=None):
                dw_eff = native.gemm_splitk(dzt if dzt is not None else native.transpose_pad(dz2, ldp), cols_t())   # [Cout, KH*KW*Cin], f32
                return native.unpack_conv_wgrad(dw_eff, s, (Cout, Cin, KH, KW), accumulate_into=into)   # * s, parameter layout

            if slot is not None:
                cls, i = slot
                if dzt is None:
                    native.transpose_pad(dz2, ldp, out=cls['dzt'][i])
                cols_t(out=cls['cols'][i])
                cls['items'].append((wp, s))
                _claim(wp)
            elif _direct['on'] and not _overlap['on'] and has_grad_buf:
                _claim(wp)                        # (added, not written: the gradient stops being zero_grad's zero either way)
                weight_gradient(into=wp.grad)     # added in place on this stream; autograd gets no gradient for the weight
            elif _overlap['on'] and has_grad_buf:
                _claim(wp)
                main, side = torch.cuda.current_stream(dz.device), _side_stream(dz.device)
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    weight_gradient(into=wp.grad)
                for t_ in (dz, xs, s):
                    t_.record_stream(side)
                side_reads_dz = True
            else:
                dw = weight_gradient()
        dt = native.colsum(dz2) if ctx.needs_input_grad[3] else None   # a trainable bias passed as the shift (RPN / 1x1 heads)
        # dz doubles as the residual's gradient.  Autograd may later add another branch's gradient INTO the tensor it is
        # handed (InputBuffer accumulates in place when it holds the only reference), on the main stream -- while the side
        # stream is still reading dz for the weight gradient.  Off-stream mode therefore hands autograd its own copy.
        dr = (dz.clone() if side_reads_dz else dz) if (has_resid and ctx.needs_input_grad[4]) else None
        return dx, dw, None, dt, dr, None, None, None, None, None


def _frozen_bn_affine(bn):
    """(scale, shift) of a frozen BatchNorm (eval statistics): s = weight / sqrt(var + eps), t = bias - mean * s.  Constant
    while the statistics and the affine stay frozen, so it is computed once and kept until one of the four tensors changes."""
    key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    hit = bn.__dict__.get('_hvr_affine')
    if hit is None or hit[0] != key:
        with torch.no_grad():
            s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            t = (bn.bias - bn.running_mean * s).float().contiguous()
        hit = (key, s, t)
        bn.__dict__['_hvr_affine'] = hit
    return hit[1], hit[2]


def conv_bn(x, conv, bn, resid=None, relu=False):
    """nn.Conv2d (bias-free) + frozen nn.BatchNorm2d (eval statistics, models/utils/norm.py eps) on an NHWC tensor."""
    if bn.training:
        raise NotImplementedError('conv_bn folds a FROZEN BatchNorm: eval statistics (norm_eval=True in both configs); its affine '
                                  'gets no gradient (norm_cfg requires_grad=False)')
    s, t = _frozen_bn_affine(bn)
    return ConvFunction.apply(x, conv.weight, s, t, resid, relu, conv.stride[0], conv.padding[0], conv.dilation[0])


def conv_bias(x, conv, relu=False):
    """nn.Conv2d with a trainable bias and no norm (RPN convs, res5's external 1x1) on an NHWC tensor."""
    return ConvFunction.apply(x, conv.weight, ones(conv.weight.shape[0], x.device), conv.bias, None, relu, conv.stride[0],
                              conv.padding[0], conv.dilation[0])


_ones = {}


def ones(n, device):
    """f32 ones [n] (the unit BatchNorm scale of a plain conv), one tensor per (n, device)."""
    key = (int(n), str(device))
    if key not in _ones:
        _ones[key] = torch.ones(int(n), dtype=torch.float32, device=device)
    return _ones[key]
