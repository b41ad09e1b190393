"""ctypes binding of ``libhvr_hip.so`` (C ABI: ``include/hvr_hip.h``).

PyTorch is used only for device memory and streams: every wrapper here takes CUDA(ROCm)
tensors, passes ``data_ptr()`` + sizes + the current stream to the C ABI and returns
tensors it allocated with the torch allocator.  There is NO CPU fallback: a missing
library or a CPU tensor raises (the reference's RoIAlign raises ``NotImplementedError`` on
CPU input the same way, ``mmdet/ops/roi_align/roi_align.py:27-28``).
"""
import contextlib
import ctypes
import math
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libhvr_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'hvr_hip.h')

HVR_F32, HVR_BF16, HVR_F16, HVR_F16S = 0, 1, 2, 3
ABI_VERSION = 6
# Split-half tensors (HVR_F16S, include/hvr_hip.h: [32 hi | 32 lo] half groups, 4 bytes per logical element) travel through
# torch as int32 tensors of the LOGICAL shape: element size, strides, row / 32-column slicing, cat, clone and zeros all mean
# the right thing on the container, and nothing but this library ever interprets the bytes.  `SPLIT` is the dtype sentinel
# (`set_compute_dtype(model, native.SPLIT)`).
SPLIT = torch.int32
COMPUTE_DTYPES = (torch.bfloat16, torch.float16, SPLIT, torch.float32)
DTYPE_NAMES = {torch.bfloat16: 'bf16', torch.float16: 'f16', SPLIT: 'f16x2', torch.float32: 'f32'}
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1

# default operand staging of the MFMA tile engine (0 register-staged, 1 global->LDS DMA)
STAGING = 1   # operand loader of the tile engine: 1 = LDS-DMA (the product path); 0 = register staging, kept as the second loader the
              # kernel tests cross-check (3 x slower); an argument of the C ABI, not an environment switch
# 0 = the library's cost model picks the tile shape; k > 0 forces shape k-1 (tools/kernel_bench.py sweeps)
TILE_HINT = 0


class HvrError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p),
                ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
                ('lda', ctypes.c_int64), ('ldb', ctypes.c_int64), ('ldc', ctypes.c_int64),
                ('bias', ctypes.c_void_p), ('resid', ctypes.c_void_p), ('ldr', ctypes.c_int64),
                ('relu', ctypes.c_int32), ('out_f32', ctypes.c_int32),
                ('dtype', ctypes.c_int32), ('staging', ctypes.c_int32), ('tile_hint', ctypes.c_int32),
                ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_size_t), ('alpha', ctypes.c_float), ('beta', ctypes.c_float)]


class ConvDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('Cin', ctypes.c_int32), ('Cout', ctypes.c_int32), ('KH', ctypes.c_int32),
                ('KW', ctypes.c_int32), ('stride', ctypes.c_int32), ('pad', ctypes.c_int32),
                ('dil', ctypes.c_int32),
                ('bias', ctypes.c_void_p), ('resid', ctypes.c_void_p),
                ('relu', ctypes.c_int32), ('out_f32', ctypes.c_int32),
                ('dtype', ctypes.c_int32), ('staging', ctypes.c_int32), ('tile_hint', ctypes.c_int32),
                ('zero', ctypes.c_void_p), ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_size_t), ('alpha', ctypes.c_float),
                ('beta', ctypes.c_float)]


class TailDesc(ctypes.Structure):
    _fields_ = [('h', ctypes.c_void_p), ('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('B', ctypes.c_int32), ('OH', ctypes.c_int32), ('OW', ctypes.c_int32), ('C1', ctypes.c_int32),
                ('H2', ctypes.c_int32), ('W2', ctypes.c_int32), ('C2', ctypes.c_int32), ('stride2', ctypes.c_int32),
                ('Cout', ctypes.c_int32), ('bias', ctypes.c_void_p), ('relu', ctypes.c_int32), ('dtype', ctypes.c_int32)]


class TailNextDesc(ctypes.Structure):
    _fields_ = [('tail', TailDesc), ('resid', ctypes.c_void_p), ('wn', ctypes.c_void_p), ('bias_n', ctypes.c_void_p),
                ('hn', ctypes.c_void_p), ('Cn', ctypes.c_int32), ('alpha', ctypes.c_float), ('beta', ctypes.c_float)]


class RpnDesc(ctypes.Structure):
    _fields_ = [('cls', ctypes.c_void_p), ('reg', ctypes.c_void_p),
                ('T', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('A', ctypes.c_int32), ('anchor_stride', ctypes.c_int32),
                ('base_anchors', ctypes.c_void_p), ('means', ctypes.c_void_p), ('stds', ctypes.c_void_p),
                ('img_h', ctypes.c_float), ('img_w', ctypes.c_float), ('wh_ratio_clip', ctypes.c_float),
                ('nms_pre', ctypes.c_int32), ('nms_post', ctypes.c_int32), ('max_num', ctypes.c_int32),
                ('nms_thr', ctypes.c_float),
                ('proposals', ctypes.c_void_p), ('counts', ctypes.c_void_p),
                ('cls_pitch', ctypes.c_int32), ('reg_pitch', ctypes.c_int32)]


# every symbol include/hvr_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t
SYMBOLS = {
    'hvr_abi_version': (_i, []),
    'hvr_last_error': (ctypes.c_char_p, []),
    'hvr_gemm': (_i, [ctypes.POINTER(GemmDesc), _vp]),
    'hvr_gemm_fewrow_workspace_bytes': (_sz, [ctypes.POINTER(GemmDesc)]),
    'hvr_gemm_splitk_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'hvr_gemm_splitk': (_i, [ctypes.POINTER(GemmDesc), _vp, _sz, _vp]),
    'hvr_gemm_splitk_batched_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'hvr_gemm_splitk_batched': (_i, [ctypes.POINTER(GemmDesc), _i, _i64, _i64, _i64, _vp, _sz, _vp]),
    'hvr_conv2d_nhwc': (_i, [ctypes.POINTER(ConvDesc), _vp]),
    'hvr_conv2d_path': (_i, [ctypes.POINTER(ConvDesc)]),
    'hvr_conv2d_splitk_workspace_bytes': (_sz, [ctypes.POINTER(ConvDesc)]),
    'hvr_bottleneck_tail': (_i, [ctypes.POINTER(TailDesc), _vp]),
    'hvr_bottleneck_tail_supported': (_i, [ctypes.POINTER(TailDesc)]),
    'hvr_bottleneck_tail_next': (_i, [ctypes.POINTER(TailNextDesc), _vp]),
    'hvr_bottleneck_tail_next_supported': (_i, [ctypes.POINTER(TailNextDesc)]),
    'hvr_im2col_stem': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'hvr_maxpool3x3s2_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'hvr_stem_fused': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'hvr_stem_fused_dtype': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'hvr_relation_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'hvr_relation_fwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    'hvr_relation_grouped_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'hvr_relation_fwd_grouped': (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    'hvr_relation_probs_workspace_bytes': (_sz, [_i, _i]),
    'hvr_relation_probs': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    'hvr_relation_dscore': (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i, _i64, _i, _f, _i, _vp]),
    'hvr_relu_bwd': (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    'hvr_im2col_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'hvr_im2col_t': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'hvr_relu_bwd_t': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    'hvr_scale_rows': (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp]),
    'hvr_sgd_workspace_bytes': (_sz, []),
    'hvr_sgd_step': (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp, _sz, _i, _vp]),
    'hvr_colsum_workspace_bytes': (_sz, [_i, _i]),
    'hvr_colsum': (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _sz, _vp]),
    'hvr_pack_conv_weight': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'hvr_pack_conv_weights_multi': (_i, [_vp, _i, _i64, _i, _vp]),
    'hvr_transpose_multi': (_i, [_vp, _i, _i, _vp]),
    'hvr_unpack_conv_wgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'hvr_unpack_conv_wgrads_multi': (_i, [_vp, _i, _i64, _i, _vp]),
    'hvr_det_loss': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _vp]),
    'hvr_det_loss_sampled': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _vp]),
    'hvr_max_iou_assign_workspace_bytes': (_sz, [_i, _i]),
    'hvr_max_iou_assign': (_i, [_vp, _i, _i, _vp, _i, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    'hvr_sample_pos_neg': (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    'hvr_box_targets': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    'hvr_rpn_loss': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    'hvr_ce_rows': (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    'hvr_triplet_margin': (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    'hvr_ingest_frame': (_i, [_vp, _i, _i, _i64, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'hvr_mining_argreduce': (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _vp]),
    'hvr_roi_align_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    'hvr_roi_align_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'hvr_nms_workspace_bytes': (_sz, [_i]),
    'hvr_nms_first': (_i, [_vp, _i, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'hvr_nms': (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    'hvr_rpn_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'hvr_rpn_proposals': (_i, [ctypes.POINTER(RpnDesc), _vp, _sz, _vp]),
    'hvr_rpn_wide_frames': (_i, [_i]),
    'hvr_det_decode': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp]),
    'hvr_multiclass_nms_workspace_bytes': (_sz, [_i, _i]),
    'hvr_multiclass_nms': (_i, [_vp, _vp, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'hvr_cast': (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    'hvr_cast_scaled': (_i, [_vp, _vp, _i64, _i, _i, _f, _vp]),
    'hvr_permute_nchw_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'hvr_transpose_pad': (_i, [_vp, _vp, _i, _i, _i64, _i64, _i, _vp]),
}

_lib = None


def build(force=False):
    """Compile the HIP sources for gfx950 into ``hvrnet_amd/libhvr_hip.so`` (in-tree)."""
    script = os.path.join(_HERE, 'csrc', 'build.sh')
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
        for f in os.listdir(os.path.join(_HERE, 'csrc', 'build')) if os.path.isdir(os.path.join(_HERE, 'csrc', 'build')) else []:
            os.remove(os.path.join(_HERE, 'csrc', 'build', f))
    subprocess.run(['bash', script], check=True)
    return LIB_PATH


def lib():
    """The loaded library; raises HvrError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HvrError('libhvr_hip.so is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no CPU or PyTorch fallback for the HVR hot path)')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.hvr_abi_version() != ABI_VERSION:
            raise HvrError('libhvr_hip.so has ABI version %d, this binding needs %d: rebuild it (hvrnet_amd/csrc/build.sh)'
                           % (handle.hvr_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HvrError('%s failed (%d): %s' % (what, rc, lib().hvr_last_error().decode()))


def _raw_stream(device=None):
    """hipStream_t of torch's current stream as an int.  torch.cuda.current_stream() builds a Stream object (and re-checks
    availability) on every call, ~3-9 us; a training iteration asks ~1 700 times, so go to the raw getter."""
    idx = torch._C._cuda_getDevice() if device is None or device.index is None else device.index
    return torch._C._cuda_getCurrentRawStream(idx)


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _dt(t):
    if t.dtype == torch.float32:
        return HVR_F32
    if t.dtype == torch.bfloat16:
        return HVR_BF16
    if t.dtype == torch.float16:
        return HVR_F16
    if t.dtype == SPLIT:
        return HVR_F16S
    raise HvrError('unsupported tensor dtype %s (float32 / bfloat16 / float16 / the split-half container int32)' % t.dtype)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise NotImplementedError('hvr_hip ops run on the GPU only; got a %s tensor' % t.device)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---- optional per-call timing with HIP events on the launch stream (used by bench.py) ----
_prof = None


class _Span(object):
    __slots__ = ('tag', 'work', 'start')

    def __init__(self, tag, work):
        self.tag, self.work = tag, work

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        end = torch.cuda.Event(enable_timing=True)
        end.record(torch.cuda.current_stream())
        _prof['spans'].append((self.tag, self.work, self.start, end))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()


def _span(tag, work=0.0):
    if _prof is None or (tag.split(' ')[0] not in _prof['tags'] and '*' not in _prof['tags']):
        return _NOSPAN
    return _Span(tag, work)


def profile_begin(tags=('*',), detail=False):
    """Start recording (tag, algorithmic work, HIP-event pair) for the C-ABI calls whose tag is in `tags`."""
    global _prof
    _prof = dict(tags=set(tags), spans=[], detail=detail)


def profile_end(raw=False):
    """-> {tag: dict(calls, ms, work)}; synchronises once.  raw: the calls in issue order, [(tag, work, ms)]."""
    global _prof
    spans, _prof = _prof['spans'], None
    torch.cuda.synchronize()
    if raw:
        return [(tag, work, s.elapsed_time(e)) for tag, work, s, e in spans]
    out = {}
    for tag, work, s, e in spans:
        d = out.setdefault(tag, dict(calls=0, ms=0.0, work=0.0))
        d['calls'] += 1
        d['ms'] += s.elapsed_time(e)
        d['work'] += work
    return out


_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()  # once per device: every stream reads this page afterwards
        _zero_pages[device] = z
    return z


def kstep(dtype):
    return 32 if dtype == torch.float32 else 64


# A split-half value below 2^-3 keeps an absolute, not a relative, error bound (its lo half is then a half subnormal), so this
# layer keeps split tensors SCALED by powers of two: weights x 2^6 (they are ~1e-2), activations x 2^4 (full precision from
# 2^-7 up, range 4094).  cast() applies / removes the activation scale, as_operand() the weight scale, and gemm() / conv2d_nhwc()
# hand the kernels the factors that keep every output in its convention: alpha on the accumulators, beta on the bias.  Every B
# operand those two are given here is a weight matrix made by as_operand().  Nothing outside this file knows about the scales.
SPLIT_WEIGHT_SCALE = 64.0
SPLIT_ACT_SCALE = 16.0


def _split_factors(out_is_f32, alpha):
    """(alpha, beta) of a split-half product A(x ACT) . W(x WEIGHT)^T whose output is a scaled split tensor or true-valued f32."""
    if alpha is not None:
        return float(alpha), 1.0
    if out_is_f32:
        return 1.0 / (SPLIT_WEIGHT_SCALE * SPLIT_ACT_SCALE), 1.0
    return 1.0 / SPLIT_WEIGHT_SCALE, SPLIT_ACT_SCALE


def as_operand(t, dtype):
    """An f32 WEIGHT tensor (pack time) in the operand format `dtype`, contiguous (split half: x SPLIT_WEIGHT_SCALE)."""
    t = t.contiguous()
    if dtype == SPLIT:
        return cast(t.float(), SPLIT, scale=SPLIT_WEIGHT_SCALE)
    return t.to(dtype)


# ----------------------------------------------------------------------------------------
def gemm(a, w, bias=None, resid=None, relu=False, out_f32=False, out=None, staging=None, tile=None, alpha=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias + resid).  a / w / resid share one dtype.  alpha (split half only): the factor on
    the accumulators (bias then unscaled); default: the factors of this layer's scaled split tensors (_split_factors)."""
    _need_cuda(a, w, bias, resid)
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], (a.shape, w.shape)
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    d = GemmDesc(A=a.data_ptr(), B=w.data_ptr(), C=out.data_ptr(), M=M, N=N, K=K,
                 lda=a.stride(0), ldb=w.stride(0), ldc=out.stride(0),
                 bias=bias.data_ptr() if bias is not None else None,
                 resid=resid.data_ptr() if resid is not None else None,
                 ldr=resid.stride(0) if resid is not None else 0,
                 relu=int(relu), out_f32=int(out.dtype == torch.float32 and a.dtype != torch.float32),
                 dtype=_dt(a), staging=STAGING if staging is None else staging,
                 tile_hint=TILE_HINT if tile is None else tile)
    if a.dtype == SPLIT:
        if resid is not None and out.dtype == torch.float32:
            # a split residual is stored x SPLIT_ACT_SCALE and the epilogue adds it as it is: right for a scaled split output, 16 x
            # too much beside a true-valued f32 one
            raise HvrError('split-half gemm: a residual cannot be combined with an f32 output (the residual carries the activation scale)')
        d.alpha, d.beta = _split_factors(out.dtype == torch.float32, alpha)
    nbytes = lib().hvr_gemm_fewrow_workspace_bytes(ctypes.byref(d)) if _fewrow[0] else 0   # few rows, long K: K slices + one reduce launch
    if nbytes:
        ws = _workspace(nbytes, a.device, 'gemm_fewrow')
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    with _span('gemm' if not (_prof and _prof['detail']) else 'gemm M%d N%d K%d' % (M, N, K), 2.0 * M * N * K):
        _check(lib().hvr_gemm(ctypes.byref(d), _stream()), 'hvr_gemm')
    return out


def gemm_splitk(a, w, staging=None, tile=None, out=None):
    """f32 out[M,N] = a[M,K] @ w[N,K]^T for few-tile / long-K products (weight gradients): K slices in one launch + a reduce.
    out: a contiguous f32 [M, N] tensor to write into (e.g. a parameter's gradient buffer) instead of a fresh one."""
    _need_cuda(a, w)
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.dtype == w.dtype, (a.shape, w.shape)
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        assert out.dtype == torch.float32 and tuple(out.shape) == (M, N) and out.is_contiguous() and out.device == a.device
    d = GemmDesc(A=a.data_ptr(), B=w.data_ptr(), C=out.data_ptr(), M=M, N=N, K=K, lda=a.stride(0), ldb=w.stride(0), ldc=N,
                 bias=None, resid=None, ldr=0, relu=0, out_f32=int(a.dtype != torch.float32), dtype=_dt(a),
                 staging=STAGING if staging is None else staging, tile_hint=TILE_HINT if tile is None else tile)
    nbytes = lib().hvr_gemm_splitk_workspace_bytes(M, N, K, _dt(a))
    ws = _workspace(nbytes, a.device, 'splitk') if nbytes else None
    with _span('gemm' if not (_prof and _prof['detail']) else 'gemm M%d N%d K%d splitk' % (M, N, K), 2.0 * M * N * K):
        _check(lib().hvr_gemm_splitk(ctypes.byref(d), _ptr(ws), nbytes, _stream()), 'hvr_gemm_splitk')
    return out


def gemm_splitk_batched(a, w, out=None):
    """f32 out[g] = a[g] @ w[g]^T for g < G: a [G, M, K], w [G, N, K] (leading slices of contiguous slabs), one launch (+ one reduce
    when K is cut into slices) for the G weight gradients of a stage's identical blocks.  -> out [G, M, N] f32."""
    _need_cuda(a, w)
    assert a.dim() == 3 and w.dim() == 3 and a.shape[0] == w.shape[0] and a.shape[2] == w.shape[2] and a.dtype == w.dtype
    assert a.stride(2) == 1 and w.stride(2) == 1 and a.stride(1) == a.shape[2] and w.stride(1) == w.shape[2]
    G, M, K = a.shape
    N = w.shape[1]
    if out is None:
        out = torch.empty((G, M, N), dtype=torch.float32, device=a.device)
    d = GemmDesc(A=a.data_ptr(), B=w.data_ptr(), C=out.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=None, resid=None, ldr=0, relu=0,
                 out_f32=int(a.dtype != torch.float32), dtype=_dt(a), staging=STAGING, tile_hint=TILE_HINT)
    nbytes = lib().hvr_gemm_splitk_batched_workspace_bytes(M, N, K, _dt(a), G)
    ws = _workspace(nbytes, a.device, 'splitk_batched') if nbytes else None
    with _span('gemm' if not (_prof and _prof['detail']) else 'gemm %dx M%d N%d K%d splitk batched' % (G, M, N, K), 2.0 * G * M * N * K):
        _check(lib().hvr_gemm_splitk_batched(ctypes.byref(d), G, a.stride(0), w.stride(0), M * N, _ptr(ws), nbytes, _stream()), 'hvr_gemm_splitk_batched')
    return out


def conv2d_nhwc(x, w, bias=None, resid=None, relu=False, stride=1, pad=0, dil=1, out_f32=False, staging=None, tile=None, out=None, alpha=None):
    """x [B,H,W,Cin] (physical NHWC), w [Cout,KH,KW,Cin] -> [B,OH,OW,Cout]; out: a contiguous tensor of that shape and the
    output dtype to write into (the caller-allocates contract of the C ABI) instead of a fresh one."""
    _need_cuda(x, w, bias, resid)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    odt = torch.float32 if out_f32 else x.dtype
    if out is not None:
        assert tuple(out.shape) == (B, OH, OW, Cout) and out.dtype == odt and out.is_contiguous() and out.device == x.device, \
            (tuple(out.shape), (B, OH, OW, Cout), out.dtype)
        y = out
    else:
        y = torch.empty((B, OH, OW, Cout), dtype=odt, device=x.device)
    d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, KH=KH, KW=KW,
                 stride=stride, pad=pad, dil=dil,
                 bias=bias.data_ptr() if bias is not None else None,
                 resid=resid.data_ptr() if resid is not None else None,
                 relu=int(relu), out_f32=int(out_f32 and x.dtype != torch.float32), dtype=_dt(x),
                 staging=STAGING if staging is None else staging, tile_hint=TILE_HINT if tile is None else tile,
                 zero=zero_page(x.device).data_ptr())
    if x.dtype == SPLIT:
        if resid is not None and out_f32:
            raise HvrError('split-half conv: a residual cannot be combined with an f32 output (the residual carries the activation scale)')
        d.alpha, d.beta = _split_factors(bool(out_f32), alpha)
    # few-row problems (one frame through the stride-16 stages): the library cuts the K loop into slices when it is handed
    # scratch for the f32 partial tiles (per stream, like every other workspace here)
    nbytes = lib().hvr_conv2d_splitk_workspace_bytes(ctypes.byref(d)) if _fewrow[0] else 0
    if nbytes:
        ws = _workspace(nbytes, x.device, 'conv_splitk')
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    tag, work = 'conv', 2.0 * B * OH * OW * Cout * KH * KW * Cin
    if _prof is not None and ('*' in _prof['tags'] or 'conv' in _prof['tags'] or 'conv_expand' in _prof['tags']):
        if lib().hvr_conv2d_path(ctypes.byref(d)) == 1:
            # the HBM-bound expand + residual convs (expand.hip) are accounted in bytes: X, residual and output once, W once
            tag, work = 'conv_expand', float((B * OH * OW * (Cin + (2 if resid is not None else 1) * Cout) + Cout * Cin) * x.element_size())
        if _prof['detail']:
            tag = '%s %dx%d %d->%d k%d s%d d%d%s' % (tag, H, W, Cin, Cout, KH, stride, dil, '+res' if resid is not None else '')
    with _span(tag, work):
        _check(lib().hvr_conv2d_nhwc(ctypes.byref(d), _stream()), 'hvr_conv2d_nhwc')
    return y


def _tail_desc(h, x, w, bias, stride2, relu, y):
    B, OH, OW, C1 = h.shape
    _, H2, W2, C2 = x.shape
    return TailDesc(h=h.data_ptr(), x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr() if y is not None else 1 << 20, B=B, OH=OH, OW=OW, C1=C1,
                    H2=H2, W2=W2, C2=C2, stride2=int(stride2), Cout=w.shape[0], bias=bias.data_ptr(), relu=int(relu), dtype=_dt(h))


def bottleneck_tail_supported(h, x, w, bias, stride2):
    """True when hvr_bottleneck_tail has a fused kernel for these shapes (bf16, C1 + C2 in {128, 384}, ...)."""
    if not (h.is_cuda and h.dtype in (torch.bfloat16, torch.float16) and x.dtype == h.dtype and h.is_contiguous() and x.is_contiguous()):
        return False
    return bool(lib().hvr_bottleneck_tail_supported(ctypes.byref(_tail_desc(h, x, w, bias, stride2, True, None))))


def bottleneck_tail(h, x, w, bias, stride2=1, relu=True, out=None):
    """relu(h W3^T + x_s Wd^T + bias): the closing 1x1 of a Bottleneck with its projection shortcut as a second K segment.
    h [B,OH,OW,C1], x [B,H2,W2,C2] physical NHWC, w [Cout, C1 + C2], bias f32 [Cout] -> [B,OH,OW,Cout]."""
    _need_cuda(h, x, w, bias)
    B, OH, OW, C1 = h.shape
    Cout = w.shape[0]
    if out is not None:
        assert tuple(out.shape) == (B, OH, OW, Cout) and out.dtype == h.dtype and out.is_contiguous()
        y = out
    else:
        y = torch.empty((B, OH, OW, Cout), dtype=h.dtype, device=h.device)
    d = _tail_desc(h, x, w, bias, stride2, relu, y)
    tag, work = 'conv_expand', float((h.numel() + B * OH * OW * x.shape[3] + y.numel() + w.numel()) * h.element_size())
    if _prof is not None and _prof['detail']:
        tag = 'conv_expand tail %dx%d %d+%d->%d s%d' % (OH, OW, C1, x.shape[3], Cout, stride2)
    with _span(tag, work):
        _check(lib().hvr_bottleneck_tail(ctypes.byref(d), _stream()), 'hvr_bottleneck_tail')
    return y


def _tail_next_desc(h, x, resid, w, bias, stride2, wn, bias_n, y, hn):
    B, OH, OW, C1 = h.shape
    ph = 1 << 20  # placeholder address for the support query
    if x is not None:
        t = _tail_desc(h, x, w, bias, stride2, True, y)
    else:
        t = TailDesc(h=h.data_ptr(), x=None, w=w.data_ptr(), y=y.data_ptr() if y is not None else ph, B=B, OH=OH, OW=OW, C1=C1,
                     H2=OH, W2=OW, C2=0, stride2=1, Cout=w.shape[0], bias=bias.data_ptr(), relu=1, dtype=_dt(h))
    d = TailNextDesc(tail=t, resid=resid.data_ptr() if resid is not None else None, wn=wn.data_ptr(), bias_n=bias_n.data_ptr(),
                     hn=hn.data_ptr() if hn is not None else ph, Cn=wn.shape[0])
    if h.dtype == SPLIT:   # both products: scaled split activations x scaled split weights -> a scaled split activation
        d.alpha, d.beta = _split_factors(False, None)
    return d


def bottleneck_tail_next_supported(h, x, resid, w, bias, stride2, wn, bias_n):
    """True when hvr_bottleneck_tail_next runs these shapes: (Cout, Cn) = (256, 64) / (512, 128), bf16 / half, contiguous maps;
    split half: the identity form (resid given, x None) with (Cout, Cn) = (256, 64)."""
    ts = [t for t in (h, x, resid) if t is not None]
    if not all(t.is_cuda and t.dtype in (torch.bfloat16, torch.float16, SPLIT) and t.dtype == h.dtype and t.is_contiguous() for t in ts):
        return False
    if h.dtype == SPLIT and (x is not None or wn.dtype != SPLIT or w.dtype != SPLIT):
        return False
    if (x is None) == (resid is None) or wn.dim() != 2 or wn.shape[1] != w.shape[0] or not wn.is_contiguous():
        return False
    return bool(lib().hvr_bottleneck_tail_next_supported(ctypes.byref(_tail_next_desc(h, x, resid, w, bias, stride2, wn, bias_n, None, None))))


def bottleneck_tail_next(h, x, resid, w, bias, wn, bias_n, stride2=1, out=None):
    """y = relu(h W3^T [+ x_s Wd^T] + bias [+ resid]) and hn = relu(y wn^T + bias_n), the next block's conv1, in one pass.
    Exactly one of x (projection block: its input map) and resid (identity block) is given.  -> (y, hn)"""
    _need_cuda(h, w, bias, wn, bias_n)
    B, OH, OW, C1 = h.shape
    Cout, Cn = w.shape[0], wn.shape[0]
    if out is not None:
        assert tuple(out.shape) == (B, OH, OW, Cout) and out.dtype == h.dtype and out.is_contiguous()
        y = out
    else:
        y = torch.empty((B, OH, OW, Cout), dtype=h.dtype, device=h.device)
    hn = torch.empty((B, OH, OW, Cn), dtype=h.dtype, device=h.device)
    d = _tail_next_desc(h, x, resid, w, bias, stride2, wn, bias_n, y, hn)
    side = x if x is not None else resid
    tag = 'conv_expand'
    work = float((h.numel() + B * OH * OW * side.shape[3] + y.numel() + hn.numel() + w.numel() + wn.numel()) * h.element_size())
    if _prof is not None and _prof['detail']:
        tag = 'conv_expand tail+next %dx%d %d+%d->%d->%d' % (OH, OW, C1, x.shape[3] if x is not None else 0, Cout, Cn)
    with _span(tag, work):
        _check(lib().hvr_bottleneck_tail_next(ctypes.byref(d), _stream()), 'hvr_bottleneck_tail_next')
    return y, hn


def conv2d_path(B, H, W, Cin, Cout, k=1, stride=1, pad=0, dil=1, dtype=torch.bfloat16, resid=True, bias=True, out_f32=False, tile=0):
    """Which kernel hvr_conv2d_nhwc would run for a conv of this shape (0 tile engine, 1 expand.hip panel kernel, < 0 rejected).
    Nothing is launched and no memory is touched: the descriptor carries placeholder (16-byte aligned) addresses."""
    fake = 1 << 20
    d = ConvDesc(x=fake, w=fake, y=fake, B=B, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=stride, pad=pad, dil=dil,
                 bias=fake if bias else None, resid=fake if resid else None, relu=1, out_f32=int(out_f32),
                 dtype={torch.bfloat16: HVR_BF16, torch.float16: HVR_F16, SPLIT: HVR_F16S}.get(dtype, HVR_F32), staging=STAGING,
                 tile_hint=tile, zero=fake)
    return int(lib().hvr_conv2d_path(ctypes.byref(d)))


def im2col_stem(img, dtype, kp=192):
    """img [B,3,H,W] f32 NCHW -> ([B*OH*OW, kp] patches, OH, OW)."""
    _need_cuda(img)
    assert img.dtype == torch.float32 and img.is_contiguous()
    B, _, H, W = img.shape
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    cols = torch.empty((B * OH * OW, kp), dtype=torch.float32 if dtype == SPLIT else dtype, device=img.device)
    _check(lib().hvr_im2col_stem(_ptr(img), _ptr(cols), B, H, W, kp, _dt(cols), _stream()), 'hvr_im2col_stem')
    return (cast(cols, SPLIT) if dtype == SPLIT else cols), OH, OW   # (split half: through cast(), which applies the activation scale)


def stem_fused(img, wpk, bias):
    """img [B,3,H,W] f32 NCHW -> conv7x7/2 + bias + ReLU + maxpool3x3/2 as physical NHWC bf16 [B,PH,PW,64]."""
    _need_cuda(img, wpk, bias)
    assert img.dtype == torch.float32 and img.is_contiguous() and wpk.dtype in (torch.bfloat16, torch.float16)
    split = wpk.dim() == 4   # [2, 64, 7, 32] half: the hi / lo planes of split-half weights (stem_split_weights)
    assert tuple(wpk.shape) == ((2, 64, 7, 32) if split else (64, 7, 32)) and (not split or wpk.dtype == torch.float16) and wpk.is_contiguous()
    B, _, H, W = img.shape
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    out = torch.empty((B, PH, PW, 64), dtype=SPLIT if split else wpk.dtype, device=img.device)
    with _span('stem', 2.0 * B * CH * CW * 64 * 147):
        _check(lib().hvr_stem_fused_dtype(_ptr(img), _ptr(wpk), _ptr(bias), _ptr(out), B, H, W, HVR_F16S if split else _dt(wpk), _stream()),
               'hvr_stem_fused')
    return out


def stem_split_weights(wf):
    """f32 fused-stem weights [64, 7, 32] -> [2, 64, 7, 32] half planes (hi, lo) of the weights x 2^6 x the activation scale: the kernel
    takes 2^6 back, so with the bias x SPLIT_ACT_SCALE (stem_split_bias) its output is a scaled split activation."""
    wf = wf * (SPLIT_WEIGHT_SCALE * SPLIT_ACT_SCALE)
    hi = wf.half()
    lo = (wf - hi.float()).half()
    return torch.stack([hi, lo], 0).contiguous()


def stem_split_bias(b):
    return (b * SPLIT_ACT_SCALE).contiguous()


def maxpool3x3s2_nhwc(x):
    _need_cuda(x)
    if x.dtype == SPLIT:   # pooled in f32 (max does not act per half plane)
        return cast(maxpool3x3s2_nhwc(cast(x, torch.float32)), SPLIT)
    B, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
    _check(lib().hvr_maxpool3x3s2_nhwc(_ptr(x), _ptr(y), B, H, W, C, _dt(x), _stream()), 'hvr_maxpool3x3s2_nhwc')
    return y


_ws_cache = {}


# Few-row split-K (hvr_gemm_fewrow_workspace_bytes / hvr_conv2d_splitk_workspace_bytes): opt-in, because it changes the f32
# summation order with the batch size -- a frame's rows computed alone would no longer equal the same rows computed in a
# batch bit for bit, a property the cached / look-ahead loops are tested for.  graphs.GraphedStream turns it on for its
# one-frame graphs (the stream loop's latency case); everything else runs the unsplit kernels.
_fewrow = [False]


@contextlib.contextmanager
def fewrow_split(on=True):
    prev = _fewrow[0]
    _fewrow[0] = bool(on)
    try:
        yield
    finally:
        _fewrow[0] = prev


BIG_TILE_HINT = 16   # gemm_params.h: kBigHint
TWO_LEVEL_HINT = 18  # gemm_params.h: kTwoLevelHint (conv2d_nhwc: two-level accumulation for f32 / split-half operands, else as hint 0)


@contextlib.contextmanager
def throughput_mode(on=True):
    """Launches issued inside prefer CU-time over latency: convs / linear layers whose 288 x 256 tile grid covers only half the
    chip (layer 3: 125 workgroups) still take the big-tile kernel (bigtile.hip) -- for callers that keep several windows in
    flight (bench.py's headline region captures its window graphs inside this).  Same results bit for bit."""
    global TILE_HINT
    prev = TILE_HINT
    if on and TILE_HINT == 0:
        TILE_HINT = BIG_TILE_HINT
    try:
        yield
    finally:
        TILE_HINT = prev


_hip_rt = [None]
_masked_streams = {}   # (device index, first, n) -> (raw handle, ExternalStream): ONE queue per mask, kept for the life of the process


def cu_masked_stream(device, first_cu, n_cus, total_cus=256):
    """A HIP stream whose launches -- direct ones and hipGraph replays issued ON it -- only use `n_cus` of the chip's CUs
    (hipExtStreamCreateWithCUMask, mask bits [first_cu, first_cu + n_cus)).  On MI355X a contiguous range of mask bits is spread
    over all eight XCDs (96 bits = 12 CUs in each: measured in round 3 with a per-CU occupancy probe, profiles/r03_stream_bench.json), and the mask belongs to the stream a graph is replayed on,
    not to the one it was captured on.  Every mask is a hardware queue of its own: a process that creates a dozen of them slows
    every one down (queue oversubscription), so streams are cached per mask.  -> torch.cuda.ExternalStream"""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(first_cu), int(n_cus))
    if key in _masked_streams:
        return _masked_streams[key][1]
    if _hip_rt[0] is None:
        rt = ctypes.CDLL('libamdhip64.so')
        rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        rt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _hip_rt[0] = rt
    if not (0 <= first_cu and n_cus > 0 and first_cu + n_cus <= total_cus):
        raise HvrError('CU mask [%d, %d) outside [0, %d)' % (first_cu, first_cu + n_cus, total_cus))
    words = (total_cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(first_cu, first_cu + n_cus):
        mask[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(key[0]):
        rc = _hip_rt[0].hipExtStreamCreateWithCUMask(ctypes.byref(handle), words, mask)
    if rc != 0:
        raise HvrError('hipExtStreamCreateWithCUMask failed (%d)' % rc)
    st = torch.cuda.ExternalStream(handle.value, device=torch.device('cuda', key[0]))
    _masked_streams[key] = (handle, st)
    return st


_ws_captured = set()   # keys whose current buffer was handed out during a stream capture
_ws_retired = []       # replaced buffers a captured graph may still address: never freed


def _workspace(nbytes, device, tag):
    # one buffer per (use, stream): windows enqueued on different HIP streams run concurrently and must not share scratch
    key = (tag, device.index if isinstance(device, torch.device) else str(device), _raw_stream(device if isinstance(device, torch.device) else None))
    ws = _ws_cache.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ws is None or ws.numel() < nbytes:
        if capturing:
            raise HvrError('scratch buffer %r must grow (%d -> %d bytes) inside a stream capture: warm the largest shapes up first'
                           % (tag, 0 if ws is None else ws.numel(), nbytes))
        if ws is not None and key in _ws_captured:
            _ws_retired.append(ws)   # a captured graph holds this address (hvrnet_amd/graphs.py): keep it alive
            _ws_captured.discard(key)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    if capturing:
        _ws_captured.add(key)
    return ws


def relation_fwd(q, k, v, scale, staging=None):
    """softmax(scale * q @ k^T, dim=1) @ v without materialising the f32 logits."""
    _need_cuda(q, k, v)
    Mq, D = q.shape
    Mk = k.shape[0]
    assert k.shape[1] == D and v.shape == (Mk, D) and q.dtype == k.dtype == v.dtype
    o = torch.empty((Mq, D), dtype=q.dtype, device=q.device)
    nbytes = lib().hvr_relation_workspace_bytes(Mq, Mk, D, _dt(q))
    ws = _workspace(nbytes, q.device, 'relation')
    if q.dtype == SPLIT:   # q, k arrive x SPLIT_ACT_SCALE each; v's scale carries over to the output
        scale = float(scale) / (SPLIT_ACT_SCALE * SPLIT_ACT_SCALE)
    with _span('relation_full' if Mq == Mk else 'relation_key', 4.0 * Mq * Mk * D):
        _check(lib().hvr_relation_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o), o.stride(0),
                                      Mq, Mk, D, float(scale), _dt(q), STAGING if staging is None else staging,
                                      _ptr(ws), ws.numel(), _stream()), 'hvr_relation_fwd')
    return o


def relation_fwd_grouped(q, k, v, scale, groups, staging=None, exact=False):
    """`groups` independent relation problems of one shape in one call (the clips a batched head has in flight): q [G * Mq, D],
    k / v [G * Mk, D] hold the groups' rows back to back (row-strided views are fine: group g starts g * rows * stride(0) elements
    behind group 0) -> o [G * Mq, D], group g's rows = relation_fwd of group g's q / k / v (hvr_relation_fwd_grouped): up to the
    association of the f32 sums by default, bit for bit with exact=True."""
    _need_cuda(q, k, v)
    G = int(groups)
    assert G >= 1 and q.shape[0] % G == 0 and k.shape[0] % G == 0
    Mq, Mk, D = q.shape[0] // G, k.shape[0] // G, q.shape[1]
    assert k.shape[1] == D and v.shape == k.shape and q.dtype == k.dtype == v.dtype
    if G == 1:
        return relation_fwd(q, k, v, scale, staging)
    o = torch.empty((G * Mq, D), dtype=q.dtype, device=q.device)
    nbytes = lib().hvr_relation_grouped_workspace_bytes(G, Mq, Mk, D, _dt(q))
    ws = _workspace(nbytes, q.device, 'relation_grouped')
    if q.dtype == SPLIT:
        scale = float(scale) / (SPLIT_ACT_SCALE * SPLIT_ACT_SCALE)
    with _span('relation_full' if Mq == Mk else 'relation_key', 4.0 * G * Mq * Mk * D):
        _check(lib().hvr_relation_fwd_grouped(_ptr(q), q.stride(0), Mq * q.stride(0), _ptr(k), k.stride(0), Mk * k.stride(0),
                                              _ptr(v), v.stride(0), Mk * v.stride(0), _ptr(o), o.stride(0), Mq * o.stride(0), G,
                                              Mq, Mk, D, float(scale), _dt(q), STAGING if staging is None else staging, int(bool(exact)),
                                              _ptr(ws), ws.numel(), _stream()), 'hvr_relation_fwd_grouped')
    return o


def relation_ldp(Mk):
    """Row length of the relation's probability matrix: keys padded to a multiple of 128."""
    return (Mk + 127) // 128 * 128


def relation_probs(q, k, scale, staging=None):
    """softmax(scale * q @ k^T, dim=1) as a [Mq, ldp] matrix of q's dtype (padding columns zero)."""
    _need_cuda(q, k)
    Mq, D = q.shape
    Mk = k.shape[0]
    ldp = relation_ldp(Mk)
    P = torch.empty((Mq, ldp), dtype=q.dtype, device=q.device)
    ws = _workspace(lib().hvr_relation_probs_workspace_bytes(Mq, Mk), q.device, 'relation_probs')
    _check(lib().hvr_relation_probs(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(P), ldp, Mq, Mk, D, float(scale), _dt(q),
                                    STAGING if staging is None else staging, _ptr(ws), ws.numel(), _stream()), 'hvr_relation_probs')
    return P


def relation_dscore(P, dP, dO, O, scale):
    """scale * P * (dP - rowsum(dO * O)): gradient w.r.t. the un-scaled logits q @ k^T, same shape / dtype as P."""
    _need_cuda(P, dP, dO, O)
    assert P.shape == dP.shape and P.is_contiguous() and dP.is_contiguous() and dO.stride(1) == 1 and O.stride(1) == 1
    dS = torch.empty_like(P)
    _check(lib().hvr_relation_dscore(_ptr(P), _ptr(dP), _ptr(dO), dO.stride(0), _ptr(O), O.stride(0), _ptr(dS), P.shape[0],
                                     P.shape[1], dO.shape[1], float(scale), _dt(P), _stream()), 'hvr_relation_dscore')
    return dS


def im2col_nhwc(x, KH, KW, pad, dil):
    """x [B,H,W,Cin] -> patch matrix [B*OH*OW, KH*KW*Cin] (stride 1, zero padding)."""
    _need_cuda(x)
    x = x.contiguous()
    B, H, W, Cin = x.shape
    OH, OW = H + 2 * pad - dil * (KH - 1), W + 2 * pad - dil * (KW - 1)
    cols = torch.empty((B * OH * OW, KH * KW * Cin), dtype=x.dtype, device=x.device)
    _check(lib().hvr_im2col_nhwc(_ptr(x), _ptr(cols), B, H, W, Cin, KH, KW, pad, dil, _dt(x), _stream()), 'hvr_im2col_nhwc')
    return cols


def scale_rows(w, scale):
    """[R, ...] * scale[R] per leading row (f32 scale)."""
    _need_cuda(w, scale)
    w = w.contiguous()
    out = torch.empty_like(w)
    R = w.shape[0]
    _check(lib().hvr_scale_rows(_ptr(w), _ptr(scale.float().contiguous()), _ptr(out), R, w.numel() // R, _dt(w), _stream()), 'hvr_scale_rows')
    return out


def sgd_step(param_flat, grad_flat, momentum_buf, lr, momentum, weight_decay, grad_scale=1.0, max_norm=0.0, first_step=False):
    """In-place SGD-with-momentum update of a flat f32 parameter buffer (see hvr_sgd_step in include/hvr_hip.h)."""
    _need_cuda(param_flat, grad_flat, momentum_buf)
    assert param_flat.dtype == grad_flat.dtype == momentum_buf.dtype == torch.float32
    assert param_flat.is_contiguous() and grad_flat.is_contiguous() and momentum_buf.is_contiguous()
    assert param_flat.numel() == grad_flat.numel() == momentum_buf.numel()
    ws = _workspace(lib().hvr_sgd_workspace_bytes(), param_flat.device, 'sgd')
    _check(lib().hvr_sgd_step(_ptr(param_flat), _ptr(grad_flat), _ptr(momentum_buf), param_flat.numel(), float(lr), float(momentum),
                              float(weight_decay), float(grad_scale), float(max_norm), _ptr(ws), ws.numel(), int(first_step),
                              _stream()), 'hvr_sgd_step')


def relu_bwd(dy, y):
    """dy where y > 0 else 0 (y: the output of a GEMM epilogue with relu=True)."""
    _need_cuda(dy, y)
    dy, y = dy.contiguous(), y.contiguous()
    assert dy.shape == y.shape and dy.dtype == y.dtype and dy.numel() % 4 == 0
    dz = torch.empty_like(dy)
    _check(lib().hvr_relu_bwd(_ptr(dy), _ptr(y), _ptr(dz), dy.numel(), _dt(dy), _stream()), 'hvr_relu_bwd')
    return dz


def relu_bwd_t(dy, y, ldt, dzt_out=None):
    """(dz, dzt): dz = dy where y > 0 ([R, C]) and its transpose [C, ldt] (columns R.. zero) from one pass (bf16 / half).
    dzt_out: a contiguous [C, ldt] tensor to write the transpose into (a slot of a stage's weight-gradient slab)."""
    _need_cuda(dy, y)
    dy, y = dy.contiguous(), y.contiguous()
    assert dy.dim() == 2 and dy.shape == y.shape and dy.dtype == y.dtype
    R, C = dy.shape
    dz = torch.empty_like(dy)
    if dzt_out is None:
        dzt = torch.empty((C, ldt), dtype=dy.dtype, device=dy.device)
    else:
        assert tuple(dzt_out.shape) == (C, ldt) and dzt_out.dtype == dy.dtype and dzt_out.is_contiguous()
        dzt = dzt_out
    _check(lib().hvr_relu_bwd_t(_ptr(dy), _ptr(y), _ptr(dz), _ptr(dzt), ldt, R, C, _dt(dy), _stream()), 'hvr_relu_bwd_t')
    return dz, dzt


def im2col_t(x, KH, KW, pad, dil, ldt, out=None):
    """x [B,H,W,Cin] (NHWC, bf16 / half) -> the TRANSPOSED patch matrix [KH*KW*Cin, ldt] of a stride-1 conv (columns B*OH*OW.. zero)."""
    _need_cuda(x)
    x = x.contiguous()
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((KH * KW * Cin, ldt), dtype=x.dtype, device=x.device)
    else:
        assert tuple(out.shape) == (KH * KW * Cin, ldt) and out.dtype == x.dtype and out.is_contiguous()
    _check(lib().hvr_im2col_t(_ptr(x), _ptr(out), ldt, B, H, W, Cin, KH, KW, pad, dil, _dt(x), _stream()), 'hvr_im2col_t')
    return out


def colsum(dy, out=None):
    """[M, N] -> f32 [N] column sums (bias gradient); out: a contiguous f32 [N] tensor to write into."""
    _need_cuda(dy)
    assert dy.dim() == 2 and dy.stride(1) == 1
    if out is None:
        db = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
    else:
        assert out.dtype == torch.float32 and tuple(out.shape) == (dy.shape[1],) and out.is_contiguous() and out.device == dy.device
        db = out
    nbytes = lib().hvr_colsum_workspace_bytes(dy.shape[0], dy.shape[1])
    ws = _workspace(nbytes, dy.device, 'colsum') if nbytes else None
    _check(lib().hvr_colsum(_ptr(dy), _ptr(db), dy.shape[0], dy.shape[1], dy.stride(0), _dt(dy), _ptr(ws), nbytes, _stream()), 'hvr_colsum')
    return db


def pack_conv_weight(w, scale, dtype):
    """nn.Conv2d weight [Cout,Cin,KH,KW] f32 * scale[Cout] -> [Cout,KH,KW,Cin] in `dtype` (the conv kernel's operand)."""
    _need_cuda(w, scale)
    assert w.dtype == torch.float32 and w.is_contiguous() and scale.dtype == torch.float32 and scale.is_contiguous()
    Cout, Cin, KH, KW = w.shape
    out = torch.empty((Cout, KH, KW, Cin), dtype=dtype, device=w.device)
    _check(lib().hvr_pack_conv_weight(_ptr(w), _ptr(scale), _ptr(out), Cout, Cin, KH, KW, _dt(out), _stream()), 'hvr_pack_conv_weight')
    return out


class PackItem(ctypes.Structure):          # hvr_pack_item
    _fields_ = [('w', ctypes.c_void_p), ('scale', ctypes.c_void_p), ('out', ctypes.c_void_p), ('first', ctypes.c_int64),
                ('Cout', ctypes.c_int32), ('Cin', ctypes.c_int32), ('KK', ctypes.c_int32), ('pad_', ctypes.c_int32)]


class TransposeItem(ctypes.Structure):     # hvr_transpose_item
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('lds', ctypes.c_int64), ('ldd', ctypes.c_int64),
                ('R', ctypes.c_int32), ('C', ctypes.c_int32), ('first_tile', ctypes.c_int32), ('tiles_c', ctypes.c_int32),
                ('dcols', ctypes.c_int32), ('pad_', ctypes.c_int32)]


def items_to_device(items, device):
    """A list of ctypes structures -> a uint8 device tensor holding the array (the descriptor tables of the *_multi entry points)."""
    arr = (type(items[0]) * len(items))(*items)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def pack_conv_weights_multi(items_dev, n, total, dtype):
    code = {torch.bfloat16: HVR_BF16, torch.float16: HVR_F16}[dtype]
    _check(lib().hvr_pack_conv_weights_multi(_ptr(items_dev), n, total, code, _stream()), 'hvr_pack_conv_weights_multi')


def transpose_multi(items_dev, n, tiles):
    _check(lib().hvr_transpose_multi(_ptr(items_dev), n, tiles, _stream()), 'hvr_transpose_multi')


def unpack_conv_wgrads_multi(items_dev, n, total, accumulate=True):
    """hvr_unpack_conv_wgrad for a device table of layers (PackItem: w = the parameter-layout gradient to add to, scale, out = the f32
    product) in one launch."""
    _check(lib().hvr_unpack_conv_wgrads_multi(_ptr(items_dev), n, total, int(accumulate), _stream()), 'hvr_unpack_conv_wgrads_multi')


def unpack_conv_wgrad(dw, scale, shape, accumulate_into=None):
    """dW_eff [Cout, KH*KW*Cin] f32 * scale[Cout] -> [Cout,Cin,KH,KW] f32 (the nn.Conv2d parameter's layout); with
    `accumulate_into` (a contiguous f32 tensor of that shape, e.g. the parameter's .grad) the result is added to it in place."""
    _need_cuda(dw, scale)
    Cout, Cin, KH, KW = shape
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == Cout * Cin * KH * KW
    if accumulate_into is not None:
        out = accumulate_into
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(shape)
    else:
        out = torch.empty(shape, dtype=torch.float32, device=dw.device)
    _check(lib().hvr_unpack_conv_wgrad(_ptr(dw), _ptr(scale), _ptr(out), Cout, Cin, KH, KW, int(accumulate_into is not None), _stream()),
           'hvr_unpack_conv_wgrad')
    return out


def det_loss(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, beta=1.0, w_cls=1.0, w_bbox=1.0):
    """BBoxHead.loss on a fused [R, ld] f32 logit matrix -> (out3 = [loss_cls, loss_bbox, acc] f32, dlogits [R, ld] f32)."""
    _need_cuda(logits, labels, label_weights, bbox_targets, bbox_weights)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.long
    out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    _check(lib().hvr_det_loss(_ptr(logits), logits.shape[1], cls_off, reg_off, ncls, _ptr(labels.contiguous()),
                              _ptr(label_weights.float().contiguous()), _ptr(bbox_targets.float().contiguous()),
                              _ptr(bbox_weights.float().contiguous()), logits.shape[0], float(beta), float(w_cls), float(w_bbox),
                              _ptr(out3), _ptr(dlogits), _stream()), 'hvr_det_loss')
    return out3, dlogits



# ---- training targets (include/hvr_hip.h "Training targets") ----
def max_iou_assign(boxes, gts, pos_iou_thr, neg_iou_thr, min_pos_iou, valid=None):
    """boxes [n, >=4] f32, gts [k,4] f32, valid [n] uint8/bool or None -> (gt_inds int64 [n], max_overlaps f32 [n]).
    neg_iou_thr: float (background = [0, thr)) or a (lo, hi) pair."""
    _need_cuda(boxes, gts, valid)
    assert boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.stride(1) == 1 and boxes.shape[1] >= 4
    gts = gts.contiguous().float()
    n, k = boxes.shape[0], gts.shape[0]
    if n == 0 or k == 0:
        raise ValueError('No gt or bboxes')  # max_iou_assigner.py:77-78
    lo, hi = (0.0, float(neg_iou_thr)) if isinstance(neg_iou_thr, (int, float)) else (float(neg_iou_thr[0]), float(neg_iou_thr[1]))
    if valid is not None:
        valid = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.contiguous()
        assert valid.dtype == torch.uint8 and valid.numel() == n
    gt_inds = torch.empty(n, dtype=torch.long, device=boxes.device)
    max_ov = torch.empty(n, dtype=torch.float32, device=boxes.device)
    nbytes = lib().hvr_max_iou_assign_workspace_bytes(n, k)
    ws = _workspace(nbytes, boxes.device, 'assign')
    _check(lib().hvr_max_iou_assign(_ptr(boxes), boxes.stride(0), n, _ptr(gts), k, _ptr(valid), float(pos_iou_thr), lo, hi,
                                    float(min_pos_iou), _ptr(gt_inds), _ptr(max_ov), _ptr(ws), nbytes, _stream()), 'hvr_max_iou_assign')
    return gt_inds, max_ov


def sample_pos_neg(cls, keys, num, num_expected_pos, neg_pos_ub=-1.0):
    """cls int64 [n] (>0 positive, ==0 negative), keys f32 [n] -> (inds int64 [num]: positives then negatives, counts int32 [2])."""
    _need_cuda(cls, keys)
    assert cls.dtype == torch.long and cls.is_contiguous() and keys.dtype == torch.float32 and keys.is_contiguous()
    assert cls.numel() == keys.numel()
    inds = torch.empty(int(num), dtype=torch.long, device=cls.device)
    counts = torch.empty(2, dtype=torch.int32, device=cls.device)
    _check(lib().hvr_sample_pos_neg(_ptr(cls), _ptr(keys), cls.numel(), int(num), int(num_expected_pos), float(neg_pos_ub), _ptr(inds),
                                    _ptr(counts), _stream()), 'hvr_sample_pos_neg')
    return inds, counts


def box_targets(boxes, gts, gt_labels, gt_inds, inds, counts, means, stds, pos_weight=-1.0, scatter=False):
    """-> (labels int64, label_weights, bbox_targets [.,4], bbox_weights [.,4]) with n rows (scatter) or num rows."""
    _need_cuda(boxes, gts, gt_labels, gt_inds, inds, counts)
    assert boxes.dtype == torch.float32 and boxes.stride(1) == 1 and gt_inds.dtype == torch.long and inds.dtype == torch.long
    gts = gts.contiguous().float()
    n, num = boxes.shape[0], inds.numel()
    rows = n if scatter else num
    dev = boxes.device
    labels = torch.empty(rows, dtype=torch.long, device=dev)
    label_w = torch.empty(rows, dtype=torch.float32, device=dev)
    bbox_t = torch.empty((rows, 4), dtype=torch.float32, device=dev)
    bbox_w = torch.empty((rows, 4), dtype=torch.float32, device=dev)
    m4, s4 = (ctypes.c_float * 4)(*[float(v) for v in means]), (ctypes.c_float * 4)(*[float(v) for v in stds])
    gl = gt_labels.contiguous() if gt_labels is not None else None
    _check(lib().hvr_box_targets(_ptr(boxes), boxes.stride(0), n, _ptr(gts), _ptr(gl), _ptr(gt_inds.contiguous()), _ptr(inds),
                                 _ptr(counts), num, ctypes.cast(m4, ctypes.c_void_p), ctypes.cast(s4, ctypes.c_void_p),
                                 float(pos_weight), int(bool(scatter)), _ptr(labels), _ptr(label_w), _ptr(bbox_t), _ptr(bbox_w),
                                 _stream()), 'hvr_box_targets')
    return labels, label_w, bbox_t, bbox_w


def rpn_loss(o, A, labels, label_weights, bbox_targets, bbox_weights, counts, beta):
    """o [rows, ldo] f32 (A objectness logits then 4A deltas per row) -> (out2 = [loss_rpn_cls, loss_rpn_bbox], d_o like o)."""
    _need_cuda(o, labels, label_weights, bbox_targets, bbox_weights, counts)
    assert o.dtype == torch.float32 and o.dim() == 2 and o.is_contiguous() and labels.dtype == torch.long
    rows = o.shape[0]
    assert labels.numel() == rows * A and counts.dtype == torch.int32
    out2 = torch.empty(2, dtype=torch.float32, device=o.device)
    d_o = torch.empty_like(o)
    _check(lib().hvr_rpn_loss(_ptr(o), o.shape[1], int(A), rows, _ptr(labels.contiguous()), _ptr(label_weights.contiguous()),
                              _ptr(bbox_targets.contiguous()), _ptr(bbox_weights.contiguous()), _ptr(counts), float(beta), _ptr(out2),
                              _ptr(d_o), _stream()), 'hvr_rpn_loss')
    return out2, d_o


def ce_rows(logits, cls_off, ncls, labels):
    """per-row softmax cross entropy of logits[:, cls_off:cls_off+ncls] -> f32 [R]."""
    _need_cuda(logits, labels)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.long
    loss = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    _check(lib().hvr_ce_rows(_ptr(logits), logits.shape[1], cls_off, ncls, _ptr(labels.contiguous()), logits.shape[0], _ptr(loss),
                             _stream()), 'hvr_ce_rows')
    return loss


def triplet_margin(q, k, anchor_idx, pos_idx, neg_idx, margin, need_grad=True):
    """Stand-in triplet margin loss (see include/hvr_hip.h) -> (out2 = [loss, active triples] f32, dq f32 [Mq,D], dk f32 [Mk,D])."""
    _need_cuda(q, k, anchor_idx, pos_idx, neg_idx)
    assert q.dim() == 2 and k.dim() == 2 and q.shape[1] == k.shape[1] and q.dtype == k.dtype and q.stride(1) == 1 and k.stride(1) == 1
    n = anchor_idx.numel()
    assert n > 0 and pos_idx.numel() == n and neg_idx.numel() == n and anchor_idx.dtype == torch.long
    D = q.shape[1]
    out2 = torch.empty(2, dtype=torch.float32, device=q.device)
    dq = torch.empty((q.shape[0], D), dtype=torch.float32, device=q.device) if need_grad else None
    dk = torch.empty((k.shape[0], D), dtype=torch.float32, device=q.device) if need_grad else None
    ws = _workspace(n * 12, q.device, 'triplet')
    _check(lib().hvr_triplet_margin(_ptr(q), q.stride(0), _ptr(k), k.stride(0), D, q.shape[0], k.shape[0], _ptr(anchor_idx.contiguous()),
                                    _ptr(pos_idx.contiguous()), _ptr(neg_idx.contiguous()), n, float(margin), _dt(q), _ptr(ws), n * 12,
                                    _ptr(out2), _ptr(dq), _ptr(dk), _stream()), 'hvr_triplet_margin')
    return out2, dq, dk


def ingest_frame(frame, new_hw, pad_hw, mean, std, to_rgb=False):
    """frame uint8 [H, W, 3] (cuda, rows contiguous) -> f32 [1, 3, pad_h, pad_w]: resize + normalise + pad in one kernel."""
    _need_cuda(frame)
    assert frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3 and frame.stride(2) == 1 and frame.stride(1) == 3
    out = torch.empty((1, 3, int(pad_hw[0]), int(pad_hw[1])), dtype=torch.float32, device=frame.device)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    _check(lib().hvr_ingest_frame(_ptr(frame), frame.shape[0], frame.shape[1], frame.stride(0), _ptr(out), int(new_hw[0]), int(new_hw[1]),
                                  int(pad_hw[0]), int(pad_hw[1]), ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p),
                                  int(bool(to_rgb)), _stream()), 'hvr_ingest_frame')
    return out


def mining_argreduce(aff, labels, all_labels):
    """aff f32 [Mq, Mk] (row stride >= Mk), labels int64 [Mq], all_labels int64 [Mk] -> int64 [Mq, 4]:
    (argmax over different-label keys, argmin over same-label keys, top-2 over different-label keys)."""
    _need_cuda(aff, labels, all_labels)
    assert aff.dtype == torch.float32 and aff.dim() == 2 and aff.stride(1) == 1
    assert labels.dtype == torch.long and all_labels.dtype == torch.long
    Mq, Mk = aff.shape
    assert labels.numel() == Mq and all_labels.numel() == Mk
    out = torch.empty((Mq, 4), dtype=torch.long, device=aff.device)
    _check(lib().hvr_mining_argreduce(_ptr(aff), Mq, Mk, aff.stride(0), _ptr(labels.contiguous()), _ptr(all_labels.contiguous()), _ptr(out),
                                      _stream()), 'hvr_mining_argreduce')
    return out


def det_loss_sampled(logits, cls_off, reg_off, ncls, labels, label_weights, bbox_targets, bbox_weights, sel_counts, beta=1.0):
    """hvr_det_loss in the OHEM form: weights are zero outside the selected rows, sel_counts int32 [2] their number."""
    _need_cuda(logits, labels, label_weights, bbox_targets, bbox_weights, sel_counts)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.long and sel_counts.dtype == torch.int32
    out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    _check(lib().hvr_det_loss_sampled(_ptr(logits), logits.shape[1], cls_off, reg_off, ncls, _ptr(labels.contiguous()),
                                      _ptr(label_weights.float().contiguous()), _ptr(bbox_targets.float().contiguous()),
                                      _ptr(bbox_weights.float().contiguous()), logits.shape[0], _ptr(sel_counts), float(beta),
                                      _ptr(out3), _ptr(dlogits), _stream()), 'hvr_det_loss_sampled')
    return out3, dlogits


def roi_align_fwd(feat, rois, out_h, out_w, spatial_scale, sample_num, layout):
    """layout NCHW: feat [B,C,H,W] -> [K,C,oh,ow]; NHWC: feat [B,H,W,C] -> [K,oh,ow,C] (physical shapes)."""
    _need_cuda(feat, rois)
    if feat.dtype == SPLIT:   # interpolated in f32, handed back in the split format
        return cast(roi_align_fwd(cast(feat, torch.float32), rois, out_h, out_w, spatial_scale, sample_num, layout), SPLIT)
    rois = rois.contiguous().float()
    K = rois.shape[0]
    if layout == LAYOUT_NCHW:
        B, C, H, W = feat.shape
        out = torch.empty((K, C, out_h, out_w), dtype=feat.dtype, device=feat.device)
    else:
        B, H, W, C = feat.shape
        out = torch.empty((K, out_h, out_w, C), dtype=feat.dtype, device=feat.device)
    assert feat.is_contiguous()
    with _span('roi_align', float((feat.numel() + out.numel()) * feat.element_size() + rois.numel() * 4)):
        _check(lib().hvr_roi_align_fwd(_ptr(feat), _ptr(rois), _ptr(out), B, C, H, W, K, out_h, out_w, float(spatial_scale),
                                       int(sample_num), _dt(feat), layout, _stream()), 'hvr_roi_align_fwd')
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale, sample_num, layout):
    _need_cuda(grad_out, rois)
    grad_out = grad_out.contiguous().float()
    rois = rois.contiguous().float()
    grad_in = torch.zeros(feat_shape, dtype=torch.float32, device=grad_out.device)
    if layout == LAYOUT_NCHW:
        B, C, H, W = feat_shape
        K, _, PH, PW = grad_out.shape
    else:
        B, H, W, C = feat_shape
        K, PH, PW, _ = grad_out.shape
    _check(lib().hvr_roi_align_bwd(_ptr(grad_out), _ptr(rois), _ptr(grad_in), B, C, H, W, K, PH, PW, float(spatial_scale),
                                   int(sample_num), layout, _stream()), 'hvr_roi_align_bwd')
    return grad_in


def nms_first(dets, iou_thr, max_keep, ge_semantics=True):
    """The first max_keep survivors (in score order) of greedy NMS: nms(dets, thr)[:max_keep] of rpn_head.py:95-97, as
    ascending input indices.  dets [n,5] f32 cuda."""
    _need_cuda(dets)
    dets = dets.contiguous().float()
    n = dets.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=dets.device)
    keep = torch.empty(n, dtype=torch.long, device=dets.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=dets.device)
    ws = _workspace(lib().hvr_nms_workspace_bytes(n), dets.device, 'nms')
    _check(lib().hvr_nms_first(_ptr(dets), n, float(iou_thr), int(ge_semantics), int(max_keep), _ptr(keep), _ptr(cnt), _ptr(ws),
                               ws.numel(), _stream()), 'hvr_nms_first')
    return keep[:int(cnt.item())]


def nms(dets, iou_thr, ge_semantics=True):
    """dets [n,5] f32 cuda -> int64 indices kept (ascending input order). One D2H of the count."""
    _need_cuda(dets)
    dets = dets.contiguous().float()
    n = dets.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=dets.device)
    keep = torch.empty(n, dtype=torch.long, device=dets.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=dets.device)
    ws = _workspace(lib().hvr_nms_workspace_bytes(n), dets.device, 'nms')
    _check(lib().hvr_nms(_ptr(dets), n, float(iou_thr), int(ge_semantics), _ptr(keep), _ptr(cnt), _ptr(ws), ws.numel(),
                         _stream()), 'hvr_nms')
    return keep[:int(cnt.item())]


def rpn_proposals(cls, reg, base_anchors, anchor_stride, means, stds, img_shape, nms_pre, nms_post, max_num, nms_thr,
                  wh_ratio_clip=16 / 1000):
    """cls [T,H,W,A] f32, reg [T,H,W,4A] f32 (physical NHWC; may be channel slices of one wider contiguous
    [T,H,W,P] tensor) -> (proposals [T,max_num,5], counts [T] int32)."""
    _need_cuda(cls, reg)
    assert cls.dtype == torch.float32 and reg.dtype == torch.float32
    T, H, W, A = cls.shape
    for t, ch in ((cls, A), (reg, 4 * A)):
        assert t.stride(3) == 1 and t.stride(1) == W * t.stride(2) and t.stride(0) == H * W * t.stride(2) and t.stride(2) >= ch
    props = torch.zeros((T, max_num, 5), dtype=torch.float32, device=cls.device)
    counts = torch.zeros(T, dtype=torch.int32, device=cls.device)
    ba = (ctypes.c_float * (A * 4))(*[float(v) for v in base_anchors.reshape(-1).tolist()])
    mm = (ctypes.c_float * 4)(*[float(v) for v in means])
    ss = (ctypes.c_float * 4)(*[float(v) for v in stds])
    d = RpnDesc(cls=cls.data_ptr(), reg=reg.data_ptr(), T=T, H=H, W=W, A=A, anchor_stride=int(anchor_stride),
                base_anchors=ctypes.cast(ba, ctypes.c_void_p), means=ctypes.cast(mm, ctypes.c_void_p),
                stds=ctypes.cast(ss, ctypes.c_void_p), img_h=float(img_shape[0]), img_w=float(img_shape[1]),
                wh_ratio_clip=float(wh_ratio_clip), nms_pre=int(nms_pre), nms_post=int(nms_post), max_num=int(max_num),
                nms_thr=float(nms_thr), proposals=props.data_ptr(), counts=counts.data_ptr(),
                cls_pitch=cls.stride(2), reg_pitch=reg.stride(2))
    ws = _workspace(lib().hvr_rpn_workspace_bytes(T, H, W, A, int(nms_pre)), cls.device, 'rpn')
    with _span('rpn_proposals', float(cls.numel() * 4 + reg.numel() * 4)):
        _check(lib().hvr_rpn_proposals(ctypes.byref(d), _ptr(ws), ws.numel(), _stream()), 'hvr_rpn_proposals')
    return props, counts


def rpn_wide_frames(frames=-1):
    """Calls of rpn_proposals with T <= frames take the chip-wide kernels (same proposals bit for bit); -> previous value."""
    return int(lib().hvr_rpn_wide_frames(int(frames)))


def det_decode(logits, cls_off, reg_off, ncls, rois, means, stds, img_shape, scale_factor, wh_ratio_clip=16 / 1000):
    """logits [R,ld] f32 -> (scores [R,ncls], boxes [R,4])."""
    _need_cuda(logits, rois)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    rois = rois.contiguous().float()
    R = rois.shape[0]
    scores = torch.empty((R, ncls), dtype=torch.float32, device=logits.device)
    boxes = torch.empty((R, 4), dtype=torch.float32, device=logits.device)
    mm = (ctypes.c_float * 4)(*[float(v) for v in means])
    ss = (ctypes.c_float * 4)(*[float(v) for v in stds])
    ih, iw = (float(img_shape[0]), float(img_shape[1])) if img_shape is not None else (0.0, 0.0)
    _check(lib().hvr_det_decode(_ptr(logits), logits.stride(0), cls_off, reg_off, ncls, _ptr(rois), R,
                                ctypes.cast(mm, ctypes.c_void_p), ctypes.cast(ss, ctypes.c_void_p), float(wh_ratio_clip),
                                ih, iw, float(scale_factor), _ptr(scores), _ptr(boxes), _stream()), 'hvr_det_decode')
    return scores, boxes


def multiclass_nms(boxes, scores, score_thr, iou_thr, max_num):
    """boxes [R,4], scores [R,ncls] f32 -> (dets [max_num,5], labels [max_num] int64, n int32[1]) device tensors."""
    _need_cuda(boxes, scores)
    R, ncls = scores.shape
    # (the merge kernel writes the count and zeroes the rows behind it: no fill launches in front of it)
    alloc = torch.empty if R > 0 else torch.zeros
    dets = alloc((max_num, 5), dtype=torch.float32, device=boxes.device)
    labels = alloc(max_num, dtype=torch.long, device=boxes.device)
    n_out = alloc(1, dtype=torch.int32, device=boxes.device)
    ws = _workspace(lib().hvr_multiclass_nms_workspace_bytes(R, ncls), boxes.device, 'mcnms')
    _check(lib().hvr_multiclass_nms(_ptr(boxes.contiguous()), _ptr(scores.contiguous()), R, ncls, float(score_thr),
                                    float(iou_thr), int(max_num), _ptr(dets), _ptr(labels), _ptr(n_out), _ptr(ws),
                                    ws.numel(), _stream()), 'hvr_multiclass_nms')
    return dets, labels, n_out


def cast(x, dtype, scale=None):
    """x in the operand format `dtype`.  Split-half tensors are kept x SPLIT_ACT_SCALE (see the note at as_operand): casting
    into the format applies the factor, casting out of it removes it; `scale` overrides that (as_operand: the weight scale)."""
    _need_cuda(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    if scale is None:
        scale = SPLIT_ACT_SCALE if dtype == SPLIT else (1.0 / SPLIT_ACT_SCALE if x.dtype == SPLIT else 1.0)
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _check(lib().hvr_cast_scaled(_ptr(x), _ptr(out), x.numel(), _dt(x), _dt(out), float(scale), _stream()), 'hvr_cast')
    return out


def transpose_pad(x, ldt, out=None):
    """[R, C] -> [C, ldt] with columns R.. zero (the relation's V^T operand)."""
    _need_cuda(x)
    x = x.contiguous()
    R, C = x.shape
    if out is None:
        out = torch.empty((C, ldt), dtype=x.dtype, device=x.device)
    else:
        assert tuple(out.shape) == (C, ldt) and out.dtype == x.dtype and out.is_contiguous()
    _check(lib().hvr_transpose_pad(_ptr(x), _ptr(out), R, C, C, ldt, _dt(x), _stream()), 'hvr_transpose_pad')
    return out


def nchw_to_nhwc(x, dtype=None):
    """[B,C,H,W] contiguous -> physical [B,H,W,C] (optionally casting)."""
    _need_cuda(x)
    x = x.contiguous()
    if dtype == SPLIT or x.dtype == SPLIT:
        assert x.dtype != SPLIT, 'split-half tensors are NHWC only'
        return cast(nchw_to_nhwc(x, torch.float32), SPLIT)
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=dtype or x.dtype, device=x.device)
    _check(lib().hvr_permute_nchw_nhwc(_ptr(x), _ptr(out), B, C, H * W, 1, _dt(x), _dt(out), _stream()), 'hvr_permute')
    return out


def nhwc_to_nchw(x, dtype=None):
    """physical [B,H,W,C] contiguous -> [B,C,H,W] contiguous."""
    _need_cuda(x)
    x = x.contiguous()
    if x.dtype == SPLIT:
        x = cast(x, torch.float32)
    assert dtype != SPLIT, 'split-half tensors are NHWC only'
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=dtype or x.dtype, device=x.device)
    _check(lib().hvr_permute_nchw_nhwc(_ptr(x), _ptr(out), B, C, H * W, 0, _dt(x), _dt(out), _stream()), 'hvr_permute')
    return out
