"""Kernel-level numerics of the precision ladder between the bf16 benchmark mode and the exact-f32 parity mode
(include/hvr_hip.h: HVR_F16 = IEEE half operands, HVR_F16S = split half -- three half MFMAs per product), on the GPU,
against a float64 statement of the same op on the same seeded inputs.  The reference computes in f32 throughout
(configs/faster_rcnn_r101_hrnmp_c5.py has no fp16 key; mmdet/core/fp16/decorators.py:9-160 are its optional islands).

Tolerances, each stated where it is used:
  * half: operands are rounded to half before both sides see them, so what is left is the f32 accumulation order and the
    half rounding of stored outputs (2^-11 relative);
  * split half: operands carry 22 significant bits (hi + lo * 2^-11), products are exact in the MFMA, sums are f32 --
    the result tracks the f64 product of the UN-rounded f32 inputs to ~1e-6 of the output scale.
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hvrnet_amd import native  # noqa: E402

DEV = 'cuda:0'
SPLIT = native.SPLIT


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _to(x, dtype):
    """f32 host tensor -> device ACTIVATION operand in `dtype` (split half goes through hvr_cast, unscaled)."""
    return native.cast(x.to(DEV).contiguous(), dtype)


def _tow(x, dtype):
    """f32 host tensor -> device WEIGHT operand (native.as_operand: split-half weights are stored x 2^6, gemm / conv2d pass alpha)."""
    return native.as_operand(x.to(DEV), dtype)


def _backw(y):
    """a weight operand back in f32 (split half: / SPLIT_WEIGHT_SCALE)."""
    return native.cast(y, torch.float32, scale=1.0 / native.SPLIT_WEIGHT_SCALE).cpu() if y.dtype == SPLIT else _back(y)


def _back(y):
    return native.cast(y, torch.float32).cpu() if y.dtype != torch.float32 else y.cpu()


def test_split_half_cast_round_trip_keeps_22_bits():
    """f32 -> split half -> f32, raw format (scale 1): |error| <= 2^-21 |x| for values in the half range (2^-24 absolute beneath
    2^-3, where lo is a half subnormal), exact zeros stay zero, values beyond 65504 saturate (no inf - inf).  native.cast keeps
    split ACTIVATIONS x 2^4: the same bounds hold from 2^-7 up, and the range ends at 4094."""
    x = _rand((37, 192), 1, 3.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, 1e5, -1e5, 1e-6])
    x[1, :4] = torch.tensor([6e-5, 3e-8, 1e-9, 1234.5678])
    s = native.cast(x.to(DEV), SPLIT, scale=1.0)
    assert s.dtype == SPLIT and s.shape == x.shape
    y = native.cast(s, torch.float32, scale=1.0).cpu()
    ok = x.abs() <= 65504
    err = (y - x).abs()
    assert (err[ok] <= x.abs()[ok] * 2.0 ** -21 + 2.0 ** -24).all(), err[ok].max()
    assert y[0, 0] == 0 and y[0, 1] == 0 and y[0, 2] == 1 and y[0, 3] == -1
    assert y[0, 5] == 65504 and y[0, 6] == -65504 and torch.isfinite(y).all()
    ya = native.cast(native.cast(x.to(DEV), SPLIT), torch.float32).cpu()     # the activation convention
    oka = x.abs() <= 65504 / native.SPLIT_ACT_SCALE
    assert ((ya - x).abs()[oka] <= x.abs()[oka] * 2.0 ** -21 + 2.0 ** -28).all()
    assert ya[0, 4] == 65504 / native.SPLIT_ACT_SCALE and ya[0, 6] == -65504 / native.SPLIT_ACT_SCALE
    # the split form is what a half cast gives for values a half holds exactly
    h = _rand((8, 64), 2).half().float()
    assert torch.equal(native.cast(native.cast(h.to(DEV), SPLIT), torch.float32).cpu(), h)
    # half <-> f32 agree with torch's casts; bf16 <-> half go through f32
    z = _rand((5, 64), 3, 10.0)
    assert torch.equal(native.cast(z.to(DEV), torch.float16).cpu(), z.half())
    assert torch.equal(native.cast(z.half().to(DEV), torch.float32).cpu(), z.half().float())
    assert torch.equal(native.cast(z.bfloat16().to(DEV), torch.float16).cpu(), z.bfloat16().float().half())
    with pytest.raises(native.HvrError):
        native.cast(torch.zeros((3, 40), device=DEV), SPLIT)   # rows must be whole 32-element groups


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize('dtype', [torch.float16, SPLIT])
def test_every_tile_shape_on_half_and_split_operands(dtype, tile):
    """hvr_gemm on half / split-half operands through every tile shape of the engine (hint 0 = the cost model's pick; the
    pipelined 3 / 4-stage shapes walk the split K loop's three passes through the same LDS ring): bias + residual + ReLU,
    against float64.  Ragged M and N."""
    M, N, K = 300, 328, 1024
    a, w, bias, resid = _rand((M, K), 11), _rand((N, K), 12, 0.05), _rand((N,), 13), _rand((M, 384), 14)
    ad, wd, rd = _to(a, dtype), _tow(w, dtype), _to(resid, dtype)
    af, wf, rf = _back(ad).double(), _backw(wd).double(), _back(rd).double()[:, :N]
    ref = torch.relu(af @ wf.t() + bias.double() + rf)
    out = torch.zeros((M, 384), dtype=dtype, device=DEV)
    # N = 328 is not a whole number of 32-column groups: split half takes N % 8 == 0 outputs inside a wider (ldc % 32 == 0) matrix
    y = native.gemm(ad, wd, bias.to(DEV), rd[:, :N], relu=True, tile=tile, out=out[:, :N])
    scale = ref.abs().max().item()
    err = (_back(out)[:, :N].double() - ref).abs().max().item()
    if dtype == SPLIT:
        # operands: 2^-21 relative each (already inside af / wf), accumulation in f32 over K = 1024, output split again
        assert err < 3e-6 * scale, (err, scale)
        # and against the UN-rounded f32 inputs: f32-grade
        ref32 = torch.relu(a.double() @ w.double().t() + bias.double() + resid.double()[:, :N])
        assert (_back(out)[:, :N].double() - ref32).abs().max().item() < 5e-6 * scale
    else:
        assert err < 2.0 ** -10 * scale, (err, scale)          # one half rounding of the output
    assert y.data_ptr() == out.data_ptr()
    assert (_back(out)[:, N:] == 0).all()                        # columns beyond N untouched
    # f32 output of the same product: no output rounding at all
    y32 = native.gemm(ad, wd, out_f32=True, tile=tile)
    assert y32.dtype == torch.float32
    e32 = (y32.cpu().double() - af @ wf.t()).abs().max().item()
    assert e32 < 2e-6 * scale, (e32, scale)


@pytest.mark.parametrize('dtype', [torch.float16, SPLIT])
@pytest.mark.parametrize('cfg', [
    dict(Cin=64, Cout=64, k=3, stride=1, pad=1, dil=1, H=19, W=23),
    dict(Cin=128, Cout=128, k=3, stride=1, pad=2, dil=2, H=17, W=21),   # res5-style dilation
    dict(Cin=256, Cout=128, k=1, stride=2, pad=0, dil=1, H=20, W=31),   # caffe-style strided 1x1
    dict(Cin=64, Cout=256, k=1, stride=1, pad=0, dil=1, H=9, W=14),
    dict(Cin=256, Cout=256, k=3, stride=1, pad=1, dil=1, H=38, W=63),   # layer 3's 3x3 at window size (pipelined shapes)
])
def test_conv_on_half_and_split_operands(cfg, dtype):
    """hvr_conv2d_nhwc (implicit GEMM: the filter-tap gather, hardware zero fill of padding taps, strides, dilation) with
    bias + residual + ReLU on half / split-half maps against float64 conv2d of the operands as the kernel sees them."""
    B = 2
    x = _rand((B, cfg['Cin'], cfg['H'], cfg['W']), 21)
    w = _rand((cfg['Cout'], cfg['Cin'], cfg['k'], cfg['k']), 22, 0.05)
    bias = _rand((cfg['Cout'],), 23)
    xn, wn = _to(x.permute(0, 2, 3, 1), dtype), _tow(w.permute(0, 2, 3, 1), dtype)
    xf, wf = _back(xn).permute(0, 3, 1, 2).double(), _backw(wn).permute(0, 3, 1, 2).double()
    ref = F.conv2d(xf, wf, bias.double(), stride=cfg['stride'], padding=cfg['pad'], dilation=cfg['dil'])
    resid = _rand(tuple(ref.shape), 24)
    rn = _to(resid.permute(0, 2, 3, 1), dtype)
    ref = torch.relu(ref + _back(rn).permute(0, 3, 1, 2).double())
    y = native.conv2d_nhwc(xn, wn, bias.to(DEV), rn, relu=True, stride=cfg['stride'], pad=cfg['pad'], dil=cfg['dil'])
    assert y.dtype == dtype and tuple(y.shape) == (B, ref.shape[2], ref.shape[3], cfg['Cout'])
    scale = ref.abs().max().item()
    err = (_back(y).permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < (3e-6 if dtype == SPLIT else 2.0 ** -10) * scale, (err, scale)
    y32 = native.conv2d_nhwc(xn, wn, bias.to(DEV), None, relu=False, stride=cfg['stride'], pad=cfg['pad'], dil=cfg['dil'], out_f32=True)
    ref32 = F.conv2d(xf, wf, bias.double(), stride=cfg['stride'], padding=cfg['pad'], dilation=cfg['dil'])
    assert (y32.cpu().permute(0, 3, 1, 2).double() - ref32).abs().max().item() < 2e-6 * scale


@pytest.mark.parametrize('dtype', [torch.float16, SPLIT])
@pytest.mark.parametrize('Mq,Mk', [(300, 300), (96, 1500), (300, 4500), (1100, 1100)])
def test_relation_on_half_and_split_operands(Mq, Mk, dtype):
    """hvr_relation_fwd = softmax(q k^T / sqrt(D)) v (selsa_bbox_head.py:166-182) on half / split-half operands against
    float64 attention over the operands as the kernel sees them.  Split half: the scores pass writes the block-relative
    exponentials in the split format, one sweep normalises them, O = P V is a plain three-pass product."""
    D = 1024
    q, k, v = _rand((Mq, D), 31, 0.5), _rand((Mk, D), 32, 0.5), _rand((Mk, D), 33)
    qd, kd, vd = _to(q, dtype), _to(k, dtype), _to(v, dtype)
    qf, kf, vf = _back(qd).double(), _back(kd).double(), _back(vd).double()
    ref = torch.softmax(qf @ kf.t() / math.sqrt(D), dim=1) @ vf
    o = native.relation_fwd(qd, kd, vd, 1.0 / math.sqrt(D))
    assert o.dtype == dtype
    scale = ref.abs().max().item()
    err = (_back(o).double() - ref).abs().max().item()
    # half: P~ is rounded to half (2^-11 relative per probability, averaged over the keys) and so is the output
    assert err < (1e-5 if dtype == SPLIT else 2e-3) * scale, (err, scale)


@pytest.mark.parametrize('G,Mq,Mk', [(1, 4500, 4500), (3, 4321, 4100), (4, 4500, 4500)])
def test_split_half_relation_window_size_on_the_big_tiles(G, Mq, Mk):
    """Window-sized split-half problems take the persistent 352 x 256 score tiles (relation_bt.hip on f16s_t: the hi / lo planes of a
    K-step are the two 64-byte halves of the staged lines, three MFMAs per fragment pair in six phases, V^T planes copied by the idle
    workgroups), G of them in ONE launch through hvr_relation_fwd_grouped, and from three groups on the 288 x 256 apply launch
    (relation_apply_bt.hip on f16s_t: integer block maxima, the block weight 2^-shift applied to both planes of the P~ fragments by
    v_pk_mul_f16) instead of the normalising sweep + plain product.
      exact=True : every group's rows are hvr_relation_fwd's bit for bit (scores of all groups in one launch, the tail per group);
      default    : against float64 attention over the operands as the kernel sees them, on sampled rows (first / last rows of the
                   first, a middle and the last row tile), at the tolerance of the single call -- with a dominant key in the last
                   (ragged, for Mk = 4 100) key tile, one in the first, and a row whose blocks span more than the 2^-25 the half
                   weights can express (their terms vanish, as they do in f32)."""
    D = 1024
    q, k, v = _rand((G * Mq, D), 61, 0.5), _rand((G * Mk, D), 62, 0.5), _rand((G * Mk, D), 63)
    for g in range(G):
        k[g * Mk + Mk - 2] = q[g * Mq + 7] * 24.0          # row 7: a late-block maximum far above the rest (shift > 25 everywhere else)
        k[g * Mk + 5] = q[g * Mq + Mq - 1] * 16.0          # the last row: an early-block maximum
        k[g * Mk + 900] = q[g * Mq + 3] * 1.2              # row 3: a block a few powers of two above its neighbours (small shifts)
    qd, kd, vd = _to(q, SPLIT), _to(k, SPLIT), _to(v, SPLIT)
    sc = 1.0 / math.sqrt(D)
    o = native.relation_fwd_grouped(qd, kd, vd, sc, G)
    oe = native.relation_fwd_grouped(qd, kd, vd, sc, G, exact=True)
    assert o.dtype == SPLIT and torch.equal(_back(o), _back(native.relation_fwd_grouped(qd, kd, vd, sc, G)))
    rows = torch.cat([torch.arange(0, 12), torch.arange(346, 358), torch.arange(2100, 2108), torch.arange(Mq - 12, Mq)])
    for g in range(G):
        single = _back(native.relation_fwd(qd[g * Mq:(g + 1) * Mq], kd[g * Mk:(g + 1) * Mk], vd[g * Mk:(g + 1) * Mk], sc))
        og, oeg = _back(o[g * Mq:(g + 1) * Mq]), _back(oe[g * Mq:(g + 1) * Mq])
        assert torch.equal(oeg, single), 'group %d: exact form differs from hvr_relation_fwd' % g
        qf = _back(qd[g * Mq:(g + 1) * Mq])[rows].double()
        kf, vf = _back(kd[g * Mk:(g + 1) * Mk]).double(), _back(vd[g * Mk:(g + 1) * Mk]).double()
        ref = torch.softmax(qf @ kf.t() / math.sqrt(D), dim=1) @ vf
        assert torch.isfinite(og).all()
        scale = ref.abs().max().item()
        err, err_single = (og[rows].double() - ref).abs().max().item(), (single[rows].double() - ref).abs().max().item()
        assert err < 1e-5 * scale and err_single < 1e-5 * scale, (g, err, err_single, scale)
        assert (og - single).abs().max().item() < 2e-5 * single.abs().max().item(), g


def test_split_half_key_stage_over_the_clips_of_a_call():
    """The key stage (300 queries x 4 500 keys per clip, hrnmp_bbox_head.py:269-278) of W = 4 clips in ONE hvr_relation_fwd_grouped call on
    split-half operands: scores and the folded apply pass as one tile-engine launch each over all clips (GemmParams::batch) -- every
    clip's rows are hvr_relation_fwd's bit for bit, exact or not."""
    G, Mq, Mk, D = 4, 300, 4500, 1024
    q, k, v = _rand((G * Mq, D), 161, 0.5), _rand((G * Mk, D), 162, 0.5), _rand((G * Mk, D), 163)
    qd, kd, vd = _to(q, SPLIT), _to(k, SPLIT), _to(v, SPLIT)
    sc = 1.0 / math.sqrt(D)
    o = native.relation_fwd_grouped(qd, kd, vd, sc, G)
    oe = native.relation_fwd_grouped(qd, kd, vd, sc, G, exact=True)
    for g in range(G):
        single = native.relation_fwd(qd[g * Mq:(g + 1) * Mq], kd[g * Mk:(g + 1) * Mk], vd[g * Mk:(g + 1) * Mk], sc)
        assert torch.equal(_back(o[g * Mq:(g + 1) * Mq]), _back(single)) and torch.equal(_back(oe[g * Mq:(g + 1) * Mq]), _back(single)), g


def test_relation_split_half_peaky_rows():
    """One key dominates a row by e^40 and sits in a different 128-key block than the runner-up (block maxima differ by far
    more than the half range of exp2): the normalising sweep works from the f32 block statistics."""
    D, Mq, Mk = 1024, 64, 700
    q, k, v = _rand((Mq, D), 41, 0.2), _rand((Mk, D), 42, 0.2), _rand((Mk, D), 43)
    k[5] = q[3] * 40.0
    k[600] = q[3] * 39.0
    qd, kd, vd = _to(q, SPLIT), _to(k, SPLIT), _to(v, SPLIT)
    qf, kf, vf = _back(qd).double(), _back(kd).double(), _back(vd).double()
    ref = torch.softmax(qf @ kf.t() / math.sqrt(D), dim=1) @ vf
    o = _back(native.relation_fwd(qd, kd, vd, 1.0 / math.sqrt(D))).double()
    assert torch.isfinite(o).all()
    assert (o - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_split_half_relation_folded_and_swept_forms_agree(tmp_path):
    """The split-half relation core picks its apply form by the query-row count (capi.hip: block weights folded into the apply pass
    below 1 024 rows, one normalising sweep + a plain product from there) and HVR_SPLIT_NORMALIZE = 0 / 1 forces one; the switch is
    read once per process, so each form runs in a child: the key stage (300 x 4 500) and a window-sized problem (1 100 x 4 500)
    in BOTH forms against the f64 softmax of the operands as the kernel sees them, and against each other."""
    import subprocess
    import sys
    code = r'''
import math, sys, torch
sys.path.insert(0, %r)
from hvrnet_amd import native
D, outs = 1024, []
for Mq, Mk, seed in ((300, 4500, 61), (1100, 4500, 71)):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn((m, D), generator=g) * sc for m, sc in ((Mq, 0.5), (Mk, 0.5), (Mk, 1.0)))
    k[4000] = q[3] * 6.0          # a late-block maximum far above the rest of row 3
    qd, kd, vd = (native.cast(t.cuda().contiguous(), native.SPLIT) for t in (q, k, v))
    back = lambda t: native.cast(t, torch.float32).cpu()
    ref = torch.softmax(back(qd).double() @ back(kd).double().t() / math.sqrt(D), dim=1) @ back(vd).double()
    o = back(native.relation_fwd(qd, kd, vd, 1.0 / math.sqrt(D))).double()
    err = (o - ref).abs().max().item()
    assert err < 2e-5 * ref.abs().max().item(), (Mq, err)
    outs.append(o)
torch.save(outs, sys.argv[1])
print('ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for form in ('0', '1'):
        path = str(tmp_path / ('split_%s.pt' % form))
        r = subprocess.run([sys.executable, '-c', code % root, path], capture_output=True, text=True, timeout=300, env=dict(os.environ, HVR_SPLIT_NORMALIZE=form))
        assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-1500:]
        res[form] = torch.load(path)
    for a, b in zip(res['0'], res['1']):
        assert (a - b).abs().max().item() < 2e-5 * b.abs().max().item()


@pytest.mark.parametrize('R,C', [(300, 1024), (4500, 1024), (65, 64)])
def test_transpose_pad_split_half_is_exact(R, C):
    x = _rand((R, C), 51)
    xs = native.cast(x.to(DEV), SPLIT)
    ldt = (R + 127) // 128 * 128
    t = native.transpose_pad(xs, ldt)
    assert t.dtype == SPLIT and tuple(t.shape) == (C, ldt)
    back = native.cast(t, torch.float32).cpu()
    assert torch.equal(back[:, :R], native.cast(xs, torch.float32).cpu().t())
    assert (back[:, R:] == 0).all()


def test_roi_align_and_pooling_on_half_and_split_maps():
    """RoIAlign (roi_align_kernel.cu:63-141) on half maps uses the bf16 kernels' gathers with half unpacking; split-half maps
    are interpolated in f32 and handed back in the split format: both against the f32 kernel on the same (rounded) map."""
    B, H, W, C = 3, 38, 63, 256
    feat = _rand((B, H, W, C), 61)
    g = torch.Generator().manual_seed(62)
    K = 200
    x1, y1 = torch.rand(K, generator=g) * 900, torch.rand(K, generator=g) * 500
    rois = torch.stack([torch.randint(0, B, (K,), generator=g).float(), x1, y1, x1 + torch.rand(K, generator=g) * 300 + 4,
                        y1 + torch.rand(K, generator=g) * 200 + 4], 1)
    for dtype, tol in ((torch.float16, 2.0 ** -10), (SPLIT, 1e-6)):
        fd = _to(feat, dtype)
        want = native.roi_align_fwd(native.cast(fd, torch.float32), rois.to(DEV), 7, 7, 1 / 16.0, 2, native.LAYOUT_NHWC).cpu()
        got = native.roi_align_fwd(fd, rois.to(DEV), 7, 7, 1 / 16.0, 2, native.LAYOUT_NHWC)
        assert got.dtype == dtype and tuple(got.shape) == (K, 7, 7, C)
        assert (_back(got) - want).abs().max().item() <= tol * want.abs().max().item()
    x = _rand((2, 21, 33, 64), 63)
    want = native.maxpool3x3s2_nhwc(x.to(DEV)).cpu()
    for dtype in (torch.float16, SPLIT):
        xd = _to(x, dtype)
        got = native.maxpool3x3s2_nhwc(xd)
        assert got.dtype == dtype
        assert torch.equal(_back(got), native.maxpool3x3s2_nhwc(native.cast(xd, torch.float32)).cpu())
        assert (_back(got) - want).abs().max().item() <= 2.0 ** -10 * want.abs().max().item()


def test_split_half_rejects_what_it_cannot_address():
    a = native.cast(torch.zeros((64, 128), device=DEV), SPLIT)
    w = native.as_operand(torch.zeros((64, 128), device=DEV), SPLIT)
    with pytest.raises(native.HvrError):     # a column slice that does not start on a 32-element group
        native.gemm(a[:, 16:80], w[:, :64])
    native.gemm(a[:, 32:96], w[:, :64])      # one that does
    out = torch.zeros((64, 100), dtype=SPLIT, device=DEV)
    with pytest.raises(native.HvrError):     # output rows must be whole groups
        native.gemm(a, w, out=out[:, :64])


def test_stem_patch_rows_in_half_and_split_formats():
    """hvr_im2col_stem writes the 7x7/2 stem's patch matrix (resnet.py:456-466) directly in the operand format: the split-half
    rows equal hvr_cast of the f32 rows bit for bit, the half rows equal torch's cast."""
    img = _rand((2, 3, 61, 95), 71, 50.0).to(DEV)
    cols32, OH, OW = native.im2col_stem(img, torch.float32)
    cols_s = torch.empty((2 * OH * OW, 192), dtype=SPLIT, device=DEV)     # the C entry point writes the raw (unscaled) split format
    assert native.lib().hvr_im2col_stem(native._ptr(img), native._ptr(cols_s), 2, 61, 95, 192, native.HVR_F16S, native._stream()) == 0
    assert torch.equal(cols_s, native.cast(cols32, SPLIT, scale=1.0))
    assert torch.equal(native.im2col_stem(img, SPLIT)[0], native.cast(cols32, SPLIT))   # the host wrapper: activation scale applied
    cols_h, _, _ = native.im2col_stem(img, torch.float16)
    assert torch.equal(cols_h, cols32.half())


@pytest.mark.parametrize('dtype', [torch.float16, SPLIT], ids=['f16', 'f16x2'])
@pytest.mark.parametrize('B,H,W,Cin,Cout,k,stride,pad,dil', [(15, 38, 63, 512, 512, 3, 1, 2, 2), (8, 38, 63, 1024, 512, 3, 1, 1, 1),
                                                              (13, 37, 61, 256, 256, 3, 1, 1, 1), (15, 38, 63, 2048, 512, 1, 1, 0, 1)])
def test_big_tile_kernel_on_half_operands_equals_the_tile_engine(B, H, W, Cin, Cout, k, stride, pad, dil, dtype):
    """bigtile.hip (288 x 256 tiles) instantiated on half and on split-half operands: the same MFMA sequence per output element as
    the tile engine (split half: B_hi x A_hi, B_lo x A_hi, B_hi x A_lo per K-step, six phases from one LDS image), so the outputs
    are bit-identical -- tile=17 forces the kernel, tile=11 the engine's 144 x 256 shape.  The residual epilogue (merged f32 residual,
    one rounding into the [hi | lo] pair) as well."""
    x = _to(_rand((B, H, W, Cin), 81), dtype)
    w = _tow(_rand((Cout, k, k, Cin), 82, 0.03), dtype)
    bias = _rand((Cout,), 83).to(DEV)
    big = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=17)
    eng = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=11)
    assert big.dtype == dtype and torch.equal(big, eng)
    tiles = ((B * H * W + 287) // 288) * (Cout // 256)     # (stride 1, same-size outputs)
    assert native.conv2d_path(B, H, W, Cin, Cout, resid=False, k=k, stride=stride, pad=pad, dil=dil, dtype=dtype) == (3 if tiles >= 170 else 0)
    r = _to(_rand((B, H, W, Cout), 84), dtype)
    assert torch.equal(native.conv2d_nhwc(x, w, bias, r, relu=True, stride=stride, pad=pad, dil=dil, tile=17),
                       native.conv2d_nhwc(x, w, bias, r, relu=True, stride=stride, pad=pad, dil=dil, tile=11))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, SPLIT], ids=['bf16', 'f16', 'f16x2'])
@pytest.mark.parametrize('B,H,W,Cin,Cout,k,pad,dil,res', [
    (60, 38, 63, 256, 256, 3, 1, 1, False),    # layer 3's 3x3 on four clips: 499 tiles = two rounds of 256 persistent workgroups
    (60, 38, 63, 1024, 256, 1, 0, 1, False),   # its reducing 1x1: 16 K-steps
    (30, 38, 63, 256, 1024, 1, 0, 1, True),    # its expand conv + residual: 1 000 tiles = four rounds, 4 K-steps per tile
    (33, 38, 63, 320, 256, 1, 0, 1, True),     # an ODD number of K-steps (5): the LDS stage a tile starts in alternates
    (31, 37, 61, 256, 512, 3, 2, 2, False),    # dilation 2, odd extents, a ragged last tile in the second round
])
def test_big_tile_persistent_rounds_equal_the_tile_engine(B, H, W, Cin, Cout, k, pad, dil, res, dtype):
    """bigtile.hip's PERSISTENT form (round 6): more tiles than the 256 workgroups of a launch -- a workgroup's second tile has its
    loader state built under the first one's epilogue, its first K-step fetched under the first one's last K-step, and the first one's
    stores drain under its loop.  Bit for bit the tile engine's result (tile=11), like the one-tile-per-workgroup launches; odd K-step
    counts (the starting LDS stage flips per tile), residual epilogues and split-half operands included."""
    if dtype == SPLIT and Cin % 32:
        pytest.skip('split-half rows are whole 32-element groups')
    x = _to(_rand((B, H, W, Cin), 181), dtype)
    w = _tow(_rand((Cout, k, k, Cin), 182, 0.03), dtype)
    bias = _rand((Cout,), 183).to(DEV)
    r = _to(_rand((B, H, W, Cout), 184), dtype) if res else None
    tiles = ((B * H * W + 287) // 288) * (Cout // 256)
    assert tiles > 256
    big = native.conv2d_nhwc(x, w, bias, r, relu=True, pad=pad, dil=dil, tile=17)
    eng = native.conv2d_nhwc(x, w, bias, r, relu=True, pad=pad, dil=dil, tile=11)
    assert big.dtype == dtype and torch.equal(big, eng)
    again = native.conv2d_nhwc(x, w, bias, r, relu=True, pad=pad, dil=dil, tile=17)
    assert torch.equal(big, again)


def test_big_tile_kernel_on_split_half_tracks_the_f64_product():
    """The split-half big tiles against a float64 convolution of the un-rounded f32 inputs (res5's 3x3, dilation 2, ragged last
    row tile): 22-bit operands, exact products, f32 sums -- 5e-6 of the output scale, like the tile engine's split K-step."""
    B, H, W, Cin, Cout = 2, 38, 63, 256, 256
    xf, wf, bf = _rand((B, H, W, Cin), 85), _rand((Cout, 3, 3, Cin), 86, 0.02), _rand((Cout,), 87)
    got = _back(native.conv2d_nhwc(_to(xf, SPLIT), _tow(wf, SPLIT), bf.to(DEV), relu=True, pad=2, dil=2, tile=17))
    want = F.relu(F.conv2d(xf.permute(0, 3, 1, 2).double(), wf.permute(0, 3, 1, 2).double(), bf.double(), padding=2, dilation=2)).permute(0, 2, 3, 1)
    assert (got.double() - want).abs().max().item() < 5e-6 * want.abs().max().item()


@pytest.mark.parametrize('dtype', [SPLIT, torch.float32])
def test_two_level_accumulation_of_a_long_k_conv(dtype):
    """tile_hint 18 (round 6, gemm_params.h: EPI_LINEAR2): the RPN's 3x3 conv (K = 9 x 1024) with every 8 K-steps summed in a block
    accumulator that then joins the running total.  Against the same conv on one accumulator: the same products, so within the f32 noise
    of a 9 216-term sum of each other; against a float64 convolution of the operands the device was handed: the two-level sum's rms error
    is smaller (what the mode is for: the conv's noise reaches the final boxes through the proposal coordinates); bias and ReLU behave;
    a format or a K it does not take (bf16; K not a multiple of 256) runs as with hint 0, bit for bit."""
    B, H, W, Cin, Cout = 2, 38, 63, 1024, 512
    xf, wf, bf = _rand((B, H, W, Cin), 91).abs(), _rand((Cout, 3, 3, Cin), 92, 0.02), _rand((Cout,), 93)
    x, w = _to(xf, dtype), _tow(wf, dtype)
    one = _back(native.conv2d_nhwc(x, w, bf.to(DEV), relu=True, pad=1, tile=1))          # the same 128 x 128 shape, one accumulator
    two = _back(native.conv2d_nhwc(x, w, bf.to(DEV), relu=True, pad=1, tile=native.TWO_LEVEL_HINT))
    x64, w64 = _back(x).double(), _backw(w).double()
    want = F.relu(F.conv2d(x64.permute(0, 3, 1, 2), w64.permute(0, 3, 1, 2), bf.double(), padding=1)).permute(0, 2, 3, 1)
    scale = want.abs().max().item()
    assert (two - one).abs().max().item() < 2e-5 * scale and not torch.equal(two, one)
    e1, e2 = (one.double() - want).pow(2).mean().sqrt().item(), (two.double() - want).pow(2).mean().sqrt().item()
    assert e2 < 0.8 * e1, (e1, e2)
    assert (two.double() - want).abs().max().item() < 5e-6 * scale
    # formats / shapes outside the mode: as hint 0
    xb, wb = _to(xf[..., :64], torch.bfloat16), _tow(wf[..., :64], torch.bfloat16)
    assert torch.equal(native.conv2d_nhwc(xb, wb, bf.to(DEV), relu=True, pad=1, tile=native.TWO_LEVEL_HINT), native.conv2d_nhwc(xb, wb, bf.to(DEV), relu=True, pad=1))
    xs, ws = _to(xf[..., :96], dtype), _tow(wf[..., :96], dtype)                          # K = 9 x 96 = 27 K-steps: not a multiple of 8
    assert torch.equal(_back(native.conv2d_nhwc(xs, ws, bf.to(DEV), relu=True, pad=1, tile=native.TWO_LEVEL_HINT)), _back(native.conv2d_nhwc(xs, ws, bf.to(DEV), relu=True, pad=1)))


# ---- the dedicated bf16 kernels instantiated on half operands: each against the tile engine on the same operands ----
H16 = torch.float16


@pytest.mark.parametrize('Cin,Cout,H,W,B,with_res', [(64, 256, 19, 23, 2, True), (128, 512, 13, 31, 3, True), (256, 1024, 38, 63, 2, True),
                                                     (512, 2048, 11, 13, 3, True), (64, 128, 12, 11, 1, False)])
def test_expand_panel_kernel_on_half_operands(Cin, Cout, H, W, B, with_res):
    """expand.hip on half operands (tile hint 13) against the tile engine (hint 1): same products, f32 sums in a different order,
    one half rounding -- and bit-exact one-hot rows (transposition detector)."""
    x, w = _to(_rand((B, H, W, Cin), 91), H16), _tow(_rand((Cout, 1, 1, Cin), 92, 0.05), H16)
    bias = _rand((Cout,), 93).to(DEV)
    r = _to(_rand((B, H, W, Cout), 94), H16) if with_res else None
    forced = native.conv2d_nhwc(x, w, bias, r, relu=True, tile=13)
    engine = native.conv2d_nhwc(x, w, bias, r, relu=True, tile=1)
    auto = native.conv2d_nhwc(x, w, bias, r, relu=True)
    assert forced.dtype == H16
    torch.testing.assert_close(forced.float(), engine.float(), rtol=2 ** -10, atol=2 ** -10)
    torch.testing.assert_close(auto.float(), engine.float(), rtol=2 ** -10, atol=2 ** -10)
    eye = torch.zeros((1, 1, 128, Cin), dtype=H16)
    eye[0, 0, torch.arange(128), torch.arange(128) % Cin] = 1
    got = native.conv2d_nhwc(eye.to(DEV), w, None, None, relu=False, tile=13).view(128, Cout)
    assert torch.equal(got, w.view(Cout, Cin).t()[(torch.arange(128) % Cin).to(DEV)])


@pytest.mark.parametrize('Cin,Cout,H,W,B,with_res', [(64, 256, 19, 23, 2, True), (128, 512, 13, 31, 3, True), (64, 256, 152, 252, 1, True),
                                                     (128, 512, 76, 126, 2, True), (64, 128, 12, 11, 1, False), (128, 256, 9, 17, 1, False)])
def test_expand_panel_kernel_on_split_half_operands(Cin, Cout, H, W, B, with_res):
    """expand_split.hip (the row-panel kernel on split-half operands: both X planes in registers, three MFMAs per fragment pair from one
    LDS image of W; tile hint 13, and the automatic choice for the residual convs of layers 1-2) against the tile engine (hint 1):
    the same products, f32 sums in another order, one rounding into the [hi | lo] pair -- 3e-6 of the output scale; against float64
    on the operands as the kernel sees them; and one-hot rows (transposition detector, ragged last panel)."""
    x, w = _to(_rand((B, H, W, Cin), 91), SPLIT), _tow(_rand((Cout, 1, 1, Cin), 92, 0.05), SPLIT)
    bias = _rand((Cout,), 93).to(DEV)
    r = _to(_rand((B, H, W, Cout), 94), SPLIT) if with_res else None
    assert native.conv2d_path(B, H, W, Cin, Cout, dtype=SPLIT, resid=with_res, tile=13) == 1
    assert native.conv2d_path(B, H, W, Cin, Cout, dtype=SPLIT, resid=with_res) == (1 if with_res else 0)
    forced = native.conv2d_nhwc(x, w, bias, r, relu=True, tile=13)
    engine = native.conv2d_nhwc(x, w, bias, r, relu=True, tile=1)
    auto = native.conv2d_nhwc(x, w, bias, r, relu=True)
    assert forced.dtype == SPLIT
    ref = _back(x).view(-1, Cin).double() @ _backw(w).view(Cout, Cin).double().t() + bias.cpu().double()
    if with_res:
        ref = ref + _back(r).view(-1, Cout).double()
    ref = torch.relu(ref)
    scale = ref.abs().max().item()
    for got in (forced, auto):
        assert (_back(got).view(-1, Cout).double() - ref).abs().max().item() < 3e-6 * scale
        assert (_back(got).double() - _back(engine).double()).abs().max().item() < 3e-6 * scale
    M = 128 + 37                                        # a ragged second panel
    eye = torch.zeros((1, 1, M, Cin))
    eye[0, 0, torch.arange(M), torch.arange(M) % Cin] = 1
    got = _back(native.conv2d_nhwc(_to(eye, SPLIT), w, None, None, relu=False, tile=13)).view(M, Cout)
    # (not bit for bit: a lo half below 2^-14 is a half subnormal -- absolute quantum 2^-24 on the x 16 stored value -- so a small weight
    # comes back to within 2^-29 of itself; a transposed or shifted fragment is off by the weights' own size)
    assert (got - _backw(w).view(Cout, Cin).t()[torch.arange(M) % Cin]).abs().max().item() < 1e-8


def test_split_half_tail_with_the_next_conv():
    """hvr_bottleneck_tail_next on split-half operands (expand_split.hip, NX > 0: stage 1 of the R-101, identity form -- the block's
    input, or its projection computed by a separate conv, comes in as the residual): y equals the separate closing conv bit for bit
    (the same kernel body), and hn -- the next block's conv1 on y's packed [hi | lo] registers -- tracks float64 on the y it wrote."""
    B, OH, OW, C1, Cout, Cn = 2, 38, 63, 64, 256, 64
    h, res = _to(_rand((B, OH, OW, C1), 106), SPLIT), _to(_rand((B, OH, OW, Cout), 107), SPLIT)
    w3, b3 = _tow(_rand((Cout, C1), 108, 0.1), SPLIT), _rand((Cout,), 110, 0.1).to(DEV)
    wn, bn = _tow(_rand((Cn, Cout), 111, 0.05), SPLIT), _rand((Cn,), 112, 0.1).to(DEV)
    assert native.bottleneck_tail_next_supported(h, None, res, w3, b3, 1, wn, bn)
    y, hn = native.bottleneck_tail_next(h, None, res, w3, b3, wn, bn)
    y_sep = native.conv2d_nhwc(h, w3.view(Cout, 1, 1, C1), b3, res, relu=True, tile=13)
    assert y.dtype == SPLIT and hn.dtype == SPLIT and tuple(hn.shape) == (B, OH, OW, Cn)
    assert torch.equal(y, y_sep)
    y_ref = torch.relu(_back(h).view(-1, C1).double() @ _backw(w3).double().t() + b3.cpu().double() + _back(res).view(-1, Cout).double())
    assert (_back(y).view(-1, Cout).double() - y_ref).abs().max().item() < 3e-6 * y_ref.abs().max().item()
    hn_ref = torch.relu(_back(y).view(-1, Cout).double() @ _backw(wn).double().t() + bn.cpu().double())
    assert (_back(hn).view(-1, Cn).double() - hn_ref).abs().max().item() < 3e-6 * hn_ref.abs().max().item()
    hn_sep = native.conv2d_nhwc(y, wn.view(Cn, 1, 1, Cout), bn, None, relu=True)
    assert (_back(hn).double() - _back(hn_sep).double()).abs().max().item() < 3e-6 * hn_ref.abs().max().item()
    # layer 2's shapes have no fused form in this format: the query says so and the host layer runs the convs one by one
    h2, r2 = _to(_rand((1, 19, 31, 128), 113), SPLIT), _to(_rand((1, 19, 31, 512), 114), SPLIT)
    assert not native.bottleneck_tail_next_supported(h2, None, r2, _tow(_rand((512, 128), 115, 0.1), SPLIT), _rand((512,), 116).to(DEV), 1,
                                                     _tow(_rand((128, 512), 117, 0.05), SPLIT), _rand((128,), 118).to(DEV))


def test_layer1_3x3_stem_and_tails_on_half_operands():
    """conv3x3.hip (bit-identical to the tile engine), the fused stem (against the patch-matrix route in half), and the fused
    Bottleneck tails (hvr_bottleneck_tail / _tail_next) against the separate convs, all on half operands."""
    x, w = _to(_rand((2, 37, 53, 64), 101), H16), _tow(_rand((64, 3, 3, 64), 102, 0.05), H16)
    bias = _rand((64,), 103).to(DEV)
    assert native.conv2d_path(2, 37, 53, 64, 64, k=3, pad=1, resid=False, dtype=H16) == 2
    assert torch.equal(native.conv2d_nhwc(x, w, bias, None, relu=True, pad=1), native.conv2d_nhwc(x, w, bias, None, relu=True, pad=1, tile=1))
    # stem
    img = _rand((2, 3, 64, 96), 104, 50.0).to(DEV)
    w7 = _rand((64, 3, 7, 7), 105, 0.05).half()
    wf = torch.zeros((64, 7, 8, 4))
    wf[:, :, :7, :3] = w7.float().permute(0, 2, 3, 1)
    wp = torch.zeros((64, 192))
    wp[:, :147] = w7.float().permute(0, 2, 3, 1).reshape(64, 147)
    y = native.stem_fused(img, wf.view(64, 7, 32).half().to(DEV), bias)
    cols, OH, OW = native.im2col_stem(img, H16)
    ref = native.maxpool3x3s2_nhwc(native.gemm(cols, wp.half().to(DEV), bias, relu=True).view(2, OH, OW, 64))
    assert y.dtype == H16 and y.shape == ref.shape
    torch.testing.assert_close(y.float(), ref.float(), rtol=2 ** -9, atol=2 ** -9 * float(ref.float().abs().max()))
    # tails: layer1.0 (projection block) with and without the next block's conv1
    B, OH, OW, C1, C2, Cout, Cn = 2, 38, 63, 64, 64, 256, 64
    h, xin = _to(_rand((B, OH, OW, C1), 106), H16), _to(_rand((B, OH, OW, C2), 107), H16)
    w3, wd = _rand((Cout, C1), 108, 0.1), _rand((Cout, C2), 109, 0.1)
    wt = _to(torch.cat([w3, wd], 1), H16)
    bt = _rand((Cout,), 110, 0.1).to(DEV)
    assert native.bottleneck_tail_supported(h, xin, wt, bt, 1)
    out = native.bottleneck_tail(h, xin, wt, bt, stride2=1, relu=True)
    ref = torch.relu(_back(h).view(-1, C1).double() @ _back(wt)[:, :C1].double().t() + _back(xin).view(-1, C2).double() @ _back(wt)[:, C1:].double().t()
                     + bt.cpu().double())
    assert (_back(out).view(-1, Cout).double() - ref).abs().max().item() < 2 ** -10 * ref.abs().max().item()
    wn, bn = _to(_rand((Cn, Cout), 111, 0.05), H16), _rand((Cn,), 112, 0.1).to(DEV)
    assert native.bottleneck_tail_next_supported(h, xin, None, wt, bt, 1, wn, bn)
    y2, hn = native.bottleneck_tail_next(h, xin, None, wt, bt, wn, bn, stride2=1)
    assert torch.equal(y2, out)
    hn_ref = torch.relu(_back(out).view(-1, Cout).double() @ _back(wn).double().t() + bn.cpu().double())
    assert (_back(hn).view(-1, Cn).double() - hn_ref).abs().max().item() < 2 ** -10 * hn_ref.abs().max().item()


@pytest.mark.parametrize('M,N,K,tile', [(300, 256, 1280, 14), (4500, 1024, 1024, 15), (145, 136, 128, 14)])
def test_producer_consumer_kernel_on_half_operands_equals_the_tile_engine(M, N, K, tile):
    a, w = _to(_rand((M, K), 121), H16), _to(_rand((N, K), 122, 0.1), H16)
    bias, r = _rand((N,), 123).to(DEV), _to(_rand((M, N), 124), H16)
    assert torch.equal(native.gemm(a, w, bias, r, relu=True, tile=tile), native.gemm(a, w, bias, r, relu=True, tile=1))


def test_relation_window_size_big_tile_path_on_half_operands():
    """The one-round 352 x 256 scores kernel (relation_bt.hip) + the pipelined apply pass on half operands at window size."""
    Mq = Mk = 4500
    D = 1024
    q, k, v = _rand((Mq, D), 131, 1.5).half(), _rand((Mk, D), 132, 1.5).half(), _rand((Mk, D), 133).half()
    k[Mk - 2] = (q[7].float() * 3).half()
    out = native.relation_fwd(q.to(DEV), k.to(DEV), v.to(DEV), 1.0 / 32)
    rows = torch.cat([torch.arange(0, 16), torch.arange(340, 370), torch.arange(Mq - 40, Mq)])
    ref = torch.softmax((q[rows].double() @ k.double().t()) / 32, dim=1) @ v.double()
    assert (out[rows.to(DEV)].float().cpu().double() - ref).abs().max().item() < 2e-3 * float(v.float().abs().max())
    ones = native.relation_fwd(q.to(DEV), k.to(DEV), torch.ones_like(v).to(DEV), 1.0 / 32)
    torch.testing.assert_close(ones.float(), torch.ones_like(ones.float()), rtol=0, atol=2e-3)


def test_fused_stem_on_split_half_operands():
    """The fused 7x7/2 conv + ReLU + 3x3/2 max-pool on split-half operands (three MFMAs per product, f32 pooling, the pooled
    pixel written as one [64 hi | 64 lo] group) against float64 conv / pool of the f32 inputs: f32-grade; and against the
    patch-matrix route in the same format."""
    img = _rand((2, 3, 75, 101), 141, 50.0)
    w7 = _rand((64, 3, 7, 7), 142, 0.05)
    bias = _rand((64,), 143)
    wf = torch.zeros((64, 7, 8, 4))
    wf[:, :, :7, :3] = w7.permute(0, 2, 3, 1)
    y = native.stem_fused(img.to(DEV), native.stem_split_weights(wf.view(64, 7, 32).to(DEV)), native.stem_split_bias(bias.to(DEV)))
    ref = F.max_pool2d(torch.relu(F.conv2d(img.double(), w7.double(), bias.double(), stride=2, padding=3)), 3, 2, 1)
    assert y.dtype == SPLIT and tuple(y.shape) == (2, ref.shape[2], ref.shape[3], 64)
    got = _back(y).permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() < 3e-6 * ref.abs().max().item()
    wp = torch.zeros((64, 192))
    wp[:, :147] = w7.permute(0, 2, 3, 1).reshape(64, 147)
    cols, OH, OW = native.im2col_stem(img.to(DEV), SPLIT)
    alt = native.maxpool3x3s2_nhwc(native.gemm(cols, _tow(wp, SPLIT), bias.to(DEV), relu=True, out_f32=True).view(2, OH, OW, 64)).cpu()
    assert (_back(y) - alt).abs().max().item() < 3e-6 * alt.abs().max().item()


@pytest.mark.parametrize('tile', [0, 8, 11, 12])
def test_split_half_product_is_the_same_with_a_second_launch_keeping_the_lds_busy(tile):
    """Two graph replays in flight on two streams, each an LDS-heavy kernel (the padded transpose) in front of a window-sized
    split-half product: every output equals the product computed alone, bit for bit.  Round 3's pipelined split K loop renamed
    its spare B_hi fragment set across iterations; the compiler resolved the rename with register copies at the loop header, in
    front of the lgkmcnt wait, and a co-resident workgroup of the OTHER launch delayed the fragment reads past those copies --
    garbage tiles only with two windows in flight (hvrnet_amd/csrc/check_asm_waits.py now scans every build for the pattern)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    Mq, Mk, D = 4500, 4500, 1024
    ldp = native.relation_ldp(Mk)
    P = native.cast(torch.rand((Mq, ldp), device=DEV, generator=g), SPLIT)
    V = native.cast(torch.randn((Mk, D), device=DEV, generator=g), SPLIT)
    fn = lambda: native.gemm(P, native.transpose_pad(V, ldp), alpha=1.0, tile=tile)   # noqa: E731
    ref = fn().clone()
    torch.cuda.synchronize()
    graphs = []
    for _ in range(2):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                out = fn()
        torch.cuda.current_stream().wait_stream(st)
        graphs.append((gr, out))
    lanes = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(8):
        for (gr, _), lane in zip(graphs, lanes):
            with torch.cuda.stream(lane):
                gr.replay()
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(out, ref) else 1 for _, out in graphs)
    assert bad == 0, '%d of 16 concurrent replays differ from the product computed alone' % bad
