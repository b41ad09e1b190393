"""Kernel-level numerics on the GPU: every HIP kernel against a plain PyTorch fp32 (CPU)
statement of the same op on the same seeded inputs.  (The reference-anchored parity tests
that go through the oracle live in test_parity_gpu.py.)

Tolerances: bf16 operands are rounded before both sides see them, so the only error left is
f32 accumulation order (+ bf16 rounding of stored outputs, 2^-9 relative).
"""
import math
import os

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hvrnet_amd import native  # noqa: E402

DEV = 'cuda:0'


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _tol(dtype, out_dtype=None):
    out_dtype = out_dtype or dtype
    return dict(rtol=2e-2, atol=2e-2) if out_dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('staging', [0, 1])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K', [(128, 128, 128), (300, 256, 1024), (4500 // 9, 64, 192), (77, 36, 64), (257, 1024, 576)])
def test_gemm_matches_torch(M, N, K, dtype, staging):
    a, w = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, 0.1)
    bias = _rand((N,), torch.float32, 3)
    resid = _rand((M, N), dtype, 4)
    ref = torch.relu(a.float() @ w.float().t() + bias + resid.float())
    out = native.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV), relu=True, staging=staging)
    assert out.dtype == dtype
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))
    # transposition-detecting plain product, f32 output
    out2 = native.gemm(a.to(DEV), w.to(DEV), out_f32=True, staging=staging)
    assert out2.dtype == torch.float32
    torch.testing.assert_close(out2.cpu(), a.float() @ w.float().t(), rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('staging', [0, 1])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('cfg', [
    dict(Cin=64, Cout=64, k=3, stride=1, pad=1, dil=1, H=19, W=23),
    dict(Cin=128, Cout=128, k=3, stride=1, pad=2, dil=2, H=17, W=21),   # res5-style dilation
    dict(Cin=256, Cout=128, k=1, stride=2, pad=0, dil=1, H=20, W=31),   # caffe-style strided 1x1
    dict(Cin=64, Cout=256, k=1, stride=1, pad=0, dil=1, H=9, W=14),
])
def test_conv_matches_torch(cfg, dtype, staging):
    B = 2
    x = _rand((B, cfg['Cin'], cfg['H'], cfg['W']), dtype, 5)
    w = _rand((cfg['Cout'], cfg['Cin'], cfg['k'], cfg['k']), dtype, 6, 0.05)
    bias = _rand((cfg['Cout'],), torch.float32, 7)
    ref = F.conv2d(x.float(), w.float(), bias, stride=cfg['stride'], padding=cfg['pad'], dilation=cfg['dil'])
    resid = _rand(ref.shape, dtype, 8)
    ref = torch.relu(ref + resid.float())
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wn = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = native.conv2d_nhwc(xn, wn, bias.to(DEV), rn, relu=True, stride=cfg['stride'], pad=cfg['pad'], dil=cfg['dil'],
                           staging=staging)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, **_tol(dtype))


@pytest.mark.parametrize('relu,with_res', [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize('Cin,Cout,H,W,B', [(64, 256, 19, 23, 2), (128, 512, 13, 31, 3), (256, 1024, 38, 63, 2), (512, 2048, 11, 13, 3),
                                            (256, 1024, 8, 16, 1), (64, 128, 12, 11, 1)])
def test_expand_conv_panel_kernel(Cin, Cout, H, W, B, relu, with_res):
    """expand.hip (row-panel kernel for the Bottleneck's 1x1 expand + residual, resnet.py:248-264): forced through tile
    hint 13 and as the automatic choice; against the f32 statement of the op and against the tile engine (hint 1) on the
    same operands.  Shapes: every K the kernel instantiates (64 / 128 / 256) with 2, 4 and 8 chunks per workgroup, a
    ragged last row panel (M % 128 != 0: rows past M are clamped, not predicated), exactly one panel (M = 128), the
    narrowest output it accepts (N = 2 chunks), no residual / no ReLU; K = 512 (res5) runs the 8-wave, one-row-fragment
    form of the kernel."""
    x = _rand((B, H, W, Cin), torch.bfloat16, 61)
    w = _rand((Cout, 1, 1, Cin), torch.bfloat16, 62, 0.05)
    bias = _rand((Cout,), torch.float32, 63)
    resid = _rand((B, H, W, Cout), torch.bfloat16, 64) if with_res else None
    ref = x.float().view(-1, Cin) @ w.float().view(Cout, Cin).t() + bias
    if with_res:
        ref = ref + resid.float().view(-1, Cout)
    if relu:
        ref = torch.relu(ref)
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV) if with_res else None
    forced = native.conv2d_nhwc(xd, wd, bd, rd, relu=relu, tile=13)
    engine = native.conv2d_nhwc(xd, wd, bd, rd, relu=relu, tile=1)
    auto = native.conv2d_nhwc(xd, wd, bd, rd, relu=relu)
    assert forced.dtype == torch.bfloat16 and forced.shape == (B, H, W, Cout)
    torch.testing.assert_close(forced.float().cpu().view(-1, Cout), ref, **_tol(torch.bfloat16))
    # same products, same f32 accumulation up to order, one bf16 rounding: the two kernels agree to an output ulp
    torch.testing.assert_close(forced.float(), engine.float(), rtol=2 ** -7, atol=2 ** -7)
    torch.testing.assert_close(auto.float().cpu().view(-1, Cout), ref, **_tol(torch.bfloat16))
    # transposition / permutation detector: one-hot input rows pick single weight columns exactly
    eye = torch.zeros((1, 1, 128, Cin), dtype=torch.bfloat16)
    eye[0, 0, torch.arange(128), torch.arange(128) % Cin] = 1
    got = native.conv2d_nhwc(eye.to(DEV), wd, None, None, relu=False, tile=13).view(128, Cout).cpu()
    assert torch.equal(got, w.view(Cout, Cin).t()[torch.arange(128) % Cin])


@pytest.mark.parametrize('Cin,Cout,k,stride,pad,dil,with_res', [(256, 256, 3, 1, 1, 1, False), (1024, 256, 1, 1, 0, 1, False),
                                                                (512, 512, 3, 1, 2, 2, False), (512, 256, 1, 2, 0, 1, False),
                                                                (1024, 512, 3, 1, 1, 1, False), (2048, 512, 1, 1, 0, 1, True),
                                                                (128, 128, 3, 2, 1, 1, False)])
def test_split_k_conv_for_one_frame(Cin, Cout, k, stride, pad, dil, with_res):
    """hvr_conv2d_nhwc with a split-K workspace (one 600x1000 frame's stride-16 maps: 38x63 = 2 394 output pixels, 76 tiles of
    128 x 64): the K loop -- filter taps included, a slice starts inside the tap sequence -- is cut into grid.y slices of f32
    partial tiles and one reduce launch applies bias / residual / ReLU.  Against the f32 statement of the op and against the
    unsplit tile engine on the same operands (a tile hint keeps the split off): same products, f32 sums in a different order,
    one bf16 rounding."""
    B, H, W = 1, (38 if stride == 1 else 76), (63 if stride == 1 else 126)
    x = _rand((B, Cin, H, W), torch.bfloat16, 91)
    w = _rand((Cout, Cin, k, k), torch.bfloat16, 92, 0.03)
    bias = _rand((Cout,), torch.float32, 93)
    ref = F.conv2d(x.float(), w.float(), bias, stride=stride, padding=pad, dilation=dil)
    resid = _rand(ref.shape, torch.bfloat16, 94) if with_res else None
    if with_res:
        ref = ref + resid.float()
    ref = torch.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wn = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV) if with_res else None
    d = native.ConvDesc(x=xn.data_ptr(), w=wn.data_ptr(), y=xn.data_ptr(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=stride,
                        pad=pad, dil=dil, bias=None, resid=None, relu=1, out_f32=0, dtype=native.HVR_BF16, staging=1, tile_hint=0,
                        zero=native.zero_page(xn.device).data_ptr())
    assert native.lib().hvr_conv2d_splitk_workspace_bytes(d) > 0, 'this shape is meant to take the split-K route'
    with native.fewrow_split(True):
        split = native.conv2d_nhwc(xn, wn, bias.to(DEV), rn, relu=True, stride=stride, pad=pad, dil=dil)
    whole = native.conv2d_nhwc(xn, wn, bias.to(DEV), rn, relu=True, stride=stride, pad=pad, dil=dil)
    assert not torch.equal(split, whole) or Cin * k * k < 1024, 'the split route was not taken'
    torch.testing.assert_close(split.float().cpu().permute(0, 3, 1, 2), ref, **_tol(torch.bfloat16))
    torch.testing.assert_close(split.float(), whole.float(), rtol=2 ** -7, atol=2 ** -6)
    # a 15-frame batch of the same layer has tiles enough: not split
    d.B = 15
    assert native.lib().hvr_conv2d_splitk_workspace_bytes(d) == 0


def big_rows(B, H, W, k, stride, pad, dil):
    return B * ((H + 2 * pad - dil * (k - 1) - 1) // stride + 1) * ((W + 2 * pad - dil * (k - 1) - 1) // stride + 1)


@pytest.mark.parametrize('B,H,W,Cin,Cout,k,stride,pad,dil', [(15, 38, 63, 256, 256, 3, 1, 1, 1), (15, 38, 63, 1024, 256, 1, 1, 0, 1),
                                                              (15, 38, 63, 512, 512, 3, 1, 2, 2), (8, 38, 63, 1024, 512, 3, 1, 1, 1),
                                                              (13, 37, 61, 256, 256, 3, 1, 1, 1), (15, 76, 126, 512, 256, 1, 2, 0, 1),
                                                              (21, 38, 63, 2048, 512, 1, 1, 0, 1)])
def test_big_tile_kernel_equals_the_tile_engine(B, H, W, Cin, Cout, k, stride, pad, dil):
    """bigtile.hip (288 x 256 tiles, phase-staggered loop, buffer-addressed gather with hardware zero fill for out-of-image
    taps) against F.conv2d and, BIT for bit, against the tile engine: same MFMA sequence per output element.  tile=17 forces the
    kernel whatever its grid (the library's own choice takes it when the 288 x 256 grid covers most of the chip: hvr_conv2d_path
    reports 3 then)."""
    x = _rand((B, H, W, Cin), torch.bfloat16, 71).to(DEV)
    w = _rand((Cout, k, k, Cin), torch.bfloat16, 72, 0.03).to(DEV)
    bias = _rand((Cout,), torch.float32, 73).to(DEV)
    tiles = ((big_rows(B, H, W, k, stride, pad, dil) + 287) // 288) * (Cout // 256)
    assert native.conv2d_path(B, H, W, Cin, Cout, resid=False, k=k, stride=stride, pad=pad, dil=dil) == (3 if tiles >= 170 else 0)
    big = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=17)
    eng = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=11)   # 144 x 256, 8 waves
    assert torch.equal(big, eng)
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, stride=stride, padding=pad, dilation=dil))
    torch.testing.assert_close(big.float().permute(0, 3, 1, 2), ref, **_tol(torch.bfloat16))
    with native.throughput_mode(True):   # (the hint changes which kernels run, never the result)
        assert torch.equal(native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil), big)
    # no ReLU / no bias, and the plain-GEMM entry point
    if k == 1 and stride == 1:
        a = x.view(-1, Cin)
        g1 = native.gemm(a, w.view(Cout, Cin), None, relu=False, tile=17)
        g0 = native.gemm(a, w.view(Cout, Cin), None, relu=False, tile=11)
        assert torch.equal(g1, g0)


@pytest.mark.parametrize('B,H,W,Cin,Cout,k,stride,pad,dil', [(2, 38, 63, 64, 256, 5, 1, 2, 1),     # 25 taps
                                                              (2, 38, 63, 64, 256, 7, 1, 3, 1),     # 49 taps: more than a 32-bit tap mask holds
                                                              (1, 23, 31, 128, 256, 3, 1, 6, 6),    # dilation 6: whole tap rows / columns off the image
                                                              (2, 41, 59, 128, 256, 3, 2, 1, 1),    # stride 2, odd extents
                                                              (1, 9, 11, 256, 256, 3, 1, 1, 1),     # fewer output pixels than one tile has rows
                                                              (3, 38, 63, 64, 512, 6, 2, 2, 1)])    # 36 taps, even filter
def test_conv_loader_tap_masks(B, H, W, Cin, Cout, k, stride, pad, dil):
    """The pipelined K-step of the tile engine and the big tiles address a conv's A operand through per-piece out-of-image BIT MASKS
    (bit t = filter tap t of this output pixel lies outside the image, built once per tile) and advance the tap by scalar selects
    (gemm_tile.h, bigtile.hip).  Shapes that stress that form -- many taps (filters with more than 32 taps must be kept off the masked
    loaders by the dispatcher and still come out right under every hint), dilations that push whole tap rows off the image, stride 2,
    a map smaller than a tile -- against F.conv2d, and bit for bit against the double-buffered base shape (tile=1: the K order per output
    element is the same in every shape)."""
    x = _rand((B, H, W, Cin), torch.bfloat16, 171).to(DEV)
    w = _rand((Cout, k, k, Cin), torch.bfloat16, 172, 0.02).to(DEV)
    bias = _rand((Cout,), torch.float32, 173).to(DEV)
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, stride=stride, padding=pad, dilation=dil))
    base = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=1)
    torch.testing.assert_close(base.float().permute(0, 3, 1, 2), ref, **_tol(torch.bfloat16))
    for hint in (0, 7, 9, 11, 12):
        got = native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=hint)
        assert torch.equal(got, base), 'tile hint %d' % hint
    if k * k <= 32:
        assert torch.equal(native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=17), base)
    else:   # the big tiles refuse what their loader cannot address instead of computing something else
        with pytest.raises(native.HvrError):
            native.conv2d_nhwc(x, w, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=17)
    xs, ws = native.cast(x.float(), native.SPLIT), native.as_operand(w.float(), native.SPLIT)
    sb = native.conv2d_nhwc(xs, ws, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=1)
    for hint in (0, 11, 12):
        assert torch.equal(native.conv2d_nhwc(xs, ws, bias, relu=True, stride=stride, pad=pad, dil=dil, tile=hint), sb), 'split half, tile hint %d' % hint
    torch.testing.assert_close(native.cast(sb, torch.float32).permute(0, 3, 1, 2), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('M,N,K,with_res', [(300, 1024, 12544, False), (300, 1024, 1024, True), (37, 256, 4096, False)])
def test_few_row_gemm_split_over_k(M, N, K, with_res):
    """hvr_gemm with the few-row scratch (one frame's 300 proposals through fc_new_1: 48 tiles walking 196 K-steps): K slices +
    one reduce launch with the epilogue; against the f32 statement and the unsplit engine."""
    a = _rand((M, K), torch.bfloat16, 95)
    w = _rand((N, K), torch.bfloat16, 96, 0.02)
    bias = _rand((N,), torch.float32, 97)
    resid = _rand((M, N), torch.bfloat16, 98) if with_res else None
    ref = a.float() @ w.float().t() + bias
    if with_res:
        ref = ref + resid.float()
    ref = torch.relu(ref)
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV) if with_res else None
    d = native.GemmDesc(A=ad.data_ptr(), B=wd.data_ptr(), C=ad.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=None, resid=None, ldr=0,
                        relu=1, out_f32=0, dtype=native.HVR_BF16, staging=1, tile_hint=0)
    assert native.lib().hvr_gemm_fewrow_workspace_bytes(d) > 0
    with native.fewrow_split(True):
        split = native.gemm(ad, wd, bd, rd, relu=True)
    whole = native.gemm(ad, wd, bd, rd, relu=True)
    torch.testing.assert_close(split.float().cpu(), ref, **_tol(torch.bfloat16))
    torch.testing.assert_close(split.float(), whole.float(), rtol=2 ** -7, atol=2 ** -6)
    d.M, d.N = 4500, 1024   # the head's full-window products: tiles enough
    assert native.lib().hvr_gemm_fewrow_workspace_bytes(d) == 0


@pytest.mark.parametrize('relu,with_bias', [(True, True), (False, False)])
@pytest.mark.parametrize('B,H,W', [(2, 37, 53), (1, 32, 32), (3, 16, 80), (1, 152, 252)])
def test_conv3x3_c64_persistent_kernel(B, H, W, relu, with_bias):
    """conv3x3.hip (layer 1's conv2: 3x3, 64 -> 64, LDS-resident weights, 16 x 16 output tiles from an 18 x 18 halo): against
    F.conv2d and, bit for bit, against the tile engine (same MFMA sequence per output element: tap-major, two 32-channel
    halves per tap).  Shapes: ragged tiles on both axes, exactly one tile per workgroup, a tile row cut by the image's
    bottom edge, several frames (the halo must not leak across frame boundaries), the path's own 152 x 252 map."""
    x = _rand((B, 64, H, W), torch.bfloat16, 71)
    w = _rand((64, 64, 3, 3), torch.bfloat16, 72, 0.05)
    bias = _rand((64,), torch.float32, 73) if with_bias else None
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    if relu:
        ref = torch.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wn = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = bias.to(DEV) if with_bias else None
    assert native.conv2d_path(B, H, W, 64, 64, k=3, pad=1, resid=False, bias=with_bias) == 2
    got = native.conv2d_nhwc(xn, wn, bd, None, relu=relu, pad=1)
    engine = native.conv2d_nhwc(xn, wn, bd, None, relu=relu, pad=1, tile=1)
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), ref, **_tol(torch.bfloat16))
    assert torch.equal(got, engine)
    # a one-hot input pixel reproduces the (flipped) filter around it exactly: transposition / tap-order detector
    one = torch.zeros((1, 40, 40, 64), dtype=torch.bfloat16)
    one[0, 17, 23, 5] = 1
    out = native.conv2d_nhwc(one.to(DEV), wn, None, None, relu=False, pad=1).cpu()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            assert torch.equal(out[0, 17 + dy, 23 + dx], w[:, 5, 1 - dy, 1 - dx])
    assert float(out.float().abs().sum()) == float(w[:, 5].float().abs().sum())


def _relation_ref(q, k, v, scale):
    p = torch.softmax(scale * (q.double() @ k.double().t()), dim=1)
    return (p @ v.double()).float()


@pytest.mark.parametrize('staging', [0, 1])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('Mq,Mk', [(96, 96), (300, 700), (37, 129)])
def test_relation_matches_torch(Mq, Mk, dtype, staging):
    D = 1024
    q, k, v = _rand((Mq, D), dtype, 11, 1.5), _rand((Mk, D), dtype, 12, 1.5), _rand((Mk, D), dtype, 13)
    ref = _relation_ref(q, k, v, 1.0 / 32)
    out = native.relation_fwd(q.to(DEV), k.to(DEV), v.to(DEV), 1.0 / 32, staging=staging)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_relation_peaky_rows_and_tile_max_jumps(dtype):
    """Forces the cross-tile rescale path: one key far above the rest, placed in a late 128-key tile."""
    Mq, Mk, D = 64, 520, 1024
    q, k, v = _rand((Mq, D), dtype, 21), _rand((Mk, D), dtype, 22), _rand((Mk, D), dtype, 23)
    k[400] = (q[5].float() * 3).to(dtype)     # logit ~ 3*|q|^2/32 ~ 96 above the rest for row 5
    k[3] = (q[9].float() * 2).to(dtype)       # early-tile spike for row 9
    ref = _relation_ref(q, k, v, 1.0 / 32)
    out = native.relation_fwd(q.to(DEV), k.to(DEV), v.to(DEV), 1.0 / 32)
    assert torch.isfinite(out.float()).all()
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize('Mq,Mk', [(4500, 4500), (3300, 4417), (4321, 4100)])
def test_relation_window_size_big_tile_path(Mq, Mk):
    """Window-sized bf16 problems take the one-round 352 x 256 scores kernel (relation_bt.hip, V^T written by the same
    launch): ragged last row / key tiles, a spike in the last 128-key block, and the same answer as the tile-engine path
    gives on a row subset (computed as its own small problem)."""
    D = 1024
    q, k, v = _rand((Mq, D), torch.bfloat16, 41, 1.5), _rand((Mk, D), torch.bfloat16, 42, 1.5), _rand((Mk, D), torch.bfloat16, 43)
    k[Mk - 2] = (q[7].float() * 3).to(torch.bfloat16)        # a late-block maximum ~ 96 above the rest of row 7
    k[5] = (q[Mq - 1].float() * 2).to(torch.bfloat16)        # an early-block maximum for the last row
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out = native.relation_fwd(qd, kd, vd, 1.0 / 32)
    assert torch.isfinite(out.float()).all()
    rows = torch.cat([torch.arange(0, 16), torch.arange(340, 370), torch.arange(Mq - 40, Mq)])  # tile seams + the ragged tail
    ref = _relation_ref(q[rows], k, v, 1.0 / 32)
    torch.testing.assert_close(out[rows.to(DEV)].float().cpu(), ref, **_tol(torch.bfloat16))
    small = native.relation_fwd(qd[rows.to(DEV)].contiguous(), kd, vd, 1.0 / 32)  # 86 rows: the tile-engine scores pass
    torch.testing.assert_close(out[rows.to(DEV)].float(), small.float(), rtol=2e-2, atol=2e-2)
    # every output row is a convex combination of V rows
    ones = native.relation_fwd(qd, kd, torch.ones_like(vd), 1.0 / 32)
    torch.testing.assert_close(ones.float(), torch.ones_like(ones.float()), rtol=0, atol=8e-3)


@pytest.mark.parametrize('G,Mq,Mk,dtype', [(4, 4500, 4500, torch.bfloat16), (2, 4500, 4500, torch.bfloat16), (3, 4321, 4100, torch.bfloat16),
                                           (4, 4500, 4500, torch.float16), (2, 300, 4500, torch.bfloat16), (3, 96, 200, torch.float32),
                                           (4, 300, 4500, torch.bfloat16), (3, 300, 4321, torch.float16),   # the key stage over the clips of a call: the tile engine's batch dimension
                                           (2, 5400, 5400, torch.bfloat16)])   # (352 tiles per group: the one-group rule sends it to the tile engine -> `exact` runs single calls)
def test_relation_grouped_equals_the_single_calls(G, Mq, Mk, dtype):
    """hvr_relation_fwd_grouped: G independent problems of one shape (the windows a batched head has in flight) in one call --
    persistent 352 x 256 score tiles over all groups (relation_bt.hip) and, from three window-sized groups on, the 288 x 256 apply
    launch (relation_apply_bt.hip) with the block weights applied on the exponent fields of P~ (bf16) or multiplied on as halves (f16).
      exact=True : every group's rows are hvr_relation_fwd's bit for bit;
      default    : the same up to the association of the f32 sums (one bf16 output ulp), and against the f64 softmax on sampled rows;
    spikes force block-weight shifts in an early and in the last 128-key block; shapes the grouped kernels do not take (key stage,
    small problems, f32) run as G single calls and are equal trivially -- the plumbing of strides and workspaces is what they test."""
    D = 1024
    q, k, v = _rand((G * Mq, D), dtype, 51, 1.5), _rand((G * Mk, D), dtype, 52, 1.5), _rand((G * Mk, D), dtype, 53)
    for g in range(G):
        k[g * Mk + Mk - 2] = (q[g * Mq + 7].float() * 3).to(dtype)           # a late-block maximum ~ 96 above the rest of row 7
        k[g * Mk + 5] = (q[g * Mq + Mq - 1].float() * 2).to(dtype)          # an early-block maximum for the last row
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else _tol(dtype)
    exact = native.relation_fwd_grouped(qd, kd, vd, 1.0 / 32, G, exact=True)
    fast = native.relation_fwd_grouped(qd, kd, vd, 1.0 / 32, G)
    again = native.relation_fwd_grouped(qd, kd, vd, 1.0 / 32, G)
    assert torch.isfinite(fast.float()).all() and torch.equal(fast, again)
    rows = torch.cat([torch.arange(0, 16), torch.arange(min(280, Mq - 1), min(300, Mq)), torch.arange(max(Mq - 40, 0), Mq)]).unique()
    for g in range(G):
        single = native.relation_fwd(qd[g * Mq:(g + 1) * Mq], kd[g * Mk:(g + 1) * Mk], vd[g * Mk:(g + 1) * Mk], 1.0 / 32)
        assert torch.equal(exact[g * Mq:(g + 1) * Mq], single), 'group %d: exact form differs from hvr_relation_fwd' % g
        torch.testing.assert_close(fast[g * Mq:(g + 1) * Mq].float(), single.float(), **tol)
        ref = _relation_ref(q[g * Mq + rows], k[g * Mk:(g + 1) * Mk], v[g * Mk:(g + 1) * Mk], 1.0 / 32)
        torch.testing.assert_close(fast[g * Mq + rows.to(DEV)].float().cpu(), ref, **tol)
    # strided operands: q / k as the two column halves of one projection output, as the heads pass them
    if Mq == Mk:
        qk = torch.cat([qd, kd], dim=1)
        assert torch.equal(native.relation_fwd_grouped(qk[:, :D], qk[:, D:], vd, 1.0 / 32, G), fast)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Mq,Mk', [(96, 96), (200, 333), (333, 130)])
def test_relation_backward_matches_autograd(Mq, Mk, dtype):
    """ops.relation's HIP backward (score pass + fused softmax backward + four tile-engine GEMMs) against torch autograd
    of the f64 statement of the same function, on the operands as stored (bf16-rounded inputs for the bf16 case)."""
    from hvrnet_amd import ops
    D = 1024
    q, k, v = _rand((Mq, D), dtype, 51, 1.2), _rand((Mk, D), dtype, 52, 1.2), _rand((Mk, D), dtype, 53)
    go = _rand((Mq, D), dtype, 54)
    k[Mk - 2] = (q[3].float() * 2).to(dtype)                  # one peaky row
    qr, kr, vr = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = torch.softmax((qr @ kr.t()) / 32, dim=1) @ vr
    ref.backward(go.double())
    qd, kd, vd = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    out = ops.relation(qd, kd, vd, 1.0 / 32)
    out.backward(go.to(DEV))
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(out.detach().float().cpu(), ref.detach().float(), **_tol(dtype))
    for name, got, want in (('dq', qd.grad, qr.grad), ('dk', kd.grad, kr.grad), ('dv', vd.grad, vr.grad)):
        assert got.shape == want.shape and got.dtype == dtype, name
        scale = want.abs().max().item()
        err = (got.float().cpu().double() - want).abs().max().item()
        assert err <= tol['atol'] + tol['rtol'] * scale, '%s: max err %g (scale %g)' % (name, err, scale)


def test_relation_backward_window_size_bf16():
    """Window size (4 500 x 4 500): dV / dQ / dK of the HIP backward against a row / column sample of the f64 statement."""
    from hvrnet_amd import ops
    M, D = 4500, 1024
    q, k, v = _rand((M, D), torch.bfloat16, 61, 1.2), _rand((M, D), torch.bfloat16, 62, 1.2), _rand((M, D), torch.bfloat16, 63)
    go = _rand((M, D), torch.bfloat16, 64)
    qd, kd, vd = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    ops.relation(qd, kd, vd, 1.0 / 32).backward(go.to(DEV))
    # f64 reference on the device (the full 4 500 x 4 500 problem is cheap in f64 there)
    qr, kr, vr = [t.to(DEV).double().requires_grad_(True) for t in (q, k, v)]
    (torch.softmax((qr @ kr.t()) / 32, dim=1) @ vr).backward(go.to(DEV).double())
    for name, got, want in (('dq', qd.grad, qr.grad), ('dk', kd.grad, kr.grad), ('dv', vd.grad, vr.grad)):
        scale = want.abs().max().item()
        err = (got.double() - want).abs().max().item()
        assert torch.isfinite(got.float()).all() and err <= 3e-2 * scale + 1e-3, '%s: max err %g (scale %g)' % (name, err, scale)


@pytest.mark.parametrize('cfg', [dict(k=3, s=1, p=1, d=1), dict(k=3, s=1, p=2, d=2), dict(k=1, s=1, p=0, d=1), dict(k=1, s=2, p=0, d=1)])
def test_conv_bn_backward_matches_autograd(cfg):
    """train_ops.ConvFunction (conv + frozen BN + residual + ReLU; dX by the conv kernel on rotated weights or a GEMM,
    dW by im2col + GEMM, both through the BN scale) against torch autograd of F.conv2d on the same f32 operands."""
    from hvrnet_amd import train_ops as TO
    B, Cin, Cout, H, W = 2, 64, 96, 13, 18
    k, st, p, d = cfg['k'], cfg['s'], cfg['p'], cfg['d']
    x = _rand((B, Cin, H, W), torch.float32, 71)
    w = _rand((Cout, Cin, k, k), torch.float32, 72, 0.05)
    sc, sh = torch.rand(Cout, generator=torch.Generator().manual_seed(73)) + 0.5, _rand((Cout,), torch.float32, 74, 0.1)
    OH, OW = (H - 1) // st + 1, (W - 1) // st + 1
    res, go = _rand((B, Cout, OH, OW), torch.float32, 75), _rand((B, Cout, OH, OW), torch.float32, 76)
    xr, wr, rr = [t.clone().requires_grad_(True) for t in (x, w, res)]
    ref = torch.relu(F.conv2d(xr, wr, None, stride=st, padding=p, dilation=d) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + rr)
    ref.backward(go)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    rd = res.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    y = TO.ConvFunction.apply(xd, wd, sc.to(DEV), sh.to(DEV), rd, True, st, p, d)
    y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV))
    torch.testing.assert_close(y.detach().cpu().permute(0, 3, 1, 2), ref.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(rd.grad.cpu().permute(0, 3, 1, 2), rr.grad, rtol=0, atol=0)


def test_maxpool_and_stem_patches():
    x = _rand((2, 64, 21, 30), torch.float32, 31)
    ref = F.max_pool2d(x, 3, 2, 1)
    y = native.maxpool3x3s2_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    torch.testing.assert_close(y.cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=0)
    img = _rand((2, 3, 37, 45), torch.float32, 32)
    w = _rand((64, 3, 7, 7), torch.float32, 33, 0.1)
    ref = F.conv2d(img, w, None, stride=2, padding=3)
    cols, OH, OW = native.im2col_stem(img.to(DEV), torch.float32)
    wp = torch.zeros(64, 192)
    wp[:, :147] = w.permute(0, 2, 3, 1).reshape(64, 147)
    out = native.gemm(cols, wp.to(DEV)).cpu().view(2, OH, OW, 64).permute(0, 3, 1, 2)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_layout_and_cast_round_trip():
    x = _rand((2, 24, 5, 7), torch.float32, 41).to(DEV)
    n = native.nchw_to_nhwc(x)
    assert torch.equal(n.cpu(), x.cpu().permute(0, 2, 3, 1))
    assert torch.equal(native.nhwc_to_nchw(n).cpu(), x.cpu())
    b = native.cast(x, torch.bfloat16)
    assert torch.equal(b.cpu(), x.cpu().to(torch.bfloat16))
    assert torch.equal(native.cast(b, torch.float32).cpu(), x.cpu().to(torch.bfloat16).float())


def _greedy_nms(dets, thr, ge=True, score_order=False):
    import numpy as np
    d = dets.numpy().astype(np.float32)
    n = len(d)
    order = sorted(range(n), key=lambda i: (-float(d[i, 4]), i))
    one = np.float32(1)
    area = (d[:, 2] - d[:, 0] + one) * (d[:, 3] - d[:, 1] + one)
    dead = np.zeros(n, bool)
    for a, i in enumerate(order):
        if dead[i]:
            continue
        rest = np.array(order[a + 1:], dtype=np.int64)
        if rest.size == 0:
            break
        w = np.maximum(np.float32(0), np.minimum(d[i, 2], d[rest, 2]) - np.maximum(d[i, 0], d[rest, 0]) + one)
        h = np.maximum(np.float32(0), np.minimum(d[i, 3], d[rest, 3]) - np.maximum(d[i, 1], d[rest, 1]) + one)
        inter = w * h
        ovr = inter / (area[i] + area[rest] - inter)
        hit = (ovr >= np.float32(thr)) if ge else (ovr > np.float32(thr))
        dead[rest[hit]] = True
    if score_order:
        return [i for i in order if not dead[i]]
    return [i for i in range(n) if not dead[i]]


def _boxes(n, seed, span=200.0):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand((n, 2), generator=g) * span
    wh = torch.rand((n, 2), generator=g) * 60 + 4
    sc = torch.rand((n, 1), generator=g)
    return torch.cat([xy, xy + wh, sc], 1)


@pytest.mark.parametrize('n,thr', [(1, 0.5), (7, 0.7), (64, 0.3), (65, 0.7), (300, 0.3), (700, 0.5)])
def test_nms_matches_greedy(n, thr):
    dets = _boxes(n, 50 + n)
    keep = native.nms(dets.to(DEV), thr).cpu().tolist()
    assert keep == _greedy_nms(dets, thr)


@pytest.mark.parametrize('n,thr,max_keep', [(1, 0.5, 1), (65, 0.7, 3), (700, 0.5, 50), (700, 0.3, 1024), (6000, 0.7, 300),
                                            (6000, 0.3, 300), (8192, 0.7, 1000)])
def test_nms_first_keeps_the_first_survivors(n, thr, max_keep):
    """hvr_nms_first == nms(dets, thr)[:max_keep] of rpn_head.py:95-97 (the survivors in score order, cut, reported as
    ascending input indices): the greedy kernel that only prices the surviving rows finds the same set as the full sweep."""
    dets = _boxes(n, 90 + n, span=200.0 if n < 2000 else 900.0)
    want = sorted(_greedy_nms(dets, thr, score_order=True)[:max_keep])
    assert native.nms_first(dets.to(DEV), thr, max_keep).cpu().tolist() == want


@pytest.mark.parametrize('ge', [True, False])
@pytest.mark.parametrize('thr', [0.5, 1.0 / 3.0, 2.0 / 3.0, 0.6, 0.25])
def test_nms_first_on_boxes_whose_ious_sit_on_the_threshold(thr, ge):
    """Small integer boxes: many pairs have an IoU of exactly 1/2, 1/3, 2/3, 3/5 ... -- the real quotient equals (or is within
    an ulp of) the threshold, where the kernel's division-free comparison has to fall back to the rounded division to agree
    with nms_cpu.cpp:46-54 (`>=`) / nms_kernel.cu (`>`) bit for bit."""
    g = torch.Generator().manual_seed(int(thr * 1000) + ge)
    n = 1500
    xy = torch.randint(0, 24, (n, 2), generator=g).float()
    wh = torch.randint(1, 12, (n, 2), generator=g).float()
    dets = torch.cat([xy, xy + wh - 1, torch.rand((n, 1), generator=g)], 1)
    for max_keep in (40, 1024):
        want = sorted(_greedy_nms(dets, thr, ge=ge, score_order=True)[:max_keep])
        assert native.nms_first(dets.to(DEV), thr, max_keep, ge_semantics=ge).cpu().tolist() == want


def test_nms_docstring_case_and_empty():
    dets = torch.tensor([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9], [49.2, 31.8, 51.0, 35.4, 0.5],
                         [35.1, 11.5, 39.1, 15.7, 0.5], [35.6, 11.8, 39.3, 14.2, 0.5], [35.3, 11.5, 39.9, 14.5, 0.4],
                         [35.2, 11.7, 39.7, 15.7, 0.3]])
    assert native.nms(dets.to(DEV), 0.7).cpu().tolist() == [0, 3, 4]
    assert native.nms(torch.zeros((0, 5), device=DEV), 0.5).numel() == 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('R,C', [(300, 1024), (129, 64), (37, 72), (4500, 1024)])
def test_transpose_pad_is_exact(R, C, dtype):
    """V -> V^T with the key axis zero-padded to a multiple of 128 (the apply pass multiplies the pad by P~ = 0)."""
    x = _rand((R, C), dtype, 81)
    ldt = (R + 127) // 128 * 128
    out = native.transpose_pad(x.to(DEV), ldt).cpu()
    assert out.shape == (C, ldt)
    assert torch.equal(out[:, :R], x.t())
    assert not out[:, R:].any()


@pytest.mark.parametrize('tile', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize('M,N,K', [(300, 256, 128), (1000, 512, 320), (145, 36, 64), (700, 264, 192), (513, 128, 1280), (2394, 256, 64)])
def test_every_tile_shape_gives_the_same_gemm(M, N, K, tile):
    """The tile menu (128x128, 128x64, 144x256, 144x128, 256x128 and their 3 / 4-stage pipelined variants) is a
    speed choice only.  K = 64 ... 1280 covers 1, 2, 3, 5 and 20 K-steps: shorter than, equal to and longer than
    the LDS ring of the pipelined variants."""
    dtype = torch.bfloat16
    a, w = _rand((M, K), dtype, 61), _rand((N, K), dtype, 62, 0.1)
    bias, resid = _rand((N,), torch.float32, 63), _rand((M, N), dtype, 64)
    ref = torch.relu(a.float() @ w.float().t() + bias + resid.float())
    out = native.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV), relu=True, staging=1, tile=tile)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))
    out32 = native.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), out_f32=True, staging=1, tile=tile)
    torch.testing.assert_close(out32.cpu(), a.float() @ w.float().t() + bias, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('tile', [1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_every_tile_shape_gives_the_same_conv(tile):
    dtype = torch.bfloat16
    x = _rand((2, 128, 17, 21), dtype, 71)
    w = _rand((256, 128, 3, 3), dtype, 72, 0.05)
    bias = _rand((256,), torch.float32, 73)
    ref = torch.relu(F.conv2d(x.float(), w.float(), bias, padding=2, dilation=2))
    y = native.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), bias.to(DEV),
                           relu=True, pad=2, dil=2, staging=1, tile=tile)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, **_tol(dtype))


@pytest.mark.parametrize('H,W', [(37, 45), (64, 96), (608, 1008)])
def test_fused_stem_matches_conv_relu_pool(H, W):
    """conv7x7/2 + bias + ReLU + maxpool3x3/2 in one kernel == the three torch ops on bf16-rounded operands."""
    img = _rand((2, 3, H, W), torch.float32, 81, 50.0)
    w = _rand((64, 3, 7, 7), torch.float32, 82, 0.05).to(torch.bfloat16)
    bias = _rand((64,), torch.float32, 83)
    ref = F.max_pool2d(torch.relu(F.conv2d(img.to(torch.bfloat16).float(), w.float(), bias, stride=2, padding=3)), 3, 2, 1)
    wf = torch.zeros((64, 7, 8, 4))
    wf[:, :, :7, :3] = w.float().permute(0, 2, 3, 1)
    y = native.stem_fused(img.to(DEV), wf.view(64, 7, 32).to(torch.bfloat16).to(DEV), bias.to(DEV))
    assert y.shape == (2, ref.shape[2], ref.shape[3], 64)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K', [(256, 1024, 7232), (128, 1152, 28736), (64, 512, 7232), (1024, 256, 7232), (36, 1024, 960),
                                   (512, 4608, 7232), (128, 128, 2048 + 64)])
def test_gemm_splitk_matches_the_single_pass_product(dtype, M, N, K):
    """hvr_gemm_splitk (K slices in one launch + ordered reduce; the weight-gradient product) against hvr_gemm with f32 output
    on the same operands: same products, different summation order -> 1e-5 of the output scale; shapes the library does not
    split (many tiles / short K) must fall through to the single pass bit for bit."""
    if dtype == torch.float32:
        K = K // 2 if K > 8000 else K          # keep the f32 case quick
        K = K // 32 * 32
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn((M, K), generator=g)).to(DEV).to(dtype)
    w = (torch.randn((N, K), generator=g)).to(DEV).to(dtype)
    want = native.gemm(a, w, out_f32=True)
    got = native.gemm_splitk(a, w)
    assert got.dtype == torch.float32 and got.shape == (M, N)
    nbytes = native.lib().hvr_gemm_splitk_workspace_bytes(M, N, K, native._dt(a))
    if nbytes == 0:
        assert torch.equal(got, want)
    else:
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-5 * scale + 1e-6
        ref = a.double() @ w.double().t()
        assert float((got.double() - ref).abs().max()) <= float((want.double() - ref).abs().max()) * 2 + 1e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('G,M,N,K', [(23, 256, 1024, 7232), (3, 512, 4608, 7232), (2, 64, 64, 2112), (5, 256, 2304, 1088)])
def test_gemm_splitk_batched_equals_the_single_products(G, M, N, K):
    """hvr_gemm_splitk_batched (round 6: the weight gradients of a stage's identical blocks in one launch through the tile engine's batch
    dimension) on slices of two slabs against hvr_gemm with f32 output problem by problem: same products, the K slices' f32 sums
    associate differently -> 1e-5 of the output scale (bit for bit where neither form slices K), no worse than the single pass against
    f64; a slab with spare capacity behind the used slices (the caller's slabs are sized for the largest count seen) is untouched."""
    g = torch.Generator().manual_seed(G + M + N + K)
    a = torch.randn((G + 1, M, K), generator=g).to(DEV).bfloat16()
    w = torch.randn((G + 1, N, K), generator=g).to(DEV).bfloat16()
    out = torch.full((G + 1, M, N), 7.0, dtype=torch.float32, device=DEV)
    got = native.gemm_splitk_batched(a[:G], w[:G], out=out[:G])
    assert got.data_ptr() == out.data_ptr() and bool((out[G] == 7.0).all())
    sliced = native.lib().hvr_gemm_splitk_batched_workspace_bytes(M, N, K, native.HVR_BF16, G) > 0
    for i in range(G):
        want = native.gemm(a[i], w[i], out_f32=True)
        scale = float(want.abs().max())
        if not sliced:
            assert torch.equal(got[i], want), i
        else:
            assert float((got[i] - want).abs().max()) <= 1e-5 * scale + 1e-6, i
            ref = a[i].double() @ w[i].double().t()
            assert float((got[i].double() - ref).abs().max()) <= float((want.double() - ref).abs().max()) * 2 + 1e-4 * scale


@pytest.mark.gpu
def test_unpack_conv_wgrads_multi_equals_the_per_layer_unpack():
    """hvr_unpack_conv_wgrads_multi: a table of layers' f32 products x scale -> parameter-layout gradients in one launch, added or written:
    bit-identical to hvr_unpack_conv_wgrad layer by layer."""
    for Cout, Cin, KH, KW, n in ((64, 32, 3, 3, 4), (16, 64, 1, 1, 3), (10, 6, 3, 3, 3)):   # (the last: Cin % 4 != 0 -> the element-wise path)
        _unpack_multi_case(Cout, Cin, KH, KW, n)


def _unpack_multi_case(Cout, Cin, KH, KW, n):
    g = torch.Generator().manual_seed(5)
    per = Cout * Cin * KH * KW
    dw = torch.randn((n, Cout, KH * KW * Cin), generator=g).to(DEV)
    scales = [torch.rand(Cout, generator=g).to(DEV) + 0.5 for _ in range(n)]
    for accumulate in (True, False):
        base = [torch.randn((Cout, Cin, KH, KW), generator=g).to(DEV) for _ in range(n)]
        want = [native.unpack_conv_wgrad(dw[i], scales[i], (Cout, Cin, KH, KW), accumulate_into=b.clone()) if accumulate
                else native.unpack_conv_wgrad(dw[i], scales[i], (Cout, Cin, KH, KW)) for i, b in enumerate(base)]
        outs = [b.clone() for b in base]
        items = [native.PackItem(w=outs[i].data_ptr(), scale=scales[i].data_ptr(), out=dw.data_ptr() + i * per * 4, first=i * per, Cout=Cout, Cin=Cin, KK=KH * KW)
                 for i in range(n)]
        native.unpack_conv_wgrads_multi(native.items_to_device(items, DEV), n, n * per, accumulate=accumulate)
        for i in range(n):
            assert torch.equal(outs[i], want[i]), (accumulate, i)


# ------------------------------------------------------------------------------- producer / consumer tile kernel
PC128, PC256 = 14, 15   # gemm_params.h: kPcHint128 / kPcHint256


@pytest.mark.parametrize('tile', [PC128, PC256])
@pytest.mark.parametrize('M,N,K', [(300, 256, 128), (1000, 512, 320), (700, 264, 192), (513, 128, 1280), (2394, 256, 128), (4500, 1024, 256)])
def test_producer_consumer_kernel_equals_the_tile_engine(M, N, K, tile):
    """pc_gemm.hip (4 compute + 4 DMA waves, 144 x 128 / 144 x 256 tiles): the MFMA sequence of an output element is the
    tile engine's (K-steps in order, two 32-wide halves each), so the results are bit-identical -- ragged M / N tiles,
    K loops shorter (2 K-steps) and longer (20) than the LDS ring, bias + residual + ReLU and the f32 output included."""
    dtype = torch.bfloat16
    a, w = _rand((M, K), dtype, 71), _rand((N, K), dtype, 72, 0.1)
    bias, resid = _rand((N,), torch.float32, 73), _rand((M, N), dtype, 74)
    ref = torch.relu(a.float() @ w.float().t() + bias + resid.float())
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV)
    out = native.gemm(ad, wd, bd, rd, relu=True, staging=1, tile=tile)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(dtype))
    assert torch.equal(out, native.gemm(ad, wd, bd, rd, relu=True, staging=1, tile=1))
    out32 = native.gemm(ad, wd, bd, out_f32=True, staging=1, tile=tile)
    assert torch.equal(out32, native.gemm(ad, wd, bd, out_f32=True, staging=1, tile=1))


def test_producer_consumer_kernel_rejects_what_it_cannot_run():
    a, w = _rand((145, 64), torch.bfloat16, 75).to(DEV), _rand((36, 64), torch.bfloat16, 76).to(DEV)
    with pytest.raises(native.HvrError):
        native.gemm(a, w, tile=PC128)          # N % 8 != 0 and a one-K-step loop: no silent fallback behind a forced shape
    out = native.gemm(a, w)                    # the default dispatch takes the tile engine
    torch.testing.assert_close(out.float().cpu(), a.float().cpu() @ w.float().cpu().t(), **_tol(torch.bfloat16))


def test_producer_consumer_apply_pass_follows_block_maximum_jumps():
    """Window-sized apply passes (Mq >= 1024) run on pc_gemm.hip (block weights from an LDS table written by the DMA waves, the
    previous block's fold riding between the next block's MFMAs): against an f64 statement of softmax(q k^T / 32) v, with a key
    whose score jumps the row maximum by 2^40 between blocks (the fold's weights must follow), and against the tile engine's apply
    pass, which the same rows take as a problem of their own (512 query rows)."""
    g = torch.Generator().manual_seed(5)
    M, D = 2200, 1024
    q = (torch.randn((M, D), generator=g) * 1.2).to(torch.bfloat16)
    k = (torch.randn((M, D), generator=g) * 1.2).to(torch.bfloat16)
    v = torch.randn((M, D), generator=g).to(torch.bfloat16)
    k[1500] = q[7] * 4.0        # a spike: row 7's maximum jumps far above the earlier blocks' at block 11
    o = native.relation_fwd(q.to(DEV), k.to(DEV), v.to(DEV), 1 / 32).float().cpu()
    ref = (torch.softmax((q.double() @ k.double().t()) / 32, 1) @ v.double()).float()
    assert (o - ref).abs().max().item() < 2e-2 * v.float().abs().max().item()
    small = native.relation_fwd(q[:512].contiguous().to(DEV), k.to(DEV), v.to(DEV), 1 / 32).float().cpu()
    # same block statistics and per-block products up to the scores kernel's shape; the weights' rounding path differs (table vs in-loop exp2)
    assert (o[:512] - small).abs().max().item() < 2e-2


# ------------------------------------------------------------------------------- Bottleneck tail (projection shortcut as a second K segment)
@pytest.mark.parametrize('C1,C2,Cout,stride,B,OH,OW', [(64, 64, 256, 1, 2, 38, 63), (128, 256, 512, 2, 3, 19, 32), (64, 64, 256, 1, 1, 152, 252),
                                                        (256, 512, 1024, 2, 2, 19, 32), (512, 1024, 2048, 1, 1, 38, 63)])
def test_bottleneck_tail_fuses_the_projection_shortcut(C1, C2, Cout, stride, B, OH, OW):
    """hvr_bottleneck_tail: relu(h W3^T + x_s Wd^T + bias) with x_s = the block input sampled at the downsample conv's stride
    (resnet.py:248-264, first block of a stage) against the same expression in f32 -- and against the two-conv path it
    replaces (downsample conv -> bf16 identity -> expand conv + residual), which differs only by the identity's rounding.
    Stages 1-2 run on the row-panel kernel (expand.hip), stage 3 (256 + 512, stride 2) and res5 (512 + 1024) on the tile engine,
    whose A operand switches from h to the sampled block input at K-step C1 / 64."""
    g = torch.Generator().manual_seed(C1 + Cout)
    H2, W2 = (OH - 1) * stride + 1 + (stride - 1), (OW - 1) * stride + 1   # a row / column beyond the last sampled pixel too
    h = (torch.randn((B, OH, OW, C1), generator=g)).to(torch.bfloat16)
    x = (torch.randn((B, H2, W2, C2), generator=g)).to(torch.bfloat16)
    w3 = (torch.randn((Cout, C1), generator=g) * 0.1).to(torch.bfloat16)
    wd = (torch.randn((Cout, C2), generator=g) * 0.1).to(torch.bfloat16)
    b3, bd = torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1
    xs = x[:, ::stride, ::stride][:, :OH, :OW]
    ref = torch.relu(h.float() @ w3.float().t() + xs.float() @ wd.float().t() + b3 + bd)
    w = torch.cat([w3, wd], 1).contiguous().to(DEV)
    hd, xd = h.to(DEV), x.to(DEV)
    bias = (b3 + bd).to(DEV)
    assert native.bottleneck_tail_supported(hd, xd, w, bias, stride)
    out = native.bottleneck_tail(hd, xd, w, bias, stride2=stride, relu=True)
    torch.testing.assert_close(out.float().cpu(), ref, **_tol(torch.bfloat16))
    ident = native.conv2d_nhwc(xd, wd.view(Cout, 1, 1, C2).to(DEV), bd.to(DEV), relu=False, stride=stride)
    two = native.conv2d_nhwc(hd, w3.view(Cout, 1, 1, C1).to(DEV), b3.to(DEV), resid=ident[:, :OH, :OW].contiguous(), relu=True)
    torch.testing.assert_close(out.float(), two.float(), rtol=2e-2, atol=6e-2)
    # writing into a caller-owned slice (frame groups)
    whole = torch.zeros((B + 1, OH, OW, Cout), dtype=torch.bfloat16, device=DEV)
    native.bottleneck_tail(hd, xd, w, bias, stride2=stride, relu=True, out=whole[1:])
    assert torch.equal(whole[1:], out) and not whole[0].any()


@pytest.mark.parametrize('C1,C2,Cout,Cn,stride,B,OH,OW', [
    (64, 64, 256, 64, 1, 2, 38, 63),      # layer1.0: projection block + layer1.1's conv1
    (64, 0, 256, 64, 1, 2, 38, 63),       # layer1.1: identity block + layer1.2's conv1
    (128, 256, 512, 128, 2, 3, 19, 32),   # layer2.0 (stride-2 shortcut) + layer2.1's conv1
    (128, 0, 512, 128, 1, 1, 76, 126),    # layer2.x at the bench frame size (a ragged last panel: 9576 = 74 * 128 + 104)
    (256, 0, 1024, 256, 1, 28, 38, 63),   # layer3.x (round 6; from 512 panels on): identity block + the next block's 1024 -> 256 conv1, 8 waves x 16 rows; 67 032 = 523 * 128 + 88
])
def test_bottleneck_tail_next_also_computes_the_next_conv1(C1, C2, Cout, Cn, stride, B, OH, OW):
    """hvr_bottleneck_tail_next: the block output y is BIT-identical to the tail / expand kernel's, and hn is the next block's
    conv1 + bn1 + ReLU (resnet.py:224-232) of that bf16 y -- against f32 arithmetic on the same bf16 operands, and against the
    separate conv launch it replaces."""
    g = torch.Generator().manual_seed(C1 + Cout + C2)
    h = torch.randn((B, OH, OW, C1), generator=g).to(torch.bfloat16).to(DEV)
    w3 = (torch.randn((Cout, C1), generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    b3 = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    wn = (torch.randn((Cn, Cout), generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bn = (torch.randn(Cn, generator=g) * 0.1).to(DEV)
    if C2:
        H2, W2 = (OH - 1) * stride + 1 + (stride - 1), (OW - 1) * stride + 1
        x = torch.randn((B, H2, W2, C2), generator=g).to(torch.bfloat16).to(DEV)
        wd = (torch.randn((Cout, C2), generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        w, resid = torch.cat([w3, wd], 1).contiguous(), None
        want_y = native.bottleneck_tail(h, x, w, b3, stride2=stride, relu=True)
    else:
        x, w = None, w3
        resid = torch.randn((B, OH, OW, Cout), generator=g).to(torch.bfloat16).to(DEV)
        want_y = native.conv2d_nhwc(h, w3.view(Cout, 1, 1, C1), b3, resid=resid, relu=True)
    assert native.bottleneck_tail_next_supported(h, x, resid, w, b3, stride, wn, bn)
    y, hn = native.bottleneck_tail_next(h, x, resid, w, b3, wn, bn, stride2=stride)
    assert torch.equal(y, want_y)
    ref = torch.relu(y.float().reshape(-1, Cout) @ wn.float().t() + bn).reshape(B, OH, OW, Cn)
    torch.testing.assert_close(hn.float(), ref, **_tol(torch.bfloat16))
    sep = native.conv2d_nhwc(y, wn.view(Cn, 1, 1, Cout), bn, relu=True)
    torch.testing.assert_close(hn.float(), sep.float(), rtol=2e-2, atol=3e-2)
    whole = torch.zeros((B + 1, OH, OW, Cout), dtype=torch.bfloat16, device=DEV)
    y2, hn2 = native.bottleneck_tail_next(h, x, resid, w, b3, wn, bn, stride2=stride, out=whole[1:])
    assert torch.equal(whole[1:], y) and torch.equal(hn2, hn) and not whole[0].any()


def test_bottleneck_tail_next_says_when_it_does_not_apply():
    bf = dict(device=DEV, dtype=torch.bfloat16)
    h, r = torch.zeros((1, 16, 16, 512), **bf), torch.zeros((1, 16, 16, 2048), **bf)
    w, b = torch.zeros((2048, 512), **bf), torch.zeros(2048, device=DEV)
    wn, bn = torch.zeros((512, 2048), **bf), torch.zeros(512, device=DEV)
    assert not native.bottleneck_tail_next_supported(h, None, r, w, b, 1, wn, bn)      # res5: no kernel (Cn = 512 accumulators)
    h3, r3 = torch.zeros((1, 16, 16, 256), **bf), torch.zeros((1, 16, 16, 1024), **bf)
    w3, b3 = torch.zeros((1024, 256), **bf), torch.zeros(1024, device=DEV)
    wn3, bn3 = torch.zeros((256, 1024), **bf), torch.zeros(256, device=DEV)
    
    h1, r1 = torch.zeros((1, 16, 16, 64), **bf), torch.zeros((1, 16, 16, 256), **bf)
    w1, b1 = torch.zeros((256, 64), **bf), torch.zeros(256, device=DEV)
    wn1, bn1 = torch.zeros((64, 256), **bf), torch.zeros(64, device=DEV)
    assert native.bottleneck_tail_next_supported(h1, None, r1, w1, b1, 1, wn1, bn1)
    assert not native.bottleneck_tail_next_supported(h1, None, None, w1, b1, 1, wn1, bn1)  # neither shortcut input nor residual
    assert not native.bottleneck_tail_next_supported(h1.float(), None, r1.float(), w1.float(), b1, 1, wn1.float(), bn1)  # f32 parity mode


def test_bottleneck_tail_says_when_it_does_not_apply():
    h = torch.zeros((1, 8, 16, 64), device=DEV)                     # f32: the parity mode keeps the two-conv path
    x = torch.zeros((1, 8, 16, 64), device=DEV)
    w, b = torch.zeros((256, 128), device=DEV), torch.zeros(256, device=DEV)
    assert not native.bottleneck_tail_supported(h, x, w, b, 1)
    hb, xb = h.bfloat16(), torch.zeros((1, 8, 16, 96), device=DEV, dtype=torch.bfloat16)
    assert not native.bottleneck_tail_supported(hb, xb, torch.zeros((256, 160), device=DEV, dtype=torch.bfloat16), b, 1)  # 64 + 96: no whole K-step


# ------------------------------------------------------------------------------- training: one-pass K-contiguous operands of the weight gradient
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W,Cin,k,pad,dil', [(2, 19, 31, 64, 3, 1, 1), (3, 38, 63, 256, 3, 2, 2), (1, 9, 14, 128, 3, 1, 1), (2, 7, 9, 72, 5, 2, 1)])
def test_im2col_t_equals_im2col_then_transpose(B, H, W, Cin, k, pad, dil, dtype):
    """hvr_im2col_t writes the TRANSPOSED patch matrix of a stride-1 conv directly (the weight-gradient product's K-contiguous operand):
    bit for bit hvr_transpose_pad(hvr_im2col_nhwc(x)), ragged pixel counts, dilation and zero padding included."""
    x = _rand((B, H, W, Cin), dtype, 91).to(DEV)
    OH, OW = H + 2 * pad - dil * (k - 1), W + 2 * pad - dil * (k - 1)
    P = B * OH * OW
    ldt = (P + 63) // 64 * 64
    got = native.im2col_t(x, k, k, pad, dil, ldt)
    want = native.transpose_pad(native.im2col_nhwc(x, k, k, pad, dil), ldt)
    assert got.shape == want.shape == (k * k * Cin, ldt) and torch.equal(got, want)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('R,C', [(7182, 256), (300, 1024), (65, 72), (4500, 36 + 4)])
def test_relu_bwd_t_equals_relu_bwd_then_transpose(R, C, dtype):
    """hvr_relu_bwd_t: the ReLU mask and the transposed masked gradient from one read -- bit for bit hvr_relu_bwd and hvr_transpose_pad of
    its output (negative, zero and negative-zero activations gate the gradient off)."""
    dy, y = _rand((R, C), dtype, 92).to(DEV), _rand((R, C), dtype, 93).to(DEV)
    y[::7] = 0.0
    y[1::11] = -0.0
    ldt = (R + 63) // 64 * 64
    dz, dzt = native.relu_bwd_t(dy, y, ldt)
    want = native.relu_bwd(dy, y)
    assert torch.equal(dz, want) and torch.equal(dzt, native.transpose_pad(want, ldt))
    assert torch.equal(dz, torch.where(y.float() > 0, dy, torch.zeros_like(dy)))
