// extern "C" surface of libhvr_hip.so (declarations + contracts: include/hvr_hip.h).
// Argument validation lives here; kernels assume validated inputs.
#include "../../include/hvr_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.h"
#include "gemm_params.h"
#include "relation_bt.h"

namespace hvr {
hipError_t run_transpose_pad(const void*, void*, int, int, long, long, int, hipStream_t);
hipError_t run_zero_fill(void*, size_t, hipStream_t);
hipError_t run_pack_conv_weights_multi(const void*, int, long, int, hipStream_t);
hipError_t run_transpose_multi(const void*, int, int, hipStream_t);
hipError_t run_im2col_t(const void*, void*, int, int, int, int, int, int, int, int, int, int, long, hipStream_t);
hipError_t run_relu_bwd_t(const void*, const void*, void*, void*, int, int, long, hipStream_t);
hipError_t run_splitk_reduce(const float*, float*, int, int, long, int, hipStream_t);
hipError_t run_unpack_conv_wgrads_multi(const void*, int, long, int, hipStream_t);
hipError_t run_splitk_reduce_bf16(const float*, void*, int, int, long, int, hipStream_t);
hipError_t run_splitk_reduce_f16(const float*, void*, int, int, long, int, hipStream_t);
hipError_t run_splitk_reduce_half_batched(const float*, void*, int, int, long, int, int, int, long, long, hipStream_t);
hipError_t run_splitk_reduce_epi_bf16(const float*, void*, int, int, long, int, const float*, const void*, long, int, hipStream_t);
hipError_t run_relation_normalize(void*, const float*, const float*, int, int, long, int, float, hipStream_t);
hipError_t run_relu_bwd(const void*, const void*, void*, long, int, hipStream_t);
hipError_t run_im2col_nhwc(const void*, void*, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
hipError_t run_scale_rows(const void*, const float*, void*, int, long, int, hipStream_t);
hipError_t run_sgd_step(float*, const float*, float*, long, float, float, float, float, float, float*, int, hipStream_t);
hipError_t run_colsum(const void*, float*, int, int, long, int, float*, hipStream_t);
int colsum_slices(int M, int N);
size_t sgd_workspace_bytes();
hipError_t run_pack_conv_weight(const float*, const float*, void*, int, int, int, int, hipStream_t);
hipError_t run_unpack_conv_wgrad(const float*, const float*, float*, int, int, int, int, hipStream_t);
hipError_t run_det_loss(const float*, int, int, int, int, const long long*, const float*, const float*, const float*, int, float, float,
                        float, float*, float*, const int*, hipStream_t);
size_t assign_workspace_bytes(int n, int k);
hipError_t run_max_iou_assign(const float*, int, int, const float*, int, const uint8_t*, float, float, float, float, long long*, float*,
                              void*, hipStream_t);
hipError_t run_sample(const long long*, const float*, int, int, int, float, long long*, int*, hipStream_t);
hipError_t run_box_targets(const float*, int, int, const float*, const long long*, const long long*, const long long*, const int*, int,
                           const float*, const float*, float, int, long long*, float*, float*, float*, hipStream_t);
hipError_t run_rpn_loss(const float*, int, int, int, const long long*, const float*, const float*, const float*, const int*, float, float*,
                        float*, hipStream_t);
hipError_t run_ce_rows(const float*, int, int, int, const long long*, int, float*, hipStream_t);
hipError_t run_triplet_margin(const void*, long, const void*, long, int, int, int, const long long*, const long long*, const long long*, int,
                              float, int, float*, float*, float*, float*, hipStream_t);
hipError_t run_ingest(const uint8_t*, int, int, long, float*, int, int, int, int, const float*, const float*, int, hipStream_t);
hipError_t run_mining_argreduce(const float*, int, int, long, const long long*, const long long*, long long*, hipStream_t);
hipError_t run_relation_dscore(const void*, const void*, const void*, const void*, void*, int, long, int, long, long, float, int, hipStream_t);
hipError_t run_im2col_stem(const float*, void*, int, int, int, int, int, int, int, hipStream_t);
hipError_t run_maxpool3x3s2(const void*, void*, int, int, int, int, int, int, int, hipStream_t);
hipError_t run_cast(const void*, void*, long, int, int, float, hipStream_t);
hipError_t run_permute(const void*, void*, int, int, int, int, int, int, hipStream_t);
hipError_t run_det_decode(const float*, int, int, int, int, const float*, int, const float*, const float*, float, float,
                          float, float, float*, float*, hipStream_t);
hipError_t run_roi_align_fwd(const void*, const float*, void*, int, int, int, int, int, int, int, float, int, int, int,
                             hipStream_t);
hipError_t run_roi_align_bwd(const float*, const float*, float*, int, int, int, int, int, int, float, int, int, hipStream_t);
size_t nms_workspace_bytes(int P, int n);
hipError_t run_nms_batched(const float*, int, int, float, int, int, int, long long*, int*, void*, hipStream_t, int band = 0,
                           int* band_done = nullptr);
hipError_t run_rpn_select(const void*, const void*, long, long, float*, const RpnParams&, int, hipStream_t);
hipError_t run_rpn_select_wide(const void*, const void*, long, long, float*, const RpnParams&, void*, hipStream_t);
size_t rpn_wide_workspace_bytes(int T, long n_anchor);
int rpn_wide_max_frames(int set);
int* rpn_wide_done_flags(void* wws, int T);
hipError_t run_rpn_gather(const float*, const long long*, const int*, int, int, int, int, int, float*, int*, hipStream_t);
hipError_t run_multiclass_nms(const float*, const float*, int, int, float, float, int, float*, long long*, int*, void*,
                              hipStream_t);
size_t multiclass_nms_workspace_bytes(int R, int ncls);
hipError_t run_stem_fused(const float*, const void*, const float*, void*, int, int, int, int, hipStream_t);
}  // namespace hvr

using namespace hvr;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
static int check_launch(hipError_t e, const char* what) {
  if (e == hipSuccess) return HVR_OK;
  if (e == hipErrorInvalidValue) return fail(HVR_EUNSUPPORTED, "%s: shape outside the implemented envelope", what);
  return fail(HVR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static inline int elem_size(int dtype) { return (dtype == HVR_BF16 || dtype == HVR_F16) ? 2 : 4; }   // (split half: 4 bytes per logical element)
static inline int kstep_elems(int dtype) { return (dtype == HVR_F32 || dtype == HVR_F16S) ? 32 : 64; }
static inline bool valid_dtype(int dtype) { return dtype >= HVR_F32 && dtype <= HVR_F16S; }
static inline bool aligned128(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 127) == 0; }   // a split-half [32 hi | 32 lo] group
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int hvr_abi_version(void) { return 6; }  // 2: hvr_gemm_desc / hvr_conv_desc carry the few-row split-K scratch (ws, ws_bytes); 3: HVR_F16 / HVR_F16S; 4: hvr_tail_next_desc carries alpha / beta
const char* hvr_last_error(void) { return g_err.c_str(); }

static int fill_linear(GemmParams& p, const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb,
                       long ldc, int dtype, int staging) {
  std::memset(&p, 0, sizeof p);
  if (!valid_dtype(dtype)) return fail(HVR_EINVAL, "dtype %d is none of HVR_F32 / HVR_BF16 / HVR_F16 / HVR_F16S", dtype);
  const int bke = kstep_elems(dtype);
  if (!A || !B || !C) return fail(HVR_EINVAL, "null operand pointer");
  if (M <= 0 || N <= 0 || K <= 0) return fail(HVR_EINVAL, "empty problem M=%d N=%d K=%d", M, N, K);
  if (K % bke) return fail(HVR_EINVAL, "K=%d is not a multiple of the %d-element K-step", K, bke);
  if (N % 4) return fail(HVR_EINVAL, "N=%d is not a multiple of 4", N);
  if (!aligned16(A) || !aligned16(B) || !aligned16(C)) return fail(HVR_EINVAL, "operands must be 16-byte aligned");
  // split half: rows are whole [64 hi | 64 lo] groups and a matrix starts on a group boundary
  if (dtype == HVR_F16S && (lda % 32 || ldb % 32 || !aligned128(A) || !aligned128(B)))
    return fail(HVR_EINVAL, "split-half operands need 128-byte aligned bases and row pitches that are multiples of 32 elements");
  // the tile loaders keep 32-bit byte offsets into A and B
  const long es = elem_size(dtype);
  if ((long)M * lda * es >= (1L << 31) || (long)N * ldb * es >= (1L << 31))
    return fail(HVR_EUNSUPPORTED, "operand of 2 GiB or more (M=%d lda=%ld N=%d ldb=%ld)", M, lda, N, ldb);
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.dtype = dtype;
  p.staging = staging ? 1 : 0;
  return HVR_OK;
}

// K slices for a tile-engine product (linear layer or conv) whose 128 x 64 tile grid leaves most of the chip idle (0 / 1 = not split): the slices bring
// the launch to ~1.5 workgroups per CU, at least 4 K-steps each
static int fewrow_mode() {   // HVR_CONV_SPLITK: 0 = few-row forms off; 1 (default) = kpar.hip where it applies, else K slices + reduce; 2 = always the latter
  static const int on = std::getenv("HVR_CONV_SPLITK") ? std::atoi(std::getenv("HVR_CONV_SPLITK")) : 1;
  return on;
}
static int fewrow_slices(const GemmParams& p) {
  const int on = fewrow_mode();
  constexpr int target = 512, mink = 8, minper = 4;
  if (!on || p.dtype != DT_BF16 || !p.staging || p.out_f32 || p.tile_hint != 0) return 1;
  if (p.N % 8 || p.ldc % 8 || (p.resid && p.ldr % 8) || !aligned16(p.C) || (p.resid && !aligned16(p.resid)) || (p.bias && !aligned16(p.bias))) return 1;
  const long tiles = (long)((p.M + 127) / 128) * ((p.N + 63) / 64);
  const int ksteps = p.K / 64;
  if (tiles > 160 || ksteps < mink) return 1;
  long s = (target + tiles / 2) / tiles;
  if (s > ksteps / minper) s = ksteps / minper;
  return s < 2 ? 1 : (int)s;
}

// split-K form of a tile-engine product: f32 partial tiles [slice][M][N] from the 128 x 64 shape (three workgroups per CU),
// then one reduce + epilogue pass
static hipError_t run_fewrow_split(const GemmParams& p, int slices, void* ws, hipStream_t stream) {
  GemmParams q = p;
  const int ksteps = p.K / 64;
  q.ksplit_steps = (ksteps + slices - 1) / slices;
  q.ksplit_count = (ksteps + q.ksplit_steps - 1) / q.ksplit_steps;
  q.csplit_bytes = (long)p.M * p.N * 4;
  q.C = ws; q.ldc = p.N; q.out_f32 = 1;
  q.bias = nullptr; q.resid = nullptr; q.relu = 0;
  q.tile_hint = 2;
  hipError_t e = run_tile_op(q, EPI_LINEAR, stream);
  if (e == hipSuccess) e = run_splitk_reduce_epi_bf16((const float*)ws, p.C, p.M, p.N, p.ldc, q.ksplit_count, p.bias, p.resid, p.ldr, p.relu, stream);
  return e;
}

static int gemm_params(const hvr_gemm_desc* d, GemmParams& p) {
  if (!d) return fail(HVR_EINVAL, "null descriptor");
  int rc = fill_linear(p, d->A, d->B, d->C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->dtype, d->staging);
  if (rc) return rc;
  const int es = elem_size(d->dtype);
  if ((d->lda * es) % 16 || (d->ldb * es) % 16) return fail(HVR_EINVAL, "lda/ldb rows must be 16-byte multiples");
  if ((d->ldc * (d->out_f32 ? 4 : es)) % 8) return fail(HVR_EINVAL, "ldc rows must be 8-byte multiples");
  if (d->dtype == HVR_F16S) {
    if (!d->out_f32 && (d->N % 8 || d->ldc % 32 || !aligned128(d->C))) return fail(HVR_EINVAL, "split-half output: N %% 8 == 0, ldc %% 32 == 0, 128-byte aligned C");
    if (d->resid && (d->N % 8 || d->ldr % 32 || !aligned128(d->resid))) return fail(HVR_EINVAL, "split-half residual: N %% 8 == 0, ldr %% 32 == 0, 128-byte aligned");
  }
  p.bias = d->bias; p.resid = d->resid; p.ldr = d->ldr; p.relu = d->relu; p.out_f32 = d->out_f32;
  p.tile_hint = d->tile_hint;
  p.alpha = d->alpha; p.beta = d->beta;
  return 0;
}

size_t hvr_gemm_fewrow_workspace_bytes(const hvr_gemm_desc* d) {
  GemmParams p;
  if (gemm_params(d, p)) return 0;
  const int s = fewrow_slices(p);
  return s > 1 ? (size_t)s * p.M * p.N * 4 : 0;
}

int hvr_gemm(const hvr_gemm_desc* d, void* stream) {
  GemmParams p;
  const int rc = gemm_params(d, p);
  if (rc) return rc;
  int hint = d->tile_hint;
  if (hint == kBigForce) {
    if (!bigtile_supported(p, true)) return fail(HVR_EUNSUPPORTED, "the big-tile kernel takes bf16 products with N %% 256 == 0 and K %% 64 == 0");
    return check_launch(run_bigtile(p, (hipStream_t)stream), "hvr_gemm(big tile)");
  }
  if (hint == kBigHint || hint == 0) {  // the 288 x 256 shape where it applies (kBigHint: a throughput caller), else as with hint 0
    const bool thr = hint == kBigHint;
    p.tile_hint = hint = 0;
    if (bigtile_supported(p, thr)) return check_launch(run_bigtile(p, (hipStream_t)stream), "hvr_gemm(big tile)");
  }
  if (d->tile_hint == kPcHint128 || d->tile_hint == kPcHint256) {
    if (!pc_supported(p, EPI_LINEAR)) return fail(HVR_EUNSUPPORTED, "the producer / consumer tile kernel takes aligned bf16 operands with N %% 8 == 0");
    return check_launch(run_pc(p, EPI_LINEAR, d->tile_hint == kPcHint256 ? 256 : 128, (hipStream_t)stream), "hvr_gemm (pc)");
  }
  // Long-K products whose 144 x 128 tile grid is one round of the chip (fc_new_1: 4500 x 1024 x 12544 -> 256 tiles of 196
  // K-steps): the producer / consumer kernel, 1.00-1.03 PF/s against the tile engine's 0.90 (tools/pc_bench.py).  Shorter K
  // loops do not amortise its one-wave-per-SIMD compute stream's prologue; there the tile engine stays ahead.
  if (hint == 0 && d->K >= 8192 && pc_supported(p, EPI_LINEAR)) {
    const long tiles = (long)((d->M + 143) / 144) * ((d->N + 127) / 128);
    if (tiles > 192 && tiles <= 256) return check_launch(run_pc(p, EPI_LINEAR, 128, (hipStream_t)stream), "hvr_gemm (pc)");
  }
  const int slices = fewrow_slices(p);
  if (slices > 1 && d->ws && aligned16(d->ws) && d->ws_bytes >= (size_t)slices * p.M * p.N * 4) {
    // (a caller that hands a few-row workspace asked for the few-row forms: K sliced across the waves of a workgroup where the
    // shape allows -- no partials, one launch -- else across workgroups + a reduce launch)
    if (fewrow_mode() == 1 && kpar_supported(p)) return check_launch(run_kpar(p, (hipStream_t)stream), "hvr_gemm(K-parallel waves)");
    return check_launch(run_fewrow_split(p, slices, d->ws, (hipStream_t)stream), "hvr_gemm(split-K)");
  }
  return check_launch(run_tile_op(p, EPI_LINEAR, (hipStream_t)stream), "hvr_gemm");
}

// slices for a GEMM whose output has too few tiles to fill the chip and whose K loop is long (weight gradients)
static int splitk_slices(int M, int N, int K, int dtype) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (dtype == HVR_F16S) return 1;   // (the three-pass K loop is not sliced)
  const int ksteps = K / kstep_elems(dtype);
  if (tiles >= 192 || ksteps < 32) return 1;
  long s = (512 + tiles - 1) / tiles;
  if (s > ksteps / 8) s = ksteps / 8;
  if (s > 64) s = 64;
  return s < 2 ? 1 : (int)s;
}

size_t hvr_gemm_splitk_workspace_bytes(int M, int N, int K, int dtype) {
  if (M <= 0 || N <= 0 || K <= 0 || !valid_dtype(dtype)) return 0;
  const int s = splitk_slices(M, N, K, dtype);
  return s > 1 ? (size_t)s * M * N * 4 : 0;
}

int hvr_gemm_splitk(const hvr_gemm_desc* d, void* ws, size_t ws_bytes, void* stream) {
  if (!d) return fail(HVR_EINVAL, "null descriptor");
  if (d->bias || d->resid || d->relu) return fail(HVR_EINVAL, "split-K products have no epilogue (bias / residual / ReLU)");
  if (d->dtype != HVR_F32 && !d->out_f32) return fail(HVR_EINVAL, "split-K products are f32 (set out_f32)");
  GemmParams p;
  int rc = fill_linear(p, d->A, d->B, d->C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->dtype, d->staging);
  if (rc) return rc;
  const int es = elem_size(d->dtype);
  if ((d->lda * es) % 16 || (d->ldb * es) % 16) return fail(HVR_EINVAL, "lda/ldb rows must be 16-byte multiples");
  if ((d->ldc * 4) % 8) return fail(HVR_EINVAL, "ldc rows must be 8-byte multiples");
  p.out_f32 = d->out_f32;
  p.tile_hint = d->tile_hint;
  const int slices = splitk_slices(d->M, d->N, d->K, d->dtype);
  if (slices <= 1) return check_launch(run_tile_op(p, EPI_LINEAR, (hipStream_t)stream), "hvr_gemm_splitk");
  const size_t need = (size_t)slices * d->M * d->N * 4;
  if (!ws || ws_bytes < need || !aligned16(ws)) return fail(HVR_EINVAL, "split-K workspace too small (%zu < %zu) or unaligned", ws_bytes, need);
  const int ksteps = d->K / kstep_elems(d->dtype);
  p.ksplit_steps = (ksteps + slices - 1) / slices;
  p.ksplit_count = (ksteps + p.ksplit_steps - 1) / p.ksplit_steps;
  p.csplit_bytes = (long)d->M * d->N * 4;
  p.C = ws;
  p.ldc = d->N;
  hipError_t e = run_tile_op(p, EPI_LINEAR, (hipStream_t)stream);
  if (e == hipSuccess) e = run_splitk_reduce((const float*)ws, (float*)d->C, d->M, d->N, d->ldc, p.ksplit_count, (hipStream_t)stream);
  return check_launch(e, "hvr_gemm_splitk");
}

// `count` products of ONE shape in one launch (the weight gradients of a stage's identical blocks): problem g's operands sit stride_a /
// stride_b elements behind problem 0's, its f32 output stride_c floats behind; the tile engine's batch dimension (gridDim.z), K slices
// (gridDim.y) while count x tiles leaves the chip idle, and one reduce over all problems' partials
static int splitk_slices_batched(int M, int N, int K, int dtype, int count) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * count;
  if (dtype == HVR_F16S) return 1;
  const int ksteps = K / kstep_elems(dtype);
  if (tiles >= 384 || ksteps < 32) return 1;
  long s = (512 + tiles - 1) / tiles;
  if (s > ksteps / 8) s = ksteps / 8;
  if (s > 64) s = 64;
  return s < 2 ? 1 : (int)s;
}

size_t hvr_gemm_splitk_batched_workspace_bytes(int M, int N, int K, int dtype, int count) {
  if (M <= 0 || N <= 0 || K <= 0 || count <= 0 || !valid_dtype(dtype)) return 0;
  const int s = splitk_slices_batched(M, N, K, dtype, count);
  return s > 1 ? (size_t)s * count * M * N * 4 : 0;
}

int hvr_gemm_splitk_batched(const hvr_gemm_desc* d, int count, int64_t stride_a, int64_t stride_b, int64_t stride_c, void* ws, size_t ws_bytes,
                            void* stream) {
  if (!d || count <= 0) return fail(HVR_EINVAL, "null descriptor / count <= 0");
  if (d->bias || d->resid || d->relu) return fail(HVR_EINVAL, "split-K products have no epilogue (bias / residual / ReLU)");
  if (d->dtype != HVR_F32 && !d->out_f32) return fail(HVR_EINVAL, "split-K products are f32 (set out_f32)");
  if (d->dtype == HVR_F16S) return fail(HVR_EUNSUPPORTED, "batched split-K products take f32 / bf16 / half operands");
  GemmParams p;
  int rc = fill_linear(p, d->A, d->B, d->C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->dtype, d->staging);
  if (rc) return rc;
  const int es = elem_size(d->dtype);
  if ((d->lda * es) % 16 || (d->ldb * es) % 16 || (stride_a * es) % 16 || (stride_b * es) % 16) return fail(HVR_EINVAL, "lda / ldb rows and the problem strides must be 16-byte multiples");
  if ((d->ldc * 4) % 8 || (stride_c * 4) % 16) return fail(HVR_EINVAL, "ldc rows must be 8-byte, stride_c 16-byte multiples");
  if (stride_a * es >= (1L << 31) || stride_b * es >= (1L << 31)) return fail(HVR_EUNSUPPORTED, "one problem's operands must stay below 2 GiB");
  p.out_f32 = d->out_f32;
  p.tile_hint = d->tile_hint;
  p.batch = count;
  p.gs_a = stride_a * es; p.gs_b = stride_b * es; p.gs_c = stride_c * 4;
  const int slices = splitk_slices_batched(d->M, d->N, d->K, d->dtype, count);
  if (slices <= 1) return check_launch(run_tile_op(p, EPI_LINEAR, (hipStream_t)stream), "hvr_gemm_splitk_batched");
  if (d->ldc != d->N || stride_c != (int64_t)d->M * d->N) return fail(HVR_EINVAL, "the sliced form writes one contiguous [count][M][N] output (ldc = N, stride_c = M * N)");
  const size_t need = (size_t)slices * count * d->M * d->N * 4;
  if (!ws || ws_bytes < need || !aligned16(ws)) return fail(HVR_EINVAL, "split-K workspace too small (%zu < %zu) or unaligned", ws_bytes, need);
  if ((long)count * d->M > 0x7fffffffL) return fail(HVR_EUNSUPPORTED, "too many output rows");
  const int ksteps = d->K / kstep_elems(d->dtype);
  p.ksplit_steps = (ksteps + slices - 1) / slices;
  p.ksplit_count = (ksteps + p.ksplit_steps - 1) / p.ksplit_steps;
  p.csplit_bytes = (long)count * d->M * d->N * 4;   // slice s of problem g: ws + s * csplit_bytes + g * M * N * 4
  p.gs_c = (long)d->M * d->N * 4;
  p.C = ws;
  p.ldc = d->N;
  hipError_t e = run_tile_op(p, EPI_LINEAR, (hipStream_t)stream);
  if (e == hipSuccess) e = run_splitk_reduce((const float*)ws, (float*)d->C, count * d->M, d->N, d->N, p.ksplit_count, (hipStream_t)stream);
  return check_launch(e, "hvr_gemm_splitk_batched");
}

// descriptor -> kernel parameters + the path it takes (0 tile engine, 1 expand.hip); a negative return is the error
static int conv_params(const hvr_conv_desc* d, GemmParams& p, int& path) {
  if (!d) return fail(HVR_EINVAL, "null descriptor");
  const int OH = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int OW = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  if (OH <= 0 || OW <= 0) return fail(HVR_EINVAL, "empty conv output %dx%d", OH, OW);
  if (!valid_dtype(d->dtype)) return fail(HVR_EINVAL, "bad dtype %d", d->dtype);
  const int bke = kstep_elems(d->dtype);
  if (d->Cin % bke) return fail(HVR_EINVAL, "Cin=%d is not a multiple of %d", d->Cin, bke);
  if (d->dtype == HVR_F16S && !d->out_f32 && (d->Cout % 32 || !aligned128(d->y))) return fail(HVR_EINVAL, "split-half conv output: Cout %% 32 == 0, 128-byte aligned y");
  if (d->dtype == HVR_F16S && d->resid && (d->Cout % 32 || !aligned128(d->resid))) return fail(HVR_EINVAL, "split-half conv residual: Cout %% 32 == 0, 128-byte aligned");
  const long M = (long)d->B * OH * OW;
  if (M > 0x7fffffffL) return fail(HVR_EUNSUPPORTED, "too many output pixels");
  if ((long)d->B * d->H * d->W * d->Cin * (long)elem_size(d->dtype) >= (1L << 31))
    return fail(HVR_EUNSUPPORTED, "conv input of 2 GiB or more");
  const int K = d->KH * d->KW * d->Cin;
  int rc = fill_linear(p, d->x, d->w, d->y, (int)M, d->Cout, K, d->Cin, K, d->Cout, d->dtype, d->staging);
  if (rc) return rc;
  const bool pointwise = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0;
  if (!pointwise) {
    if (!d->zero) return fail(HVR_EINVAL, "conv needs the zero page");
    p.conv = 1; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = OH; p.OW = OW; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.zero = d->zero;
  }
  p.bias = d->bias; p.resid = d->resid; p.ldr = d->Cout; p.relu = d->relu; p.out_f32 = d->out_f32;
  p.tile_hint = d->tile_hint;
  p.alpha = d->alpha; p.beta = d->beta;
  if (pointwise) p.zero = d->zero;  // (expand.hip reads it in place of a missing shift)
  // the expand convs of a Bottleneck (1x1, K <= 256, + residual) are HBM-bound: row-panel kernel (expand.hip)
  static const int use_expand = std::getenv("HVR_EXPAND") ? std::atoi(std::getenv("HVR_EXPAND")) : 1;
  if (p.tile_hint == kBigForce) {
    path = 0;
    return 0;
  }
  const bool hint0 = p.tile_hint == 0 || p.tile_hint == kBigHint;  // (the big-tile hint leaves the dedicated kernels their shapes)
  bool big_first = false;  // the 288 x 256 tiles are ahead of the row-panel kernel from K = 256 on (layer 3's and res5's expand convs)
  if (hint0 && pointwise) {
    GemmParams q = p;
    q.tile_hint = 0;
    big_first = bigtile_supported(q, p.tile_hint == kBigHint);
    // (layer 3's expand + residual, K = 256: the big tiles are ahead on one clip -- 40 against 43 us for 15 frames -- and behind on a
    // batch of clips -- 60 frames: 178 - 198 against 155 - 167 us, profiles/r06_expand_ablation_raw.txt / r06_l3_fused.txt)
    if (big_first && p.resid && p.K == 256 && p.dtype != DT_F16S && p.M >= 65536 && expand_supported(p)) big_first = false;
  }
  path = (pointwise && (expand_supported(p) || expand_split_supported(p)) &&
          (p.tile_hint == kExpandHint || (hint0 && use_expand && p.resid && p.N >= 2 * p.K && !big_first))) ? 1 : 0;
  if (p.tile_hint == kExpandHint) p.tile_hint = 0;
  // layer 1's 3x3 (64 -> 64): persistent kernel with the weights resident in the LDS (conv3x3.hip)
  static const int use_c3 = std::getenv("HVR_CONV3") ? std::atoi(std::getenv("HVR_CONV3")) : 1;
  if (path == 0 && hint0 && use_c3 && conv3x3_c64_supported(p)) path = 2;
  if (path != 0 && p.tile_hint == kBigHint) p.tile_hint = 0;
  return 0;
}

size_t hvr_conv2d_splitk_workspace_bytes(const hvr_conv_desc* d) {
  GemmParams p;
  int path = 0;
  if (conv_params(d, p, path)) return 0;
  const int s = (path == 0 ? fewrow_slices(p) : 1);
  return s > 1 ? (size_t)s * p.M * p.N * 4 : 0;
}

int hvr_conv2d_nhwc(const hvr_conv_desc* d, void* stream) {
  GemmParams p;
  int path = 0;
  const int rc = conv_params(d, p, path);
  if (rc) return rc;
  if (path == 1) return check_launch(p.dtype == DT_F16S ? run_expand_split(p, (hipStream_t)stream) : run_expand(p, (hipStream_t)stream), "hvr_conv2d_nhwc(expand)");
  if (path == 2) return check_launch(run_conv3x3_c64(p, (hipStream_t)stream), "hvr_conv2d_nhwc(conv3x3_c64)");
  if (path == 0 && d->tile_hint == kBigForce) {
    if (!bigtile_supported(p, true)) return fail(HVR_EUNSUPPORTED, "the big-tile kernel takes bf16 convs with Cout %% 256 == 0 and Cin %% 64 == 0");
    return check_launch(run_bigtile(p, (hipStream_t)stream), "hvr_conv2d_nhwc(big tile)");
  }
  if (path == 0 && d->tile_hint == kTwoLevelHint) {
    // two-level accumulation (gemm_params.h: EPI_LINEAR2) where the format carries the f32 tolerance; any other format runs as with hint 0
    p.tile_hint = 0;
    if (two_level_supported(p)) return check_launch(run_tile_op(p, EPI_LINEAR2, (hipStream_t)stream), "hvr_conv2d_nhwc(two-level)");
  }
  if (path == 0 && (d->tile_hint == kBigHint || d->tile_hint == 0 || d->tile_hint == kTwoLevelHint)) {
    p.tile_hint = 0;
    if (bigtile_supported(p, d->tile_hint == kBigHint)) return check_launch(run_bigtile(p, (hipStream_t)stream), "hvr_conv2d_nhwc(big tile)");
  }
  const int slices = (path == 0 ? fewrow_slices(p) : 1);
  if (slices > 1 && d->ws && aligned16(d->ws) && d->ws_bytes >= (size_t)slices * p.M * p.N * 4) {
    if (fewrow_mode() == 1 && kpar_supported(p)) return check_launch(run_kpar(p, (hipStream_t)stream), "hvr_conv2d_nhwc(K-parallel waves)");
    const hipError_t e = run_fewrow_split(p, slices, d->ws, (hipStream_t)stream);
    return check_launch(e, "hvr_conv2d_nhwc(split-K)");
  }
  return check_launch(run_tile_op(p, EPI_LINEAR, (hipStream_t)stream), "hvr_conv2d_nhwc");
}

static int tail_params(const hvr_tail_desc* d, GemmParams& p) {
  if (!d) return fail(HVR_EINVAL, "null descriptor");
  if (d->dtype != HVR_BF16 && d->dtype != HVR_F16) return fail(HVR_EUNSUPPORTED, "hvr_bottleneck_tail takes bf16 or half operands");
  if (d->B <= 0 || d->OH <= 0 || d->OW <= 0 || d->stride2 <= 0) return fail(HVR_EINVAL, "empty tail problem");
  if ((d->OH - 1) * d->stride2 >= d->H2 || (d->OW - 1) * d->stride2 >= d->W2) return fail(HVR_EINVAL, "the sampled pixels fall outside x");
  const long M = (long)d->B * d->OH * d->OW;
  if (M > 0x7fffffffL) return fail(HVR_EUNSUPPORTED, "too many output pixels");
  const int K = d->C1 + d->C2;
  int rc = fill_linear(p, d->h, d->w, d->y, (int)M, d->Cout, K, d->C1, K, d->Cout, d->dtype, 1);
  if (rc) return rc;
  if (!d->x || !aligned16(d->x)) return fail(HVR_EINVAL, "x must be a 16-byte aligned device pointer");
  if ((long)d->B * d->H2 * d->W2 * d->C2 * 2 >= (1L << 31)) return fail(HVR_EUNSUPPORTED, "block input of 2 GiB or more");
  p.bias = d->bias; p.relu = d->relu;
  p.A2 = d->x; p.K1 = d->C1; p.H2 = d->H2; p.W2 = d->W2; p.s2 = d->stride2; p.OH = d->OH; p.OW = d->OW;
  return 0;
}

// the tile engine's form of the fused tail (K too long for the row-panel kernel: stage 3's 256 + 512 and res5's 512 + 1024):
// one GEMM over K = C1 + C2 whose A operand switches from h to the sampled block input at K-step C1 / 64
static bool tail_on_tile_engine(const GemmParams& p) {
  return (p.dtype == DT_BF16 || p.dtype == DT_F16) && p.K1 % 64 == 0 && (p.K - p.K1) % 64 == 0 && p.K1 >= 64 && p.K - p.K1 >= 64 && p.N % 8 == 0 && p.ldc % 8 == 0 &&
         (long)p.M * (p.K - p.K1) * 2 < (1L << 31);
}

int hvr_bottleneck_tail_supported(const hvr_tail_desc* d) {
  GemmParams p;
  if (tail_params(d, p)) return 0;
  return (expand_supported(p) || tail_on_tile_engine(p)) ? 1 : 0;
}

int hvr_bottleneck_tail(const hvr_tail_desc* d, void* stream) {
  GemmParams p;
  const int rc = tail_params(d, p);
  if (rc) return rc;
  if (expand_supported(p)) return check_launch(run_expand(p, (hipStream_t)stream), "hvr_bottleneck_tail");
  if (tail_on_tile_engine(p)) {
    // the 288 x 256 tiles take the second K segment too (same MFMA order: bit-identical).  Round 3 measured them slightly behind the tile
    // engine here; with persistent workgroups (round 6) they are ahead: res5's 512 + 1024 -> 2048 tail 973 -> 857 us for four clips (230 ->
    // 200 for one), layer 3's 256 + 512 -> 1024 311 -> 256 (72 -> 69)
    if (bigtile_supported(p, false)) return check_launch(run_bigtile(p, (hipStream_t)stream), "hvr_bottleneck_tail(big tile)");
    return check_launch(run_tile_op(p, EPI_LINEAR, (hipStream_t)stream), "hvr_bottleneck_tail(tile engine)");
  }
  return fail(HVR_EUNSUPPORTED, "no fused tail kernel for C1=%d C2=%d Cout=%d", d->C1, d->C2, d->Cout);
}

// closing 1x1 (+ projection shortcut or identity residual) + the next block's reducing 1x1 (expand.hip, NX > 0)
static int tail_next_params(const hvr_tail_next_desc* d, GemmParams& p) {
  if (!d) return fail(HVR_EINVAL, "null descriptor");
  const hvr_tail_desc& t = d->tail;
  if (t.C2 > 0) {
    const int rc = tail_params(&t, p);
    if (rc) return rc;
  } else {
    if (t.dtype != HVR_BF16 && t.dtype != HVR_F16 && t.dtype != HVR_F16S)
      return fail(HVR_EUNSUPPORTED, "hvr_bottleneck_tail_next takes bf16, half or (identity form) split-half operands");
    if (t.B <= 0 || t.OH <= 0 || t.OW <= 0) return fail(HVR_EINVAL, "empty tail problem");
    const long M = (long)t.B * t.OH * t.OW;
    if (M > 0x7fffffffL) return fail(HVR_EUNSUPPORTED, "too many output pixels");
    const int rc = fill_linear(p, t.h, t.w, t.y, (int)M, t.Cout, t.C1, t.C1, t.C1, t.Cout, t.dtype, 1);
    if (rc) return rc;
    const bool split = t.dtype == HVR_F16S;
    if (!d->resid || !(split ? aligned128(d->resid) : aligned16(d->resid)))
      return fail(HVR_EINVAL, "an identity block needs its (16-byte aligned; split half: 128-byte aligned) residual map");
    if (split && (t.Cout % 32 || !aligned128(t.y))) return fail(HVR_EINVAL, "split-half tail: Cout %% 32 == 0, 128-byte aligned y");
    p.bias = t.bias; p.relu = t.relu; p.resid = d->resid; p.ldr = t.Cout;
    if (t.dtype == HVR_F16S) { p.alpha = d->alpha; p.beta = d->beta; }   // (documented as split-half only: include/hvr_hip.h)
  }
  if (!t.relu) return fail(HVR_EUNSUPPORTED, "the next block reads the activated output (relu = 1)");
  p.Wn = d->wn; p.bias_n = d->bias_n; p.Hn = d->hn; p.Cn = d->Cn;
  return 0;
}

int hvr_bottleneck_tail_next_supported(const hvr_tail_next_desc* d) {
  GemmParams p;
  if (tail_next_params(d, p)) return 0;
  return (expand_next_supported(p) || expand_split_next_supported(p)) ? 1 : 0;
}

int hvr_bottleneck_tail_next(const hvr_tail_next_desc* d, void* stream) {
  GemmParams p;
  const int rc = tail_next_params(d, p);
  if (rc) return rc;
  if (expand_split_next_supported(p)) return check_launch(run_expand_split(p, (hipStream_t)stream), "hvr_bottleneck_tail_next(split half)");
  if (!expand_next_supported(p))
    return fail(HVR_EUNSUPPORTED, "no fused tail + next conv kernel for C1=%d C2=%d Cout=%d Cn=%d", d->tail.C1, d->tail.C2, d->tail.Cout, d->Cn);
  return check_launch(run_expand(p, (hipStream_t)stream), "hvr_bottleneck_tail_next");
}

int hvr_conv2d_path(const hvr_conv_desc* d) {
  GemmParams p;
  int path = 0;
  const int rc = conv_params(d, p, path);
  if (rc) return rc;
  if (path == 0 && (d->tile_hint == 0 || d->tile_hint == kBigHint)) {
    p.tile_hint = 0;
    if (bigtile_supported(p, d->tile_hint == kBigHint)) return 3;
  }
  return path;
}

int hvr_im2col_stem(const float* img, void* cols, int B, int H, int W, int KP, int dtype, void* stream) {
  if (!img || !cols || B <= 0) return fail(HVR_EINVAL, "bad stem arguments");
  if (dtype == HVR_F16S && (KP % 32 || !aligned128(cols))) return fail(HVR_EINVAL, "split-half patch rows: KP %% 32 == 0, 128-byte aligned");
  if (KP < 147 || KP % kstep_elems(dtype)) return fail(HVR_EINVAL, "KP=%d must be >= 147 and a K-step multiple", KP);
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  return check_launch(run_im2col_stem(img, cols, B, H, W, OH, OW, KP, dtype, (hipStream_t)stream), "hvr_im2col_stem");
}

int hvr_stem_fused_dtype(const float* img, const void* wpk, const float* bias, void* out, int B, int H, int W, int dtype, void* stream) {
  if (!img || !wpk || !bias || !out || B <= 0 || H < 7 || W < 7) return fail(HVR_EINVAL, "bad fused-stem arguments");
  if (dtype != HVR_BF16 && dtype != HVR_F16 && dtype != HVR_F16S) return fail(HVR_EUNSUPPORTED, "the fused stem computes on bf16, half or split-half operands");
  if (!aligned16(wpk) || !aligned16(out) || !aligned16(bias)) return fail(HVR_EINVAL, "fused stem operands must be 16-byte aligned");
  if (dtype == HVR_F16S && !aligned128(out)) return fail(HVR_EINVAL, "split-half stem output must be 128-byte aligned");
  return check_launch(run_stem_fused(img, wpk, bias, out, B, H, W, dtype, (hipStream_t)stream), "hvr_stem_fused");
}
int hvr_stem_fused(const float* img, const void* wpk, const float* bias, void* out, int B, int H, int W, void* stream) {
  return hvr_stem_fused_dtype(img, wpk, bias, out, B, H, W, HVR_BF16, stream);
}

int hvr_maxpool3x3s2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  if (!x || !y || C % 4) return fail(HVR_EINVAL, "bad maxpool arguments (C %% 4 == 0 required)");
  if (dtype == HVR_F16S) return fail(HVR_EUNSUPPORTED, "max pooling runs on f32 / bf16 / half maps");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  return check_launch(run_maxpool3x3s2(x, y, B, H, W, C, OH, OW, dtype, (hipStream_t)stream), "hvr_maxpool3x3s2_nhwc");
}

// ---- relation ----
static int env_tile(const char* name) {  // tuning override, read once
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : 0;
}
static inline long rel_ldp(int Mk) { return ((long)Mk + 127) / 128 * 128; }

// Key-slices of the apply pass when there are few query rows (the key-frame-only stage: Mq = 300 against Mk = 4 500 gives a
// 3 x 8 output tile grid with a 72-step K loop -- 24 workgroups on 256 CUs): the 128-key blocks are dealt to `slices` workgroups
// per output tile (at least 2 blocks each), every slice writes an f32 partial, one reduce launch sums and rounds them.  The
// block weights g = 2^(m_t - m*) / L are global per row, so the partials simply add.  1 = no split.
static int apply_slice_blocks() { return 4; }   // 128-key blocks per slice of a sliced apply pass
static int apply_slices(int Mq, int Mk, int D) {
  const long tiles = (long)((Mq + 127) / 128) * ((D + 127) / 128);
  const int nblk = (int)(rel_ldp(Mk) / 128);
  if (tiles >= 96 || nblk < 8 || D % 8) return 1;
  // a FIXED four blocks (512 keys) per slice: the partition -- and with it the order of the f32 sum -- depends on Mk only,
  // so AMONG the calls that take this path (fewer than 96 output tiles: up to 1 536 query rows at D = 1024) a query row's result
  // does not depend on how many other rows are in the call.  Whether the path is taken does depend on Mq: the same row computed
  // in a 4 500-row call (unsplit, one running f32 sum over the 36 blocks) may differ from it in the last f32 bit.  The loops that
  // are tested for batch invariance (cached / look-ahead stream) always call the stages with the shapes that ship -- 300 and
  // 4 500 rows -- so each stage stays on one side of the threshold
  const int per = apply_slice_blocks();
  return (nblk + per - 1) / per;
}

// The apply pass of ONE problem on two-byte / f32 operands: O = sum_t g_t (P~_t V_t) from the scores pass's P~ and block statistics
// (the tile engine's / pc_gemm.hip's EPI_APPLY, key slices for few query rows).  Shared by hvr_relation_fwd and the per-group leg of
// hvr_relation_fwd_grouped.
static int relation_apply_pass(const void* P, const void* Vt, float* mstat, float* lstat, float* partial, void* O, int64_t ldo,
                               int Mq, int Mk, int D, long ldp, int nt, int dtype, int staging, hipStream_t s) {
  const bool two_byte = dtype == HVR_BF16 || dtype == HVR_F16;
  constexpr int tile_apply = 0, gm_apply = 1;
#ifdef HVR_DEBUG_KNOBS
  static const int dbg_ld0 = env_tile("HVR_DBG_LD0");
#endif
  GemmParams p;
  int rc;
  rc = fill_linear(p, P, Vt, O, Mq, D, (int)ldp, ldp, ldp, ldo, dtype, staging);
  if (rc) return rc;
  p.mstat = mstat; p.lstat = lstat; p.ntile = nt;
  p.tile_hint = tile_apply;
  p.group_m = gm_apply;
#ifdef HVR_DEBUG_KNOBS
  if (dbg_ld0 & 2) { p.lda = 0; p.ldb = 0; }
#endif
  const int slices = apply_slices(Mq, Mk, D);
  if (slices > 1 && tile_apply == 0 && (!two_byte || (ldo % 8 == 0 && aligned16(O)))) {
    const int steps_per_blk = two_byte ? 2 : 4;
    const int per = apply_slice_blocks();  // apply_slices: 128-key blocks per slice
    p.ksplit_steps = per * steps_per_blk;
    p.ksplit_count = slices;
    p.csplit_bytes = (long)Mq * D * 4;
    p.C = partial; p.ldc = D; p.out_f32 = two_byte ? 1 : 0;
    p.tile_hint = 1;  // 128 x 128 tiles; the slices are latency chains of 8 K-steps (the pipelined shapes measure the same)
    // (an in-launch merge of the partials by the slice that reaches an output tile last -- tickets, agent-scope fences -- was built in
    // round 3 and measured SLOWER than the 5 us reduce launch it saved: 49.3 -> 83.6 us for the stage, 27.6 us of it fences
    // (profiles/r03_key_stage_merge_ab.txt); removed in round 5)
    hipError_t e = run_tile_op(p, EPI_APPLY, s);
    if (e == hipSuccess)
      e = dtype == HVR_BF16 ? run_splitk_reduce_bf16(partial, O, Mq, D, ldo, slices, s)
          : dtype == HVR_F16 ? run_splitk_reduce_f16(partial, O, Mq, D, ldo, slices, s) : run_splitk_reduce(partial, (float*)O, Mq, D, ldo, slices, s);
    return check_launch(e, "relation: apply (key slices)");
  }
  // The producer / consumer form of the apply pass (pc_gemm.hip, 144 x 128 tiles, block weights from an LDS table).  Round 2 measured
  // it behind the tile engine (60 against 57 us) and left it opt-in; on round 4's boxes it is AHEAD, alone (tools/rel_bench.py: 0.1073 /
  // 0.1102 ms per relation call against 0.1164 / 0.1200) and inside the window (three alternating runs of tools/window_breakdown.py:
  // 109.4 / 110.3 / 110.5 us per call against 114.1 / 113.0 / 113.1) -- profiles/r04_relation_apply.txt -- so it is the default for
  // window-sized problems.
  if (Mq >= 1024 && pc_supported(p, EPI_APPLY))
    return check_launch(run_pc(p, EPI_APPLY, 128, s), "relation: apply (pc)");
  return check_launch(run_tile_op(p, EPI_APPLY, s), "relation: apply");
}

// What follows the scores pass of ONE split-half problem: V^T (V == nullptr: the scores launch wrote it), then either one normalising sweep over P~ + a plain product, or the apply
// pass with the block weights g = 2^(m_t - m*) / L folded in per 128-key block (EPI_APPLY: a block is four split K-steps of three
// MFMAs, its un-scaled partial joins the running total by one FMA per accumulator register).  Shared by hvr_relation_fwd and the
// per-group leg of hvr_relation_fwd_grouped.
static int relation_split_tail(void* P, void* Vt, float* mstat, float* lstat, const void* V, int64_t ldv, void* O, int64_t ldo, int Mq,
                               int Mk, int D, long ldp, int nt, int staging, hipStream_t s) {
  constexpr int tile_apply_s = 0;
  int rc = V ? check_launch(run_transpose_pad(V, Vt, Mk, D, ldv, ldp, HVR_F16S, s), "relation: V transpose (split half)") : HVR_OK;
  if (rc) return rc;
  GemmParams p;
  rc = fill_linear(p, P, Vt, O, Mq, D, (int)ldp, ldp, ldp, ldo, HVR_F16S, staging);
  if (rc) return rc;
  p.alpha = 1.f / kSplitProbScale;   // the probabilities are stored x 2^12 (gemm_tile.h / relation_bt.hip, scores epilogue)
  p.mstat = mstat; p.lstat = lstat; p.ntile = nt;
  p.tile_hint = tile_apply_s;
#ifdef HVR_DEBUG_KNOBS
  static const int dbg_apply_s = env_tile("HVR_DBG_APPLY_S");
  if (dbg_apply_s) p.tile_hint = dbg_apply_s;
#endif
  // Which form: measured on one MI355X (tools/rel_bench.py --dtype f16x2; profiles/r04_split_relation.txt): the folded form wins
  // the key stage (300 x 4 500: 0.146 against 0.163 ms), the full stage goes the other way (4 500 x 4 500: 0.32-0.36 against
  // 0.29 ms) -- the fold needs the double-buffered apply shapes (the pipelined loop is written for two K-steps per block), and one
  // normalising sweep over P~ (22 us at 7.5 TB/s out of the Infinity Cache) + the pipelined plain product is cheaper than they are.
  // HVR_SPLIT_NORMALIZE = 0 / 1 forces the folded / the swept form.
  static const int split_normalize = std::getenv("HVR_SPLIT_NORMALIZE") ? std::atoi(std::getenv("HVR_SPLIT_NORMALIZE")) : -1;
  if (split_normalize == 1 || (split_normalize < 0 && Mq >= 1024)) {
    rc = check_launch(run_relation_normalize(P, mstat, lstat, Mq, nt, ldp, HVR_F16S, 1.f, s), "relation: normalise (split half)");
    if (rc) return rc;
    return check_launch(run_tile_op(p, EPI_LINEAR, s), "relation: apply (split half, plain product)");
  }
  return check_launch(run_tile_op(p, EPI_APPLY, s), "relation: apply (split half)");
}

size_t hvr_relation_workspace_bytes(int Mq, int Mk, int D, int dtype) {
  const long ldp = rel_ldp(Mk), nt = ldp / 128;
  const size_t es = elem_size(dtype);
  const int slices = apply_slices(Mq, Mk, D);
  return align256((size_t)Mq * ldp * es) + align256((size_t)D * ldp * es) + 2 * align256((size_t)Mq * nt * 4) +
         (slices > 1 ? align256((size_t)slices * Mq * D * 4) : 0);
}

int hvr_relation_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                     int64_t ldo, int Mq, int Mk, int D, float scale, int dtype, int staging, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!Q || !K || !V || !O || !ws) return fail(HVR_EINVAL, "null pointer");
  if (Mq <= 0 || Mk <= 0) return fail(HVR_EINVAL, "empty relation Mq=%d Mk=%d", Mq, Mk);
  if (!(scale > 0.f)) return fail(HVR_EINVAL, "relation scale must be positive (the reference uses 1/sqrt(D)), got %g", (double)scale);
  if (ws_bytes < hvr_relation_workspace_bytes(Mq, Mk, D, dtype))
    return fail(HVR_EWORKSPACE, "relation workspace %zu < %zu", ws_bytes, hvr_relation_workspace_bytes(Mq, Mk, D, dtype));
  const long ldp = rel_ldp(Mk);
  const int nt = (int)(ldp / 128);
  const size_t es = elem_size(dtype);
  char* w = (char*)ws;
  void* P = w;                w += align256((size_t)Mq * ldp * es);
  void* Vt = w;               w += align256((size_t)D * ldp * es);
  float* mstat = (float*)w;   w += align256((size_t)Mq * nt * 4);
  float* lstat = (float*)w;   w += align256((size_t)Mq * nt * 4);
  float* partial = (float*)w;  // apply_slices(...) > 1: the slices' f32 partial outputs
  hipStream_t s = (hipStream_t)stream;

  GemmParams p;
  int rc;
  constexpr int tile_scores_s = 0;
  if (dtype == HVR_F16S) {
    // split half: the scores pass (block-local maxima, P~ stored x 2^12 in the split format) -- on the 352 x 256 persistent tiles of
    // relation_bt.hip for window-sized problems, else on the tile engine -- then relation_split_tail
    if (ldv % 64 || !aligned128(V) || ldo % 32 || !aligned128(O) || D % 64) return fail(HVR_EINVAL, "split-half relation: D, ldv multiples of 64, ldo of 32, 128-byte aligned V / O");
    const bool bt = staging && scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt, 1, true);
    if (bt) {   // (V^T is written by the same launch)
      ScoresBTParams b;
      b.Q = (const bf16_t*)Q; b.K = (const bf16_t*)K; b.P = (bf16_t*)P; b.mstat = mstat; b.lstat = lstat;
      b.V = (const bf16_t*)V; b.Vt = (bf16_t*)Vt; b.Mq = Mq; b.Mk = Mk; b.D = D; b.ntile = nt;
      b.ldq = ldq; b.ldk = ldk; b.ldv = ldv; b.ldp = ldp; b.sl2 = scale * 1.4426950408889634f; b.f16 = 2;
      b.groups = 1; b.gs_q = b.gs_k = b.gs_v = b.gs_p = b.gs_vt = b.gs_stat = 0; b.int_max = 0;
      rc = check_launch(run_scores_bt(b, s), "relation: scores (split half, big tile)");
    } else {
      rc = fill_linear(p, Q, K, P, Mq, (Mk + 3) / 4 * 4, D, ldq, ldk, ldp, dtype, staging);
      if (rc) return rc;
      p.N = Mk; p.scale = scale; p.mstat = mstat; p.lstat = lstat; p.ntile = nt; p.group_m = 8;
      p.tile_hint = tile_scores_s;
      rc = check_launch(run_tile_op(p, EPI_SCORES, s), "relation: scores (split half)");
    }
    if (rc) return rc;
    return relation_split_tail(P, Vt, mstat, lstat, bt ? nullptr : V, ldv, O, ldo, Mq, Mk, D, ldp, nt, staging, s);
  }
  const bool two_byte = dtype == HVR_BF16 || dtype == HVR_F16;
  // tuning overrides, read once
  constexpr int tile_scores = 0, gm_scores = 8;
#ifdef HVR_DEBUG_KNOBS  // tuning builds only (tools/build_dbg.sh): alias every operand row to row 0 (no memory-system cost)
  static const int dbg_ld0 = env_tile("HVR_DBG_LD0");
#endif
  const bool bt = two_byte && staging && scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt);
  if (bt) {
    // window-sized problems: one 336 x 256 score tile per CU, V^T written by the same launch (relation_bt.hip)
    ScoresBTParams b;
    b.Q = (const bf16_t*)Q; b.K = (const bf16_t*)K; b.P = (bf16_t*)P; b.mstat = mstat; b.lstat = lstat;
    b.V = (const bf16_t*)V; b.Vt = (bf16_t*)Vt; b.Mq = Mq; b.Mk = Mk; b.D = D; b.ntile = nt;
    b.ldq = ldq; b.ldk = ldk; b.ldv = ldv; b.ldp = ldp; b.sl2 = scale * 1.4426950408889634f; b.f16 = dtype == HVR_F16;
    b.groups = 1; b.gs_q = b.gs_k = b.gs_v = b.gs_p = b.gs_vt = b.gs_stat = 0; b.int_max = 0;
    rc = check_launch(run_scores_bt(b, s), "relation: scores (big tile)");
    if (rc) return rc;
  } else {
    // few query rows (the key stage, 300 x 4 500: 108 score tiles on 256 CUs): V^T is written by workgroups of the SCORES launch that sit
    // behind its tiles, on the CUs the score grid leaves idle -- one launch and ~8 us less than the transpose launch in front of it
    // (GemmParams::tr_*, gemm_tile.h)
    const bool fold = two_byte && staging && Mq <= 1024 && D % 8 == 0 && ldv % 8 == 0 && aligned16(V) && aligned16(Vt);
    if (!fold) {
      rc = check_launch(run_transpose_pad(V, Vt, Mk, D, ldv, ldp, dtype, s), "relation: V transpose");
      if (rc) return rc;
    }
    rc = fill_linear(p, Q, K, P, Mq, (Mk + 3) / 4 * 4, D, ldq, ldk, ldp, dtype, staging);
    if (rc) return rc;
    if (fold) {
      p.tr_in = V; p.tr_out = Vt; p.tr_R = Mk; p.tr_C = D; p.tr_ldx = ldv; p.tr_ldt = ldp;
      const int score_tiles = ((Mq + 127) / 128) * ((Mk + 127) / 128);   // (the 128 x 128 shape below)
      const int vt_tiles = ((D + 63) / 64) * (int)((ldp + 63) / 64);
      int nb = 256 - score_tiles % 256;   // one workgroup per CU the last round of score tiles leaves free
      if (nb < 64) nb += 256;
      p.tr_blocks = nb < vt_tiles ? nb : vt_tiles;
    }
    p.N = Mk;  // keys beyond Mk are masked inside the score epilogue
    p.scale = scale; p.mstat = mstat; p.lstat = lstat; p.ntile = nt;
    // (few query rows -- the key stage, 300 x 4 500: 108 tiles walking 16 K-steps each -- are a latency chain: the 4-stage ring
    // of the 128 x 128 shape keeps three K-steps of DMA in flight, 54.3 -> 50.8 us for the stage)
    p.tile_hint = tile_scores ? tile_scores : ((two_byte && staging && Mq <= 1024) ? 9 : 0);
    p.group_m = gm_scores;
#ifdef HVR_DEBUG_KNOBS
    if (dbg_ld0 & 1) { p.lda = 0; p.ldb = 0; }
#endif
    rc = check_launch(run_tile_op(p, EPI_SCORES, s), "relation: scores");
    if (rc) return rc;
  }
  return relation_apply_pass(P, Vt, mstat, lstat, partial, O, ldo, Mq, Mk, D, ldp, nt, dtype, staging, s);
}

// ---- relation, G independent problems of one shape in one call (the windows a caller has in flight) ----
size_t hvr_relation_grouped_workspace_bytes(int groups, int Mq, int Mk, int D, int dtype) {
  return (size_t)(groups > 0 ? groups : 0) * align256(hvr_relation_workspace_bytes(Mq, Mk, D, dtype));
}

int hvr_relation_fwd_grouped(const void* Q, int64_t ldq, int64_t gsq, const void* K, int64_t ldk, int64_t gsk, const void* V, int64_t ldv,
                             int64_t gsv, void* O, int64_t ldo, int64_t gso, int groups, int Mq, int Mk, int D, float scale, int dtype,
                             int staging, int exact, void* ws, size_t ws_bytes, void* stream) {
  if (!Q || !K || !V || !O || !ws) return fail(HVR_EINVAL, "null pointer");
  if (groups <= 0) return fail(HVR_EINVAL, "relation groups must be positive, got %d", groups);
  if (Mq <= 0 || Mk <= 0) return fail(HVR_EINVAL, "empty relation Mq=%d Mk=%d", Mq, Mk);
  if (!(scale > 0.f)) return fail(HVR_EINVAL, "relation scale must be positive (the reference uses 1/sqrt(D)), got %g", (double)scale);
  if (gsq < 0 || gsk < 0 || gsv < 0 || gso < 0) return fail(HVR_EINVAL, "negative group stride");
  const size_t per = align256(hvr_relation_workspace_bytes(Mq, Mk, D, dtype));
  if (ws_bytes < per * (size_t)groups) return fail(HVR_EWORKSPACE, "grouped relation workspace %zu < %zu", ws_bytes, per * (size_t)groups);
  const size_t es = elem_size(dtype);
  hipStream_t s = (hipStream_t)stream;
  // HVR_REL_GROUPED (tuning): 0 = one hvr_relation_fwd per group; 1 = the persistent scores launch over all groups, apply per group;
  // 2 (default) = + the 288 x 256 apply launch over all groups where it applies (bf16, >= 3 window-sized groups)
  static const int mode = std::getenv("HVR_REL_GROUPED") ? std::atoi(std::getenv("HVR_REL_GROUPED")) : 2;
  const long ldp = rel_ldp(Mk);
  const int nt = (int)(ldp / 128);
  const bool two_byte = dtype == HVR_BF16 || dtype == HVR_F16;
  // group 0's workspace laid out as hvr_relation_fwd lays out its own; group g's `per` bytes further
  char* w = (char*)ws;
  void* P = w;                w += align256((size_t)Mq * ldp * es);
  void* Vt = w;               w += align256((size_t)D * ldp * es);
  float* mstat = (float*)w;   w += align256((size_t)Mq * nt * 4);
  float* lstat = (float*)w;   w += align256((size_t)Mq * nt * 4);
  float* partial = (float*)w;
  const bool split = dtype == HVR_F16S;
  const bool strides_ok = split ? (gsq % 32 == 0 && gsk % 32 == 0 && gsv % 32 == 0 && gso % 32 == 0)
                                : (two_byte && gsq % 8 == 0 && gsk % 8 == 0 && gsv % 8 == 0 && gso % 8 == 0);
  if (split && (ldv % 64 || !aligned128(V) || ldo % 32 || !aligned128(O) || D % 64)) return fail(HVR_EINVAL, "split-half relation: D, ldv multiples of 64, ldo of 32, 128-byte aligned V / O");
  // (exact: "bit for bit hvr_relation_fwd's" only holds when the single call takes the SAME scores kernel -- a per-group tile count
  // the one-group rule sends to the tile engine, e.g. Mq = Mk = 5 400: 352 tiles, runs as per-group single calls)
  // ---- few query rows (the key stage, hrnmp_bbox_head.py:269-278,888-891: 300 x 4 500 per clip), two-byte operands: ONE launch per pass
  // over all groups -- the tile engine's batch dimension (GemmParams::batch, gridDim.z = group): scores + V^T (the transposing workgroups
  // behind every group's score tiles), the key-sliced apply pass, one reduce.  Every group's arithmetic is hvr_relation_fwd's (same
  // tiles, same slices, same order): bit-identical to G single calls, in one sixth of the launches.  VERDICT r05 item 4.
  const int key_slices = apply_slices(Mq, Mk, D);
  if (groups > 1 && mode != 0 && staging && strides_ok && two_byte && Mq <= 1024 && key_slices > 1 && D % 8 == 0 && ldv % 8 == 0 &&
      ldo % 8 == 0 && aligned16(V) && aligned16(Vt) && aligned16(O) && per % 16 == 0 &&
      !scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt, 1, false)) {
    GemmParams p;
    int rc = fill_linear(p, Q, K, P, Mq, (Mk + 3) / 4 * 4, D, ldq, ldk, ldp, dtype, staging);
    if (rc) return rc;
    p.batch = groups;
    p.gs_a = gsq * (long)es; p.gs_b = gsk * (long)es; p.gs_c = (long)per; p.gs_stat = (long)(per / 4);
    p.tr_in = V; p.tr_out = Vt; p.tr_R = Mk; p.tr_C = D; p.tr_ldx = ldv; p.tr_ldt = ldp;
    p.gs_tr_in = gsv * (long)es; p.gs_tr_out = (long)per;
    const int score_tiles = ((Mq + 127) / 128) * ((Mk + 127) / 128);
    const int vt_tiles = ((D + 63) / 64) * (int)((ldp + 63) / 64);
    // transposing workgroups per group: together with the score tiles of all groups about two rounds of the chip's 512 resident
    // workgroups (128 x 128 tiles, two per CU); at least 32, at most one per 64 x 64 tile
    int nb = (1024 - groups * score_tiles) / groups;
    nb = nb < 32 ? 32 : nb;
    p.tr_blocks = nb < vt_tiles ? nb : vt_tiles;
    p.N = Mk;
    p.scale = scale; p.mstat = mstat; p.lstat = lstat; p.ntile = nt;
    // the double-buffered 128 x 128 shape (two workgroups per CU: the 1 024 workgroups of four clips are two rounds) -- the single call's
    // four-stage ring holds one workgroup per CU and is the better latency chain for ONE clip's 108 tiles; same MFMA order, same P~
    // (sweep: 112 us per four clips against 122; profiles/r06_key_stage.txt)
    p.tile_hint = 1; p.group_m = 8;
    rc = check_launch(run_tile_op(p, EPI_SCORES, s), "relation (grouped key stage): scores");
    if (rc) return rc;
    GemmParams a;
    rc = fill_linear(a, P, Vt, O, Mq, D, (int)ldp, ldp, ldp, ldo, dtype, staging);
    if (rc) return rc;
    a.mstat = mstat; a.lstat = lstat; a.ntile = nt; a.group_m = 1;
    a.batch = groups; a.gs_a = (long)per; a.gs_b = (long)per; a.gs_c = (long)per; a.gs_stat = (long)(per / 4);
    const int per_blk = apply_slice_blocks();
    a.ksplit_steps = per_blk * 2; a.ksplit_count = key_slices; a.csplit_bytes = (long)Mq * D * 4;
    a.C = partial; a.ldc = D; a.out_f32 = 1; a.tile_hint = 1;
    hipError_t e = run_tile_op(a, EPI_APPLY, s);
    if (e == hipSuccess) e = run_splitk_reduce_half_batched(partial, O, Mq, D, ldo, key_slices, dtype == HVR_F16, groups, (long)(per / 4), gso, s);
    return check_launch(e, "relation (grouped key stage): apply");
  }
  // the same stage on split-half operands: the scores pass and the folded apply pass (relation_split_tail's form for few query rows) as one
  // launch each over all groups, the V^T copies per group in between -- 24 apply tiles per clip are 96 on the chip at once
  if (groups > 1 && mode != 0 && staging && strides_ok && split && Mq < 1024 &&
      !scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt, 1, true)) {
    GemmParams p;
    int rc = fill_linear(p, Q, K, P, Mq, (Mk + 3) / 4 * 4, D, ldq, ldk, ldp, dtype, staging);
    if (rc) return rc;
    p.N = Mk; p.scale = scale; p.mstat = mstat; p.lstat = lstat; p.ntile = nt; p.group_m = 8; p.tile_hint = 0;
    p.batch = groups; p.gs_a = gsq * (long)es; p.gs_b = gsk * (long)es; p.gs_c = (long)per; p.gs_stat = (long)(per / 4);
    rc = check_launch(run_tile_op(p, EPI_SCORES, s), "relation (grouped key stage, split half): scores");
    if (rc) return rc;
    for (int g = 0; g < groups; ++g) {
      rc = check_launch(run_transpose_pad((const char*)V + (size_t)g * gsv * es, (char*)Vt + (size_t)g * per, Mk, D, ldv, ldp, HVR_F16S, s),
                        "relation (grouped key stage, split half): V transpose");
      if (rc) return rc;
    }
    GemmParams a;
    rc = fill_linear(a, P, Vt, O, Mq, D, (int)ldp, ldp, ldp, ldo, HVR_F16S, staging);
    if (rc) return rc;
    a.alpha = 1.f / kSplitProbScale;
    a.mstat = mstat; a.lstat = lstat; a.ntile = nt; a.tile_hint = 0;
    a.batch = groups; a.gs_a = (long)per; a.gs_b = (long)per; a.gs_c = gso * (long)es; a.gs_stat = (long)(per / 4);
    return check_launch(run_tile_op(a, EPI_APPLY, s), "relation (grouped key stage, split half): apply");
  }
  if (groups == 1 || mode == 0 || !staging || !strides_ok ||
      !scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt, groups, split) ||
      (exact && !scores_bt_supported(Mq, Mk, D, ldq, ldk, ldv, ldp, Q, K, V, P, Vt, 1, split))) {
    for (int g = 0; g < groups; ++g) {
      const int rc = hvr_relation_fwd((const char*)Q + (size_t)g * gsq * es, ldq, (const char*)K + (size_t)g * gsk * es, ldk,
                                      (const char*)V + (size_t)g * gsv * es, ldv, (char*)O + (size_t)g * gso * es, ldo, Mq, Mk, D, scale,
                                      dtype, staging, (char*)ws + (size_t)g * per, per, stream);
      if (rc) return rc;
    }
    return HVR_OK;
  }
  // exact: every group's result is hvr_relation_fwd's bit for bit (one scores launch over all groups, its arithmetic per tile
  // unchanged; the apply pass per group) -- what the batched head's equality tests run
  const bool bt_apply = !exact && mode >= 2 && (two_byte || split) && apply_bt_supported(Mq, Mk, D, ldp, ldo, P, Vt, O, groups, split);
  ScoresBTParams b;
  b.Q = (const bf16_t*)Q; b.K = (const bf16_t*)K; b.P = (bf16_t*)P; b.mstat = mstat; b.lstat = lstat;
  b.V = (const bf16_t*)V; b.Vt = (bf16_t*)Vt; b.Mq = Mq; b.Mk = Mk; b.D = D; b.ntile = nt;
  b.ldq = ldq; b.ldk = ldk; b.ldv = ldv; b.ldp = ldp; b.sl2 = scale * 1.4426950408889634f; b.f16 = split ? 2 : dtype == HVR_F16;
  b.groups = groups; b.gs_q = gsq; b.gs_k = gsk; b.gs_v = gsv; b.gs_p = (long)(per / es); b.gs_vt = (long)(per / es); b.gs_stat = (long)(per / 4);
  b.int_max = bt_apply ? 1 : 0;
  int rc = check_launch(run_scores_bt(b, s), "relation (grouped): scores");
  if (rc) return rc;
  if (bt_apply) {
    ApplyBTParams a;
    a.P = (const bf16_t*)P; a.Vt = (const bf16_t*)Vt; a.mstat = mstat; a.lstat = lstat; a.O = (bf16_t*)O;
    a.Mq = Mq; a.D = D; a.ntile = nt; a.ldp = ldp; a.ldo = ldo; a.groups = groups;
    a.gs_p = b.gs_p; a.gs_vt = b.gs_vt; a.gs_stat = b.gs_stat; a.gs_o = gso; a.split = split ? 1 : (dtype == HVR_F16 ? 2 : 0);
    return check_launch(run_apply_bt(a, s), "relation (grouped): apply");
  }
  for (int g = 0; g < groups; ++g) {
    char* wg = (char*)ws + (size_t)g * per;
    if (split) {   // one persistent scores launch over all groups (V^T included); the normalising sweep and the product per group
      rc = relation_split_tail(wg + ((char*)P - (char*)ws), wg + ((char*)Vt - (char*)ws), (float*)(wg + ((char*)mstat - (char*)ws)),
                               (float*)(wg + ((char*)lstat - (char*)ws)), nullptr, ldv,
                               (char*)O + (size_t)g * gso * es, ldo, Mq, Mk, D, ldp, nt, staging, s);
      if (rc) return rc;
      continue;
    }
    rc = relation_apply_pass(wg + ((char*)P - (char*)ws), wg + ((char*)Vt - (char*)ws), (float*)(wg + ((char*)mstat - (char*)ws)),
                             (float*)(wg + ((char*)lstat - (char*)ws)), (float*)(wg + ((char*)partial - (char*)ws)),
                             (char*)O + (size_t)g * gso * es, ldo, Mq, Mk, D, ldp, nt, dtype, staging, s);
    if (rc) return rc;
  }
  return HVR_OK;
}

// ---- relation backward (attention backward of one stage) ----
size_t hvr_relation_probs_workspace_bytes(int Mq, int Mk) {
  const long nt = rel_ldp(Mk) / 128;
  return 2 * align256((size_t)Mq * nt * 4);
}

int hvr_relation_probs(const void* Q, int64_t ldq, const void* K, int64_t ldk, void* P, int64_t ldp, int Mq, int Mk, int D,
                       float scale, int dtype, int staging, void* ws, size_t ws_bytes, void* stream) {
  if (!Q || !K || !P || !ws) return fail(HVR_EINVAL, "null pointer");
  if (Mq <= 0 || Mk <= 0) return fail(HVR_EINVAL, "empty relation Mq=%d Mk=%d", Mq, Mk);
  if (!(scale > 0.f)) return fail(HVR_EINVAL, "relation scale must be positive, got %g", (double)scale);
  if (ldp != rel_ldp(Mk)) return fail(HVR_EINVAL, "P rows must be %ld elements (keys padded to 128), got %ld", rel_ldp(Mk), (long)ldp);
  if (ws_bytes < hvr_relation_probs_workspace_bytes(Mq, Mk)) return fail(HVR_EWORKSPACE, "relation probs workspace too small");
  const int nt = (int)(ldp / 128);
  float* mstat = (float*)ws;
  float* lstat = (float*)((char*)ws + align256((size_t)Mq * nt * 4));
  hipStream_t s = (hipStream_t)stream;
  GemmParams p;
  int rc = fill_linear(p, Q, K, P, Mq, (Mk + 3) / 4 * 4, D, ldq, ldk, ldp, dtype, staging);
  if (rc) return rc;
  p.N = Mk;
  p.scale = scale; p.mstat = mstat; p.lstat = lstat; p.ntile = nt; p.group_m = 8;
  rc = check_launch(run_tile_op(p, EPI_SCORES, s), "relation probs: scores");
  if (rc) return rc;
  return check_launch(run_relation_normalize(P, mstat, lstat, Mq, nt, ldp, dtype, 1.f / kSplitProbScale, s), "relation probs: normalise");
}

int hvr_relation_dscore(const void* P, const void* dP, const void* dO, int64_t ldgo, const void* O, int64_t ldo, void* dS,
                        int Mq, int64_t ldp, int D, float scale, int dtype, void* stream) {
  if (!P || !dP || !dO || !O || !dS) return fail(HVR_EINVAL, "null pointer");
  if (Mq <= 0 || D <= 0 || D % 4 || ldp % 4 || ldgo % 4 || ldo % 4) return fail(HVR_EINVAL, "bad relation dscore shape");
  return check_launch(run_relation_dscore(P, dP, dO, O, dS, Mq, ldp, D, ldgo, ldo, scale, dtype, (hipStream_t)stream),
                      "hvr_relation_dscore");
}

int hvr_im2col_nhwc(const void* x, void* cols, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int dtype, void* stream) {
  if (!x || !cols || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 4 || KH <= 0 || KW <= 0 || pad < 0 || dil <= 0)
    return fail(HVR_EINVAL, "bad im2col arguments (Cin %% 4 == 0 required)");
  const int OH = H + 2 * pad - dil * (KH - 1), OW = W + 2 * pad - dil * (KW - 1);
  if (OH <= 0 || OW <= 0) return fail(HVR_EINVAL, "empty im2col output");
  return check_launch(run_im2col_nhwc(x, cols, B, H, W, Cin, KH, KW, pad, dil, OH, OW, dtype, (hipStream_t)stream), "hvr_im2col_nhwc");
}

int hvr_im2col_t(const void* x, void* colsT, int64_t ldt, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int dtype, void* stream) {
  if (!x || !colsT || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 8 || KH <= 0 || KW <= 0 || pad < 0 || dil <= 0)
    return fail(HVR_EINVAL, "bad im2col_t arguments (Cin %% 8 == 0 required)");
  if (dtype != HVR_BF16 && dtype != HVR_F16) return fail(HVR_EUNSUPPORTED, "hvr_im2col_t takes bf16 / half maps (f32: hvr_im2col_nhwc + hvr_transpose_pad)");
  const int OH = H + 2 * pad - dil * (KH - 1), OW = W + 2 * pad - dil * (KW - 1);
  if (OH <= 0 || OW <= 0) return fail(HVR_EINVAL, "empty im2col output");
  if (ldt % 8 || ldt < (int64_t)B * OH * OW || !aligned16(x) || !aligned16(colsT)) return fail(HVR_EINVAL, "im2col_t: ldt a multiple of 8 >= B*OH*OW, 16-byte aligned buffers");
  return check_launch(run_im2col_t(x, colsT, B, H, W, Cin, KH, KW, pad, dil, OH, OW, (long)ldt, (hipStream_t)stream), "hvr_im2col_t");
}

int hvr_relu_bwd_t(const void* dY, const void* Y, void* dZ, void* dZt, int64_t ldt, int R, int C, int dtype, void* stream) {
  if (!dY || !Y || !dZ || !dZt || R <= 0 || C <= 0 || C % 8 || ldt % 8 || ldt < R) return fail(HVR_EINVAL, "bad relu_bwd_t arguments (C %% 8 == 0, ldt %% 8 == 0, ldt >= R)");
  if (dtype != HVR_BF16 && dtype != HVR_F16) return fail(HVR_EUNSUPPORTED, "hvr_relu_bwd_t takes bf16 / half maps (f32: hvr_relu_bwd + hvr_transpose_pad)");
  if (!aligned16(dY) || !aligned16(Y) || !aligned16(dZ) || !aligned16(dZt)) return fail(HVR_EINVAL, "relu_bwd_t: 16-byte aligned buffers");
  return check_launch(run_relu_bwd_t(dY, Y, dZ, dZt, R, C, (long)ldt, (hipStream_t)stream), "hvr_relu_bwd_t");
}

int hvr_scale_rows(const void* w, const float* scale, void* out, int R, int64_t C, int dtype, void* stream) {
  if (!w || !scale || !out || R <= 0 || C <= 0 || C % 4) return fail(HVR_EINVAL, "bad scale_rows arguments (C %% 4 == 0 required)");
  return check_launch(run_scale_rows(w, scale, out, R, C, dtype, (hipStream_t)stream), "hvr_scale_rows");
}

size_t hvr_sgd_workspace_bytes(void) { return sgd_workspace_bytes(); }

// ---- head training helpers ----
int hvr_relu_bwd(const void* dY, const void* Y, void* dZ, int64_t n, int dtype, void* stream) {
  if (n == 0) return HVR_OK;
  if (!dY || !Y || !dZ || n < 0 || n % 4) return fail(HVR_EINVAL, "bad relu_bwd arguments (n %% 4 == 0 required)");
  return check_launch(run_relu_bwd(dY, Y, dZ, n, dtype, (hipStream_t)stream), "hvr_relu_bwd");
}

size_t hvr_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const int s = colsum_slices(M, N);
  return s > 1 ? (size_t)s * N * sizeof(float) : 0;
}

int hvr_colsum(const void* dY, float* db, int M, int N, int64_t ld, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!dY || !db || M <= 0 || N <= 0) return fail(HVR_EINVAL, "bad colsum arguments");
  const size_t need = hvr_colsum_workspace_bytes(M, N);
  if (need && (!ws || ws_bytes < need)) return fail(HVR_EWORKSPACE, "colsum workspace too small");
  return check_launch(run_colsum(dY, db, M, N, ld, dtype, (float*)ws, (hipStream_t)stream), "hvr_colsum");
}

int hvr_pack_conv_weight(const float* w, const float* scale, void* out, int Cout, int Cin, int KH, int KW, int out_dtype, void* stream) {
  if (!w || !scale || !out || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return fail(HVR_EINVAL, "bad pack_conv_weight arguments");
  if (out_dtype != HVR_F32 && out_dtype != HVR_BF16 && out_dtype != HVR_F16) return fail(HVR_EINVAL, "bad dtype");
  return check_launch(run_pack_conv_weight(w, scale, out, Cout, Cin, KH * KW, out_dtype, (hipStream_t)stream), "hvr_pack_conv_weight");
}

int hvr_pack_conv_weights_multi(const hvr_pack_item* items_dev, int n, int64_t total, int out_dtype, void* stream) {
  if (!items_dev || n <= 0 || total <= 0) return fail(HVR_EINVAL, "bad pack_conv_weights_multi arguments");
  if (out_dtype != HVR_BF16 && out_dtype != HVR_F16) return fail(HVR_EUNSUPPORTED, "hvr_pack_conv_weights_multi writes bf16 / half operands");
  return check_launch(run_pack_conv_weights_multi(items_dev, n, (long)total, out_dtype, (hipStream_t)stream), "hvr_pack_conv_weights_multi");
}

int hvr_transpose_multi(const hvr_transpose_item* items_dev, int n, int tiles, void* stream) {
  if (!items_dev || n <= 0 || tiles <= 0) return fail(HVR_EINVAL, "bad transpose_multi arguments");
  return check_launch(run_transpose_multi(items_dev, n, tiles, (hipStream_t)stream), "hvr_transpose_multi");
}

int hvr_unpack_conv_wgrads_multi(const hvr_unpack_item* items_dev, int n, int64_t total, int accumulate, void* stream) {
  if (!items_dev || n <= 0 || total <= 0) return fail(HVR_EINVAL, "bad unpack_conv_wgrads_multi arguments");
  return check_launch(run_unpack_conv_wgrads_multi(items_dev, n, (long)total, accumulate, (hipStream_t)stream), "hvr_unpack_conv_wgrads_multi");
}

int hvr_unpack_conv_wgrad(const float* dw, const float* scale, float* out, int Cout, int Cin, int KH, int KW, int accumulate, void* stream) {
  if (!dw || !scale || !out || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return fail(HVR_EINVAL, "bad unpack_conv_wgrad arguments");
  return check_launch(run_unpack_conv_wgrad(dw, scale, out, Cout, Cin, KH * KW, accumulate, (hipStream_t)stream), "hvr_unpack_conv_wgrad");
}

int hvr_det_loss(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const int64_t* labels, const float* label_weights,
                 const float* bbox_targets, const float* bbox_weights, int R, float beta, float w_cls, float w_bbox, float* out3,
                 float* dlogits, void* stream) {
  if (!logits || !labels || !label_weights || !bbox_targets || !bbox_weights || !out3 || !dlogits) return fail(HVR_EINVAL, "null pointer");
  if (R <= 0 || ncls <= 1 || cls_off < 0 || reg_off < 0 || cls_off + ncls > ldl || reg_off + 4 > ldl || !(beta > 0.f))
    return fail(HVR_EINVAL, "bad det_loss shape");
  return check_launch(run_det_loss(logits, ldl, cls_off, reg_off, ncls, (const long long*)labels, label_weights, bbox_targets,
                                   bbox_weights, R, beta, w_cls, w_bbox, out3, dlogits, nullptr, (hipStream_t)stream),
                      "hvr_det_loss");
}

int hvr_det_loss_sampled(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const int64_t* labels,
                         const float* label_weights, const float* bbox_targets, const float* bbox_weights, int R,
                         const int32_t* sel_counts, float beta, float* out3, float* dlogits, void* stream) {
  if (!logits || !labels || !label_weights || !bbox_targets || !bbox_weights || !out3 || !dlogits || !sel_counts)
    return fail(HVR_EINVAL, "null pointer");
  if (R <= 0 || ncls <= 1 || cls_off < 0 || reg_off < 0 || cls_off + ncls > ldl || reg_off + 4 > ldl || !(beta > 0.f))
    return fail(HVR_EINVAL, "bad det_loss shape");
  return check_launch(run_det_loss(logits, ldl, cls_off, reg_off, ncls, (const long long*)labels, label_weights, bbox_targets,
                                   bbox_weights, R, beta, 1.f, 1.f, out3, dlogits, sel_counts, (hipStream_t)stream),
                      "hvr_det_loss_sampled");
}

size_t hvr_max_iou_assign_workspace_bytes(int n, int k) { return n > 0 && k > 0 ? assign_workspace_bytes(n, k) : 0; }

int hvr_max_iou_assign(const float* boxes, int ldb, int n, const float* gts, int k, const uint8_t* valid, float pos_iou_thr,
                       float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int64_t* gt_inds, float* max_overlaps, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!boxes || !gts || !gt_inds || !max_overlaps || !ws) return fail(HVR_EINVAL, "null pointer");
  if (n <= 0 || k <= 0) return fail(HVR_EINVAL, "No gt or bboxes");  // max_iou_assigner.py:77-78 raises ValueError
  if (k > 256) return fail(HVR_EUNSUPPORTED, "at most 256 ground-truth boxes per assignment (got %d)", k);
  if (ldb < 4) return fail(HVR_EINVAL, "boxes need 4 coordinates per row");
  if (ws_bytes < assign_workspace_bytes(n, k)) return fail(HVR_EINVAL, "assign workspace too small");
  return check_launch(run_max_iou_assign(boxes, ldb, n, gts, k, valid, pos_iou_thr, neg_iou_lo, neg_iou_hi, min_pos_iou,
                                         (long long*)gt_inds, max_overlaps, ws, (hipStream_t)stream),
                      "hvr_max_iou_assign");
}

int hvr_sample_pos_neg(const int64_t* cls, const float* keys, int n, int num, int num_expected_pos, float neg_pos_ub, int64_t* inds,
                       int32_t* counts, void* stream) {
  if (!cls || !keys || !inds || !counts) return fail(HVR_EINVAL, "null pointer");
  if (n <= 0 || num <= 0 || num_expected_pos < 0 || num_expected_pos > num) return fail(HVR_EINVAL, "bad sampler sizes");
  return check_launch(run_sample((const long long*)cls, keys, n, num, num_expected_pos, neg_pos_ub, (long long*)inds, counts,
                                 (hipStream_t)stream),
                      "hvr_sample_pos_neg");
}

int hvr_box_targets(const float* boxes, int ldb, int n, const float* gts, const int64_t* gt_labels, const int64_t* gt_inds,
                    const int64_t* inds, const int32_t* counts, int num, const float* means4, const float* stds4, float pos_weight,
                    int scatter, int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights, void* stream) {
  if (!boxes || !gts || !gt_inds || !inds || !counts || !means4 || !stds4 || !labels || !label_weights || !bbox_targets || !bbox_weights)
    return fail(HVR_EINVAL, "null pointer");
  if (n <= 0 || num <= 0 || ldb < 4) return fail(HVR_EINVAL, "bad box_targets sizes");
  for (int i = 0; i < 4; ++i)
    if (stds4[i] == 0.f) return fail(HVR_EINVAL, "target_stds must be non-zero");
  return check_launch(run_box_targets(boxes, ldb, n, gts, (const long long*)gt_labels, (const long long*)gt_inds, (const long long*)inds,
                                      counts, num, means4, stds4, pos_weight, scatter, (long long*)labels, label_weights, bbox_targets,
                                      bbox_weights, (hipStream_t)stream),
                      "hvr_box_targets");
}

int hvr_rpn_loss(const float* o, int ldo, int A, int rows, const int64_t* labels, const float* label_weights, const float* bbox_targets,
                 const float* bbox_weights, const int32_t* counts, float beta, float* out2, float* d_o, void* stream) {
  if (!o || !labels || !label_weights || !bbox_targets || !bbox_weights || !counts || !out2 || !d_o) return fail(HVR_EINVAL, "null pointer");
  if (A <= 0 || rows <= 0 || ldo < 5 * A || !(beta > 0.f)) return fail(HVR_EINVAL, "bad rpn_loss shape");
  return check_launch(run_rpn_loss(o, ldo, A, rows, (const long long*)labels, label_weights, bbox_targets, bbox_weights, counts, beta,
                                   out2, d_o, (hipStream_t)stream),
                      "hvr_rpn_loss");
}

int hvr_mining_argreduce(const float* aff, int Mq, int Mk, int64_t ld, const int64_t* labels, const int64_t* all_labels, int64_t* out4,
                         void* stream) {
  if (!aff || !labels || !all_labels || !out4) return fail(HVR_EINVAL, "null pointer");
  if (Mq <= 0 || Mk <= 0 || ld < Mk) return fail(HVR_EINVAL, "bad mining shape");
  return check_launch(run_mining_argreduce(aff, Mq, Mk, ld, (const long long*)labels, (const long long*)all_labels, (long long*)out4,
                                           (hipStream_t)stream),
                      "hvr_mining_argreduce");
}

int hvr_triplet_margin(const void* q, int64_t ldq, const void* k, int64_t ldk, int D, int Mq, int Mk, const int64_t* anchor_idx,
                       const int64_t* pos_idx, const int64_t* neg_idx, int n, float margin, int dtype, float* ws, size_t ws_bytes,
                       float* out2, float* dq, float* dk, void* stream) {
  if (!q || !k || !anchor_idx || !pos_idx || !neg_idx || !ws || !out2) return fail(HVR_EINVAL, "null pointer");
  if (D <= 0 || Mq <= 0 || Mk <= 0 || n <= 0 || ldq < D || ldk < D) return fail(HVR_EINVAL, "bad triplet shape");
  if (dtype != HVR_F32 && dtype != HVR_BF16) return fail(HVR_EINVAL, "bad dtype");
  if (ws_bytes < (size_t)n * 3 * sizeof(float)) return fail(HVR_EWORKSPACE, "triplet workspace too small");
  return check_launch(run_triplet_margin(q, ldq, k, ldk, D, Mq, Mk, (const long long*)anchor_idx, (const long long*)pos_idx,
                                         (const long long*)neg_idx, n, margin, dtype == HVR_BF16, ws, out2, dq, dk, (hipStream_t)stream),
                      "hvr_triplet_margin");
}

int hvr_ingest_frame(const uint8_t* src, int src_h, int src_w, int64_t src_pitch, float* dst, int new_h, int new_w, int pad_h, int pad_w,
                     const float* mean3, const float* std3, int to_rgb, void* stream) {
  if (!src || !dst || !mean3 || !std3) return fail(HVR_EINVAL, "null pointer");
  if (src_h <= 0 || src_w <= 0 || new_h <= 0 || new_w <= 0 || pad_h < new_h || pad_w < new_w || src_pitch < 3L * src_w)
    return fail(HVR_EINVAL, "bad ingest shape");
  for (int c = 0; c < 3; ++c)
    if (std3[c] == 0.f) return fail(HVR_EINVAL, "std must be non-zero");
  return check_launch(run_ingest(src, src_h, src_w, src_pitch, dst, new_h, new_w, pad_h, pad_w, mean3, std3, to_rgb, (hipStream_t)stream),
                      "hvr_ingest_frame");
}

int hvr_ce_rows(const float* logits, int ldl, int cls_off, int ncls, const int64_t* labels, int R, float* loss, void* stream) {
  if (!logits || !labels || !loss) return fail(HVR_EINVAL, "null pointer");
  if (R <= 0 || ncls <= 1 || cls_off < 0 || cls_off + ncls > ldl) return fail(HVR_EINVAL, "bad ce_rows shape");
  return check_launch(run_ce_rows(logits, ldl, cls_off, ncls, (const long long*)labels, R, loss, (hipStream_t)stream), "hvr_ce_rows");
}

int hvr_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                 float grad_scale, float max_norm, float* ws, size_t ws_bytes, int first_step, void* stream) {
  if (n == 0) return HVR_OK;
  if (!param || !grad || !momentum_buf || !ws || n < 0) return fail(HVR_EINVAL, "bad sgd_step arguments");
  if (ws_bytes < hvr_sgd_workspace_bytes()) return fail(HVR_EWORKSPACE, "sgd workspace too small");
  return check_launch(run_sgd_step(param, grad, momentum_buf, n, lr, momentum, weight_decay, grad_scale, max_norm, ws, first_step,
                                   (hipStream_t)stream),
                      "hvr_sgd_step");
}

// ---- RoIAlign ----
int hvr_roi_align_fwd(const void* feat, const float* rois, void* out, int B, int C, int H, int W, int K, int PH, int PW,
                      float spatial_scale, int sample_num, int dtype, int layout, void* stream) {
  if (K == 0) return HVR_OK;
  if (!feat || !rois || !out) return fail(HVR_EINVAL, "null pointer");
  if (K < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return fail(HVR_EINVAL, "bad roi_align shape");
  if (layout != HVR_LAYOUT_NCHW && layout != HVR_LAYOUT_NHWC) return fail(HVR_EINVAL, "bad layout %d", layout);
  return check_launch(run_roi_align_fwd(feat, rois, out, B, C, H, W, K, PH, PW, spatial_scale, sample_num, dtype, layout,
                                        (hipStream_t)stream),
                      "hvr_roi_align_fwd");
}

int hvr_roi_align_bwd(const float* grad_out, const float* rois, float* grad_feat, int B, int C, int H, int W, int K, int PH,
                      int PW, float spatial_scale, int sample_num, int layout, void* stream) {
  (void)B;
  if (K == 0) return HVR_OK;
  if (!grad_out || !rois || !grad_feat) return fail(HVR_EINVAL, "null pointer");
  return check_launch(run_roi_align_bwd(grad_out, rois, grad_feat, C, H, W, K, PH, PW, spatial_scale, sample_num, layout,
                                        (hipStream_t)stream),
                      "hvr_roi_align_bwd");
}

// ---- NMS ----
size_t hvr_nms_workspace_bytes(int n) { return nms_workspace_bytes(1, n) + 256; }

int hvr_nms(const float* dets, int n, float thr, int ge_semantics, int64_t* keep, int32_t* n_keep, void* ws,
            size_t ws_bytes, void* stream) {
  if (!n_keep) return fail(HVR_EINVAL, "null n_keep");
  if (n == 0) {
    (void)run_zero_fill(n_keep, sizeof(int32_t), (hipStream_t)stream);
    return HVR_OK;
  }
  if (!dets || !keep || !ws) return fail(HVR_EINVAL, "null pointer");
  if (n < 0 || n > 8192) return fail(HVR_EUNSUPPORTED, "hvr_nms supports 0 <= n <= 8192, got %d", n);
  if (ws_bytes < hvr_nms_workspace_bytes(n)) return fail(HVR_EWORKSPACE, "nms workspace too small");
  return check_launch(run_nms_batched(dets, 1, n, thr, ge_semantics, 0, 0, (long long*)keep, n_keep, ws, (hipStream_t)stream),
                      "hvr_nms");
}

int hvr_nms_first(const float* dets, int n, float thr, int ge_semantics, int max_keep, int64_t* keep, int32_t* n_keep, void* ws,
                  size_t ws_bytes, void* stream) {
  if (max_keep <= 0) return fail(HVR_EINVAL, "hvr_nms_first needs max_keep > 0 (hvr_nms keeps every survivor)");
  if (!n_keep) return fail(HVR_EINVAL, "null n_keep");
  if (n == 0) {
    (void)run_zero_fill(n_keep, sizeof(int32_t), (hipStream_t)stream);
    return HVR_OK;
  }
  if (!dets || !keep || !ws) return fail(HVR_EINVAL, "null pointer");
  if (n < 0 || n > 8192) return fail(HVR_EUNSUPPORTED, "hvr_nms_first supports 0 <= n <= 8192, got %d", n);
  if (ws_bytes < hvr_nms_workspace_bytes(n)) return fail(HVR_EWORKSPACE, "nms workspace too small");
  return check_launch(run_nms_batched(dets, 1, n, thr, ge_semantics, 0, max_keep, (long long*)keep, n_keep, ws, (hipStream_t)stream),
                      "hvr_nms_first");
}

// ---- RPN ----
static inline int rpn_npre(int H, int W, int A, int nms_pre) {
  const long n = (long)H * W * A;
  return (nms_pre > 0 && n > nms_pre) ? nms_pre : (int)n;
}

size_t hvr_rpn_workspace_bytes(int T, int H, int W, int A, int nms_pre) {
  const int npre = rpn_npre(H, W, A, nms_pre);
  return align256((size_t)T * npre * 5 * 4) + align256((size_t)T * npre * 8) + align256((size_t)T * 4) +
         nms_workspace_bytes(T, npre) + rpn_wide_workspace_bytes(T, (long)H * W * A) + 256;
}

int hvr_rpn_wide_frames(int frames) { return rpn_wide_max_frames(frames); }

int hvr_rpn_proposals(const hvr_rpn_desc* d, void* ws, size_t ws_bytes, void* stream) {
  if (!d || !ws) return fail(HVR_EINVAL, "null pointer");
  if (d->A > 32) return fail(HVR_EUNSUPPORTED, "at most 32 base anchors");
  const long n_anchor = (long)d->H * d->W * d->A;
  const int npre = rpn_npre(d->H, d->W, d->A, d->nms_pre);
  if (npre > 8192) return fail(HVR_EUNSUPPORTED, "nms_pre (or anchor count) above 8192");
  if (d->nms_post > 4096) return fail(HVR_EUNSUPPORTED, "nms_post above 4096");
  if (ws_bytes < hvr_rpn_workspace_bytes(d->T, d->H, d->W, d->A, d->nms_pre)) return fail(HVR_EWORKSPACE, "rpn workspace too small");
  RpnParams rp;
  rp.T = d->T; rp.H = d->H; rp.W = d->W; rp.A = d->A; rp.npre = npre; rp.n_anchor = (int)n_anchor;
  rp.cls_pitch = d->cls_pitch > 0 ? d->cls_pitch : d->A;
  rp.reg_pitch = d->reg_pitch > 0 ? d->reg_pitch : 4 * d->A;
  if (rp.cls_pitch < d->A || rp.reg_pitch < 4 * d->A || (rp.reg_pitch & 3)) return fail(HVR_EINVAL, "bad rpn pixel pitch");
  rp.stride = d->anchor_stride; rp.img_h = d->img_h; rp.img_w = d->img_w;
  for (int i = 0; i < 4; ++i) { rp.m[i] = d->means[i]; rp.s[i] = d->stds[i]; }
  rp.max_ratio = std::fabs(std::log(d->wh_ratio_clip));
  for (int a = 0; a < d->A; ++a)
    for (int i = 0; i < 4; ++i) rp.base[a][i] = d->base_anchors[a * 4 + i];
  char* w = (char*)ws;
  float* props = (float*)w;          w += align256((size_t)d->T * npre * 5 * 4);
  long long* keep = (long long*)w;   w += align256((size_t)d->T * npre * 8);
  int* n_keep = (int*)w;             w += align256((size_t)d->T * 4);
  hipStream_t s = (hipStream_t)stream;
  const int presorted = n_anchor > npre ? 1 : 0;
  // few frames (stream mode: one): the chip-wide kernels -- the same proposals, bit for bit (nms.hip, "the chip-wide form")
  const bool wide = presorted && d->T <= rpn_wide_max_frames(-1) && d->nms_post > 0 && d->nms_post <= 1024;
  void* wws = w + nms_workspace_bytes(d->T, npre);
  int rc;
  if (wide)
    rc = check_launch(run_rpn_select_wide(d->cls, d->reg, (long)d->H * d->W * rp.cls_pitch, (long)d->H * d->W * rp.reg_pitch, props, rp, wws, s), "rpn: select (chip-wide)");
  else
    rc = check_launch(run_rpn_select(d->cls, d->reg, (long)d->H * d->W * rp.cls_pitch, (long)d->H * d->W * rp.reg_pitch, props, rp, HVR_F32, s), "rpn: select");
  if (rc) return rc;
  // with score-sorted input the first nms_post survivors are known after nms_post keeps
  rc = check_launch(run_nms_batched(props, d->T, npre, d->nms_thr, 1, presorted, presorted ? d->nms_post : 0, keep, n_keep, w, s,
                                    wide ? 4096 : 0, wide ? rpn_wide_done_flags(wws, d->T) : nullptr),
                    "rpn: nms");
  if (rc) return rc;
  return check_launch(run_rpn_gather(props, keep, n_keep, d->T, npre, d->nms_post, d->max_num, presorted, d->proposals, d->counts, s),
                      "rpn: gather");
}

// ---- RCNN read-out ----
int hvr_det_decode(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const float* rois, int R,
                   const float* means, const float* stds, float wh_ratio_clip, float img_h, float img_w,
                   float scale_factor, float* scores, float* boxes, void* stream) {
  if (R == 0) return HVR_OK;
  if (!logits || !rois || !scores || !boxes || !means || !stds) return fail(HVR_EINVAL, "null pointer");
  return check_launch(run_det_decode(logits, ldl, cls_off, reg_off, ncls, rois, R, means, stds,
                                     std::fabs(std::log(wh_ratio_clip)), img_h, img_w, scale_factor, scores, boxes,
                                     (hipStream_t)stream),
                      "hvr_det_decode");
}

size_t hvr_multiclass_nms_workspace_bytes(int R, int ncls) { return multiclass_nms_workspace_bytes(R, ncls); }

int hvr_multiclass_nms(const float* boxes, const float* scores, int R, int ncls, float score_thr, float iou_thr, int max_num,
                       float* dets, int64_t* labels, int32_t* n_out, void* ws, size_t ws_bytes, void* stream) {
  if (!n_out) return fail(HVR_EINVAL, "null n_out");
  if (R == 0) {
    (void)run_zero_fill(n_out, sizeof(int32_t), (hipStream_t)stream);
    return HVR_OK;
  }
  if (!boxes || !scores || !dets || !labels || !ws) return fail(HVR_EINVAL, "null pointer");
  if (R > 512) return fail(HVR_EUNSUPPORTED, "hvr_multiclass_nms supports R <= 512, got %d", R);
  if (ws_bytes < hvr_multiclass_nms_workspace_bytes(R, ncls)) return fail(HVR_EWORKSPACE, "multiclass nms workspace too small");
  return check_launch(run_multiclass_nms(boxes, scores, R, ncls, score_thr, iou_thr, max_num, dets, (long long*)labels, n_out,
                                         ws, (hipStream_t)stream),
                      "hvr_multiclass_nms");
}

// ---- plumbing ----
int hvr_cast(const void* in, void* out, int64_t n, int from_dtype, int to_dtype, void* stream) {
  return hvr_cast_scaled(in, out, n, from_dtype, to_dtype, 1.f, stream);
}

int hvr_cast_scaled(const void* in, void* out, int64_t n, int from_dtype, int to_dtype, float scale, void* stream) {
  if (n == 0) return HVR_OK;
  if (!in || !out) return fail(HVR_EINVAL, "null pointer");
  if (!valid_dtype(from_dtype) || !valid_dtype(to_dtype) || (from_dtype == to_dtype && (scale == 1.f || from_dtype != HVR_F32)))
    return fail(HVR_EINVAL, "bad cast %d -> %d", from_dtype, to_dtype);
  if (!(scale > 0.f)) return fail(HVR_EINVAL, "cast scale must be positive");
  const bool classic = scale == 1.f && ((from_dtype == HVR_F32 && to_dtype == HVR_BF16) || (from_dtype == HVR_BF16 && to_dtype == HVR_F32));
  if (!classic) {   // 8 elements per thread; split half moves whole 64-element groups
    const bool split = from_dtype == HVR_F16S || to_dtype == HVR_F16S;   // split half moves whole 32-element groups
    if (n % (split ? 32 : 8) || !aligned16(in) || !aligned16(out) || (from_dtype == HVR_F16S && !aligned128(in)) || (to_dtype == HVR_F16S && !aligned128(out)))
      return fail(HVR_EINVAL, "hvr_cast to / from half formats: n %% 8 == 0 (split half: n %% 32 == 0, 128-byte aligned), 16-byte aligned buffers");
  }
  return check_launch(run_cast(in, out, n, from_dtype, to_dtype, scale, (hipStream_t)stream), "hvr_cast");
}

int hvr_permute_nchw_nhwc(const void* in, void* out, int B, int C, int HW, int to_nhwc, int from_dtype, int to_dtype,
                          void* stream) {
  if (!in || !out || B <= 0 || C <= 0 || HW <= 0) return fail(HVR_EINVAL, "bad permute arguments");
  return check_launch(run_permute(in, out, B, C, HW, to_nhwc, from_dtype, to_dtype, (hipStream_t)stream), "hvr_permute");
}

int hvr_transpose_pad(const void* in, void* out, int R, int C, int64_t ldx, int64_t ldt, int dtype, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || ldt < R) return fail(HVR_EINVAL, "bad transpose arguments");
  return check_launch(run_transpose_pad(in, out, R, C, ldx, ldt, dtype, (hipStream_t)stream), "hvr_transpose_pad");
}

}  // extern "C"
