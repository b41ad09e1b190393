// Kernel-argument block of the MFMA tile engine (internal; the public ABI is include/hvr_hip.h).
#pragma once
#include <hip/hip_runtime.h>

namespace hvr {

// EPI_LINEAR2: EPI_LINEAR with TWO-LEVEL accumulation -- every 8 K-steps' products sum in a block accumulator that is then added to the running
// total (the apply pass's two accumulator sets without the block weights): a long-K product's f32 rounding noise no longer grows with the
// number of MFMAs chained on one accumulator (the RPN's 3x3 conv: K = 9 216 = 864 chained additions in split half; tile_hint kTwoLevelHint)
enum { EPI_LINEAR = 0, EPI_SCORES = 1, EPI_APPLY = 2, EPI_LINEAR2 = 3 };

// RPN proposal selection parameters (nms.hip)
struct RpnParams {
  int T, H, W, A, npre, n_anchor;
  int cls_pitch, reg_pitch;  // channels per pixel of the maps holding objectness / deltas
  int stride;
  float img_h, img_w;
  float m[4], s[4];
  float max_ratio;
  float base[32][4];  // base anchors (<= 32)
};

struct GemmParams {
  const void* A;  // [M][lda] or NHWC activation when conv != 0
  const void* B;  // [N][ldb], K contiguous
  void* C;        // [M][ldc]
  int M, N, K;    // K is a multiple of 128 bytes / sizeof(elem)
  long lda, ldb, ldc;
  int dtype;    // DT_F32 / DT_BF16 (operands; accumulation is always f32)
  int staging;  // 0 = register-staged global->LDS, 1 = direct global_load_lds
  int tile_hint;  // 0 = choose by cost model, k > 0 = force tile shape k-1 (tuning / tests)
  // implicit-GEMM gather on the A side
  int conv, H, W, Cin, OH, OW, KH, KW, stride, pad, dil;
  const void* zero;  // >= 16 readable zero bytes (padding taps)
  // EPI_LINEAR
  const float* bias;  // [N] or null
  const void* resid;  // [M][ldr] (operand dtype) or null
  long ldr;
  int relu, out_f32;
  float alpha;  // split-half products (EPI_LINEAR): the accumulators are multiplied by alpha (a power of two: the caller stores small-
                // magnitude operands -- weights, probabilities -- scaled up so that their lo halves are normal numbers); 0 = 1
  float beta;   // split-half products: factor on the bias (a caller that keeps its activations scaled hands the bias in true units); 0 = 1
  // EPI_SCORES / EPI_APPLY
  float scale;
  float* mstat;    // [M][ntile] tile max (log2 units)
  float* lstat;    // [M][ntile] tile sum
  int ntile;
  int group_m;  // > 1: tile order walks `group_m` row tiles per B panel (see tile_kernel)
  // second K segment of the expand kernel (expand.hip, a Bottleneck's projection shortcut folded into its closing 1x1):
  // columns K1 .. K - 1 of a row come from A2, an NHWC map [.][H2][W2][K - K1] sampled at stride s2 (0 = no second segment)
  const void* A2;
  int K1, H2, W2, s2;
  // the NEXT block's reducing 1x1 computed on the expand kernel's output while it is still in registers (expand.hip, NX > 0):
  // Hn [M][Cn] = relu(out Wn^T + bias_n), Wn [Cn][N] (K contiguous), Cn = 64 / 128; null = off
  const void* Wn;
  const float* bias_n;
  void* Hn;
  int Cn;
  // split-K (EPI_LINEAR, no conv): ksplit_count slices of ksplit_steps K-steps, partial s at C + s * csplit_bytes (0 = off)
  int ksplit_steps, ksplit_count;
  long csplit_bytes;
  // EPI_SCORES, two-byte operands: tr_blocks > 0 workgroups behind the score tiles write tr_out[C][tr_ldt] = tr_in[R][tr_ldx]^T (zero for
  // columns R .. tr_ldt - 1) -- the apply pass's V^T, made by the CUs a few-row score grid leaves idle instead of by a launch of its own
  const void* tr_in;
  void* tr_out;
  int tr_R, tr_C, tr_blocks;
  long tr_ldx, tr_ldt;
  // batch > 1 (tile engine, plain products and the relation passes): `batch` independent problems of one shape in one launch, gridDim.z
  // selects the problem; problem g's operands sit gs_* BYTES behind problem 0's (A, B, C / the split-K partials, tr_in, tr_out) and its
  // block statistics gs_stat FLOATS behind (the key stage of hvr_relation_fwd_grouped: the clips of a call).  0 / 1 = one problem
  int batch;
  long gs_a, gs_b, gs_c, gs_stat, gs_tr_in, gs_tr_out;
};

struct TileShape { int bm, bn, wg_per_cu; float eff; };
constexpr int kNumTileShapes = 12;
constexpr int kNumBaseShapes = 5;  // shapes 5.. are deep-pipeline (3 / 4 LDS stage) variants of the base shapes
extern const TileShape kTileShapes[kNumTileShapes];
int choose_tile(const GemmParams& p, int epi);

hipError_t run_tile_op(const GemmParams& p, int epi, hipStream_t stream);
bool two_level_supported(const GemmParams& p);   // EPI_LINEAR2: exact-f32 / split-half operands, K a multiple of kTwoLevelSteps K-steps, no K slices

// Row-panel kernel for the channel-expanding 1x1 convs with a residual (expand.hip); tile_hint kExpandHint forces it
constexpr int kExpandHint = kNumTileShapes + 1;
bool expand_supported(const GemmParams& p);
// Persistent 3x3 / 64 -> 64 channel kernel with LDS-resident weights (conv3x3.hip: the layer-1 conv2)
bool conv3x3_c64_supported(const GemmParams& p);
hipError_t run_conv3x3_c64(const GemmParams& p, hipStream_t stream);
hipError_t run_expand(const GemmParams& p, hipStream_t stream);
bool expand_next_supported(const GemmParams& p);
// the same row-panel design on split-half operands (expand_split.hip): K = 64 / 128; with the next block's conv1 for stage 1 (N = 256, Cn = 64)
bool expand_split_supported(const GemmParams& p);
bool expand_split_next_supported(const GemmParams& p);
hipError_t run_expand_split(const GemmParams& p, hipStream_t stream);
// Producer / consumer tile kernel (pc_gemm.hip): 144 x 128 / 144 x 256 tiles, 4 compute + 4 DMA waves; EPI_LINEAR (plain
// GEMM) and EPI_APPLY.  tile_hint kPcHint128 / kPcHint256 force it for a plain GEMM (tuning / tests)
constexpr int kPcHint128 = kNumTileShapes + 2, kPcHint256 = kNumTileShapes + 3;
bool pc_supported(const GemmParams& p, int epi);
// Big-tile kernel (bigtile.hip): 288 x 256 tiles, one workgroup per CU on HALF the grid of the 144-row shapes -- for callers that
// keep several windows in flight (throughput, not latency: tile_hint kBigHint), and by default where its grid fills the chip by
// itself (N = 512); the tile engine runs everything else as with tile_hint 0
constexpr int kBigHint = kNumTileShapes + 4, kBigForce = kNumTileShapes + 5, kTwoLevelHint = kNumTileShapes + 6;   // (kTwoLevelHint: EPI_LINEAR2 on the 128 x 128 shape)
constexpr int kTwoLevelSteps = 8;   // K-steps per block of the two-level form  // (kBigForce: whatever the grid / epilogue -- tuning, tests)
bool bigtile_supported(const GemmParams& p, bool throughput);
hipError_t run_bigtile(const GemmParams& p, hipStream_t stream);
hipError_t run_pc(const GemmParams& p, int epi, int bn, hipStream_t stream);
// Few-row products with K sliced across the waves of one workgroup per 64 x 64 tile (kpar.hip): no partials in memory, no reduce launch
bool kpar_supported(const GemmParams& p);
hipError_t run_kpar(const GemmParams& p, hipStream_t stream);

}  // namespace hvr
