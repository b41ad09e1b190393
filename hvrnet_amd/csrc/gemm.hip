// MFMA tile engine for every dense contraction on the HVR forward path:
//   * fc layers / 1x1 convs / Q,K and output projections  -> EPI_LINEAR
//   * 3x3 / strided / dilated convs (implicit GEMM, NHWC)   -> EPI_LINEAR + conv gather
//   * relation scores  P~ = exp(scale*QK^T - tilemax)        -> EPI_SCORES
//   * relation apply   O  = sum_t g[:,t] * (P~_t V_t)        -> EPI_APPLY
//
// Reference op sequence being replaced (no reference kernel exists; the reference
// delegates to ATen/cuDNN): mmdet/models/bbox_heads/selsa_bbox_head.py:156-190,
// mmdet/models/backbones/resnet.py:220-266, mmdet/models/anchor_heads/rpn_head.py:30-35.
//
// Layout: A is [M][K] (or an NHWC activation gathered per filter tap), B is [N][K]
// (weights, K contiguous).  A K-step is 128 bytes of K per row (64 bf16 / 32 f32), so
// the LDS image, the loader and the XOR swizzle are byte-identical for both dtypes;
// only the fragment->MFMA step differs (16x16x32 bf16 vs exact-f32 16x16x4).
// Operands are fed to the MFMA swapped (weights as the "A" operand) so that each lane
// ends up with 4 consecutive n of one output row m -> 8/16-byte row-major stores.
#include "gemm_tile.h"

namespace hvr {

// Tile menu (BM x BN, waves).  Only the bf16 + global_load_lds path carries the whole menu; the
// register-staged and f32 paths (tests, parity mode) keep the two base shapes.
//   0: 128x128 (2x2 waves)   2 workgroups / CU      1: 128x64  (4x1)   3 / CU
//   2: 144x256 (3x2)         1 / CU                 3: 144x128 (3x2)   2 / CU
//   4: 256x128 (4x2)         1 / CU
// Deep-pipeline variants (NS LDS stages, NS - 1 K-steps of DMA in flight):
//   5: 256x128 NS=3  1 / CU    6: 144x256 NS=3  1 / CU    7: 144x128 NS=4  1 / CU
//   8: 128x128 NS=4  1 / CU    9: 128x64  NS=3  2 / CU
//  10: 144x256 NS=3, 8 waves as 1 x 8 (two waves on every SIMD; the 6-wave shapes leave two SIMDs with one)
//  11: 144x128 NS=4, 8 waves as 1 x 8
const TileShape kTileShapes[kNumTileShapes] = {
    {128, 128, 2, 1.25f}, {128, 64, 3, 0.80f}, {144, 256, 1, 0.90f}, {144, 128, 2, 1.00f}, {256, 128, 1, 0.80f},
    {256, 128, 1, 1.00f}, {144, 256, 1, 1.00f}, {144, 128, 1, 1.00f}, {128, 128, 1, 1.00f}, {128, 64, 2, 1.00f},
    {144, 256, 1, 1.00f}, {144, 128, 1, 1.00f}};

int choose_tile(const GemmParams& p, int epi) {
  if (p.tile_hint > 0 && p.tile_hint <= kNumTileShapes) {
    const int t = p.tile_hint - 1;
    if (epi == EPI_SCORES && kTileShapes[t].bn != 128) return 0;
    if (epi == EPI_APPLY && (t == 2 || t == 6 || t == 10)) return 3;
    if (epi == EPI_APPLY && t == 5) return 4;  // 256x128 with two accumulator sets has no room for the pipeline's registers
    if (epi == EPI_APPLY && t == 8) return 7;  // (the 128 x 128 ring has no apply instantiation: the 144 x 128 ring)
    if (epi == EPI_APPLY && p.dtype == DT_F16S && t >= kNumBaseShapes) return 3;
    if (epi != EPI_LINEAR && t == 9) return 0;
    if (p.conv && p.KH * p.KW > 32 && t >= kNumBaseShapes) return 0;   // (the pipelined loader keeps one out-of-image bit per filter tap)
    return t;
  }
  const bool full_menu = p.dtype != DT_F32 && p.staging == 1;
  if (!full_menu) return (epi == EPI_LINEAR && p.N <= 64) ? 1 : 0;
  // Cost model (fitted to tools/kernel_bench.py sweeps on MI355X, profiles/r01_tile_sweep.txt):
  // the chip runs `slots = 256 CUs x wg_per_cu` workgroups at a time; one full round of a shape costs
  // area x wg_per_cu / eff; a trailing partial round that leaves CUs with fewer co-resident workgroups
  // is cheaper, but not proportionally (a lone workgroup cannot saturate a CU).
  static const double kPartial[4][4] = {{0, 0, 0, 0}, {0, 1.0, 0, 0}, {0, 0.8, 1.0, 0}, {0, 0.5, 0.8, 1.0}};
  const int ksteps = (p.dtype == DT_F32 || p.dtype == DT_F16S) ? p.K / 32 : p.K / 64;
  int best = 0;
  double best_cost = 1e300;
  // candidates: the base shapes plus the pipelined variants that win on this path's problems
  // (5 / 8 / 9 stay reachable through tile_hint for tuning)
  static const int kCandidates[] = {0, 1, 2, 3, 4, 6, 7, 10, 11};
  for (int t : kCandidates) {
    const TileShape& s = kTileShapes[t];
    if (epi == EPI_SCORES && (s.bn != 128 || t >= kNumBaseShapes)) continue;
    if (epi == EPI_APPLY && (t == 2 || t == 6 || t >= 10)) continue;  // two accumulator sets do not fit 144x256;
                                                                       // 1 x 8 waves re-read the whole P~ tile per wave
    if (p.N <= 64 && s.bn > 64 && t != 0) continue;
    if (p.conv && p.KH * p.KW > 32 && t >= kNumBaseShapes) continue;   // (one out-of-image bit per filter tap in the pipelined loader)
    if (p.dtype == DT_F16S && t == 6) continue;   // (the 6-wave 144 x 256 ring has no registers for the split K-step's third B set)
    if (p.dtype == DT_F16S && epi == EPI_APPLY && t >= kNumBaseShapes) continue;   // (split apply: the double-buffered shapes, gemm_tile.h)
    const long tiles = (long)((p.M + s.bm - 1) / s.bm) * ((p.N + s.bn - 1) / s.bn);
    const long slots = 256L * s.wg_per_cu;
    const long full = tiles / slots, rem = tiles % slots;
    const double rounds = (double)full + (rem ? kPartial[s.wg_per_cu][(rem + 255) / 256] : 0.0);
    // short K loops expose the prologue/epilogue of shapes that run one workgroup per CU;
    // 144x256 only pays off on long K loops, and with the 3-slot pipeline from ~24 K-steps on;
    // 144x128 with the 4-slot pipeline (one workgroup per CU) is the long-K shape for M x N that fit one round
    double eff = s.eff * ((s.wg_per_cu == 1 && ksteps <= 8) ? 0.7 : 1.0);
    if (t == 2) eff = ksteps >= 64 ? 1.33 : s.eff;
    if (t == 6) eff = ksteps >= 24 ? 1.40 : 0.85;
    if (t == 7) eff = ksteps >= 64 ? 1.15 : (epi == EPI_APPLY && ksteps >= 8 ? 1.05 : 0.70);
    // the 8-wave 1 x 8 layouts keep two waves on every SIMD: ahead of the 6-wave shapes on every K length measured
    if (t == 10) eff = 1.50;
    if (t == 11) eff = 1.20;
    const double cost = rounds * s.bm * s.bn * s.wg_per_cu / eff;
    if (cost < best_cost * 0.999) { best_cost = cost; best = t; }
  }
  return best;
}

hipError_t run_tile_op_f16(const GemmParams& p, int epi, hipStream_t stream);  // gemm_f16.hip

// EPI_LINEAR2 is instantiated for the formats that carry the f32 tolerance (exact f32, split half) on the 128 x 128 double-buffered shape;
// the caller checked two_level_supported()
bool two_level_supported(const GemmParams& p) {
  if (p.dtype != DT_F32 && p.dtype != DT_F16S) return false;
  if (p.dtype == DT_F16S && p.staging != 1) return false;
  if (p.ksplit_steps > 0 || p.batch > 1 || p.s2 > 0 || p.Wn) return false;
  return (p.K / 32) % kTwoLevelSteps == 0;
}

hipError_t run_tile_op(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.dtype == DT_F16 || p.dtype == DT_F16S) return run_tile_op_f16(p, epi, stream);
  if (epi == EPI_LINEAR2) {
    if (p.dtype != DT_F32) return hipErrorInvalidValue;
    return p.staging == 1 ? launch_tile<float, 2, 2, 4, 4, EPI_LINEAR2, true>(p, stream) : launch_tile<float, 2, 2, 4, 4, EPI_LINEAR2, false>(p, stream);
  }
  if (p.dtype == DT_BF16) {
    if (epi == EPI_LINEAR) return dispatch_shape<bf16_t, EPI_LINEAR>(p, stream);
    if (epi == EPI_SCORES) return dispatch_shape<bf16_t, EPI_SCORES>(p, stream);
    return dispatch_shape<bf16_t, EPI_APPLY>(p, stream);
  }
  if (epi == EPI_LINEAR) return dispatch_shape<float, EPI_LINEAR>(p, stream);
  if (epi == EPI_SCORES) return dispatch_shape<float, EPI_SCORES>(p, stream);
  return dispatch_shape<float, EPI_APPLY>(p, stream);
}

}  // namespace hvr
