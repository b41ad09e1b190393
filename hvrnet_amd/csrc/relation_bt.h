// Big-tile relation core (relation_bt.hip: scores pass; relation_apply_bt.hip: apply pass); internal, the public ABI is
// include/hvr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace hvr {

// One launch covers `groups` independent relation problems of the same shape (the windows a caller has in flight): group g's
// operands sit gs_* ELEMENTS behind group 0's.  The 352 x 256 score tiles of all groups form one list that 256 persistent workgroups
// walk (relation_bt.hip).
struct ScoresBTParams {
  const bf16_t* Q;  // [Mq][ldq]
  const bf16_t* K;  // [Mk][ldk]
  bf16_t* P;        // [Mq][ldp]   exp2(scale*log2e*s - blockmax), bf16
  float* mstat;     // [Mq][ntile] block max (log2 units)
  float* lstat;     // [Mq][ntile] block sum
  const bf16_t* V;  // [Mk][ldv]
  bf16_t* Vt;       // [D][ldp]    V^T, zero-filled for keys Mk .. ldp - 1
  int Mq, Mk, D, ntile;
  long ldq, ldk, ldv, ldp;
  float sl2;        // scale * log2(e)
  int f16;          // 0: bf16 operands; 1: IEEE half (HVR_F16); 2: split half (HVR_F16S: 4-byte elements, Q / K / P in the
                    // [32 hi | 32 lo] layout of common.h, Q / K / V / P / Vt alike; P~ stored x 2^12)
  int groups;       // >= 1
  long gs_q, gs_k, gs_v, gs_p, gs_vt, gs_stat;   // group strides in elements (0 with one group)
  int int_max;      // block maxima rounded UP to integers (log2 units): a block's weight relative to the row's largest block is then an
                    // exact power of two, which relation_apply_bt.hip applies on the exponent fields of the P~ fragments
};

// true when the 352 x 256 tiling applies to ONE group of this shape (two-byte or, with `split`, split-half operands, aligned, a tile
// grid that fills most of the chip in whole rounds); with `groups` > 1 any tile count from 160 up qualifies -- the persistent workgroups take several tiles each
bool scores_bt_supported(int Mq, int Mk, int D, long ldq, long ldk, long ldv, long ldp, const void* Q, const void* K,
                         const void* V, const void* P, const void* Vt, int groups = 1, bool split = false);
hipError_t run_scores_bt(const ScoresBTParams& p, hipStream_t stream);

// Apply pass on 288 x 256 tiles over the whole key axis (relation_apply_bt.hip), bf16 / half / split half, scores written with int_max = 1:
//   O[g] = diag(1 / L) . (P~[g] with every 128-key block's exponent lowered by m* - m_block) . V^T[g]^T
struct ApplyBTParams {
  const bf16_t* P;      // [Mq][ldp]
  const bf16_t* Vt;     // [D][ldp]
  const float* mstat;   // [Mq][ntile] integer-valued block maxima
  const float* lstat;   // [Mq][ntile]
  bf16_t* O;            // [Mq][ldo]
  int Mq, D, ntile;
  long ldp, ldo;
  int groups;
  long gs_p, gs_vt, gs_stat, gs_o;
  int split;            // 0: bf16; 1: split half (P~ x 2^12, Vt, O in the [32 hi | 32 lo] layout; block weights by v_pk_mul_f16 on both planes);
                        // 2: IEEE half (two-byte operands, block weights by v_pk_mul_f16)
};
bool apply_bt_supported(int Mq, int Mk, int D, long ldp, long ldo, const void* P, const void* Vt, const void* O, int groups, bool split = false);
hipError_t run_apply_bt(const ApplyBTParams& p, hipStream_t stream);

}  // namespace hvr
