// Big-tile relation scores pass (relation_bt.hip); internal, the public ABI is include/hvr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace hvr {

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
  int tile0, tiles_here;  // set by run_scores_bt per launch: this launch's slice of the tile grid
  int f16;                // operands are IEEE half (HVR_F16) instead of bf16
};

// true when the one-round 352 x 256 tiling applies (bf16, aligned operands, a tile grid that fills most of the chip)
bool scores_bt_supported(int Mq, int Mk, int D, long ldq, long ldk, long ldv, long ldp, const void* Q, const void* K,
                         const void* V, const void* P, const void* Vt);
hipError_t run_scores_bt(const ScoresBTParams& p, hipStream_t stream);

}  // namespace hvr
