// The tile engine instantiated on IEEE half operands (DT_F16: v_mfma_f32_16x16x32_f16, the bf16 kernels with an 8 x finer
// mantissa) and on split-half operands (DT_F16S, common.h: three half passes per product, f32-grade results) -- the precision
// ladder between the bf16 benchmark mode and the exact-f32 parity mode.  The reference computes in f32 throughout
// (configs/faster_rcnn_r101_hrnmp_c5.py has no fp16 key; its mixed-precision islands are mmdet/core/fp16/decorators.py:9-160).
#include "gemm_tile.h"

namespace hvr {

hipError_t run_tile_op_f16(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.dtype == DT_F16) {
    if (epi == EPI_LINEAR2) return hipErrorInvalidValue;
    if (epi == EPI_LINEAR) return dispatch_shape<f16_t, EPI_LINEAR>(p, stream);
    if (epi == EPI_SCORES) return dispatch_shape<f16_t, EPI_SCORES>(p, stream);
    return dispatch_shape<f16_t, EPI_APPLY>(p, stream);
  }
  if (epi == EPI_LINEAR2) return p.staging == 1 ? launch_tile<f16s_t, 2, 2, 4, 4, EPI_LINEAR2, true>(p, stream) : hipErrorInvalidValue;
  if (epi == EPI_LINEAR) return dispatch_shape<f16s_t, EPI_LINEAR>(p, stream);
  if (epi == EPI_SCORES) return dispatch_shape<f16s_t, EPI_SCORES>(p, stream);
  return dispatch_shape<f16s_t, EPI_APPLY>(p, stream);   // (the double-buffered shapes: gemm.hip choose_tile)
}

}  // namespace hvr
