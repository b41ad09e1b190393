// Frame ingest for gfx950: decoded BGR uint8 frame [H][W][3] -> normalised, zero-padded f32 planes
// [3][padH][padW] in one pass -- the reference's test pipeline Resize(keep_ratio) -> Normalize -> Pad(size_divisor) ->
// ImageToTensor (mmdet/datasets/pipelines/transforms.py:111-124,260-269,308-313), which runs on the CPU in DataLoader
// workers through mmcv / OpenCV (neither is part of the reference tree).  The resize restates OpenCV's published
// INTER_LINEAR algorithm for 8-bit images (modules/imgproc/src/resize.cpp: coordinates (d + 0.5) * scale - 0.5 computed in
// double and narrowed to float, 11-bit fixed-point coefficients, horizontal pass in int, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2): integer arithmetic, so the CPU checker and this kernel agree bit
// for bit; against a real OpenCV build the parity is unpinned (no cv2 in the image).
// HBM traffic: 3 source bytes per source pixel touched + 12 output bytes per padded pixel; every output pixel is independent.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hvr {

namespace {

struct Tap { int i0, i1; int a0, a1; };

// source index pair and 11-bit coefficients of destination coordinate d (resize.cpp, INTER_LINEAR, 8U)
__device__ __forceinline__ Tap linear_tap(int d, double scale, int n_src) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
  Tap t;
  t.i0 = s;
  t.i1 = s + 1 < n_src ? s + 1 : n_src - 1;
  t.a0 = (int)lrintf((1.f - f) * 2048.f);   // saturate_cast<short>(cbuf * INTER_RESIZE_COEF_SCALE): round to nearest even
  t.a1 = (int)lrintf(f * 2048.f);
  return t;
}

}  // namespace

__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ src, int sh, int sw, long pitch, float* __restrict__ dst,
                                                     int nh, int nw, int ph, int pw, double scale_x, double scale_y, float m0, float m1,
                                                     float m2, float s0, float s1, float s2, int to_rgb) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= pw || y >= ph) return;
  const long plane = (long)ph * pw, o = (long)y * pw + x;
  if (x >= nw || y >= nh) {   // Pad(size_divisor): zeros AFTER normalisation
    dst[o] = 0.f; dst[plane + o] = 0.f; dst[2 * plane + o] = 0.f;
    return;
  }
  const Tap tx = linear_tap(x, scale_x, sw), ty = linear_tap(y, scale_y, sh);
  const uint8_t* r0 = src + (long)ty.i0 * pitch;
  const uint8_t* r1 = src + (long)ty.i1 * pitch;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = r0[tx.i0 * 3 + c] * tx.a0 + r0[tx.i1 * 3 + c] * tx.a1;
    const int S1 = r1[tx.i0 * 3 + c] * tx.a0 + r1[tx.i1 * 3 + c] * tx.a1;
    const int v = (((ty.a0 * (S0 >> 4)) >> 16) + ((ty.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
    const int oc = to_rgb ? 2 - c : c;        // mmcv.imnormalize: BGR -> RGB first, then (img - mean) / std per channel
    dst[oc * plane + o] = ((float)(v < 0 ? 0 : (v > 255 ? 255 : v)) - mean[oc]) / stdv[oc];
  }
}

hipError_t run_ingest(const uint8_t* src, int sh, int sw, long pitch, float* dst, int nh, int nw, int ph, int pw, const float* mean3,
                      const float* std3, int to_rgb, hipStream_t s) {
  // cv::resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
  const double scale_x = 1.0 / ((double)nw / (double)sw), scale_y = 1.0 / ((double)nh / (double)sh);
  hipLaunchKernelGGL(ingest_kernel, dim3((pw + 63) / 64, (ph + 3) / 4), dim3(256), 0, s, src, sh, sw, pitch, dst, nh, nw, ph, pw, scale_x,
                     scale_y, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], to_rgb);
  return hipGetLastError();
}

}  // namespace hvr
