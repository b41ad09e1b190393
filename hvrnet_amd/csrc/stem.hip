// Fused ResNet stem: 7x7/2 conv (3 -> 64, frozen BN folded) + ReLU + 3x3/2 max-pool in one pass.
// Replaces conv1 / bn1 / relu / maxpool of mmdet/models/backbones/resnet.py:456-466,522-526 (four
// launches and a 294 MB-per-window intermediate in the reference; a 0.9 GB patch matrix in this
// library's generic im2col + GEMM route, which stays as the f32 / any-shape path).
//
// One workgroup produces a 4 x 16 block of POOLED pixels for all 64 channels:
//   * the 23 x 71 x 3 input patch it needs is read once from the NCHW f32 image (coalesced along x),
//     rounded to bf16 and laid out [row][col][4] in LDS (4th channel = 0),
//   * with that layout the 7 taps of one filter row are 28 contiguous values, so the im2col row of
//     a conv pixel is 7 aligned 64-byte LDS segments: K = 7 x 32 (4 zero-weight pads per row), i.e.
//     seven v_mfma_f32_16x16x32_bf16 per 16-pixel x 16-channel fragment, A operand by ds_read_b128,
//   * the 28 weight fragments (64 x 224 bf16) stay in registers for the whole workgroup,
//   * the 9 x 33 conv pixels are written (bias, ReLU, bf16) to LDS and max-pooled from there.
// MFMA-bound in principle (12.6 GF/frame at K = 224); HBM traffic = image in + pooled map out.
#include "common.h"

namespace hvr {

#ifndef HVR_ST_PH
#define HVR_ST_PH 3
#endif
#ifndef HVR_ST_PW
#define HVR_ST_PW 16
#endif
constexpr int ST_PH = HVR_ST_PH, ST_PW = HVR_ST_PW;        // pooled tile
constexpr int ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1;  // conv region 9 x 33
constexpr int ST_NCONV = ST_CH * ST_CW;                    // 297
constexpr int ST_FRAGS = (ST_NCONV + 15) / 16;             // 19
constexpr int ST_IR = 2 * (ST_CH - 1) + 7, ST_IC = 2 * (ST_CW - 1) + 7;  // 23 x 71 input patch
constexpr int ST_PCOLS = 72;                               // patch row pitch in pixels (col 71 = zero pad)
constexpr int ST_CROW = 68;                                // conv-out row pitch in bf16 (64 + 4: conflict-free 8-byte writes)
constexpr int ST_PATCH_BYTES = (ST_IR + 3) * ST_PCOLS * 4 * 2;  // +3 rows: fragments past pixel 296 read (and discard) them
constexpr int ST_CONV_BYTES = ST_FRAGS * 16 * ST_CROW * 2;

template <typename T>   // bf16_t / f16_t
__global__ __launch_bounds__(256) void stem_fused_kernel(const float* __restrict__ img, const uint4* __restrict__ wpk,
                                                         const float* __restrict__ bias, T* __restrict__ out, int H,
                                                         int W, int CH, int CW, int PH, int PW) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* patch = reinterpret_cast<T*>(smem);
  T* cbuf = reinterpret_cast<T*>(smem + ST_PATCH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px0 = blockIdx.x * ST_PW, py0 = blockIdx.y * ST_PH, b = blockIdx.z;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // first conv pixel of the region (may be -1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;  // first input pixel of the patch

  // weights: fragment (nf, ky) = 16 bytes per lane: channel nf*16 + (lane & 15), k' = (lane >> 4)*8 ..
  uint4 wf[4][7];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) wf[nf][ky] = wpk[((nf * 16 + (lane & 15)) * 7 + ky) * 4 + (lane >> 4)];

  // input patch -> LDS [row][col][4] bf16
  const float* ib = img + (long)b * 3 * H * W;
  // branch-free: out-of-image / padding items read a clamped (valid) address and are zeroed afterwards, so that the
  // unrolled loop puts all of a thread's loads in flight before the first LDS write (the phase is latency-bound)
  constexpr int ST_ITEMS = (ST_IR + 3) * ST_PCOLS, ST_ITERS = (ST_ITEMS + 255) / 256;
  float pv[ST_ITERS][3];
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int i = it * 256 + tid;
    const int r = i / ST_PCOLS, c = i - r * ST_PCOLS;
    const int iy = iy0 + r, ix = ix0 + c;
    const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
    const long o = (long)cy * W + cx;
    pv[it][0] = ib[o];
    pv[it][1] = ib[o + (long)H * W];
    pv[it][2] = ib[o + 2L * H * W];
  }
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int i = it * 256 + tid;
    const int r = i / ST_PCOLS, c = i - r * ST_PCOLS;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = r < ST_IR && c < ST_IC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const float v0 = ok ? pv[it][0] : 0.f, v1 = ok ? pv[it][1] : 0.f, v2 = ok ? pv[it][2] : 0.f;
    if (i < ST_ITEMS) *reinterpret_cast<uint2*>(patch + (long)i * 4) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, 0.f));
  }
  __syncthreads();

  // conv fragments, round-robin over the 4 waves
  for (int f = wave; f < ST_FRAGS; f += 4) {
    const int p = f * 16 + (lane & 15);
    const int cy = p / ST_CW, cx = p - cy * ST_CW;
    const char* abase = reinterpret_cast<const char*>(patch) + ((2 * cy) * ST_PCOLS + 2 * cx) * 8 + (lane >> 4) * 16;
    f32x4 acc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const uint4 a = *reinterpret_cast<const uint4*>(abase + ky * ST_PCOLS * 8);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        acc[nf] = mfma_half<T>(wf[nf][ky], a, acc[nf]);
    }
    // lane holds channels nf*16 + (lane>>4)*4 .. +3 of pixel p
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int n = nf * 16 + (lane >> 4) * 4;
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      const float v[4] = {fmaxf(acc[nf][0] + bv.x, 0.f), fmaxf(acc[nf][1] + bv.y, 0.f), fmaxf(acc[nf][2] + bv.z, 0.f),
                          fmaxf(acc[nf][3] + bv.w, 0.f)};
      store4(cbuf + p * ST_CROW + n, v);
    }
  }
  __syncthreads();

  // 3x3/2 max-pool (pad 1 = taps outside the conv map are skipped), 8 channels per work item
  for (int i = tid; i < ST_PH * ST_PW * 8; i += 256) {
    const int ch = i & 7, q = i >> 3, qy = q / ST_PW, qx = q - qy * ST_PW;
    const int py = py0 + qy, px = px0 + qx;
    if (py >= PH || px >= PW) continue;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int gy = 2 * py - 1 + dy;
      if ((unsigned)gy >= (unsigned)CH) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int gx = 2 * px - 1 + dx;
        if ((unsigned)gx >= (unsigned)CW) continue;
        const T* src = cbuf + ((gy - cy0) * ST_CW + (gx - cx0)) * ST_CROW + ch * 8;
        float v[8];
        load4(src, v);
        load4(src + 4, v + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], v[e]);
      }
    }
    T* dst = out + (((long)b * PH + py) * PW + px) * 64 + ch * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(pack2<T>(best[0], best[1]), pack2<T>(best[2], best[3]), pack2<T>(best[4], best[5]),
                                                pack2<T>(best[6], best[7]));
  }
}

// The same stem on split-half operands (common.h: x ~ hi + lo * 2^-11): the input patch and the weights as two half planes each,
// three MFMAs per product (w_hi x a_hi, w_lo x a_hi, w_hi x a_lo into one f32 accumulator),
// the conv pixels kept in f32 in the LDS (the pooling maximum does not act per plane), the pooled pixel written as two
// [32 hi | 32 lo] groups.  wpk: [2][64][7][32] half, plane 0 = hi(w), plane 1 = half(w - hi).  The weight fragments of
// two 16-channel groups stay in registers at a time (both planes: 112 registers); the tile is produced in two passes of 32 output
// channels -- conv pixels to the LDS, pooled, written as one [32 hi | 32 lo] group per pixel -- so that the f32 conv buffer is
// half as large and two workgroups share a CU.
constexpr int ST_CROWF = 36;                                // conv-out row pitch in f32: 32 channels of one pass + 4
constexpr int ST_PLANE_BYTES = (ST_IR + 3) * ST_PCOLS * 4 * 2;
constexpr int ST_SPLIT_LDS = 2 * ST_PLANE_BYTES + ST_FRAGS * 16 * ST_CROWF * 4;   // 74 KB: two workgroups per CU

__global__ __launch_bounds__(256) void stem_fused_split_kernel(const float* __restrict__ img, const uint4* __restrict__ wpk,
                                                               const float* __restrict__ bias, char* __restrict__ out, int H, int W,
                                                               int CH, int CW, int PH, int PW) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch_hi = smem;
  char* patch_lo = smem + ST_PLANE_BYTES;
  float* cbuf = reinterpret_cast<float*>(smem + 2 * ST_PLANE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px0 = blockIdx.x * ST_PW, py0 = blockIdx.y * ST_PH, b = blockIdx.z;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;
  const float* ib = img + (long)b * 3 * H * W;
  constexpr int ST_ITEMS = (ST_IR + 3) * ST_PCOLS, ST_ITERS = (ST_ITEMS + 255) / 256;
  float pv[ST_ITERS][3];
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int i = it * 256 + tid;
    const int r = i / ST_PCOLS, c = i - r * ST_PCOLS;
    const int iy = iy0 + r, ix = ix0 + c;
    const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
    const long o = (long)cy * W + cx;
    pv[it][0] = ib[o];
    pv[it][1] = ib[o + (long)H * W];
    pv[it][2] = ib[o + 2L * H * W];
  }
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int i = it * 256 + tid;
    const int r = i / ST_PCOLS, c = i - r * ST_PCOLS;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = r < ST_IR && c < ST_IC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const float v0 = ok ? pv[it][0] : 0.f, v1 = ok ? pv[it][1] : 0.f, v2 = ok ? pv[it][2] : 0.f;
    uint32_t h01, l01, h2, l2;
    split2(v0, v1, h01, l01);
    split2(v2, 0.f, h2, l2);
    if (i < ST_ITEMS) {
      *reinterpret_cast<uint2*>(patch_hi + (long)i * 8) = make_uint2(h01, h2);
      *reinterpret_cast<uint2*>(patch_lo + (long)i * 8) = make_uint2(l01, l2);
    }
  }
  __syncthreads();

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {   // output channels 32 half .. 32 half + 31
    uint4 wh[2][7], wl[2][7];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int idx = (((half * 2 + nf) * 16 + (lane & 15)) * 7 + ky) * 4 + (lane >> 4);
        wh[nf][ky] = wpk[idx];
        wl[nf][ky] = wpk[64 * 7 * 4 + idx];
      }
    for (int f = wave; f < ST_FRAGS; f += 4) {
      const int p = f * 16 + (lane & 15);
      const int cy = p / ST_CW, cx = p - cy * ST_CW;
      const int aoff = ((2 * cy) * ST_PCOLS + 2 * cx) * 8 + (lane >> 4) * 16;
      f32x4 acc[2];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const uint4 ah = *reinterpret_cast<const uint4*>(patch_hi + aoff + ky * ST_PCOLS * 8);
        const uint4 al = *reinterpret_cast<const uint4*>(patch_lo + aoff + ky * ST_PCOLS * 8);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          acc[nf] = mfma_half<f16_t>(wl[nf][ky], ah, acc[nf]);
          acc[nf] = mfma_half<f16_t>(wh[nf][ky], al, acc[nf]);
          acc[nf] = mfma_half<f16_t>(wh[nf][ky], ah, acc[nf]);
        }
      }
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int n = (half * 2 + nf) * 16 + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        constexpr float kInvW = 1.f / 64.f;   // the weights arrive x 2^6 (native.stem_split_weights: their lo halves stay normal numbers)
        const float4 v = make_float4(fmaxf(fmaf(acc[nf][0], kInvW, bv.x), 0.f), fmaxf(fmaf(acc[nf][1], kInvW, bv.y), 0.f),
                                     fmaxf(fmaf(acc[nf][2], kInvW, bv.z), 0.f), fmaxf(fmaf(acc[nf][3], kInvW, bv.w), 0.f));
        *reinterpret_cast<float4*>(cbuf + p * ST_CROWF + nf * 16 + (lane >> 4) * 4) = v;
      }
    }
  __syncthreads();

  for (int i = tid; i < ST_PH * ST_PW * 4; i += 256) {
    const int ch = i & 3, q = i >> 2, qy = q / ST_PW, qx = q - qy * ST_PW;
    const int py = py0 + qy, px = px0 + qx;
    if (py >= PH || px >= PW) continue;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int gy = 2 * py - 1 + dy;
      if ((unsigned)gy >= (unsigned)CH) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int gx = 2 * px - 1 + dx;
        if ((unsigned)gx >= (unsigned)CW) continue;
        const float* src = cbuf + ((gy - cy0) * ST_CW + (gx - cx0)) * ST_CROWF + ch * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        best[0] = fmaxf(best[0], a.x); best[1] = fmaxf(best[1], a.y); best[2] = fmaxf(best[2], a.z); best[3] = fmaxf(best[3], a.w);
        best[4] = fmaxf(best[4], c.x); best[5] = fmaxf(best[5], c.y); best[6] = fmaxf(best[6], c.z); best[7] = fmaxf(best[7], c.w);
      }
    }
    uint4 h, l;
    split2(best[0], best[1], h.x, l.x); split2(best[2], best[3], h.y, l.y);
    split2(best[4], best[5], h.z, l.z); split2(best[6], best[7], h.w, l.w);
    char* dst = out + (((long)b * PH + py) * PW + px) * 256 + half * 128 + ch * 16;   // this pass's 32 channels = one [32 hi | 32 lo] group
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + kSplitPlane) = l;
  }
  __syncthreads();   // the next pass overwrites the conv buffer
  }
}

hipError_t run_stem_fused(const float* img, const void* wpk, const float* bias, void* out, int B, int H, int W, int dtype, hipStream_t s) {
  const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;
  const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;
  constexpr int lds = ST_PATCH_BYTES + ST_CONV_BYTES;
  static std::atomic<unsigned> attr_dev{0};   // (the attribute is per device)
  per_device_once(attr_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  dim3 grid((PW + ST_PW - 1) / ST_PW, (PH + ST_PH - 1) / ST_PH, B);
  if (dtype == DT_F16S) {
    static std::atomic<unsigned> sattr_dev{0};   // (the attribute is per device)
    per_device_once(sattr_dev, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ST_SPLIT_LDS);
    });
    hipLaunchKernelGGL(stem_fused_split_kernel, grid, dim3(256), ST_SPLIT_LDS, s, img, (const uint4*)wpk, bias, (char*)out, H, W, CH, CW, PH, PW);
    return hipGetLastError();
  }
  if (dtype == DT_F16)
    hipLaunchKernelGGL(stem_fused_kernel<f16_t>, grid, dim3(256), lds, s, img, (const uint4*)wpk, bias, (f16_t*)out, H, W, CH, CW, PH, PW);
  else
    hipLaunchKernelGGL(stem_fused_kernel<bf16_t>, grid, dim3(256), lds, s, img, (const uint4*)wpk, bias, (bf16_t*)out, H, W, CH, CW, PH, PW);
  return hipGetLastError();
}

}  // namespace hvr
