// RoIAlign forward / backward, legacy "+1 pixel" (non-aligned) convention.
// Replaces mmdet/ops/roi_align/src/roi_align_kernel.cu:16-141 (forward) and :143-282
// (backward).  Same arithmetic per sample point; different parallelisation:
//   * one workgroup per RoI, lanes along channels (NHWC: 16 B of channels per lane per
//     bilinear tap -> coalesced gathers, coalesced [roi][bin][C] stores),
//   * all frames of a window in ONE launch (the roi's batch index selects the frame),
//     where the reference launches once per frame (detectors/hnmb_rcnn.py:596-598).
// HBM-bound: bytes = feature map once + output once; taps hit L2.
#include "common.h"

namespace hvr {

struct BilinearTap {
  int y_low, x_low, y_high, x_high;
  float w1, w2, w3, w4;  // all zero when the point is outside (-1, H) x (-1, W)
  bool inside;
};

// roi_align_kernel.cu:16-61 (value) / :143-183 (gradient weights): identical index rule
__device__ __forceinline__ BilinearTap make_tap(int height, int width, float y, float x) {
  BilinearTap t;
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    t.y_low = t.x_low = t.y_high = t.x_high = 0;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    t.inside = false;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  t.y_low = y_low; t.x_low = x_low; t.y_high = y_high; t.x_high = x_high;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  t.inside = true;
  return t;
}

// one axis of make_tap: the two cells a coordinate falls between and their weights (ok = false: outside (-1, size), no contribution)
struct AxisTap { int lo, hi; float wl, wh; bool ok; };
__device__ __forceinline__ AxisTap axis_tap(int size, float v) {
  AxisTap t;
  t.ok = !(v < -1.0f || v > (float)size);
  if (v <= 0.f) v = 0.f;
  int lo = (int)v, hi;
  if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
  if (!t.ok) { lo = hi = 0; v = 0.f; }
  const float l = v - lo;
  t.lo = lo; t.hi = hi; t.wl = 1.f - l; t.wh = l;
  return t;
}

struct RoiGeom {
  int batch;
  float start_w, start_h, bin_w, bin_h;
  int sn_h, sn_w;
};

// roi_align_kernel.cu:76-99
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float spatial_scale, int sample_num, int ph, int pw) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.start_w = roi[1] * spatial_scale;
  g.start_h = roi[2] * spatial_scale;
  const float end_w = (roi[3] + 1.f) * spatial_scale, end_h = (roi[4] + 1.f) * spatial_scale;
  const float roi_w = fmaxf(end_w - g.start_w, 0.f), roi_h = fmaxf(end_h - g.start_h, 0.f);
  g.bin_h = roi_h / ph;
  g.bin_w = roi_w / pw;
  g.sn_h = sample_num > 0 ? sample_num : (int)ceilf(roi_h / ph);
  g.sn_w = sample_num > 0 ? sample_num : (int)ceilf(roi_w / pw);
  return g;
}

// layout 1: features [B][H][W][C], output [K][PH][PW][C]; CV channels per lane
template <typename T, int CV>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const T* __restrict__ feat, const float* __restrict__ rois,
                                                          T* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                          float spatial_scale, int sample_num) {
  const int k = blockIdx.x;
  const int lanes_c = C / CV;  // threads along channels
  const int groups = blockDim.x / lanes_c;
  const int cl = threadIdx.x % lanes_c, grp = threadIdx.x / lanes_c;
  if (grp >= groups) return;
  const RoiGeom g = roi_geom(rois + (long)k * 5, spatial_scale, sample_num, PH, PW);
  const T* fb = feat + (long)g.batch * H * W * C + cl * CV;
  for (int bin = grp; bin < PH * PW; bin += groups) {
    const int ph = bin / PW, pw = bin - ph * PW;
    float acc[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[e] = 0.f;
    for (int iy = 0; iy < g.sn_h; ++iy) {
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.sn_h;
      for (int ix = 0; ix < g.sn_w; ++ix) {
        const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.sn_w;
        const BilinearTap t = make_tap(H, W, y, x);
        if (!t.inside) continue;
        float lt[CV], rt[CV], lb[CV], rb[CV];
#pragma unroll
        for (int e = 0; e < CV; e += 4) {
          load4(fb + ((long)t.y_low * W + t.x_low) * C + e, lt + e);
          load4(fb + ((long)t.y_low * W + t.x_high) * C + e, rt + e);
          load4(fb + ((long)t.y_high * W + t.x_low) * C + e, lb + e);
          load4(fb + ((long)t.y_high * W + t.x_high) * C + e, rb + e);
        }
#pragma unroll
        for (int e = 0; e < CV; ++e) acc[e] += (t.w1 * lt[e] + t.w2 * rt[e] + t.w3 * lb[e] + t.w4 * rb[e]);
      }
    }
    const float cnt = (float)(g.sn_h * g.sn_w);
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[e] /= cnt;
    T* dst = out + ((long)k * PH * PW + bin) * C + cl * CV;
#pragma unroll
    for (int e = 0; e < CV; e += 4) store4(dst + e, acc + e);
  }
}

// The path's own configuration (sample_num = 2, bf16, 16 B of channels per lane): the same arithmetic in the same
// order, with the four sample points of a bin resolved first and their 16 gathers issued back to back -- the generic
// kernel's data-dependent `continue` keeps the compiler from overlapping one sample's loads with the next sample's
// (the kernel is latency-bound: 4 500 workgroups of short dependent gather chains).  A sample outside the map keeps
// the reference's "skip" (roi_align_kernel.cu:29-35): its taps read pixel (0, 0) and are not accumulated.
template <typename T> __device__ __forceinline__ void unpack8(const uint4& v, float f[8]) {
  unpack2<T>(v.x, f[0], f[1]); unpack2<T>(v.y, f[2], f[3]); unpack2<T>(v.z, f[4], f[5]); unpack2<T>(v.w, f[6], f[7]);
}

template <typename T>   // bf16_t / f16_t
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_bf16_s2(const T* __restrict__ feat, const float* __restrict__ rois,
                                                                  T* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                                  float spatial_scale) {
  const int k = blockIdx.x;
  const int lanes_c = C / 8;
  const int groups = blockDim.x / lanes_c;
  const int cl = threadIdx.x % lanes_c, grp = threadIdx.x / lanes_c;
  if (grp >= groups) return;
  const RoiGeom g = roi_geom(rois + (long)k * 5, spatial_scale, 2, PH, PW);
  const T* fb = feat + (long)g.batch * H * W * C + cl * 8;
  for (int bin = grp; bin < PH * PW; bin += groups) {
    const int ph = bin / PW, pw = bin - ph * PW;
    BilinearTap t[4];
    uint4 v[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int iy = s >> 1, ix = s & 1;
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / 2.f;
      const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / 2.f;
      t[s] = make_tap(H, W, y, x);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      v[s][0] = *reinterpret_cast<const uint4*>(fb + ((long)t[s].y_low * W + t[s].x_low) * C);
      v[s][1] = *reinterpret_cast<const uint4*>(fb + ((long)t[s].y_low * W + t[s].x_high) * C);
      v[s][2] = *reinterpret_cast<const uint4*>(fb + ((long)t[s].y_high * W + t[s].x_low) * C);
      v[s][3] = *reinterpret_cast<const uint4*>(fb + ((long)t[s].y_high * W + t[s].x_high) * C);
    }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float lt[8], rt[8], lb[8], rb[8];
      unpack8<T>(v[s][0], lt); unpack8<T>(v[s][1], rt); unpack8<T>(v[s][2], lb); unpack8<T>(v[s][3], rb);
      if (t[s].inside) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (t[s].w1 * lt[e] + t[s].w2 * rt[e] + t[s].w3 * lb[e] + t[s].w4 * rb[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] /= 4.f;
    T* dst = out + ((long)k * PH * PW + bin) * C + cl * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(pack2<T>(acc[0], acc[1]), pack2<T>(acc[2], acc[3]), pack2<T>(acc[4], acc[5]), pack2<T>(acc[6], acc[7]));
  }
}

// layout 0: features [B][C][H][W], output [K][C][PH][PW] -- the reference's own layout
// (kept for drop-in callers that hand NCHW tensors; one thread per output element).
template <typename T>
__global__ void roi_align_fwd_nchw(const T* __restrict__ feat, const float* __restrict__ rois, T* __restrict__ out,
                                   long total, int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < total; index += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const RoiGeom g = roi_geom(rois + (long)n * 5, spatial_scale, sample_num, PH, PW);
    const T* fb = feat + ((long)g.batch * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.sn_h; ++iy) {
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.sn_h;
      for (int ix = 0; ix < g.sn_w; ++ix) {
        const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.sn_w;
        const BilinearTap t = make_tap(H, W, y, x);
        if (!t.inside) continue;
        const float lt = ElemTraits<T>::load(fb + t.y_low * W + t.x_low), rt = ElemTraits<T>::load(fb + t.y_low * W + t.x_high);
        const float lb = ElemTraits<T>::load(fb + t.y_high * W + t.x_low), rb = ElemTraits<T>::load(fb + t.y_high * W + t.x_high);
        acc += (t.w1 * lt + t.w2 * rt + t.w3 * lb + t.w4 * rb);
      }
    }
    acc /= (float)(g.sn_h * g.sn_w);
    ElemTraits<T>::store(out + index, acc);
  }
}

// backward (f32 only, like the reference: roi_align_kernel.cu:269-272 rejects double and
// the training path runs f32): scatter grad * w / count with atomics.
// layout 1: grad_out [K][PH][PW][C] -> grad_in [B][H][W][C]; layout 0: NCHW both.
__global__ void roi_align_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                                     float* __restrict__ gin, long total, int C, int H, int W, int PH, int PW,
                                     float spatial_scale, int sample_num, int nhwc) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < total; index += (long)gridDim.x * blockDim.x) {
    int pw, ph, c, n;
    if (nhwc) {
      c = (int)(index % C); pw = (int)((index / C) % PW); ph = (int)((index / C / PW) % PH); n = (int)(index / C / PW / PH);
    } else {
      pw = (int)(index % PW); ph = (int)((index / PW) % PH); c = (int)((index / PW / PH) % C); n = (int)(index / PW / PH / C);
    }
    const RoiGeom g = roi_geom(rois + (long)n * 5, spatial_scale, sample_num, PH, PW);
    const float go = gout[index];
    const float count = (float)(g.sn_h * g.sn_w);
    if (nhwc && g.sn_h == 2 && g.sn_w == 2) {
      // 2 x 2 samples per bin (both configs), NHWC: the bilinear weights are separable, and the two sample rows (columns) of a bin mostly
      // fall into the same or adjacent feature rows -- their taps are merged per axis first, so a bin costs rows x cols = 4 .. 9 atomics
      // instead of 16 (round 6: the scatter is 0.6 - 0.9 ms of a training iteration).  Same sums, associated per axis.
      int ry[4], rx[4];
      float wy[4], wx[4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const AxisTap ty = axis_tap(H, g.start_h + ph * g.bin_h + (s + .5f) * g.bin_h * .5f);
        const AxisTap tx = axis_tap(W, g.start_w + pw * g.bin_w + (s + .5f) * g.bin_w * .5f);
        ry[2 * s] = ty.lo; ry[2 * s + 1] = ty.hi; wy[2 * s] = ty.ok ? ty.wl : 0.f; wy[2 * s + 1] = ty.ok ? ty.wh : 0.f;
        rx[2 * s] = tx.lo; rx[2 * s + 1] = tx.hi; wx[2 * s] = tx.ok ? tx.wl : 0.f; wx[2 * s + 1] = tx.ok ? tx.wh : 0.f;
      }
#pragma unroll
      for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i) {
          if (ry[j] == ry[i] && wy[j] != 0.f) { wy[i] += wy[j]; wy[j] = 0.f; }
          if (rx[j] == rx[i] && wx[j] != 0.f) { wx[i] += wx[j]; wx[j] = 0.f; }
        }
      float* base = gin + (long)g.batch * H * W * C + c;
      const float gs = go / count;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (wy[a] == 0.f) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (wx[b] != 0.f) atomicAdd(base + ((long)ry[a] * W + rx[b]) * C, gs * wy[a] * wx[b]);
      }
      continue;
    }
    for (int iy = 0; iy < g.sn_h; ++iy) {
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.sn_h;
      for (int ix = 0; ix < g.sn_w; ++ix) {
        const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.sn_w;
        const BilinearTap t = make_tap(H, W, y, x);
        if (!t.inside) continue;
        const float g1 = go * t.w1 / count, g2 = go * t.w2 / count, g3 = go * t.w3 / count, g4 = go * t.w4 / count;
        if (nhwc) {
          float* base = gin + (long)g.batch * H * W * C + c;
          atomicAdd(base + ((long)t.y_low * W + t.x_low) * C, g1);
          atomicAdd(base + ((long)t.y_low * W + t.x_high) * C, g2);
          atomicAdd(base + ((long)t.y_high * W + t.x_low) * C, g3);
          atomicAdd(base + ((long)t.y_high * W + t.x_high) * C, g4);
        } else {
          float* base = gin + ((long)g.batch * C + c) * H * W;
          atomicAdd(base + t.y_low * W + t.x_low, g1);
          atomicAdd(base + t.y_low * W + t.x_high, g2);
          atomicAdd(base + t.y_high * W + t.x_low, g3);
          atomicAdd(base + t.y_high * W + t.x_high, g4);
        }
      }
    }
  }
}

template <typename T>
static hipError_t launch_roi_align_nhwc_h16(const void* feat, const float* rois, void* out, int C, int H, int W, int K, int PH, int PW, float scale,
                                            int sample_num, hipStream_t s) {
  const bool al16 = ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && sample_num == 2 && al16) {
    hipLaunchKernelGGL(roi_align_fwd_nhwc_bf16_s2<T>, dim3(K), dim3(256), 0, s, (const T*)feat, rois, (T*)out, C, H, W, PH, PW, scale);
  } else if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0) {
    hipLaunchKernelGGL((roi_align_fwd_nhwc<T, 8>), dim3(K), dim3(256), 0, s, (const T*)feat, rois, (T*)out, C, H, W, PH, PW, scale, sample_num);
  } else if (C % 4 == 0 && C / 4 <= 256) {
    hipLaunchKernelGGL((roi_align_fwd_nhwc<T, 4>), dim3(K), dim3(256), 0, s, (const T*)feat, rois, (T*)out, C, H, W, PH, PW, scale, sample_num);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t run_roi_align_fwd(const void* feat, const float* rois, void* out, int B, int C, int H, int W, int K, int PH,
                             int PW, float scale, int sample_num, int dtype, int layout, hipStream_t s) {
  (void)B;
  if (K == 0) return hipSuccess;
  if (dtype == DT_F16S) return hipErrorInvalidValue;   // (split-half callers interpolate in f32 and cast)
  if (layout == 1) {
    if (dtype == DT_BF16) return launch_roi_align_nhwc_h16<bf16_t>(feat, rois, out, C, H, W, K, PH, PW, scale, sample_num, s);
    if (dtype == DT_F16) return launch_roi_align_nhwc_h16<f16_t>(feat, rois, out, C, H, W, K, PH, PW, scale, sample_num, s);
    if (C % 4 != 0 || C / 4 > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((roi_align_fwd_nhwc<float, 4>), dim3(K), dim3(256), 0, s, (const float*)feat, rois, (float*)out, C, H, W, PH, PW, scale, sample_num);
  } else {
    const long total = (long)K * C * PH * PW;
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    if (dtype == DT_BF16)
      hipLaunchKernelGGL(roi_align_fwd_nchw<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)feat, rois, (bf16_t*)out, total, C, H, W, PH, PW, scale, sample_num);
    else if (dtype == DT_F16)
      hipLaunchKernelGGL(roi_align_fwd_nchw<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)feat, rois, (f16_t*)out, total, C, H, W, PH, PW, scale, sample_num);
    else
      hipLaunchKernelGGL(roi_align_fwd_nchw<float>, dim3(grid), dim3(256), 0, s, (const float*)feat, rois, (float*)out, total, C, H, W, PH, PW, scale, sample_num);
  }
  return hipGetLastError();
}

hipError_t run_roi_align_bwd(const float* gout, const float* rois, float* gin, int C, int H, int W, int K, int PH, int PW,
                             float scale, int sample_num, int layout, hipStream_t s) {
  if (K == 0) return hipSuccess;
  const long total = (long)K * C * PH * PW;
  const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(grid), dim3(256), 0, s, gout, rois, gin, total, C, H, W, PH, PW, scale, sample_num, layout);
  return hipGetLastError();
}

}  // namespace hvr
