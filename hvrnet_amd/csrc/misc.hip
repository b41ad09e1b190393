// HBM-bound helper kernels around the MFMA tile engine: layout changes, casts, the
// 7x7 stem's patch gather, 3x3/2 max pooling, the class softmax and box decode.
// All are coalesced 16-byte-per-lane streaming kernels (no LDS reuse to exploit), except
// the transpose which stages a 64x64 tile through LDS.
#include <type_traits>
#include "common.h"

namespace hvr {

// ---- out[C][ldt] = in[R][ldx]^T, zero-filled for columns R..ldt-1 (relation V^T) ----
template <typename T>
__global__ void transpose_pad_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int C, long ldx, long ldt) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    T v = T(0);
    if (r < R && c < C) v = in[(long)r * ldx + c];
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < ldt) out[(long)c * ldt + r] = tile[tx][i];
  }
}

// ---- stem patch gather (reference conv: mmdet/models/backbones/resnet.py:456-466) ----
// img NCHW f32 [B][3][H][W] -> cols [B*OH*OW][KP], k = (ky*7+kx)*3 + c, zero for k >= 147
template <typename T>
__global__ void im2col_stem_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int H, int W, int OH,
                                   int OW, int KP) {
  constexpr int EPC = ElemTraits<T>::kPerChunk;
  const int chunks = KP / EPC;
  const long total = (long)B * OH * OW * chunks;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % chunks);
    const long m = idx / chunks;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long)OW * OH));
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int k = ch * EPC + e;
      float x = 0.f;
      if (k < 147) {
        const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) x = img[(((long)b * 3 + c) * H + iy) * W + ix];
      }
      v[e] = x;
    }
    T* dst = cols + m * KP + ch * EPC;
#pragma unroll
    for (int e = 0; e < EPC; e += 4) store4(dst + e, v + e);
  }
}

// ---- 3x3 stride-2 pad-1 max pooling, NHWC (resnet.py:466 nn.MaxPool2d(3, 2, 1)) ----
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int OH,
                                    int OW) {
  const int c4 = C / 4;
  const long total = (long)B * OH * OW * c4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    const long px = idx / c4;
    const int ox = (int)(px % OW), oy = (int)((px / OW) % OH), b = (int)(px / ((long)OW * OH));
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v[4];
        load4(in + (((long)b * H + iy) * W + ix) * C + cq * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v[e]);
      }
    }
    store4(out + px * C + cq * 4, best);
  }
}

// ---- element casts ----
__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      float v[4];
      load4(in + i, v);
      store4(out + i, v);
    } else {
      for (long j = i; j < n; ++j) out[j] = f2bf(in[j]);
    }
  }
}
__global__ void cast_bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      float v[4];
      load4(in + i, v);
      store4(out + i, v);
    } else {
      for (long j = i; j < n; ++j) out[j] = bf2f(in[j]);
    }
  }
}

// ---- casts between any two operand formats, 8 logical elements per thread (n % 8 == 0; split half: n % 32 == 0) ----
template <typename T> __device__ __forceinline__ void load8(const char* base, long i, float v[8]) {
  if constexpr (std::is_same<T, float>::value) {
    load4(reinterpret_cast<const float*>(base) + i, v);
    load4(reinterpret_cast<const float*>(base) + i + 4, v + 4);
  } else if constexpr (std::is_same<T, f16s_t>::value) {
    const char* q = base + split_col_bytes(i);
    const uint4 h = *reinterpret_cast<const uint4*>(q), l = *reinterpret_cast<const uint4*>(q + kSplitPlane);
    merge2(h.x, l.x, v[0], v[1]); merge2(h.y, l.y, v[2], v[3]); merge2(h.z, l.z, v[4], v[5]); merge2(h.w, l.w, v[6], v[7]);
  } else {
    const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(base) + i);
    unpack2<T>(t.x, v[0], v[1]); unpack2<T>(t.y, v[2], v[3]); unpack2<T>(t.z, v[4], v[5]); unpack2<T>(t.w, v[6], v[7]);
  }
}
template <typename T> __device__ __forceinline__ void store8(char* base, long i, const float v[8]) {
  if constexpr (std::is_same<T, float>::value) {
    store4(reinterpret_cast<float*>(base) + i, v);
    store4(reinterpret_cast<float*>(base) + i + 4, v + 4);
  } else if constexpr (std::is_same<T, f16s_t>::value) {
    char* q = base + split_col_bytes(i);
    uint4 h, l;
    split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y); split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
    *reinterpret_cast<uint4*>(q) = h;
    *reinterpret_cast<uint4*>(q + kSplitPlane) = l;
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(base) + i) =
        make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
  }
}
template <typename TI, typename TO>
__global__ void cast8_kernel(const char* __restrict__ in, char* __restrict__ out, long n, float scale) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    float v[8];
    load8<TI>(in, i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= scale;   // (a power of two in every use: exact)
    store8<TO>(out, i, v);
  }
}

// the stem's patch rows (im2col_stem_kernel) written straight in the split-half format: 8 elements of a row per thread
__global__ void im2col_stem_split_kernel(const float* __restrict__ img, char* __restrict__ cols, int B, int H, int W, int OH, int OW, int KP) {
  const int chunks = KP / 8;
  const long total = (long)B * OH * OW * chunks;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % chunks);
    const long m = idx / chunks;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long)OW * OH));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ch * 8 + e;
      float x = 0.f;
      if (k < 147) {
        const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) x = img[(((long)b * 3 + c) * H + iy) * W + ix];
      }
      v[e] = x;
    }
    store8<f16s_t>(cols + m * KP * 4, ch * 8, v);
  }
}

// ---- [B][C][HW] <-> [B][HW][C] (API-boundary layout changes only) ----
template <typename TI, typename TO>
__global__ void permute_bchw_kernel(const TI* __restrict__ in, TO* __restrict__ out, int C, int HW, int to_nhwc) {
  // treats each image as a [C][HW] (to_nhwc) or [HW][C] matrix and transposes it through LDS
  __shared__ float tile[64][65];
  const int R = to_nhwc ? C : HW, Cn = to_nhwc ? HW : C;
  const long img = (long)blockIdx.z * C * HW;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cn) ? ElemTraits<TI>::load(in + img + (long)r * Cn + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cn && r < R) ElemTraits<TO>::store(out + img + (long)c * R + r, tile[tx][i]);
  }
}

// ---- class softmax + box decode for the key frame's rois ----
// reference: mmdet/models/bbox_heads/bbox_head.py:141-158, mmdet/core/bbox/transforms.py:78-110.
// logits [R][ldl] f32 (cls at col cls_off.., 4 deltas at col reg_off..), rois [R][5]
// -> scores [R][ncls], boxes [R][4]
__global__ void det_decode_kernel(const float* __restrict__ logits, int ldl, int cls_off, int reg_off, int ncls,
                                  const float* __restrict__ rois, int R, float m0, float m1, float m2, float m3,
                                  float s0, float s1, float s2, float s3, float max_ratio, float img_h, float img_w,
                                  float scale_factor, float* __restrict__ scores, float* __restrict__ boxes) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* l = logits + (long)r * ldl;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, l[cls_off + c]);
  float sum = 0.f;
  for (int c = 0; c < ncls; ++c) sum += expf(l[cls_off + c] - mx);
  for (int c = 0; c < ncls; ++c) scores[(long)r * ncls + c] = expf(l[cls_off + c] - mx) / sum;

  const float* roi = rois + (long)r * 5;
  const float dx = l[reg_off + 0] * s0 + m0, dy = l[reg_off + 1] * s1 + m1;
  float dw = l[reg_off + 2] * s2 + m2, dh = l[reg_off + 3] * s3 + m3;
  dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
  dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
  const float px = (roi[1] + roi[3]) * 0.5f, py = (roi[2] + roi[4]) * 0.5f;
  const float pw = roi[3] - roi[1] + 1.0f, ph = roi[4] - roi[2] + 1.0f;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f;
  float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f;
  if (img_w > 0.f) {
    x1 = fminf(fmaxf(x1, 0.f), img_w - 1.f);
    y1 = fminf(fmaxf(y1, 0.f), img_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), img_w - 1.f);
    y2 = fminf(fmaxf(y2, 0.f), img_h - 1.f);
  }
  float* b = boxes + (long)r * 4;
  // rescale=True divides by scale_factor (bbox_head.py:152-158); scale_factor <= 0 means rescale=False
  if (scale_factor > 0.f) { x1 /= scale_factor; y1 /= scale_factor; x2 /= scale_factor; y2 /= scale_factor; }
  b[0] = x1; b[1] = y1; b[2] = x2; b[3] = y2;
}

// ---------------- launchers ----------------
static inline int grid_for(long work, int block, int cap = 256 * 16) {
  long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// bf16 fast path: 16-byte global accesses on both sides (a 64 x 64 tile = 512 chunks in, 512 chunks out)
__global__ __launch_bounds__(256) void transpose_pad_bf16x8_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C,
                                                                    long ldx, long ldt) {
  constexpr int PITCH = 64 * 2 + 16;  // bytes per staged row
  __shared__ __attribute__((aligned(16))) char tile[64 * PITCH];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, i = s >> 3, q = s & 7;  // row i, chunk q of the input tile
    const int r = r0 + i, c = c0 + q * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R && c < C) v = *reinterpret_cast<const uint4*>(in + (long)r * ldx + c);  // C % 8 == 0
    *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, q = s >> 6, i = s & 63;  // output row c0 + i, its chunk q (rows r0 + 8q .. + 7)
    const int c = c0 + i, r = r0 + q * 8;
    if (c >= C || r >= ldt) continue;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
      const uint32_t hi = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
      w[e] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(out + (long)c * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);  // ldt % 8 == 0
  }
}

// ---- training: the two K-contiguous operands of a conv's weight-gradient product dW = dZ^T . cols, written in one pass each ----
// (round 4's step built them as relu_bwd -> transpose_pad and im2col -> transpose_pad: 291 transposes per iteration, 13 % of its kernel
// time, each re-reading what the kernel before it had just written.)  16-bit operands (bf16 / half as raw words), 64 x 64 tiles, 16-byte
// global accesses on both sides, as transpose_pad_bf16x8_kernel.
//
// colsT[(tap * Cin + c)][p] = x[b][oy - pad + ky dil][ox - pad + kx dil][c] (zero outside the image and for p >= P), p = (b OH + oy) OW + ox,
// tap = ky KW + kx = blockIdx.z: the transposed patch matrix of a stride-1 KH x KW conv, without the row-major patch matrix in between.
__global__ __launch_bounds__(256) void im2col_t_x8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int Cin,
                                                           int KW, int pad, int dil, int OH, int OW, long ldt) {
  constexpr int PITCH = 64 * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tile[64 * PITCH];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x, tap = blockIdx.z;
  const int ky = tap / KW, kx = tap - ky * KW;
  const long P = (long)B * OH * OW;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, i = s >> 3, q = s & 7;
    const long p = (long)r0 + i;
    const int c = c0 + q * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (p < P && c < Cin) {
      const int ox = (int)(p % OW);
      const long t = p / OW;
      const int oy = (int)(t % OH), b = (int)(t / OH);
      const int iy = oy - pad + ky * dil, ix = ox - pad + kx * dil;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * W + ix) * Cin + c);
    }
    *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, q = s >> 6, i = s & 63;
    const int c = c0 + i;
    const long r = (long)r0 + q * 8;
    if (c >= Cin || r >= ldt) continue;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
      const uint32_t hi = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
      w[e] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(out + ((long)tap * Cin + c) * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// dZ = dY where Y > 0 else 0 ([R][C], row-major) AND dZ^T [C][ldt] (columns R .. ldt - 1 zero) from one read of dY and Y.  A 16-bit float
// (bf16 or half) is positive when its sign bit is clear and the rest is not zero.  (A NaN with a clear sign bit therefore passes the
// gate, where torch's `y > 0` is false: Y is the output of this library's own ReLU epilogues -- fmaxf(x, 0), which returns 0 for a NaN x --
// so it never holds one; the test is format-independent, which is why bf16 and half share the kernel.)
__global__ __launch_bounds__(256) void relu_bwd_t_x8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ dz,
                                                             bf16_t* __restrict__ dzt, int R, int C, long ldt) {
  constexpr int PITCH = 64 * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tile[64 * PITCH];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
  auto mask2 = [](uint32_t g, uint32_t a) -> uint32_t {   // two 16-bit lanes of gradient g gated by activation a
    const uint32_t lo = ((a & 0x7fffu) != 0u && !(a & 0x8000u)) ? 0x0000ffffu : 0u;
    const uint32_t hi = ((a & 0x7fff0000u) != 0u && !(a & 0x80000000u)) ? 0xffff0000u : 0u;
    return g & (lo | hi);
  };
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, i = s >> 3, q = s & 7;
    const int r = r0 + i, c = c0 + q * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R && c < C) {
      const uint4 g = *reinterpret_cast<const uint4*>(dy + (long)r * C + c), a = *reinterpret_cast<const uint4*>(y + (long)r * C + c);
      v = make_uint4(mask2(g.x, a.x), mask2(g.y, a.y), mask2(g.z, a.z), mask2(g.w, a.w));
      *reinterpret_cast<uint4*>(dz + (long)r * C + c) = v;
    }
    *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, q = s >> 6, i = s & 63;
    const int c = c0 + i, r = r0 + q * 8;
    if (c >= C || r >= ldt) continue;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
      const uint32_t hi = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
      w[e] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(dzt + (long)c * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// split half: the hi and the lo plane of a 64 x 64 logical tile are two independent 64 x 64 transposes of 16-bit words
// (blockIdx.z = plane); only the addressing knows about the [32 hi | 32 lo] groups.  R, C, ldx, ldt in logical elements; C, ldx,
// ldt multiples of 64 here (the relation's V^T: D and the padded key count).
__global__ __launch_bounds__(256) void transpose_pad_split_kernel(const char* __restrict__ in, char* __restrict__ out, int R, int C,
                                                                  long ldx, long ldt) {
  constexpr int PITCH = 64 * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tile[64 * PITCH];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x, plane = blockIdx.z * kSplitPlane;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, i = s >> 3, q = s & 7;
    const int r = r0 + i;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R) v = *reinterpret_cast<const uint4*>(in + (long)r * ldx * 4 + split_col_bytes(c0 + q * 8) + plane);
    *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 256 + tid, q = s >> 6, i = s & 63;
    const int c = c0 + i;
    if (c >= C) continue;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = *reinterpret_cast<const unsigned short*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
      const uint32_t hi = *reinterpret_cast<const unsigned short*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
      w[e] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(out + (long)c * ldt * 4 + split_col_bytes(r0 + q * 8) + plane) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---- relation backward helpers (attention backward of one relation stage; the products are tile-engine GEMMs) ----
// P[m][:] *= 2^(m_t - m*) / L per 128-key block t: turns the score pass's block-relative exponentials into the softmax
// probabilities (selsa_bbox_head.py:172, nn.Softmax(dim=2)); padding columns of P are zero and stay zero.
template <typename T>
__global__ __launch_bounds__(256) void relation_normalize_kernel(T* __restrict__ P, const float* __restrict__ mstat,
                                                                 const float* __restrict__ lstat, int ntile, long ldp) {
  __shared__ float g[128];
  const int m = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    float mx = -INFINITY;
    for (int t = tid; t < ntile; t += 64) mx = fmaxf(mx, mstat[(long)m * ntile + t]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float l = 0.f;
    for (int t = tid; t < ntile; t += 64) l += lstat[(long)m * ntile + t] * exp2f(mstat[(long)m * ntile + t] - mx);
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    for (int t = tid; t < ntile; t += 64) g[t] = exp2f(mstat[(long)m * ntile + t] - mx) / l;
  }
  __syncthreads();
  T* row = P + (long)m * ldp;
  for (long j = (long)tid * 4; j < ldp; j += 256 * 4) {
    float v[4];
    load4(row + j, v);
    const float w = g[j >> 7];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= w;
    store4(row + j, v);
  }
}

// the same on a split-half P (rows of ldp / 64 [hi | lo] groups): 8 logical columns per thread step
__global__ __launch_bounds__(256) void relation_normalize_split_kernel(char* __restrict__ P, const float* __restrict__ mstat,
                                                                       const float* __restrict__ lstat, int ntile, long ldp, float post) {
  __shared__ float g[128];
  const int m = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    float mx = -INFINITY;
    for (int t = tid; t < ntile; t += 64) mx = fmaxf(mx, mstat[(long)m * ntile + t]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float l = 0.f;
    for (int t = tid; t < ntile; t += 64) l += lstat[(long)m * ntile + t] * exp2f(mstat[(long)m * ntile + t] - mx);
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    for (int t = tid; t < ntile; t += 64) g[t] = exp2f(mstat[(long)m * ntile + t] - mx) / l * post;
  }
  __syncthreads();
  char* row = P + (long)m * ldp * 4;
  for (long j = (long)tid * 8; j < ldp; j += 256 * 8) {
    float v[8];
    load8<f16s_t>(row, j, v);
    const float w = g[j >> 7];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= w;
    store8<f16s_t>(row, j, v);
  }
}

// dS[m][:] = scale * P[m][:] * (dP[m][:] - sum_d dO[m][d] * O[m][d])   (softmax backward folded with the logit scale)
template <typename T>
__global__ __launch_bounds__(256) void relation_dscore_kernel(const T* __restrict__ P, const T* __restrict__ dP,
                                                              const T* __restrict__ dO, const T* __restrict__ O,
                                                              T* __restrict__ dS, long ldp, int D, long ldgo, long ldo,
                                                              float scale) {
  __shared__ float part[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int d = tid * 4; d < D; d += 256 * 4) {
    float a[4], b[4];
    load4(dO + (long)m * ldgo + d, a);
    load4(O + (long)m * ldo + d, b);
    acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  const float delta = (part[0] + part[1]) + (part[2] + part[3]);
  for (long j = (long)tid * 4; j < ldp; j += 256 * 4) {
    float p[4], g[4], o[4];
    load4(P + (long)m * ldp + j, p);
    load4(dP + (long)m * ldp + j, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = scale * p[e] * (g[e] - delta);
    store4(dS + (long)m * ldp + j, o);
  }
}

// ---- training-path helpers (head backward) ----
// dZ = dY where Y > 0 else 0: backward of the ReLU fused into a GEMM epilogue (Y is that epilogue's output)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ Y, T* __restrict__ dZ, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    float g[4], y[4];
    load4(dY + i, g);
    load4(Y + i, y);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
    store4(dZ + i, g);
  }
}

// db[n] = sum_m dY[m][n] (f32): bias gradient of a linear layer.  Two stages, fixed summation order: workgroup (x, y) sums
// rows y, y + S, ... of 64 columns (4 row lanes) into part[y][n]; the second kernel adds the S partials.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dY, float* __restrict__ part_out, int M, int N, long ld) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < N)
    for (int m = blockIdx.y * 4 + r0; m < M; m += 4 * gridDim.y) acc += ElemTraits<T>::load(dY + (long)m * ld + c);
  part[r0][threadIdx.x & 63] = acc;
  __syncthreads();
  if (r0 == 0 && c < N)
    part_out[(long)blockIdx.y * N + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ db, int N, int S) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float acc = 0.f;
  for (int y = 0; y < S; ++y) acc += part[(long)y * N + c];
  db[c] = acc;
}

// BBoxHead.loss for class-agnostic regression (bbox_heads/bbox_head.py:100-130): softmax cross entropy over ncls logits
// (row weight, sum / avg_cls), smooth-L1(beta) on the rows with label > 0 (element weight, sum / R), top-1 accuracy in
// percent, and the gradient of  w_cls * loss_cls + w_bbox * loss_bbox  w.r.t. the logit matrix [R][ldl] whose columns
// cls_off .. +ncls are class logits and reg_off .. +4 box deltas.  One workgroup, fixed reduction order.
__global__ __launch_bounds__(256) void det_loss_kernel(const float* __restrict__ logits, int ldl, int cls_off, int reg_off, int ncls,
                                                       const long long* __restrict__ labels, const float* __restrict__ label_w,
                                                       const float* __restrict__ bbox_t, const float* __restrict__ bbox_w, int R,
                                                       float beta, float w_cls, float w_bbox, float* __restrict__ out3,
                                                       float* __restrict__ dlogits, const int* __restrict__ sel_counts) {
  // sel_counts != null: the OHEM form of the loss (selsa_rcnn.py:224-232) -- the reference evaluates it on the gathered rows
  // cat(pos_inds, neg_inds); rows outside carry zero weights here, so only the smooth-L1 denominator (the number of gathered
  // rows) and the accuracy's row set differ from the plain form.
  __shared__ float red[3][256];
  __shared__ float sh_avg;
  const int tid = threadIdx.x;
  float npos = 0.f;
  for (int r = tid; r < R; r += 256) npos += label_w[r] > 0.f ? 1.f : 0.f;
  red[0][tid] = npos;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[0][tid] += red[0][tid + o];
    __syncthreads();
  }
  if (tid == 0) sh_avg = fmaxf(red[0][0], 1.f);
  __syncthreads();
  const float avg = sh_avg;
  const float rows_bbox = sel_counts ? fmaxf((float)(sel_counts[0] + sel_counts[1]), 1.f) : (float)R;
  __syncthreads();
  float lc = 0.f, lb = 0.f, hit = 0.f;
  for (int r = tid; r < R; r += 256) {
    const float* row = logits + (long)r * ldl;
    float* drow = dlogits + (long)r * ldl;
    const int lab = (int)labels[r];
    float mx = -INFINITY;
    int arg = 0;
    for (int c = 0; c < ncls; ++c) {
      const float v = row[cls_off + c];
      if (v > mx) { mx = v; arg = c; }
    }
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) se += expf(row[cls_off + c] - mx);
    const float lse = mx + logf(se), w = label_w[r];
    lc += (lse - row[cls_off + lab]) * w;
    hit += (arg == lab && (!sel_counts || w > 0.f)) ? 1.f : 0.f;
    for (int c = 0; c < ldl; ++c) drow[c] = 0.f;
    for (int c = 0; c < ncls; ++c)
      drow[cls_off + c] = w_cls * w / avg * (expf(row[cls_off + c] - lse) - (c == lab ? 1.f : 0.f));
    if (lab > 0) {
      for (int e = 0; e < 4; ++e) {
        const float d = row[reg_off + e] - bbox_t[r * 4 + e], a = fabsf(d), bw = bbox_w[r * 4 + e];
        lb += (a < beta ? 0.5f * a * a / beta : a - 0.5f * beta) * bw;
        drow[reg_off + e] = w_bbox * bw / rows_bbox * (a < beta ? d / beta : (d > 0.f ? 1.f : -1.f));
      }
    }
  }
  red[0][tid] = lc; red[1][tid] = lb; red[2][tid] = hit;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    out3[0] = red[0][0] / avg;
    out3[1] = red[1][0] / rows_bbox;
    out3[2] = red[2][0] * (100.f / rows_bbox);
  }
}

// ---- conv backward helpers ----
// cols[p][(ky*KW + kx)*Cin + c] = x[b][oy - pad + ky*dil][ox - pad + kx*dil][c] (zero outside), p = (b*OH + oy)*OW + ox,
// stride 1: the patch matrix whose transpose times dY^T is the weight gradient of a KHxKW conv (torch: conv2d backward)
template <typename T>
__global__ void im2col_nhwc_kernel(const T* __restrict__ x, T* __restrict__ cols, int B, int H, int W, int Cin, int KH, int KW,
                                   int pad, int dil, int OH, int OW) {
  const int c4 = Cin / 4;
  const long total = (long)B * OH * OW * KH * KW * c4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    long r = idx / c4;
    const int tap = (int)(r % (KH * KW));
    const long pix = r / (KH * KW);
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));
    const int ky = tap / KW, kx = tap - ky * KW;
    const int iy = oy - pad + ky * dil, ix = ox - pad + kx * dil;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) load4(x + (((long)b * H + iy) * W + ix) * Cin + cq * 4, v);
    store4(cols + (pix * (KH * KW) + tap) * Cin + cq * 4, v);
  }
}

// out[r][:] = w[r][:] * s[r]: folds a frozen BatchNorm's per-channel scale into conv weights (and back into their gradient)
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ w, const float* __restrict__ s, T* __restrict__ out, int R, long C) {
  const long total = (long)R * (C / 4);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / (C / 4), c = (idx - r * (C / 4)) * 4;
    float v[4];
    load4(w + r * C + c, v);
    const float k = s[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= k;
    store4(out + r * C + c, v);
  }
}

// nn.Conv2d weight [Cout][Cin][KH][KW] (f32 master) * s[Cout] -> the conv kernel's operand [Cout][KH][KW][Cin] in T:
// the permute, the frozen BatchNorm scale and the rounding to the compute dtype in one pass
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const float* __restrict__ s, T* __restrict__ out, int Cout, int Cin,
                                        int KK) {
  const long total = (long)Cout * KK * Cin;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin);
    const long r = idx / Cin;
    const int t = (int)(r % KK), o = (int)(r / KK);
    ElemTraits<T>::store(out + idx, w[((long)o * Cin + c) * KK + t] * s[o]);
  }
}

// the way back for the gradient: dW_eff [Cout][KH*KW][Cin] (f32) * s[Cout] -> [Cout][Cin][KH][KW]
__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ dw, const float* __restrict__ s, float* __restrict__ out, int Cout,
                                         int Cin, int KK, int accumulate) {
  const long total = (long)Cout * KK * Cin;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % KK);
    const long r = idx / KK;
    const int c = (int)(r % Cin), o = (int)(r / Cin);
    const float a = dw[((long)o * KK + t) * Cin + c];   // (an explicit fused multiply-add in every form of this kernel, so that the
    out[idx] = accumulate ? __builtin_fmaf(a, s[o], out[idx]) : a * s[o];   //  table-driven one below is bit-identical to this one whatever the compiler contracts)   // accumulate: straight into the parameter's .grad (weight gradients computed off the main stream)
  }
}

// ---- the weight side of a training iteration in two launches (round 5) ----
// Every trainable conv / linear layer needs, once per iteration: its f32 master weight x the frozen BatchNorm scale in the kernels'
// operand layout and dtype (pack_conv_weight), and for the input-gradient product that operand transposed (1x1 / linear: [Cin][ldn]) or
// rotated by 180 degrees with the channel axes swapped (KxK: [Cin][KH][KW][Cout]).  Round 4 made them per layer inside forward / backward
// (96 + 65 + 2 x 31 launches per iteration); here a descriptor table on the device lists every layer and one launch packs all of them,
// a second one writes all the transposed / rotated copies as per-tap 64 x 64 tile transposes of the packed operands.
struct PackItem {       // mirrors hvr_pack_item (include/hvr_hip.h)
  const float* w;       // [Cout][Cin][KK] f32
  const float* s;       // [Cout]
  void* eff;            // [Cout][KK][Cin], 16-bit operand
  long first;           // index of this item's first element in the flat work list
  int Cout, Cin, KK, pad_;
};
struct TransItem {      // mirrors hvr_transpose_item: dst[c][r] = src[r][c], r < R, c < C (16-bit words), 64 x 64 tiles
  const void* src;
  void* dst;
  long lds, ldd;        // row pitches in elements
  int R, C;
  int first_tile, tiles_c;
  int dcols, pad_;      // columns of dst written per row (R .. dcols - 1: zeros)
};

// One thread per EIGHT consecutive output elements (round 6: one table search, one set of divisions and one 16-byte store per eight -- the
// element-per-thread form ran at 0.8 TB/s, 0.5 ms per SELSA iteration for 68 M parameters): eight input channels of one (output channel,
// tap), read at stride KK from the parameter layout (contiguous for 1x1 / linear layers).  An item whose Cin is not a multiple of 8, or a
// group that straddles two items, takes the element-wise path.  Same values: w * s rounded once.
template <typename T>
__device__ __forceinline__ void pack_one(const PackItem* __restrict__ items, int n, long idx) {
  int lo = 0, hi = n - 1;                       // the item that holds element idx (first[] ascending)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first <= idx) lo = mid; else hi = mid - 1;
  }
  const PackItem it = items[lo];
  const long e = idx - it.first;
  const int c = (int)(e % it.Cin);
  const long r = e / it.Cin;
  const int t = (int)(r % it.KK), o = (int)(r / it.KK);
  ElemTraits<T>::store(reinterpret_cast<T*>(it.eff) + e, it.w[((long)o * it.Cin + c) * it.KK + t] * it.s[o]);
}

template <typename T>
__global__ void pack_conv_weights_multi_kernel(const PackItem* __restrict__ items, int n, long total) {
  const long groups = (total + 7) / 8;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < groups; v += (long)gridDim.x * blockDim.x) {
    const long idx = v * 8;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].first <= idx) lo = mid; else hi = mid - 1;
    }
    const PackItem it = items[lo];
    const long e = idx - it.first, item_total = (long)it.Cout * it.Cin * it.KK;
    if ((it.Cin & 7) == 0 && (e & 7) == 0 && e + 8 <= item_total && (reinterpret_cast<uintptr_t>(it.eff) & 15) == 0) {
      const int c = (int)(e % it.Cin);
      const long r = e / it.Cin;
      const int t = (int)(r % it.KK), o = (int)(r / it.KK);
      const float sc = it.s[o];
      const float* src = it.w + ((long)o * it.Cin + c) * it.KK + t;
      float lo4[4], hi4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { lo4[j] = src[(long)j * it.KK] * sc; hi4[j] = src[(long)(j + 4) * it.KK] * sc; }
      T* dst = reinterpret_cast<T*>(it.eff) + e;
      store4(dst, lo4);
      store4(dst + 4, hi4);
    } else {
      for (int j = 0; j < 8 && idx + j < total; ++j) pack_one<T>(items, n, idx + j);
    }
  }
}

__global__ __launch_bounds__(256) void transpose_multi_x8_kernel(const TransItem* __restrict__ items, int n) {
  constexpr int PITCH = 64 * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tile[64 * PITCH];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TransItem it = items[lo];
  const int tl = (int)blockIdx.x - it.first_tile;
  const int r0 = (tl / it.tiles_c) * 64, c0 = (tl % it.tiles_c) * 64, tid = threadIdx.x;
  const bf16_t* in = reinterpret_cast<const bf16_t*>(it.src);
  bf16_t* out = reinterpret_cast<bf16_t*>(it.dst);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int sidx = k * 256 + tid, i = sidx >> 3, q = sidx & 7;
    const int r = r0 + i, c = c0 + q * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < it.R && c < it.C) v = *reinterpret_cast<const uint4*>(in + (long)r * it.lds + c);     // C % 8 == 0
    *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int sidx = k * 256 + tid, q = sidx >> 6, i = sidx & 63;
    const int c = c0 + i, r = r0 + q * 8;
    if (c >= it.C || r >= it.dcols) continue;                                                     // (columns R .. dcols - 1: zero padding)
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t l = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
      const uint32_t h = *reinterpret_cast<const bf16_t*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
      w[e] = l | (h << 16);
    }
    *reinterpret_cast<uint4*>(out + (long)c * it.ldd + r) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// the way back for a whole group of layers: item i's f32 product dW_eff [Cout][KK][Cin] (at `eff`) x s -> += (or =) the parameter-layout
// gradient [Cout][Cin][KK] (at `w`); same table walk as the pack kernel (first[] ascending)
__device__ __forceinline__ void unpack_one(const PackItem* __restrict__ items, int n, long idx, int accumulate) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first <= idx) lo = mid; else hi = mid - 1;
  }
  const PackItem it = items[lo];
  const long e = idx - it.first;                // element of the parameter layout [Cout][Cin][KK]
  const int t = (int)(e % it.KK);
  const long r = e / it.KK;
  const int c = (int)(r % it.Cin), o = (int)(r / it.Cin);
  const float a = reinterpret_cast<const float*>(it.eff)[((long)o * it.KK + t) * it.Cin + c];
  float* dst = const_cast<float*>(it.w) + e;
  *dst = accumulate ? __builtin_fmaf(a, it.s[o], *dst) : a * it.s[o];
}

// one thread per FOUR consecutive elements of the PARAMETER layout [Cout][Cin][KK] (one table search and one set of divisions per four, one
// 16-byte read-modify-write; the product is read at stride Cin -- the other way round, contiguous reads and strided read-modify-writes,
// measured slower: 40 against 33 us per launch); a group that straddles two items or is not 16-byte aligned: element-wise
__global__ void unpack_conv_wgrads_multi_kernel(const PackItem* __restrict__ items, int n, long total, int accumulate) {
  const long groups = (total + 3) / 4;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < groups; v += (long)gridDim.x * blockDim.x) {
    const long idx = v * 4;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].first <= idx) lo = mid; else hi = mid - 1;
    }
    const PackItem it = items[lo];
    const long e = idx - it.first, item_total = (long)it.Cout * it.Cin * it.KK;
    float* dst = const_cast<float*>(it.w) + e;
    if (e + 4 <= item_total && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      int t = (int)(e % it.KK);
      const long r = e / it.KK;
      int c = (int)(r % it.Cin), o = (int)(r / it.Cin);
      const float* src = reinterpret_cast<const float*>(it.eff);
      float av[4], sv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[j] = src[((long)o * it.KK + t) * it.Cin + c];
        sv[j] = it.s[o];
        if (++t == it.KK) { t = 0; if (++c == it.Cin) { c = 0; ++o; } }
      }
      float4 q = accumulate ? *reinterpret_cast<const float4*>(dst) : make_float4(0.f, 0.f, 0.f, 0.f);
      q.x = accumulate ? __builtin_fmaf(av[0], sv[0], q.x) : av[0] * sv[0];
      q.y = accumulate ? __builtin_fmaf(av[1], sv[1], q.y) : av[1] * sv[1];
      q.z = accumulate ? __builtin_fmaf(av[2], sv[2], q.z) : av[2] * sv[2];
      q.w = accumulate ? __builtin_fmaf(av[3], sv[3], q.w) : av[3] * sv[3];
      *reinterpret_cast<float4*>(dst) = q;
    } else {
      for (int j = 0; j < 4 && idx + j < total; ++j) unpack_one(items, n, idx + j, accumulate);
    }
  }
}

hipError_t run_unpack_conv_wgrads_multi(const void* items, int n, long total, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(unpack_conv_wgrads_multi_kernel, dim3(grid_for((total + 3) / 4, 256)), dim3(256), 0, s, (const PackItem*)items, n, total, accumulate);
  return hipGetLastError();
}

hipError_t run_pack_conv_weights_multi(const void* items, int n, long total, int dtype, hipStream_t s) {
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_conv_weights_multi_kernel<bf16_t>, dim3(grid_for((total + 7) / 8, 256)), dim3(256), 0, s, (const PackItem*)items, n, total);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(pack_conv_weights_multi_kernel<f16_t>, dim3(grid_for((total + 7) / 8, 256)), dim3(256), 0, s, (const PackItem*)items, n, total);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t run_transpose_multi(const void* items, int n, int tiles, hipStream_t s) {
  hipLaunchKernelGGL(transpose_multi_x8_kernel, dim3(tiles), dim3(256), 0, s, (const TransItem*)items, n);
  return hipGetLastError();
}

hipError_t run_pack_conv_weight(const float* w, const float* sc, void* out, int Cout, int Cin, int KK, int dtype, hipStream_t s) {
  const long total = (long)Cout * Cin * KK;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, w, sc, (bf16_t*)out, Cout, Cin, KK);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<f16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, w, sc, (f16_t*)out, Cout, Cin, KK);
  else if (dtype != DT_F32)
    return hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, w, sc, (float*)out, Cout, Cin, KK);
  return hipGetLastError();
}

hipError_t run_unpack_conv_wgrad(const float* dw, const float* sc, float* out, int Cout, int Cin, int KK, int accumulate, hipStream_t s) {
  const long total = (long)Cout * Cin * KK;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, dw, sc, out, Cout, Cin, KK, accumulate);
  return hipGetLastError();
}

hipError_t run_im2col_nhwc(const void* x, void* cols, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int OH, int OW,
                           int dtype, hipStream_t s) {
  const long work = (long)B * OH * OW * KH * KW * (Cin / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(im2col_nhwc_kernel<bf16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)cols, B, H, W, Cin, KH, KW, pad, dil, OH, OW);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(im2col_nhwc_kernel<f16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const f16_t*)x, (f16_t*)cols, B, H, W, Cin, KH, KW, pad, dil, OH, OW);
  else if (dtype != DT_F32)
    return hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(im2col_nhwc_kernel<float>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const float*)x, (float*)cols, B, H, W, Cin, KH, KW, pad, dil, OH, OW);
  return hipGetLastError();
}

hipError_t run_scale_rows(const void* w, const float* sc, void* out, int R, long C, int dtype, hipStream_t s) {
  const long work = (long)R * (C / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const bf16_t*)w, sc, (bf16_t*)out, R, C);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(scale_rows_kernel<f16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const f16_t*)w, sc, (f16_t*)out, R, C);
  else if (dtype != DT_F32)
    return hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const float*)w, sc, (float*)out, R, C);
  return hipGetLastError();
}

// ---- optimizer step (configs/faster_rcnn_r101_selsa_c5.py:215-222: SGD momentum 0.9, weight decay 1e-4, grad clip 35) ----
// partial sums of squares of a flat f32 gradient buffer: one value per workgroup, fixed order (the clip norm)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[256];
  float acc = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);   // the flat buffers are 256-byte aligned
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = g4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += g[n4 * 4 + threadIdx.x] * g[n4 * 4 + threadIdx.x];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[1 + blockIdx.x] = red[0];
}

// part[0] = sum of part[1 .. nparts] in a fixed tree order
__global__ __launch_bounds__(256) void sumsq_final_kernel(float* __restrict__ part, int nparts) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[1 + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[0] = red[0];
}

// torch.optim.SGD(momentum, weight_decay, dampening 0, nesterov off) on a flat f32 buffer, with the two scalings the
// reference applies to the gradient first folded in: 1 / world_size (dist_utils.py:24-25) and the clip coefficient
// min(1, max_norm / (norm + 1e-6)) of clip_grad_norm_ over the AVERAGED gradient (max_norm <= 0: no clipping).
//   d = g * gscale * clip + wd * p;   buf = first ? d : mom * buf + d;   p -= lr * buf
__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                                       float lr, float mom, float wd, float gscale, float max_norm,
                                                       const float* __restrict__ part, int nparts, int first) {
  __shared__ float sh_clip;
  if (threadIdx.x == 0) {
    float clip = 1.f;
    if (max_norm > 0.f) {
      float ss = 0.f;
      for (int i = 0; i < nparts; ++i) ss += part[i];
      const float norm = sqrtf(ss) * gscale;  // norm of the averaged gradient
      const float c = max_norm / (norm + 1e-6f);
      clip = c < 1.f ? c : 1.f;
    }
    sh_clip = clip;
  }
  __syncthreads();
  const float k = gscale * sh_clip;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = g[i] * k + wd * p[i];
    const float b = first ? d : mom * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * b;
  }
}

constexpr int kSumsqParts = 1024;
size_t sgd_workspace_bytes() { return (size_t)(kSumsqParts + 1) * sizeof(float); }
hipError_t run_sgd_step(float* p, const float* g, float* buf, long n, float lr, float mom, float wd, float gscale, float max_norm,
                        float* part, int first, hipStream_t s) {
  if (max_norm > 0.f) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(kSumsqParts), dim3(256), 0, s, g, n, part);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, part, kSumsqParts);
  }
  hipLaunchKernelGGL(sgd_step_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, buf, n, lr, mom, wd, gscale, max_norm, part,
                     1, first);
  return hipGetLastError();
}

hipError_t run_relu_bwd(const void* dY, const void* Y, void* dZ, long n, int dtype, hipStream_t s) {
  const int g = grid_for((n + 3) / 4, 256);
  if (dtype == DT_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dY, (const bf16_t*)Y, (bf16_t*)dZ, n);
  else if (dtype == DT_F16) hipLaunchKernelGGL(relu_bwd_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)dY, (const f16_t*)Y, (f16_t*)dZ, n);
  else if (dtype != DT_F32) return hipErrorInvalidValue;
  else hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dY, (const float*)Y, (float*)dZ, n);
  return hipGetLastError();
}

int colsum_slices(int M, int N) {
  const int cols = (N + 63) / 64;
  int s = (1024 + cols - 1) / cols;          // ~1024 workgroups in all
  if (s > (M + 63) / 64) s = (M + 63) / 64;  // at least 16 rows per row lane
  return s < 1 ? 1 : (s > 256 ? 256 : s);
}
hipError_t run_colsum(const void* dY, float* db, int M, int N, long ld, int dtype, float* ws, hipStream_t s) {
  const int S = colsum_slices(M, N);
  float* part = S == 1 ? db : ws;
  const dim3 grid((N + 63) / 64, S);
  if (dtype == DT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dY, part, M, N, ld);
  else if (dtype == DT_F16) hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)dY, part, M, N, ld);
  else if (dtype != DT_F32) return hipErrorInvalidValue;
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)dY, part, M, N, ld);
  if (S > 1) hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, ws, db, N, S);
  return hipGetLastError();
}

// C[m][n] = sum over the S split-K partials ws[s][m][n] (fixed order); N % 4 == 0
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N, long ldc,
                                                            int S) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x, per_row = N / 4, total = (long)M * per_row;
  if (q >= total) return;
  const long m = q / per_row, n = (q - m * per_row) * 4, stride = (long)M * N;
  const float* src = ws + m * N + n;
  float4 acc = *reinterpret_cast<const float4*>(src);
  for (int s = 1; s < S; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(src + s * stride);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  float* dst = C + m * ldc + n;
  dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
}

// the same sum rounded once to bf16 (the relation apply pass's split-K form); N % 8 == 0, 16-byte aligned rows
// (gridDim.y > 1: that many problems of one shape, problem g's partials gs_ws floats and its output gs_c elements behind problem 0's)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ ws, T* __restrict__ C, int M, int N, long ldc,
                                                                 int S, long gs_ws, long gs_c) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x, per_row = N / 8, total = (long)M * per_row;
  if (q >= total) return;
  ws += (long)blockIdx.y * gs_ws;
  C += (long)blockIdx.y * gs_c;
  const long m = q / per_row, n = (q - m * per_row) * 8, stride = (long)M * N;
  const float* src = ws + m * N + n;
  float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  for (int s = 1; s < S; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(src + s * stride), v = *reinterpret_cast<const float4*>(src + s * stride + 4);
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  *reinterpret_cast<uint4*>(C + m * ldc + n) = make_uint4(pack2<T>(a.x, a.y), pack2<T>(a.z, a.w), pack2<T>(b.x, b.y), pack2<T>(b.z, b.w));
}

// the sum with a conv / linear epilogue: C = act(sum_s ws[s] + bias + resid), rounded once to bf16 (split-K convs of few-row
// problems: one frame through the stride-16 stages); N % 8 == 0, 16-byte aligned rows of C and resid
__global__ __launch_bounds__(256) void splitk_reduce_epi_bf16_kernel(const float* __restrict__ ws, bf16_t* __restrict__ C, int M, int N, long ldc,
                                                                     int S, const float* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                                     long ldr, int relu) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x, per_row = N / 8, total = (long)M * per_row;
  if (q >= total) return;
  const long m = q / per_row, n = (q - m * per_row) * 8, stride = (long)M * N;
  const float* src = ws + m * N + n;
  float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  for (int s = 1; s < S; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(src + s * stride), v = *reinterpret_cast<const float4*>(src + s * stride + 4);
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  if (bias) {
    const float4 u = *reinterpret_cast<const float4*>(bias + n), v = *reinterpret_cast<const float4*>(bias + n + 4);
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  if (resid) {
    const uint4 t = *reinterpret_cast<const uint4*>(resid + m * ldr + n);
    a.x += __uint_as_float(t.x << 16); a.y += __uint_as_float(t.x & 0xffff0000u);
    a.z += __uint_as_float(t.y << 16); a.w += __uint_as_float(t.y & 0xffff0000u);
    b.x += __uint_as_float(t.z << 16); b.y += __uint_as_float(t.z & 0xffff0000u);
    b.z += __uint_as_float(t.w << 16); b.w += __uint_as_float(t.w & 0xffff0000u);
  }
  if (relu) {
    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
  }
  *reinterpret_cast<uint4*>(C + m * ldc + n) = make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
}

hipError_t run_splitk_reduce_epi_bf16(const float* ws, void* C, int M, int N, long ldc, int S, const float* bias, const void* resid, long ldr,
                                      int relu, hipStream_t s) {
  const long total = (long)M * (N / 8);
  hipLaunchKernelGGL(splitk_reduce_epi_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, (bf16_t*)C, M, N, ldc, S, bias,
                     (const bf16_t*)resid, ldr, relu);
  return hipGetLastError();
}

hipError_t run_splitk_reduce_half_batched(const float* ws, void* C, int M, int N, long ldc, int S, int f16, int batch, long gs_ws, long gs_c, hipStream_t s) {
  const long total = (long)M * (N / 8);
  const dim3 grid((unsigned)((total + 255) / 256), (unsigned)(batch > 1 ? batch : 1));
  if (f16) hipLaunchKernelGGL(splitk_reduce_bf16_kernel<f16_t>, grid, dim3(256), 0, s, ws, (f16_t*)C, M, N, ldc, S, gs_ws, gs_c);
  else hipLaunchKernelGGL(splitk_reduce_bf16_kernel<bf16_t>, grid, dim3(256), 0, s, ws, (bf16_t*)C, M, N, ldc, S, gs_ws, gs_c);
  return hipGetLastError();
}
hipError_t run_splitk_reduce_bf16(const float* ws, void* C, int M, int N, long ldc, int S, hipStream_t s) {
  return run_splitk_reduce_half_batched(ws, C, M, N, ldc, S, 0, 1, 0, 0, s);
}
hipError_t run_splitk_reduce_f16(const float* ws, void* C, int M, int N, long ldc, int S, hipStream_t s) {
  return run_splitk_reduce_half_batched(ws, C, M, N, ldc, S, 1, 1, 0, 0, s);
}

hipError_t run_splitk_reduce(const float* ws, float* C, int M, int N, long ldc, int S, hipStream_t s) {
  const long total = (long)M * (N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, C, M, N, ldc, S);
  return hipGetLastError();
}

hipError_t run_det_loss(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const long long* labels, const float* label_w,
                        const float* bbox_t, const float* bbox_w, int R, float beta, float w_cls, float w_bbox, float* out3,
                        float* dlogits, const int* sel_counts, hipStream_t s) {
  hipLaunchKernelGGL(det_loss_kernel, dim3(1), dim3(256), 0, s, logits, ldl, cls_off, reg_off, ncls, labels, label_w, bbox_t, bbox_w, R,
                     beta, w_cls, w_bbox, out3, dlogits, sel_counts);
  return hipGetLastError();
}

// post (split half only): an extra factor on the probabilities -- the score pass stores them x kSplitProbScale; 1 keeps that
// scale (the relation's own apply product takes it back), 1 / kSplitProbScale hands out true probabilities (hvr_relation_probs)
hipError_t run_relation_normalize(void* P, const float* mstat, const float* lstat, int Mq, int ntile, long ldp, int dtype, float post, hipStream_t s) {
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(relation_normalize_kernel<bf16_t>, dim3(Mq), dim3(256), 0, s, (bf16_t*)P, mstat, lstat, ntile, ldp);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(relation_normalize_kernel<f16_t>, dim3(Mq), dim3(256), 0, s, (f16_t*)P, mstat, lstat, ntile, ldp);
  else if (dtype == DT_F16S)
    hipLaunchKernelGGL(relation_normalize_split_kernel, dim3(Mq), dim3(256), 0, s, (char*)P, mstat, lstat, ntile, ldp, post);
  else
    hipLaunchKernelGGL(relation_normalize_kernel<float>, dim3(Mq), dim3(256), 0, s, (float*)P, mstat, lstat, ntile, ldp);
  return hipGetLastError();
}

hipError_t run_relation_dscore(const void* P, const void* dP, const void* dO, const void* O, void* dS, int Mq, long ldp, int D,
                               long ldgo, long ldo, float scale, int dtype, hipStream_t s) {
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(relation_dscore_kernel<bf16_t>, dim3(Mq), dim3(256), 0, s, (const bf16_t*)P, (const bf16_t*)dP,
                       (const bf16_t*)dO, (const bf16_t*)O, (bf16_t*)dS, ldp, D, ldgo, ldo, scale);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(relation_dscore_kernel<f16_t>, dim3(Mq), dim3(256), 0, s, (const f16_t*)P, (const f16_t*)dP,
                       (const f16_t*)dO, (const f16_t*)O, (f16_t*)dS, ldp, D, ldgo, ldo, scale);
  else if (dtype != DT_F32)
    return hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(relation_dscore_kernel<float>, dim3(Mq), dim3(256), 0, s, (const float*)P, (const float*)dP, (const float*)dO,
                       (const float*)O, (float*)dS, ldp, D, ldgo, ldo, scale);
  return hipGetLastError();
}

hipError_t run_transpose_pad(const void* in, void* out, int R, int C, long ldx, long ldt, int dtype, hipStream_t s) {
  dim3 grid((C + 63) / 64, (int)((ldt + 63) / 64));
  // (the 2-byte kernels move raw 16-bit words: bf16 and half alike)
  const bool two = dtype == DT_BF16 || dtype == DT_F16;
  const bool wide = two && C % 8 == 0 && ldx % 8 == 0 && ldt % 8 == 0 &&
                    ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (dtype == DT_F16S) {
    if (C % 64 || ldx % 32 || ldt % 64 || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15)) return hipErrorInvalidValue;
    grid.z = 2;
    hipLaunchKernelGGL(transpose_pad_split_kernel, grid, dim3(256), 0, s, (const char*)in, (char*)out, R, C, ldx, ldt);
  } else if (wide)
    hipLaunchKernelGGL(transpose_pad_bf16x8_kernel, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, R, C, ldx, ldt);
  else if (two)
    hipLaunchKernelGGL(transpose_pad_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, R, C, ldx, ldt);
  else
    hipLaunchKernelGGL(transpose_pad_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, R, C, ldx, ldt);
  return hipGetLastError();
}

hipError_t run_im2col_stem(const float* img, void* cols, int B, int H, int W, int OH, int OW, int KP, int dtype, hipStream_t s) {
  if (dtype == DT_BF16) {
    const long work = (long)B * OH * OW * (KP / 8);
    hipLaunchKernelGGL(im2col_stem_kernel<bf16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, img, (bf16_t*)cols, B, H, W, OH, OW, KP);
  } else if (dtype == DT_F16) {
    const long work = (long)B * OH * OW * (KP / 8);
    hipLaunchKernelGGL(im2col_stem_kernel<f16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, img, (f16_t*)cols, B, H, W, OH, OW, KP);
  } else if (dtype == DT_F16S) {
    const long work = (long)B * OH * OW * (KP / 8);
    hipLaunchKernelGGL(im2col_stem_split_kernel, dim3(grid_for(work, 256)), dim3(256), 0, s, img, (char*)cols, B, H, W, OH, OW, KP);
  } else if (dtype != DT_F32) {
    return hipErrorInvalidValue;
  } else {
    const long work = (long)B * OH * OW * (KP / 4);
    hipLaunchKernelGGL(im2col_stem_kernel<float>, dim3(grid_for(work, 256)), dim3(256), 0, s, img, (float*)cols, B, H, W, OH, OW, KP);
  }
  return hipGetLastError();
}

hipError_t run_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int OH, int OW, int dtype, hipStream_t s) {
  const long work = (long)B * OH * OW * (C / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, H, W, C, OH, OW);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<f16_t>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const f16_t*)in, (f16_t*)out, B, H, W, C, OH, OW);
  else if (dtype != DT_F32)
    return hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for(work, 256)), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C, OH, OW);
  return hipGetLastError();
}

hipError_t run_cast(const void* in, void* out, long n, int from, int to, float scale, hipStream_t s) {
  const int g = grid_for((n + 3) / 4, 256);
  if (scale == 1.f && from == DT_F32 && to == DT_BF16)
    hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(g), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n);
  else if (scale == 1.f && from == DT_BF16 && to == DT_F32)
    hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(g), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n);
  else {
    // every other pair, 8 elements per thread (capi.hip checks n % 8 / n % 64 and the alignment)
    const int g8 = grid_for((n + 7) / 8, 256);
    const char* ip = (const char*)in;
    char* op = (char*)out;
#define HVR_CAST8(FI, TI, FO, TO) \
    if (from == FI && to == FO) { hipLaunchKernelGGL((cast8_kernel<TI, TO>), dim3(g8), dim3(256), 0, s, ip, op, n, scale); return hipGetLastError(); }
    HVR_CAST8(DT_F32, float, DT_BF16, bf16_t) HVR_CAST8(DT_BF16, bf16_t, DT_F32, float) HVR_CAST8(DT_F32, float, DT_F32, float)
    HVR_CAST8(DT_F32, float, DT_F16, f16_t) HVR_CAST8(DT_F16, f16_t, DT_F32, float)
    HVR_CAST8(DT_F32, float, DT_F16S, f16s_t) HVR_CAST8(DT_F16S, f16s_t, DT_F32, float)
    HVR_CAST8(DT_BF16, bf16_t, DT_F16, f16_t) HVR_CAST8(DT_F16, f16_t, DT_BF16, bf16_t)
    HVR_CAST8(DT_BF16, bf16_t, DT_F16S, f16s_t) HVR_CAST8(DT_F16S, f16s_t, DT_BF16, bf16_t)
    HVR_CAST8(DT_F16, f16_t, DT_F16S, f16s_t) HVR_CAST8(DT_F16S, f16s_t, DT_F16, f16_t)
#undef HVR_CAST8
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t run_permute(const void* in, void* out, int B, int C, int HW, int to_nhwc, int from, int to, hipStream_t s) {
  const int R = to_nhwc ? C : HW, Cn = to_nhwc ? HW : C;
  dim3 grid((Cn + 63) / 64, (R + 63) / 64, B);
  if (from == DT_F16S || to == DT_F16S) return hipErrorInvalidValue;
#define HVR_PERM(FI, TI, FO, TO) \
  if (from == FI && to == FO) { hipLaunchKernelGGL((permute_bchw_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)in, (TO*)out, C, HW, to_nhwc); return hipGetLastError(); }
  HVR_PERM(DT_F32, float, DT_F16, f16_t) HVR_PERM(DT_F16, f16_t, DT_F32, float) HVR_PERM(DT_F16, f16_t, DT_F16, f16_t)
  HVR_PERM(DT_BF16, bf16_t, DT_F16, f16_t) HVR_PERM(DT_F16, f16_t, DT_BF16, bf16_t)
#undef HVR_PERM
  if (from == DT_F32 && to == DT_F32)
    hipLaunchKernelGGL((permute_bchw_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, C, HW, to_nhwc);
  else if (from == DT_F32 && to == DT_BF16)
    hipLaunchKernelGGL((permute_bchw_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)in, (bf16_t*)out, C, HW, to_nhwc);
  else if (from == DT_BF16 && to == DT_F32)
    hipLaunchKernelGGL((permute_bchw_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)in, (float*)out, C, HW, to_nhwc);
  else
    hipLaunchKernelGGL((permute_bchw_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, C, HW, to_nhwc);
  return hipGetLastError();
}

hipError_t run_det_decode(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const float* rois, int R,
                          const float* means, const float* stds, float max_ratio, float img_h, float img_w,
                          float scale_factor, float* scores, float* boxes, hipStream_t s) {
  hipLaunchKernelGGL(det_decode_kernel, dim3((R + 63) / 64), dim3(64), 0, s, logits, ldl, cls_off, reg_off, ncls, rois, R,
                     means[0], means[1], means[2], means[3], stds[0], stds[1], stds[2], stds[3], max_ratio, img_h, img_w,
                     scale_factor, scores, boxes);
  return hipGetLastError();
}

hipError_t run_im2col_t(const void* x, void* out, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int OH, int OW, long ldt, hipStream_t s) {
  const long P = (long)B * OH * OW;
  dim3 grid((unsigned)((Cin + 63) / 64), (unsigned)((ldt + 63) / 64), (unsigned)(KH * KW));
  (void)P;
  hipLaunchKernelGGL(im2col_t_x8_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, Cin, KW, pad, dil, OH, OW, ldt);
  return hipGetLastError();
}
hipError_t run_relu_bwd_t(const void* dy, const void* y, void* dz, void* dzt, int R, int C, long ldt, hipStream_t s) {
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((ldt + 63) / 64));
  hipLaunchKernelGGL(relu_bwd_t_x8_kernel, grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dz, (bf16_t*)dzt, R, C, ldt);
  return hipGetLastError();
}

// ---- zero fill ----
// Every zeroing inside a capturable C-ABI path goes through this kernel, not hipMemsetAsync: a captured one-chain graph holding a memset
// node took a GPU memory fault when replayed behind ordinary launches without host synchronisation (ROCm 7.2;
// profiles/r04_graph_memset_fault.txt).  `bytes` a multiple of 4, p 4-byte aligned; 16-byte stores once p is 16-byte aligned.
__global__ void zero_fill_kernel(uint32_t* __restrict__ p, long words) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const long n4 = words >> 2;
    for (long j = i; j < n4; j += stride) q[j] = make_uint4(0u, 0u, 0u, 0u);
    for (long j = (n4 << 2) + i; j < words; j += stride) p[j] = 0u;
  } else {
    for (; i < words; i += stride) p[i] = 0u;
  }
}
hipError_t run_zero_fill(void* p, size_t bytes, hipStream_t s) {
  if (!p || bytes == 0) return hipSuccess;
  const long words = (long)((bytes + 3) / 4);
  long blocks = (words / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), words);
  return hipGetLastError();
}

}  // namespace hvr
