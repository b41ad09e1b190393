// Shared device helpers for the hvr_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // raw bf16 bits in memory

// dtype codes of the C ABI (include/hvr_hip.h)
enum { DT_F32 = 0, DT_BF16 = 1 };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, same rounding as torch's float->bfloat16 cast
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kCode = DT_F32;
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kCode = DT_BF16;
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

// 4 consecutive elements, vectorised
__device__ __forceinline__ void store4(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive logical ids
// land on the same XCD (block b runs on XCD b % 8) so tiles that share an
// operand panel hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace hvr
