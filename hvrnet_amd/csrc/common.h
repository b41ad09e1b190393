// Shared device helpers for the hvr_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>

namespace hvr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // raw bf16 bits in memory

// dtype codes of the C ABI (include/hvr_hip.h)
enum { DT_F32 = 0, DT_BF16 = 1 };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, same rounding as torch's float->bfloat16 cast (v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

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

// Reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) -- the MFMA fragment layout keeps one
// output row in those four lanes.  gfx950's v_permlane{32,16}_swap exchange half-waves / 16-lane rows between two
// registers in one VALU op each (no LDS round trip as with ds_bpermute).
__device__ __forceinline__ float quad_group_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // {lo,lo} , {hi,hi}
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned v = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // rows {0,0,2,2} , {1,1,3,3}
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_group_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned v = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace hvr
