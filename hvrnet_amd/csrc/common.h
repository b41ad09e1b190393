// Shared device helpers for the hvr_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <utility>
#include <type_traits>
#include <stdint.h>

namespace hvr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // raw bf16 bits in memory

// dtype codes of the C ABI (include/hvr_hip.h)
enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2, DT_F16S = 3 };

// IEEE half operands (v_mfma_f32_16x16x32_f16: the bf16 rate with a 10-bit mantissa): raw bits, a type of its own so that the
// kernels can be instantiated on it next to bf16_t
enum class f16_t : unsigned short {};
// "Split half" operands (DT_F16S): a logical element x is the pair  hi = half(x),  lo = half(x - hi)  -- 22 significant bits for
// |x| >= 2^-3, an absolute error below 2^-24 beneath that (lo is then a half subnormal, which the MFMA honours) -- and a product
// is three half MFMAs, hi*hi + hi*lo + lo*hi, accumulated in f32: f32-grade results at a third of the half rate instead of the
// exact-f32 MFMA's sixteenth.  Memory layout of a row of C elements (C a multiple of 32): C / 32 groups of
// [32 hi halves (64 bytes)][32 lo halves (64 bytes)], i.e. 4 bytes per logical element, and one 128-byte line holds BOTH planes
// of a 32-element K-step: the tile engine stages it with the loader it uses for every other format and issues the three MFMAs
// from one LDS image (gemm_tile.h).  f16s_t is the 4-byte container element.
struct f16s_t { uint32_t v; };
constexpr int kSplitGroup = 32;     // logical elements per [hi | lo] group
constexpr int kSplitPlane = 64;     // byte distance from a hi chunk to its lo chunk
constexpr float kSplitProbScale = 4096.f;   // the relation's probabilities are stored x 2^12 in the split format (gemm_tile.h, capi.hip)
constexpr float kHalfMax = 65504.f;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, same rounding as torch's float->bfloat16 cast (v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

typedef _Float16 hw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {  // round-to-nearest-even, as torch's float->half cast
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2));
}
__device__ __forceinline__ void unpack2h(uint32_t u, float& lo, float& hi) {
  const hw_f16x2 h = __builtin_bit_cast(hw_f16x2, u);
  lo = (float)h[0];
  hi = (float)h[1];
}
__device__ __forceinline__ f16_t f2h(float f) { return (f16_t)(pack2h(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float h2f(f16_t v) {
  float lo, hi;
  unpack2h((uint32_t)(unsigned short)v, lo, hi);
  return lo;
}
// two f32 values -> their split-half pair words (values beyond the half range saturate instead of turning into inf - inf)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  a = fminf(fmaxf(a, -kHalfMax), kHalfMax);
  b = fminf(fmaxf(b, -kHalfMax), kHalfMax);
  hi = pack2h(a, b);
  float ha, hb;
  unpack2h(hi, ha, hb);
  lo = pack2h(a - ha, b - hb);
}
__device__ __forceinline__ void merge2(uint32_t hi, uint32_t lo, float& a, float& b) {
  float ha, hb, la, lb;
  unpack2h(hi, ha, hb);
  unpack2h(lo, la, lb);
  a = ha + la;
  b = hb + lb;
}
// byte offset of logical column n inside a split-half row (hi plane; the lo plane is kSplitPlane bytes further)
__device__ __host__ __forceinline__ long split_col_bytes(long n) { return (n >> 5) * 128 + (n & 31) * 2; }

// 2-byte operand types: pack / unpack of a pair
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack2bf(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack2h(lo, hi); }
template <typename T> __device__ __forceinline__ void unpack2(uint32_t u, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t u, float& lo, float& hi) { unpack2h(u, lo, hi); }

// one 16x16x32 MFMA on 2-byte operands: bf16_t -> v_mfma_f32_16x16x32_bf16, f16_t -> v_mfma_f32_16x16x32_f16
template <typename T> __device__ __forceinline__ f32x4 mfma_half(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (std::is_same<T, bf16_t>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<f16_t> {
  static constexpr int kCode = DT_F16;
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float load(const f16_t* p) { return h2f(*p); }
  __device__ static __forceinline__ void store(f16_t* p, float v) { *p = f2h(v); }
};
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
__device__ __forceinline__ void store4(f16_t* p, const float v[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
}
__device__ __forceinline__ void load4(const f16_t* p, float v[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  unpack2h(t.x, v[0], v[1]);
  unpack2h(t.y, v[2], v[3]);
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

// One-time per-device kernel attribute setup (hipFuncSetAttribute is per device: a process driving several GPUs needs it on each).
// `done` holds one bit per device 0..31; a device beyond that has no bit and runs the setup on every launch instead of borrowing
// another device's (an aliased slot would leave it without its attribute).  Two threads that race on a fresh device both run the
// setup (it is idempotent) -- neither launches before its own call has returned.
constexpr int kMaxDevices = 32;
template <typename F> inline void per_device_once(std::atomic<unsigned>& done, F&& setup) {
  int d = -1;
  (void)hipGetDevice(&d);
  const bool slot = d >= 0 && d < kMaxDevices;
  if (slot && (done.load(std::memory_order_acquire) & (1u << d))) return;
  setup();
  if (slot) done.fetch_or(1u << d, std::memory_order_release);
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
