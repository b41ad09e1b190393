// Relation apply pass, producer / consumer form with the block weights applied by the PRODUCERS (bf16 / half, gfx950):
//     O[m, :] = (1 / L_m) sum_t 2^(m_t - M_m) (P~_t V_t)[m, :]        selsa_bbox_head.py:182, hrnmp_bbox_head.py:342
// P~ [Mq][ldp] comes from the scores pass with INTEGER block maxima m_t (log2 units, relation_bt.hip): the weight of a 128-key block
// relative to the row's largest block, 2^(m_t - M_m), is then an exact power of two, and scaling a bf16 / half number by it is a
// subtraction on its exponent field -- exact, no second rounding.
//
// Why this form.  The tile engine's apply pass (gemm_tile.h, EPI_APPLY) and pc_gemm.hip's keep two accumulator sets and fold a block's
// un-scaled partial into the running total with one FMA per accumulator register per block: 72 VALU operations per 72 MFMAs in a
// compute wave that is alone on its SIMD, i.e. ~30 % on top of the bare product (pc_gemm.hip as a plain GEMM of this shape: 45.6 us,
// with the fold: 57).  Here the four producer waves -- idle most of the time -- load the P~ rows into registers (compiler-counted
// buffer loads, four K-steps in flight), subtract the block's exponent shift from each 16-bit word (v_pk_sub_u16 with clamp: a word
// whose exponent would go below zero becomes 0), and copy them into the LDS ring; V^T goes through the same registers unscaled.  The
// compute waves run a PLAIN product (one accumulator set, pc_gemm.hip's stream: x-fragment register ring, counted lgkmcnt waits, one
// s_barrier per K-step) and multiply a row by 1 / L_m = 1 / sum_t l_t 2^(m_t - M_m) once, in the epilogue.  The per-row combine
// (M_m, L_m, the shifts of the tile's 144 rows) is computed by the compute waves while the producers' first loads are in flight.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int RA_FM = 9, RA_BM = RA_FM * 16, RA_CW = 4, RA_PW = 4, RA_NT = 64 * (RA_CW + RA_PW);
constexpr int RA_FN = 2, RA_BN = RA_CW * RA_FN * 16, RA_NS = 3;
constexpr int RA_ROWS = RA_BM + RA_BN, RA_STAGE = RA_ROWS * 128;
constexpr int RA_AP = RA_BM / 8, RA_BP = RA_BN / 8;                 // 1 KiB pieces of a K-step: 18 P~-row pieces, 16 V^T-row pieces
constexpr int RA_APW = (RA_AP + RA_PW - 1) / RA_PW, RA_BPW = RA_BP / RA_PW, RA_PPW = RA_APW + RA_BPW;   // per producer wave: 5 + 4
constexpr int RA_D = 4;                                             // K-steps a producer wave holds in registers
constexpr int RA_AHEAD = 7, RA_RING = RA_FM, RA_KB_AT = 1;          // consumer x-fragment ring (pc_gemm.hip)
constexpr int RA_TAB = RA_NS * RA_STAGE;                            // byte offset of the row table: rinv [144] f32, then shift [nt][144] u8
static_assert(RA_AHEAD + 2 == RA_RING && RA_TAB + RA_BM * 4 <= 160 * 1024, "shape");

typedef uint32_t rau32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short rau16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ra_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 ra_read128(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void ra_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ rau32x4 ra_load16(const void* base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  return __builtin_bit_cast(rau32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
#else
  (void)base; (void)voff; (void)soff;
  return rau32x4{0u, 0u, 0u, 0u};
#endif
}
// both 16-bit words of w minus the shift pattern dd, saturating at 0 (v_pk_sub_u16 clamp): x 2^-d on two positive bf16 / half values
__device__ __forceinline__ uint32_t ra_shift2(uint32_t w, uint32_t dd) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(rau16x2, w), __builtin_bit_cast(rau16x2, dd)));
}
constexpr int ra_pending(int t, int ahead_left, int fn, bool kb_issued) {
  int n = ahead_left;
  if (kb_issued && t > RA_KB_AT && t <= RA_KB_AT + RA_AHEAD) n += fn;
  return n < 15 ? n : 15;
}

}  // namespace

// HT: bf16_t (exponent field at bit 7 of a word) / f16_t (bit 10)
template <typename HT>
__global__ __launch_bounds__(RA_NT) void relation_apply_pc_kernel(const GemmParams p) {
  constexpr int FN = RA_FN, BN = RA_BN, NS = RA_NS, STAGE = RA_STAGE, D = RA_D;
  constexpr int EXP_LSB = std::is_same<HT, bf16_t>::value ? 7 : 10;
  constexpr int EXP_MAX = std::is_same<HT, bf16_t>::value ? 255 : 31;   // a shift of the whole exponent range zeroes the value
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / BN, tiles_m = (p.M + RA_BM - 1) / RA_BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int pid_m = tile / tiles_n, pid_n = tile - pid_m * tiles_n;  // n fastest: the P~ panel is shared by neighbouring tiles
  const int m0 = pid_m * RA_BM, n0 = pid_n * BN;
  const int nk = p.K / 64;   // a multiple of D (relation_apply_pc_supported); two K-steps per 128-key block
  float* const rinv = reinterpret_cast<float*>(smem + RA_TAB);
  unsigned char* const shift = reinterpret_cast<unsigned char*>(smem + RA_TAB + RA_BM * 4);   // [nt][144]

  if (wave >= RA_CW) {
    // ======================================= producer =======================================
    const int pw = wave - RA_CW;
    const char* const rs_a = (const char*)p.A;
    const char* const rs_b = (const char*)p.B;
    unsigned a_off[RA_APW];
    uint32_t a_dst[RA_APW];
    int a_row[RA_APW];
#pragma unroll
    for (int i = 0; i < RA_APW; ++i) {
      int q = i * RA_PW + pw;
      q = q < RA_AP ? q : q - RA_PW;   // (a wave without a piece in the last slot repeats its previous one: the same bytes to the same place)
      const int row = q * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      a_row[i] = row;
      a_off[i] = (unsigned)((int)((long)m * p.lda * 2) + c * 16);
      a_dst[i] = (uint32_t)q * 1024u + (uint32_t)lane * 16u;
    }
    unsigned b_off[RA_BPW];
#pragma unroll
    for (int i = 0; i < RA_BPW; ++i) {
      const int q = i * RA_PW + pw, row = q * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      b_off[i] = (unsigned)((int)((long)(n0 + row) * p.ldb * 2) + c * 16);
    }
    const uint32_t b_dst0 = (uint32_t)RA_BM * 128u + (uint32_t)pw * 1024u + (uint32_t)lane * 16u;   // + i * 4096

    rau32x4 R[D][RA_PPW];
    auto request = [&](auto DD, int kt) {   // (clamped: the tail re-requests the last K-step, nobody copies it)
      constexpr int d = decltype(DD)::value;
      kt = kt < nk ? kt : nk - 1;
      const int sk = __builtin_amdgcn_readfirstlane(kt * 128);
#pragma unroll
      for (int i = 0; i < RA_APW; ++i) R[d][i] = ra_load16(rs_a, a_off[i], sk);
#pragma unroll
      for (int i = 0; i < RA_BPW; ++i) R[d][RA_APW + i] = ra_load16(rs_b, b_off[i], sk);
    };
    auto commit = [&](auto DD, int kt) {
      constexpr int d = decltype(DD)::value;
      char* stage = smem + (kt % NS) * STAGE;
      const unsigned char* sh_t = shift + (kt >> 1) * RA_BM;
#pragma unroll
      for (int i = 0; i < RA_APW; ++i) {
        const uint32_t dd = (uint32_t)sh_t[a_row[i]] << EXP_LSB;
        const uint32_t pat = dd | (dd << 16);
        rau32x4 v = R[d][i];
        v[0] = ra_shift2(v[0], pat); v[1] = ra_shift2(v[1], pat); v[2] = ra_shift2(v[2], pat); v[3] = ra_shift2(v[3], pat);
        *reinterpret_cast<rau32x4*>(stage + a_dst[i]) = v;
      }
#pragma unroll
      for (int i = 0; i < RA_BPW; ++i) *reinterpret_cast<rau32x4*>(stage + b_dst0 + i * 4096) = R[d][RA_APW + i];
    };
    static_for<D>([&](auto DD) { request(DD, decltype(DD)::value); });
    __builtin_amdgcn_s_barrier();   // T: the compute waves have written the row table
    // Step j's slot (j mod NS) is free after B_{j-1} (pc_gemm.hip): copy, wait for the LDS writes, B_j, re-request the freed registers
    for (int j0 = 0; j0 < nk; j0 += D) {
      static_for<D>([&](auto DD) {
        const int j = j0 + decltype(DD)::value;
        commit(DD, j);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // B_j
        request(DD, j + D);
      });
    }
    return;
  }

  // ========================================= compute waves =========================================
  // ---- the tile's row table, while the producers' first loads are on their way: M = max_t m_t (integers), L = sum_t l_t 2^(m_t - M),
  // shift[t][row] = min(M - m_t, whole exponent range) ----
  if (tid < RA_BM) {
    const int m = m0 + tid;
    const long row = (long)(m < p.M ? m : p.M - 1) * p.ntile;
    constexpr int TB = 12;   // loads in batches: one memory round trip per 12 blocks
    float mx = -INFINITY;
    for (int t0 = 0; t0 < p.ntile; t0 += TB) {
      float mt[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) mt[u] = p.mstat[row + (t0 + u < p.ntile ? t0 + u : p.ntile - 1)];
#pragma unroll
      for (int u = 0; u < TB; ++u) mx = fmaxf(mx, mt[u]);
    }
    float l = 0.f;
    for (int t0 = 0; t0 < p.ntile; t0 += TB) {
      float mt[TB], lt[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int tt = t0 + u < p.ntile ? t0 + u : p.ntile - 1;
        mt[u] = p.mstat[row + tt];
        lt[u] = p.lstat[row + tt];
      }
#pragma unroll
      for (int u = 0; u < TB; ++u)
        if (t0 + u < p.ntile) {
          const float dlt = mx - mt[u];   // a non-negative integer (or +inf for a block without keys: weight 0)
          l += lt[u] * __builtin_amdgcn_exp2f(-dlt);
          shift[(t0 + u) * RA_BM + tid] = (unsigned char)(dlt < (float)EXP_MAX ? (int)dlt : EXP_MAX);
        }
    }
    rinv[tid] = 1.f / l;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // T

  const int wn = wave;
  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = ra_lds_off(smem) + frag_row * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = ra_lds_off(smem) + RA_BM * 128 + (wn * FN * 16 + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  f32x4 acc[RA_FM][FN];
#pragma unroll
  for (int i = 0; i < RA_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 fb[2][FN];
  uint4 fa[RA_RING];
  auto read_a = [&](auto SLOT, auto I, auto KK, uint32_t soff) {
    constexpr int slot = decltype(SLOT)::value, i = decltype(I)::value, kk = decltype(KK)::value;
    fa[slot] = ra_read128<i * 2048>((a_lane + soff) ^ (kk ? 64u : 0u));
  };
  auto read_b = [&](auto KK, uint32_t soff) {
    constexpr int kk = decltype(KK)::value;
    static_for<FN>([&](auto J) { fb[kk][decltype(J)::value] = ra_read128<decltype(J)::value * 2048>((b_lane + soff) ^ (kk ? 64u : 0u)); });
  };
  auto half = [&](auto KK, auto MORE, uint32_t soff, uint32_t soff_ahead) {
    constexpr int kk = decltype(KK)::value;
    constexpr bool more = decltype(MORE)::value;
    static_for<RA_FM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int in_half = (i + RA_AHEAD - 1 < RA_FM) ? RA_AHEAD - 1 : RA_FM - 1 - i;
      constexpr int ahead_left = more ? RA_AHEAD - 1 : in_half;
      __builtin_amdgcn_sched_barrier(0);
      ra_wait_lgkm<ra_pending(i, ahead_left, FN, more)>();
      __builtin_amdgcn_sched_barrier(0);
      static_for<FN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        acc[i][j] = mfma_half<HT>(fb[kk][j], fa[i], acc[i][j]);
      });
      __builtin_amdgcn_sched_barrier(0);
      constexpr int t = i + RA_AHEAD;
      constexpr int nslot = t % RA_RING;
      if constexpr (t < RA_FM) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t>{}, KK, soff);
      } else if constexpr (more) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t - RA_FM>{}, std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      }
      if constexpr (i == RA_KB_AT && more) read_b(std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  __builtin_amdgcn_s_barrier();  // B_0
  read_b(std::integral_constant<int, 0>{}, 0u);
  static_for<RA_AHEAD>([&](auto T) { read_a(T, T, std::integral_constant<int, 0>{}, 0u); });
  constexpr std::true_type Y{};
  constexpr std::false_type N{};
  constexpr std::integral_constant<int, 0> K0{};
  constexpr std::integral_constant<int, 1> K1{};
  {
    int k = 0;
    for (; k + 1 < nk; ++k) {
      const uint32_t soff = (uint32_t)(k % NS) * STAGE, snext = (uint32_t)((k + 1) % NS) * STAGE;
      half(K0, Y, soff, soff);
      __builtin_amdgcn_s_barrier();  // B_{k+1}
      half(K1, Y, soff, snext);
    }
    const uint32_t soff = (uint32_t)(k % NS) * STAGE;
    half(K0, Y, soff, soff);
    half(K1, N, soff, soff);
  }

  // ---------------- epilogue (compute waves only: the producers have exited): rows x 1 / L, whole 16-byte row segments out ----------------
  constexpr int LDW = BN + 4, CH = BN / 8, CT = RA_CW * 64;
  float* ebuf = reinterpret_cast<float*>(smem);
  __builtin_amdgcn_s_barrier();  // every compute wave is done reading the ring
#pragma unroll
  for (int i = 0; i < RA_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      *reinterpret_cast<f32x4*>(ebuf + (i * 16 + frag_row) * LDW + (wn * FN + j) * 16 + frag_grp * 4) = acc[i][j];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < (RA_BM * CH + CT - 1) / CT; ++it) {
    const int c = it * CT + tid;
    if (c >= RA_BM * CH) continue;
    const int r = c / CH, cc = c - r * CH;
    const int m = m0 + r, n = n0 + cc * 8;
    if (m >= p.M) continue;
    const float s = rinv[r];
    const float4 lo = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8);
    const float4 hi = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8 + 4);
    HT* cp = reinterpret_cast<HT*>(p.C) + (long)m * p.ldc + n;
    *reinterpret_cast<uint4*>(cp) = make_uint4(pack2<HT>(lo.x * s, lo.y * s), pack2<HT>(lo.z * s, lo.w * s), pack2<HT>(hi.x * s, hi.y * s), pack2<HT>(hi.z * s, hi.w * s));
  }
}

// window-sized bf16 / half apply passes whose statistics carry INTEGER block maxima: whole 128-column tiles of O, a K loop of whole
// register-ring rounds, the row table behind the ring
bool relation_apply_pc_supported(const GemmParams& p) {
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16) || !p.staging || p.out_f32 || p.ksplit_steps > 0) return false;
  if (p.N % RA_BN || p.K % (64 * RA_D) || p.K / 128 != p.ntile || p.ldc % 8 || p.lda % 8 || p.ldb % 8 || !p.mstat || !p.lstat) return false;
  if ((size_t)RA_TAB + RA_BM * 4 + (size_t)p.ntile * RA_BM > (size_t)160 * 1024) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C);
  if (al & 15) return false;
  if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  return true;
}

template <typename T>
static hipError_t launch_relation_apply_pc(const GemmParams& p, hipStream_t stream) {
  const size_t lds = (size_t)RA_TAB + RA_BM * 4 + (((size_t)p.ntile * RA_BM + 15) & ~(size_t)15);
  auto kern = relation_apply_pc_kernel<T>;
  static std::atomic<unsigned> attr_set{0};   // (the attribute is per device)
  per_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int tiles = ((p.M + RA_BM - 1) / RA_BM) * (p.N / RA_BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(RA_NT), lds, stream, p);
  return hipGetLastError();
}

hipError_t run_relation_apply_pc(const GemmParams& p, hipStream_t stream) {
  return p.dtype == DT_F16 ? launch_relation_apply_pc<f16_t>(p, stream) : launch_relation_apply_pc<bf16_t>(p, stream);
}

}  // namespace hvr
