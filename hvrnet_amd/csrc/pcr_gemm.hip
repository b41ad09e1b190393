// Producer / consumer MFMA tile kernel with REGISTER-STAGED producers (bf16 / half, gfx950): the N = 256 products of layer 3 --
// conv1 (1x1, 1024 -> 256) and conv2 (3x3, 256 -> 256) of its 23 Bottlenecks, mmdet/models/backbones/resnet.py:224-246 -- whose
// one-round grid leaves a CU a 144 x 256 output tile.
//
// Why a third engine for this shape.  The tile engine (gemm.hip) runs these at 0.34 MFMA-busy: every wave issues MFMAs, fragment reads
// (0.61 per MFMA on 144 x 32 wave tiles) AND its share of the K-step's LDS-DMA, and a 3-slot LDS ring (3 x 51 KB: all the LDS there is)
// keeps the DMA one K-step ahead at best.  pc_gemm.hip takes the DMA issue off the MFMA waves (4 producer + 4 consumer waves, 144 x 64
// consumer tiles: 0.36 reads per MFMA) but its producers also feed the ring by LDS-DMA, so the look-ahead is still what the LDS
// holds: with three slots B_{j+1} waits for a DMA issued one consumer K-step earlier -- measured 31.8 / 54.3 us against the engine's
// 27.5 / 44.9 on the two products as plain GEMMs.  Here the producers load into REGISTERS (ordinary buffer loads the compiler
// counts: out-of-image taps are offsets past the resource and come back as zeros) and copy a K-step into the ring when its slot is
// free: their 4 x 200 idle registers hold FOUR more K-steps in flight, the ring is only the hand-over, and no wait of the pipeline is
// ever shorter than four consumer K-steps.  The consumer stream is pc_gemm.hip's (x-fragment register ring, counted lgkmcnt waits,
// one s_barrier in the middle of a K-step); so is the MFMA order per output element, i.e. the tile engine's: bit-identical outputs.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int PR_FM = 9, PR_BM = PR_FM * 16, PR_CW = 4, PR_PW = 4, PR_NT = 64 * (PR_CW + PR_PW);
constexpr int PR_FN = 4, PR_BN = PR_CW * PR_FN * 16, PR_NS = 3;
constexpr int PR_ROWS = PR_BM + PR_BN, PR_STAGE = PR_ROWS * 128;
constexpr int PR_AP = PR_BM / 8, PR_BP = PR_BN / 8;                 // 1 KiB pieces of a K-step: 18 x-row pieces, 32 weight-row pieces
constexpr int PR_APW = (PR_AP + PR_PW - 1) / PR_PW, PR_BPW = PR_BP / PR_PW, PR_PPW = PR_APW + PR_BPW;   // per producer wave: 5 + 8
constexpr int PR_D = 4;                                             // K-steps a producer wave holds in registers
constexpr int PR_AHEAD = 7, PR_RING = PR_FM, PR_KB_AT = 1;          // consumer x-fragment ring (pc_gemm.hip)
static_assert(PR_AHEAD + 2 == PR_RING && PR_NS * PR_STAGE <= 160 * 1024, "shape");

typedef uint32_t pru32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pr_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 pr_read128(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void pr_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// 16 bytes per lane through buffer addressing: base (uniform) + voff (per lane) + soff (uniform); offsets from 2^31 up are outside
// the resource and read as zeros (a conv's out-of-image taps)
__device__ __forceinline__ pru32x4 pr_load16(const void* base, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  return __builtin_bit_cast(pru32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
#else
  (void)base; (void)voff; (void)soff;
  return pru32x4{0u, 0u, 0u, 0u};
#endif
}
constexpr int pr_pending(int t, int ahead_left, int fn, bool kb_issued) {
  int n = ahead_left;
  if (kb_issued && t > PR_KB_AT && t <= PR_KB_AT + PR_AHEAD) n += fn;
  return n < 15 ? n : 15;
}

}  // namespace

// CONV: the x operand is an NHWC map gathered per filter tap (implicit GEMM); a template parameter so that the plain product's
// producer loop carries no tap arithmetic
template <typename HT, bool CONV>
__global__ __launch_bounds__(PR_NT) void pcr_tile_kernel(const GemmParams p) {
  constexpr int FN = PR_FN, BN = PR_BN, NS = PR_NS, STAGE = PR_STAGE, D = PR_D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / BN, tiles_m = (p.M + PR_BM - 1) / PR_BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int pid_m = tile / tiles_n, pid_n = tile - pid_m * tiles_n;
  const int m0 = pid_m * PR_BM, n0 = pid_n * BN;
  const int nk = p.K / 64;   // a multiple of D (pcr_supported)

  if (wave >= PR_CW) {
    // ======================================= producer =======================================
    const int pw = wave - PR_CW;
    constexpr unsigned kOob = 0x80000000u;
    int a_bias = 0;
    if constexpr (CONV) a_bias = (int)(((long)p.pad * p.W + p.pad) * p.Cin * 2);
    const char* const rs_a = (const char*)p.A - a_bias;
    const char* const rs_b = (const char*)p.B;
    // slot i < APW: x-row piece q = i * 4 + pw (a wave without a piece in the last slot repeats its previous one: the same bytes to
    // the same place); slot APW + i: weight-row piece i * 4 + pw
    unsigned a_off[PR_APW];
    int a_yx[PR_APW];
    uint32_t a_dst[PR_APW];
#pragma unroll
    for (int i = 0; i < PR_APW; ++i) {
      int q = i * PR_PW + pw;
      q = q < PR_AP ? q : q - PR_PW;
      const int row = q * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      if constexpr (CONV) {
        const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
        const int iy = oy * p.stride - p.pad, ix = ox * p.stride - p.pad;
        a_yx[i] = (iy << 16) | (ix & 0xffff);
        a_off[i] = (unsigned)((int)((((long)b * p.H + iy) * p.W + ix) * (long)p.Cin * 2) + c * 16 + a_bias);
      } else {
        a_yx[i] = 0;
        a_off[i] = (unsigned)((int)((long)m * p.lda * 2) + c * 16);
      }
      a_dst[i] = (uint32_t)q * 1024u + (uint32_t)lane * 16u;
    }
    unsigned b_off[PR_BPW];
#pragma unroll
    for (int i = 0; i < PR_BPW; ++i) {
      const int q = i * PR_PW + pw, row = q * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      b_off[i] = (unsigned)((int)((long)(n0 + row) * p.ldb * 2) + c * 16);
    }
    const uint32_t b_dst0 = (uint32_t)PR_BM * 128u + (uint32_t)pw * 1024u + (uint32_t)lane * 16u;   // + i * 4096

    pru32x4 R[D][PR_PPW];
    // requests K-step kt (clamped: the tail re-requests the last one, nobody copies it) into register set DD
    auto request = [&](auto DD, int kt) {
      constexpr int d = decltype(DD)::value;
      kt = kt < nk ? kt : nk - 1;
      int a_koff, dy = 0, dx = 0;
      if constexpr (CONV) {
        const int k = kt * 64, tap = k / p.Cin, cin0 = k - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        dy = ky * p.dil;
        dx = kx * p.dil;
        a_koff = ((dy * p.W + dx) * p.Cin + cin0) * 2;
      } else {
        a_koff = kt * 128;
      }
      const int sa = __builtin_amdgcn_readfirstlane(a_koff), sb = __builtin_amdgcn_readfirstlane(kt * 128);
#pragma unroll
      for (int i = 0; i < PR_APW; ++i) {
        unsigned voff = a_off[i];
        if constexpr (CONV) {
          const bool ok = (unsigned)((a_yx[i] >> 16) + dy) < (unsigned)p.H && (unsigned)((short)a_yx[i] + dx) < (unsigned)p.W;
          voff = ok ? voff : kOob;
        }
        R[d][i] = pr_load16(rs_a, voff, sa);
      }
#pragma unroll
      for (int i = 0; i < PR_BPW; ++i) R[d][PR_APW + i] = pr_load16(rs_b, b_off[i], sb);
    };
    // copies register set DD (K-step kt) into its ring slot
    auto commit = [&](auto DD, int kt) {
      constexpr int d = decltype(DD)::value;
      char* stage = smem + (kt % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < PR_APW; ++i) *reinterpret_cast<pru32x4*>(stage + a_dst[i]) = R[d][i];
#pragma unroll
      for (int i = 0; i < PR_BPW; ++i) *reinterpret_cast<pru32x4*>(stage + b_dst0 + i * 4096) = R[d][PR_APW + i];
    };
    static_for<D>([&](auto DD) { request(DD, decltype(DD)::value); });
    // Step j's slot (j mod NS) is free after B_{j-1}: the consumers took step j - NS's last fragment before they reached it (pc_gemm.hip).
    // Per step: copy, wait for the LDS writes, B_j, re-request the freed register set four steps ahead.
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

  // ========================================= consumer (pc_gemm.hip's stream) =========================================
  const int wn = wave;
  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = pr_lds_off(smem) + frag_row * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = pr_lds_off(smem) + PR_BM * 128 + (wn * FN * 16 + frag_row) * 128 + ((frag_grp ^ swz) * 16);

  f32x4 acc[PR_FM][FN];
#pragma unroll
  for (int i = 0; i < PR_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 fb[2][FN];
  uint4 fa[PR_RING];
  auto read_a = [&](auto SLOT, auto I, auto KK, uint32_t soff) {
    constexpr int slot = decltype(SLOT)::value, i = decltype(I)::value, kk = decltype(KK)::value;
    fa[slot] = pr_read128<i * 2048>((a_lane + soff) ^ (kk ? 64u : 0u));
  };
  auto read_b = [&](auto KK, uint32_t soff) {
    constexpr int kk = decltype(KK)::value;
    static_for<FN>([&](auto J) { fb[kk][decltype(J)::value] = pr_read128<decltype(J)::value * 2048>((b_lane + soff) ^ (kk ? 64u : 0u)); });
  };
  auto half = [&](auto KK, auto MORE, uint32_t soff, uint32_t soff_ahead) {
    constexpr int kk = decltype(KK)::value;
    constexpr bool more = decltype(MORE)::value;
    static_for<PR_FM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int in_half = (i + PR_AHEAD - 1 < PR_FM) ? PR_AHEAD - 1 : PR_FM - 1 - i;
      constexpr int ahead_left = more ? PR_AHEAD - 1 : in_half;
      __builtin_amdgcn_sched_barrier(0);
      pr_wait_lgkm<pr_pending(i, ahead_left, FN, more)>();
      __builtin_amdgcn_sched_barrier(0);
      static_for<FN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        acc[i][j] = mfma_half<HT>(fb[kk][j], fa[i], acc[i][j]);
      });
      __builtin_amdgcn_sched_barrier(0);
      constexpr int t = i + PR_AHEAD;
      constexpr int nslot = t % PR_RING;
      if constexpr (t < PR_FM) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t>{}, KK, soff);
      } else if constexpr (more) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t - PR_FM>{}, std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      }
      if constexpr (i == PR_KB_AT && more) read_b(std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  __builtin_amdgcn_s_barrier();  // B_0
  read_b(std::integral_constant<int, 0>{}, 0u);
  static_for<PR_AHEAD>([&](auto T) { read_a(T, T, std::integral_constant<int, 0>{}, 0u); });
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

  // ---------------- epilogue (compute waves only: the producers have exited): pc_gemm.hip's ----------------
  constexpr int LDW = BN + 4, CH = BN / 8, CT = PR_CW * 64;
  float* ebuf = reinterpret_cast<float*>(smem);
  __builtin_amdgcn_s_barrier();  // every compute wave is done reading the ring
  float4 bvj[FN];
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < FN; ++j) bvj[j] = *reinterpret_cast<const float4*>(p.bias + n0 + (wn * FN + j) * 16 + frag_grp * 4);
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) bvj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < PR_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = (wn * FN + j) * 16 + frag_grp * 4;
      f32x4 v = acc[i][j];
      v[0] += bvj[j].x; v[1] += bvj[j].y; v[2] += bvj[j].z; v[3] += bvj[j].w;
      *reinterpret_cast<f32x4*>(ebuf + (i * 16 + frag_row) * LDW + col) = v;
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < (PR_BM * CH + CT - 1) / CT; ++it) {
    const int c = it * CT + tid;
    if (c >= PR_BM * CH) continue;
    const int r = c / CH, cc = c - r * CH;
    const int m = m0 + r, n = n0 + cc * 8;
    if (m >= p.M) continue;
    float v[8];
    const float4 lo = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8);
    const float4 hi = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8 + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    if (p.resid) {
      const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const HT*>(p.resid) + (long)m * p.ldr + n);
      float rv[8];
      unpack2<HT>(t.x, rv[0], rv[1]); unpack2<HT>(t.y, rv[2], rv[3]); unpack2<HT>(t.z, rv[4], rv[5]); unpack2<HT>(t.w, rv[6], rv[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    HT* cp = reinterpret_cast<HT*>(p.C) + (long)m * p.ldc + n;
    *reinterpret_cast<uint4*>(cp) = make_uint4(pack2<HT>(v[0], v[1]), pack2<HT>(v[2], v[3]), pack2<HT>(v[4], v[5]), pack2<HT>(v[6], v[7]));
  }
}

// bf16 / half products (plain or implicit-GEMM conv) with whole 256-column tiles, whole K-steps inside one filter tap, a K loop of
// whole register-ring rounds, bf16 / half output
bool pcr_supported(const GemmParams& p) {
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16) || !p.staging || p.out_f32 || p.ksplit_steps > 0 || p.s2 > 0) return false;
  if (p.N % PR_BN || p.K % (64 * PR_D) || p.ldc % 8 || p.lda % 8 || p.ldb % 8) return false;
  if (p.resid && (p.ldr % 8 || (reinterpret_cast<uintptr_t>(p.resid) & 15))) return false;
  if (p.conv && p.Cin % 64) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.bias);
  if (al & 15) return false;
  if ((long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  return true;
}

template <typename T, bool CONV>
static hipError_t launch_pcr(const GemmParams& p, hipStream_t stream) {
  constexpr size_t ring = (size_t)PR_NS * PR_STAGE, epi = (size_t)PR_BM * (PR_BN + 4) * 4;
  constexpr size_t lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = pcr_tile_kernel<T, CONV>;
  static std::atomic<unsigned> attr_set{0};   // (the attribute is per device)
  per_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int tiles = ((p.M + PR_BM - 1) / PR_BM) * (p.N / PR_BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PR_NT), lds, stream, p);
  return hipGetLastError();
}

hipError_t run_pcr(const GemmParams& p, hipStream_t stream) {
  if (p.conv) return p.dtype == DT_F16 ? launch_pcr<f16_t, true>(p, stream) : launch_pcr<bf16_t, true>(p, stream);
  return p.dtype == DT_F16 ? launch_pcr<f16_t, false>(p, stream) : launch_pcr<bf16_t, false>(p, stream);
}

}  // namespace hvr
