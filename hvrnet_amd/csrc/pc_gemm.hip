// Producer / consumer MFMA tile kernel (bf16, gfx950): the dense contractions whose one-round tile grid leaves a CU
// a 144-row output tile -- the relation apply pass O = sum_t g[:,t] (P~_t V_t) (selsa_bbox_head.py:182,
// hrnmp_bbox_head.py:342) and the plain products of the same shape (fc layers, 1x1 convs).
//
// Why a second engine next to gemm.hip's tile_kernel: there every wave does everything -- fragment reads, MFMAs AND
// its share of the K-step's global->LDS DMA.  Measured on this chip (DESIGN.md section 3, round-1 elimination builds of the
// apply pass): 57 us with both, 54 without the DMA, 48 without the fragment reads, 31 with neither -- each
// `global_load_lds` costs its wave 60-100 cycles of issue (the CU's address path takes one 1 KiB piece per ~16 cycles and
// all waves arrive together), during which that wave's MFMA stream stands still, and with every wave running the same
// stream in step there is nobody to fill the hole.  Here the roles are split:
//   * 4 COMPUTE waves, one per SIMD, own 144 x (16 FN) output columns each.  Their instruction stream is MFMAs and
//     fragment reads only: the 9 x-fragments of a half K-step run through a small register ring (fragment t + AHEAD is
//     requested when fragment t is consumed, counted lgkmcnt waits -- the LDS returns in order), the other half's
//     weight fragments are requested in the shadow of the first items;
//   * 4 PRODUCER waves (one per SIMD beside its compute wave: DMA issue goes down the vector-memory port, MFMAs down
//     the matrix port) issue every 1 KiB DMA piece of the NS-slot LDS ring, wait for them with counted vmcnt and
//     certify a landed K-step to the consumers at ONE s_barrier per K-step.  They never touch a fragment, exit after
//     the last K-step, and leave the epilogue to the compute waves.
// Barrier B_k (k >= 1) sits in the MIDDLE of the consumers' K-step k - 1 and says "K-step k is in the LDS"; the second
// half of step k - 1 already requests step k's first fragments, so no K-step starts with a fragment round trip.  A ring
// slot is handed back to the producers one barrier after its last fragment read was CONSUMED (not merely issued):
// after B_j they refill the slot of step j - 2 with step j - 2 + NS.
//
// LDS image, XOR swizzle and the MFMA operand order (weights as the "A" operand: a lane ends with 4 consecutive n of
// one output row) are gemm.hip's; so are the apply pass's two accumulator sets and block weights g = 2^(m_t - m*) / L.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int PC_FM = 9, PC_BM = PC_FM * 16, PC_CW = 4, PC_PW = 4, PC_NT = 64 * (PC_CW + PC_PW);
#ifndef HVR_PC_AHEAD
#define HVR_PC_AHEAD 7
#endif
// the x-fragment ring has one slot per item of a half (slot = item index: static for every half), read-ahead AHEAD items:
// item t + AHEAD lands in the slot of item t - 2, whose MFMAs were issued two items ago
constexpr int PC_AHEAD = HVR_PC_AHEAD, PC_RING = PC_FM;
static_assert(PC_AHEAD + 2 == PC_RING, "a requested fragment replaces the one consumed two items earlier");
constexpr int PC_KB_AT = 1;  // the other half's weight fragments are requested after this item

__device__ __forceinline__ uint32_t pc_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 pc_read128(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ float pc_read32(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void pc_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void pc_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ float pc_load_untracked(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void pc_landed(float& v) { asm volatile("" : "+v"(v)); }

template <typename HT, bool ZERO> __device__ __forceinline__ void pc_mma(const uint4& w, const uint4& x, f32x4& acc) {
  const f32x4 c = ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc;
  acc = mfma_half<HT>(w, x, c);
}

// fragment reads that may still be outstanding when item t's x-fragment is needed (t = position inside a half of PC_FM
// items): the read-ahead behind it -- `ahead_left` of them exist (PC_AHEAD - 1 except at the very end of the K loop) -- and,
// around PC_KB_AT, the FN weight fragments of the other half, which are requested after item PC_KB_AT's own read-ahead
// (and, in the last half of an apply block, the 9 block-weight reads requested after item 0's read-ahead).  lgkmcnt is a
// 4-bit count: 15 stands for "15 or more", which then waits for a few reads more than necessary -- never fewer
constexpr int pc_pending(int t, int ahead_left, int fn, bool kb_issued, bool g_issued) {
  int n = ahead_left;
  if (kb_issued && t > PC_KB_AT && t <= PC_KB_AT + PC_AHEAD) n += fn;
  if (g_issued && t > 0 && t <= PC_AHEAD) n += PC_FM;
  return n < 15 ? n : 15;
}

}  // namespace

template <typename HT, int FN, int EPI, int NS>   // HT: bf16_t / f16_t
__global__ __launch_bounds__(PC_NT) void pc_tile_kernel(const GemmParams p) {
  constexpr int BN = PC_CW * FN * 16;
  constexpr int ROWS = PC_BM + BN;                       // rows of one LDS stage image: x rows, then weight rows
  constexpr int STAGE = ROWS * 128;
  constexpr int PIECES = ROWS / 8;                       // 1 KiB DMA pieces per K-step
  constexpr int PPW = (PIECES + PC_PW - 1) / PC_PW;      // per producer wave (padded: a wave's spare slot repeats its last piece)
  static_assert(NS >= 3 && NS * STAGE <= 160 * 1024, "LDS budget");
  static_assert(PC_AHEAD >= 1 && PC_AHEAD + 1 <= PC_FM && PC_KB_AT + PC_AHEAD < PC_FM, "ring shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + PC_BM - 1) / PC_BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int pid_m = tile / tiles_n, pid_n = tile - pid_m * tiles_n;  // n fastest: the x panel is shared by neighbouring tiles
  const int m0 = pid_m * PC_BM, n0 = pid_n * BN;
  const int nk = p.K / 64;

  if (wave >= PC_CW) {
    // ======================================= producer =======================================
    const int pw = wave - PC_CW;
    const char* base[PPW];
    uint32_t dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int q = i * PC_PW + pw;
      q = q < PIECES ? q : PIECES - 1;
      const int row = q * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
      if (q < PC_BM / 8) {
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        base[i] = (const char*)p.A + ((long)m * p.lda * 2 + c * 16);
      } else {
        int n = n0 + row - PC_BM;
        n = n < p.N ? n : p.N - 1;
        base[i] = (const char*)p.B + ((long)n * p.ldb * 2 + c * 16);
      }
      dst[i] = (uint32_t)q * 1024u;
    }
    auto issue = [&](int kt) {
      char* stage = smem + (kt % NS) * STAGE;
#ifdef HVR_DBG_PC_NODMA
      if (kt >= NS) return;   // tuning build: only the first ring fill is loaded
#endif
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base[i] + (long)kt * 128),
                                         (__attribute__((address_space(3))) void*)(stage + dst[i]), 16, 0, 0);
    };
#ifdef HVR_DBG_PC_FREERUN
    // tuning build: the producers stream the whole K loop into the ring on their own counted waits, no barriers, nobody reads
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt);
      pc_wait_vm<HVR_DBG_PC_FREERUN * PPW>();
    }
    pc_wait_vm<0>();
    return;
#endif
    // steps 0 .. NS - 3 ahead of B_0; after B_j: step j + NS - 2 (the slot of step j - 2, consumed before B_j)
#pragma unroll
    for (int s = 0; s < NS - 2; ++s)
      if (s < nk) issue(s);
    if constexpr (EPI == EPI_APPLY) {
      // Block weights of this tile's 144 rows, g[t][row] = 2^(m_t - m*) / L with (m*, L) the per-row combine of the score
      // pass's block statistics (L = sum_t l_t 2^(m_t - m*)): one row per producer lane, computed while the first K-steps are
      // in flight and left in the LDS behind the ring -- the compute waves read 9 floats per block instead of combining
      // 9 rows x ntile / 4 blocks each in front of their first MFMA and exponentiating inside the K loop.
      float* g_lds = reinterpret_cast<float*>(smem + NS * STAGE);
      const int ptid = tid - PC_CW * 64;
      if (ptid < PC_BM) {
        const int m = m0 + ptid;
        const long row = (long)(m < p.M ? m : p.M - 1) * p.ntile;
        float mx = -INFINITY, l = 0.f;
        constexpr int TB = 12;   // loads go out in batches: one memory round trip per 12 tiles, not per tile
        for (int t0 = 0; t0 < p.ntile; t0 += TB) {
          float mt[TB], lt[TB];
#pragma unroll
          for (int u = 0; u < TB; ++u) {
            const bool ok = t0 + u < p.ntile;
            const int tt = ok ? t0 + u : p.ntile - 1;
            const float a = p.mstat[row + tt], b = p.lstat[row + tt];
            mt[u] = ok ? a : -INFINITY;
            lt[u] = ok ? b : 0.f;
          }
#pragma unroll
          for (int u = 0; u < TB; ++u) {
            const float mn = fmaxf(mx, mt[u]);
            l = l * __builtin_amdgcn_exp2f(mx - mn) + lt[u] * __builtin_amdgcn_exp2f(mt[u] - mn);  // 0 * exp2(-inf) = 0
            mx = mn;
          }
        }
        const float gref = mx + __builtin_amdgcn_logf(l);
        for (int t0 = 0; t0 < p.ntile; t0 += TB) {
          float mt[TB];
#pragma unroll
          for (int u = 0; u < TB; ++u) mt[u] = p.mstat[row + (t0 + u < p.ntile ? t0 + u : p.ntile - 1)];
#pragma unroll
          for (int u = 0; u < TB; ++u)
            if (t0 + u < p.ntile) g_lds[(t0 + u) * PC_BM + ptid] = __builtin_amdgcn_exp2f(mt[u] - gref);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table is written before B_0 releases its readers
    }
    for (int j = 0; j < nk; ++j) {
      // step j must have landed; steps j + 1 .. j + NS - 3 may stay in flight
      if (j + NS - 3 < nk) pc_wait_vm<(NS - 3) * PPW>();
      else pc_wait_vm<0>();
      __builtin_amdgcn_s_barrier();  // B_j
      if (j + NS - 2 < nk) issue(j + NS - 2);
    }
    return;
  }

  // ========================================= consumer =========================================
#ifdef HVR_DBG_PC_FREERUN
  return;
#endif
  const int wn = wave;
  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = pc_lds_off(smem) + frag_row * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = pc_lds_off(smem) + PC_BM * 128 + (wn * FN * 16 + frag_row) * 128 + ((frag_grp ^ swz) * 16);

  f32x4 acc[PC_FM][FN];
#pragma unroll
  for (int i = 0; i < PC_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int GN = EPI == EPI_APPLY ? PC_FM : 1;
  f32x4 pacc[GN][EPI == EPI_APPLY ? FN : 1];
  float gcur[GN];   // weight of the block whose partial sits in pacc
  const uint32_t g_lane = pc_lds_off(smem) + NS * STAGE + frag_row * 4;
  // the 9 weights of block t (one per row fragment) -> gcur, requested in line with the fragment reads (in-order LDS)
  auto read_g = [&](int t) {
    const uint32_t a = g_lane + (uint32_t)t * (PC_BM * 4);
    static_for<GN>([&](auto I) { gcur[decltype(I)::value] = pc_read32<decltype(I)::value * 64>(a); });
  };
  uint4 fb[2][FN];      // weight fragments: [half][column fragment]
  uint4 fa[PC_RING];    // x-fragment ring: item i of a half lives in slot i

  // One half K-step = PC_FM items (x fragment i, FN MFMAs).  `soff`: byte offset of this K-step's stage; `soff_ahead`: of
  // the stage the read-ahead that crosses into the next half comes from (same step for kk = 0, next step for kk = 1).
  // MORE: items exist behind this half.
  auto read_a = [&](auto SLOT, auto I, auto KK, uint32_t soff) {
    constexpr int slot = decltype(SLOT)::value, i = decltype(I)::value, kk = decltype(KK)::value;
#ifdef HVR_DBG_PC_NOLDS
    (void)soff;
#else
    fa[slot] = pc_read128<i * 2048>((a_lane + soff) ^ (kk ? 64u : 0u));
#endif
  };
  auto read_b = [&](auto KK, uint32_t soff) {
    constexpr int kk = decltype(KK)::value;
#ifdef HVR_DBG_PC_NOLDS
    (void)soff;
#else
    static_for<FN>([&](auto J) { fb[kk][decltype(J)::value] = pc_read128<decltype(J)::value * 2048>((b_lane + soff) ^ (kk ? 64u : 0u)); });
#endif
  };

  // FOLD (apply pass, first half of a block): item i first adds the PREVIOUS block's partial, acc[i] += g * pacc[i], then
  // starts the new block's partial over it (zero-C MFMAs): the FMAs ride between the MFMAs instead of stopping the stream
  // GREAD (apply pass, last half of a block): request this block's weights after item 0 -- they are older than the next
  // half's first fragment, so they have landed when that half's item 0 starts folding
  auto half = [&](auto KK, auto FIRSTBLK, auto FOLD, auto GREAD, auto MORE, uint32_t soff, uint32_t soff_ahead, int gblk) {
    constexpr int kk = decltype(KK)::value;
    constexpr bool zero_c = decltype(FIRSTBLK)::value, more = decltype(MORE)::value, fold = decltype(FOLD)::value, gread = decltype(GREAD)::value;
    static_for<PC_FM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int slot = i;
      // read-ahead items behind item i that exist at wait time: i + 1 .. i + AHEAD - 1 (those past the end of the half
      // belong to the next half and exist only when `more`)
      constexpr int in_half = (i + PC_AHEAD - 1 < PC_FM) ? PC_AHEAD - 1 : PC_FM - 1 - i;
      constexpr int ahead_left = more ? PC_AHEAD - 1 : in_half;
      __builtin_amdgcn_sched_barrier(0);
      pc_wait_lgkm<pc_pending(i, ahead_left, FN, more, gread)>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (EPI == EPI_APPLY && fold) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(gcur[i], pacc[i][j][r], acc[i][j][r]);
      }
#ifndef HVR_DBG_PC_NOMMA
      static_for<FN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (EPI == EPI_APPLY) pc_mma<HT, zero_c>(fb[kk][j], fa[slot], pacc[i][j]);
        else pc_mma<HT, false>(fb[kk][j], fa[slot], acc[i][j]);
      });
#endif
      __builtin_amdgcn_sched_barrier(0);
      // request item i + AHEAD: same half -> this half's stage and kk; next half -> (kk ^ 1), stage `soff_ahead`
      constexpr int t = i + PC_AHEAD;
      constexpr int nslot = t % PC_RING;
      if constexpr (t < PC_FM) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t>{}, KK, soff);
      } else if constexpr (more) {
        read_a(std::integral_constant<int, nslot>{}, std::integral_constant<int, t - PC_FM>{}, std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      }
      if constexpr (i == 0 && gread) read_g(gblk);
      if constexpr (i == PC_KB_AT && more) read_b(std::integral_constant<int, kk ^ 1>{}, soff_ahead);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: B_0, then the first half's weight fragments and the first AHEAD x fragments ----
  __builtin_amdgcn_s_barrier();  // B_0: K-step 0 is in the LDS (and, for the apply pass, the rows' gref)
  read_b(std::integral_constant<int, 0>{}, 0u);
  static_for<PC_AHEAD>([&](auto T) { read_a(T, T, std::integral_constant<int, 0>{}, 0u); });

  constexpr std::true_type Y{};
  constexpr std::false_type N{};
  constexpr std::integral_constant<int, 0> K0{};
  constexpr std::integral_constant<int, 1> K1{};

  // one K-step; FIRST: first K-step of a 128-key block (EPI_APPLY: two K-steps per block), FOLD: a previous block exists;
  // TAIL: the last K-step
  auto kstep = [&](int k, auto FIRST, auto FOLD, auto TAIL) {
    constexpr bool first = decltype(FIRST)::value, tail = decltype(TAIL)::value;
    constexpr bool glast = EPI == EPI_APPLY && !first;   // second K-step of an apply block: its last half fetches the weights
    const uint32_t soff = (uint32_t)(k % NS) * STAGE, snext = (uint32_t)((k + 1) % NS) * STAGE;
    half(K0, std::integral_constant<bool, (EPI == EPI_APPLY) && first>{}, FOLD, N, Y, soff, soff, 0);
    if constexpr (!tail) {
      __builtin_amdgcn_s_barrier();  // B_{k+1}: K-step k + 1 is in the LDS
      half(K1, N, N, std::integral_constant<bool, glast>{}, Y, soff, snext, k >> 1);
    } else {
      half(K1, N, N, std::integral_constant<bool, glast>{}, N, soff, soff, k >> 1);
    }
  };
  if constexpr (EPI == EPI_APPLY) {  // nk is even: 128-key blocks of two K-steps
    kstep(0, Y, N, N);
    if (nk > 2) {
      kstep(1, N, N, N);
      int k = 2;
      for (; k + 2 < nk; k += 2) {
        kstep(k, Y, Y, N);
        kstep(k + 1, N, N, N);
      }
      kstep(k, Y, Y, N);
      kstep(k + 1, N, N, Y);
    } else {
      kstep(1, N, N, Y);
    }
    // the last block's partial (its weights were requested in the last half)
    pc_wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PC_FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(gcur[i], pacc[i][j][r], acc[i][j][r]);
  } else {
    int k = 0;
    for (; k + 1 < nk; ++k) kstep(k, N, N, N);
    kstep(k, N, N, Y);
  }

  // ---------------- epilogue (compute waves only: the producers have exited) ----------------
  // the f32 tile goes through the LDS so that residual loads and output stores are whole 16-byte row segments
  constexpr int LDW = BN + 4, CH = BN / 8, CT = PC_CW * 64;
  float* ebuf = reinterpret_cast<float*>(smem);
  __builtin_amdgcn_s_barrier();  // every compute wave is done reading the ring
  // the shifts of the wave's FN column fragments in one batch (inside the (i, j) loop each was a load + vmcnt(0): see gemm_tile.h)
  float4 bvj[FN];
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      int c = n0 + (wn * FN + j) * 16 + frag_grp * 4;
      c = c < p.N ? c : p.N - 4;
      bvj[j] = *reinterpret_cast<const float4*>(p.bias + c);
    }
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) bvj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < PC_FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = (wn * FN + j) * 16 + frag_grp * 4;
      f32x4 v = acc[i][j];
      v[0] += bvj[j].x; v[1] += bvj[j].y; v[2] += bvj[j].z; v[3] += bvj[j].w;
      *reinterpret_cast<f32x4*>(ebuf + (i * 16 + frag_row) * LDW + col) = v;
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < (PC_BM * CH + CT - 1) / CT; ++it) {
    const int c = it * CT + tid;
    if (c >= PC_BM * CH) continue;
    const int r = c / CH, cc = c - r * CH;
    const int m = m0 + r, n = n0 + cc * 8;
    if (m >= p.M || n >= p.N) continue;
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
    if (p.out_f32) {
      float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
      store4(cp, v);
      store4(cp + 4, v + 4);
    } else {
      HT* cp = reinterpret_cast<HT*>(p.C) + (long)m * p.ldc + n;
      *reinterpret_cast<uint4*>(cp) = make_uint4(pack2<HT>(v[0], v[1]), pack2<HT>(v[2], v[3]), pack2<HT>(v[4], v[5]), pack2<HT>(v[6], v[7]));
    }
  }
}

// ---------------- host side ----------------
bool pc_supported(const GemmParams& p, int epi) {
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16) || !p.staging || p.conv || p.ksplit_steps > 0) return false;
  if (epi != EPI_LINEAR && epi != EPI_APPLY) return false;
  if (p.K % 64 || p.K < 128 || p.N % 8 || p.M < 1) return false;
  if (epi == EPI_APPLY && (p.K % 128 || p.K / 128 != p.ntile)) return false;
  // the block-weight table [ntile][144] f32 must fit behind the 4-slot ring of 144 + 128 rows
  if (epi == EPI_APPLY && (size_t)p.ntile * PC_BM * 4 > (size_t)160 * 1024 - 4 * (PC_BM + 128) * 128) return false;
  if (p.lda % 8 || p.ldb % 8 || (p.ldc * (p.out_f32 ? 4 : 2)) % 16 || (p.resid && p.ldr % 8)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.resid) | reinterpret_cast<uintptr_t>(p.bias);
  if (al & 15) return false;
  if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  return true;
}

template <typename T, int FN, int EPI, int NS>
static hipError_t launch_pc(const GemmParams& p, hipStream_t stream) {
  constexpr int BN = PC_CW * FN * 16;
  // the apply pass keeps its block-weight table [ntile][144] f32 behind the ring: everything the LDS has left
  constexpr size_t ring = EPI == EPI_APPLY ? (size_t)160 * 1024 : (size_t)NS * (PC_BM + BN) * 128, epi = (size_t)PC_BM * (BN + 4) * 4;
  constexpr size_t lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = pc_tile_kernel<T, FN, EPI, NS>;
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int tiles = ((p.M + PC_BM - 1) / PC_BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PC_NT), lds, stream, p);
  return hipGetLastError();
}

// bn: 128 (FN = 2, 4-slot ring) or 256 (FN = 4, 3-slot ring)
hipError_t run_pc(const GemmParams& p, int epi, int bn, hipStream_t stream) {
  if (p.dtype == DT_F16) {
    if (epi == EPI_APPLY) return launch_pc<f16_t, 2, EPI_APPLY, 4>(p, stream);
    if (bn == 256) return launch_pc<f16_t, 4, EPI_LINEAR, 3>(p, stream);
    return launch_pc<f16_t, 2, EPI_LINEAR, 4>(p, stream);
  }
  if (epi == EPI_APPLY) return launch_pc<bf16_t, 2, EPI_APPLY, 4>(p, stream);
  if (bn == 256) return launch_pc<bf16_t, 4, EPI_LINEAR, 3>(p, stream);
  return launch_pc<bf16_t, 2, EPI_LINEAR, 4>(p, stream);
}

}  // namespace hvr
