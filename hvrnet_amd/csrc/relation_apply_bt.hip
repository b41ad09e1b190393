// Relation apply pass, big-tile grouped form (bf16, gfx950):   O[g] = softmax-weighted sum of the values, from the scores pass's
// P~ / block statistics (relation_bt.hip with int_max = 1), for G independent problems of one shape in ONE launch.
//
// Replaces `torch.mm(aff_softmax, v_data)` of one relation stage (mmdet/models/bbox_heads/selsa_bbox_head.py:178-182,
// hrnmp_bbox_head.py:332-346) for the window-sized problem (Mq = Mk = 4 500, D = 1 024), G windows at a time.
//
// Why a second apply kernel next to pc_gemm.hip / the tile engine's EPI_APPLY.  The product P~ [4 500 x 4 608] . V^T [1 024 x 4 608]^T of
// ONE window is 16 x 4 tiles of 288 x 256 -- a quarter of the chip -- which is why the one-window pass runs 256 tiles of 144 x 128 with
// 0.5-0.6 LDS fragment reads per MFMA and two accumulator sets (the un-scaled partial of the current 128-key block + the running total,
// folded by one FMA per accumulator register per block) at 0.32 of the MFMA peak.  With the FOUR windows a caller has in flight the
// same product is 256 tiles of 288 x 256 over the whole key axis: bigtile.hip's shape and phase-staggered K loop (8 waves as 2 x 4 of
// 144 x 64, 0.36 fragment reads per MFMA, 72 K-steps: prologue and epilogue are 3 % of the launch), which that file's res5 3x3 conv --
// the same K = 4 608 -- runs at 0.53 of peak.  What does not fit that shape is the second accumulator set (2 x 144 registers), so the
// block weights go where they cost one VALU operation per MFMA and no register: the scores pass rounds its block maxima UP to integers
// (log2 units), the weight of block t relative to the row's largest block is then the exact power of two 2^-(m* - m_t), and this
// kernel lowers the EXPONENT FIELDS of the P~ fragments by m* - m_t on their way from the LDS to the MFMA (v_pk_sub_u16 with clamp:
// a bf16 is sign | 8 exponent bits | 7 mantissa bits, P~ >= 0, and a value scaled below 2^-126 saturates to zero).  One accumulator
// set, a plain product, rows x 1 / L in the epilogue, L = sum_t 2^-(m* - m_t) l_t built in the prologue from the block sums.  The
// scaling is exact, so the result differs from the folded form's only by the association of the f32 sum (one running sum over all
// keys here; per-block partials there).  A half's 5-bit exponent cannot carry the block weights as a field shift: IEEE half operands
// (HVR_F16) take the multiplied weights described next for split half, on their one plane.
//
// Split half (HT = f16s_t, round 5): P~ x 2^12 and V^T arrive as [32 hi | 32 lo] groups (common.h), a K-step is 32 keys = one 128-byte
// line per row holding both planes, and a fragment pair takes three MFMAs -- V_hi P_hi, V_hi P_lo, V_lo P_hi, six phases as in
// relation_bt.hip.  The block weight 2^-s is applied by v_pk_mul_f16 on BOTH planes of the P~ fragments: exact while the product is a
// normal half; a product below 2^-14 is rounded as a subnormal (absolute error <= 2^-25 against a row whose largest block holds a
// 2^11..2^12, i.e. <= 2^-36 relative per term) and 2^-s itself is zero from s = 25 on -- such a block's terms are below 2^-13 / 2^12 of
// the row's largest.  This replaces the normalising sweep over P~ (24 us per window) + the plain product on 144 x 128 tiles (111 us).
#include <type_traits>
#include "common.h"
#include "relation_bt.h"

namespace hvr {

namespace {

constexpr int AB_BM = 288, AB_BN = 256, AB_NT = 512, AB_WN = 4, AB_FN = 4, AB_FM = 9, AB_WCOLS = AB_FN * 16;
constexpr int AB_A_BYTES = AB_BM * 128, AB_STAGE = AB_A_BYTES + AB_BN * 128;     // 69 632 B per stage
constexpr int AB_A_SLOTS = (AB_BM * 8 + AB_NT - 1) / AB_NT;                       // 5 (the last one: waves 0..3)
constexpr int AB_LAST_WAVES = (AB_BM * 8 - (AB_A_SLOTS - 1) * AB_NT) / 64;
constexpr int AB_B_SLOTS = AB_BN * 8 / AB_NT;                                     // 4
constexpr int AB_MAXBLK = 72;                                                     // 128-key blocks per row the shift table holds (Mk <= 9 216)
constexpr int AB_TAB = 2 * AB_STAGE, AB_RINV = AB_TAB + AB_BM * AB_MAXBLK, AB_LDS = AB_RINV + AB_BM * 4;
static_assert(AB_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ uint32_t ab_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 ab_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ uint32_t ab_read8(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// both bf16 halves of a word: exponent field minus s (sub = s * 0x00800080), saturating at zero
__device__ __forceinline__ uint32_t ab_lower(uint32_t w, uint32_t sub) {
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(w), "v"(sub));
  return d;
}
// both IEEE halves of a word times the half pair in `w2` (a power of two in both lanes)
__device__ __forceinline__ uint32_t ab_scale_h(uint32_t w, uint32_t w2) {
  uint32_t d;
  asm("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(w), "v"(w2));
  return d;
}
__device__ __forceinline__ void ab_load_lds16(const void* base, char* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
#else
  (void)base; (void)lds; (void)voff; (void)soff;
#endif
}

}  // namespace

template <typename HT>   // bf16_t, f16_t (IEEE half: the split-half instance's multiplied weights on its one plane) or f16s_t (split half)
__global__ __launch_bounds__(AB_NT) void relation_apply_bt_kernel(const ApplyBTParams p) {
  constexpr bool SPLIT = std::is_same<HT, f16s_t>::value;
  constexpr bool HALFW = !std::is_same<HT, bf16_t>::value;   // block weights as a half multiplied on (v_pk_mul_f16), not as an exponent-field shift
  constexpr int EB = SPLIT ? 4 : 2;         // bytes per logical element in memory
  constexpr int BKE = SPLIT ? 32 : 64;      // keys per K-step (one 128-byte line per row)
  constexpr int BLK_SHIFT = SPLIT ? 2 : 1;  // K-steps per 128-key block, log2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / AB_WN, wn = wave % AB_WN;
  const int wrow0 = wm * AB_FM * 16;
  const int tiles_n = p.D / AB_BN, tiles_m = (p.Mq + AB_BM - 1) / AB_BM, tpg = tiles_m * tiles_n;
  // n fastest inside an XCD's contiguous range: the four column tiles of a 288-row panel share its P~ rows in one L2
  const int tile = xcd_remap(blockIdx.x, (int)gridDim.x);
  const int g = tile / tpg, tr = tile - g * tpg;
  const int pid_m = tr / tiles_n, pid_n = tr - pid_m * tiles_n;
  const int m0 = pid_m * AB_BM, n0 = pid_n * AB_BN;
  const char* const rs_a = (const char*)p.P + (long)g * p.gs_p * EB;
  const char* const rs_b = (const char*)p.Vt + (long)g * p.gs_vt * EB;

#ifdef HVR_DBG_AB_CLK
  long long dbg_t[5];
  dbg_t[0] = wall_clock64();
#endif
  // ---- loader (bigtile.hip's, plain product): a thread's pieces sit 64 rows apart, all in the same swizzled 16-byte chunk ----
  const int l_row = tid >> 3, l_chunk = ((tid & 7) ^ (l_row & 7)) * 16;
  int a_off[AB_A_SLOTS], b_off[AB_B_SLOTS];
#pragma unroll
  for (int i = 0; i < AB_A_SLOTS; ++i) {
    int m = m0 + i * 64 + l_row;
    m = m < p.Mq ? m : p.Mq - 1;
    a_off[i] = (int)((long)m * p.ldp * EB) + l_chunk;
  }
#pragma unroll
  for (int i = 0; i < AB_B_SLOTS; ++i) b_off[i] = (int)((long)(n0 + i * 64 + l_row) * p.ldp * EB) + l_chunk;
  auto dma_a = [&](auto I, char* stage, int koff) {
    constexpr int i = decltype(I)::value;
    if (i < AB_A_SLOTS - 1 || wave < AB_LAST_WAVES) ab_load_lds16(rs_a, stage + (i * AB_NT + wave * 64) * 16, (unsigned)a_off[i], koff);
  };
  auto dma_b = [&](auto I, char* stage, int koff) {
    constexpr int i = decltype(I)::value;
    ab_load_lds16(rs_b, stage + AB_A_BYTES + (i * AB_NT + wave * 64) * 16, (unsigned)b_off[i], koff);
  };
  // first K-step into stage 0
  static_for<AB_A_SLOTS>([&](auto I) { dma_a(I, smem, 0); });
  static_for<AB_B_SLOTS>([&](auto I) { dma_b(I, smem, 0); });

  // ---- prologue, under the first K-step's flight: per row of the tile the exponent shifts m* - m_t of its blocks (one byte each,
  // LDS table [288][72]) and 1 / L ----
  {
    unsigned char* tab = reinterpret_cast<unsigned char*>(smem + AB_TAB);
    float* rinv = reinterpret_cast<float*>(smem + AB_RINV);
    if (tid < AB_BM) {
      int m = m0 + tid;
      m = m < p.Mq ? m : p.Mq - 1;
      const float* ms = p.mstat + (long)g * p.gs_stat + (long)m * p.ntile;
      const float* ls = p.lstat + (long)g * p.gs_stat + (long)m * p.ntile;
      if ((p.ntile & 3) == 0) {
        // the row's statistics as 16-byte loads, ALL requested before the first use (clamped indices, nothing predicated): two
        // memory round trips for the whole prologue (a loop of scalar loads over the blocks took 11 us of a 160 us launch)
        constexpr int NC = AB_MAXBLK / 4;
        const int nt4 = p.ntile >> 2;
        float4 mv[NC], lv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) mv[c] = reinterpret_cast<const float4*>(ms)[c < nt4 ? c : nt4 - 1];
#pragma unroll
        for (int c = 0; c < NC; ++c) lv[c] = reinterpret_cast<const float4*>(ls)[c < nt4 ? c : nt4 - 1];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) mx = fmaxf(fmaxf(mx, fmaxf(mv[c].x, mv[c].y)), fmaxf(mv[c].z, mv[c].w));   // (clamped copies repeat a real chunk)
        float L = 0.f;
        uint32_t* tab32 = reinterpret_cast<uint32_t*>(tab + tid * AB_MAXBLK);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float mm[4] = {mv[c].x, mv[c].y, mv[c].z, mv[c].w}, ll[4] = {lv[c].x, lv[c].y, lv[c].z, lv[c].w};
          uint32_t w = 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float d = mx - mm[e];               // a non-negative integer (both are integer-valued: relation_bt.hip, int_max)
            d = d < 255.f ? d : 255.f;
            // table byte: bf16 -- the shift; half / split half -- 25 - min(shift, 25), from which the K loop forms the half 2^-shift
            w |= (uint32_t)(int)(HALFW ? 25.f - fminf(d, 25.f) : d) << (8 * e);
            L += c < nt4 ? __builtin_amdgcn_exp2f(-d) * ll[e] : 0.f;
          }
          if (c < nt4) tab32[c] = w;
        }
        rinv[tid] = 1.f / L;
      } else {
        float mx = -INFINITY;
        for (int t = 0; t < p.ntile; ++t) mx = fmaxf(mx, ms[t]);
        float L = 0.f;
        for (int t = 0; t < p.ntile; ++t) {
          float d = mx - ms[t];
          d = d < 255.f ? d : 255.f;
          tab[tid * AB_MAXBLK + t] = (unsigned char)(int)(HALFW ? 25.f - fminf(d, 25.f) : d);
          L += __builtin_amdgcn_exp2f(-d) * ls[t];
        }
        rinv[tid] = 1.f / L;
      }
    }
  }

#ifdef HVR_DBG_AB_CLK
  dbg_t[1] = wall_clock64();
#endif
  f32x4 acc[AB_FM][AB_FN];
#pragma unroll
  for (int i = 0; i < AB_FM; ++i)
#pragma unroll
    for (int j = 0; j < AB_FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = ab_lds_off(smem) + (wrow0 + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = ab_lds_off(smem) + AB_A_BYTES + (wn * AB_WCOLS + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t t_lane = ab_lds_off(smem) + AB_TAB + (wrow0 + frag_row) * AB_MAXBLK;   // + block index; fragment i: offset i * 16 * 72

  const int nk = (int)(p.ldp / BKE);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef HVR_DBG_AB_CLK
  dbg_t[2] = wall_clock64();
#endif

  // ---- phase-staggered K loop (relation_bt.hip / bigtile.hip): a K-step is four phases -- (K half h, row fragments 0..4) and
  // (h, 5..8) for h = 0, 1 -- each an L section (fragment reads + this wave's share of the next K-step's DMA) and a C section (the
  // phase's exponent adjustments and MFMAs) with an s_barrier behind each; wave group 1 runs one barrier behind group 0 ----
  {
    constexpr int FM = AB_FM, G0 = (FM + 1) / 2;
    constexpr int DMA_TOTAL = AB_A_SLOTS + AB_B_SLOTS, DMA_FIRST = DMA_TOTAL / 2;
    auto kloop = [&](auto WMC) __attribute__((always_inline)) {
    constexpr int wmc = decltype(WMC)::value;
    constexpr int dma_ph = wmc ? 0 : 1;
    if constexpr (wmc != 0) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t soff = (uint32_t)(kt & 1) * AB_STAGE;
      char* nxt = smem + ((kt + 1) & 1) * AB_STAGE;
      const int koff = (kt + 1 < nk ? kt + 1 : kt) * 128;   // (the last step re-fetches itself into the idle stage: one uniform stream)
      const uint32_t a0 = a_lane + soff, b0 = b_lane + soff;
      const uint32_t tb = t_lane + (uint32_t)(kt >> BLK_SHIFT);       // this K-step's 128-key block
      uint4 kb[AB_FN], qa[G0];
      uint32_t sh[FM];
      // phases: (64-byte half of the V^T lines, of the P~ lines, row fragments 0..4 / 5..8); two-byte formats: the halves are the K
      // halves, (0,0) (0,0) (1,1) (1,1); split half: the halves are the hi / lo planes, V_hi P_hi, V_hi P_lo, V_lo P_hi in two phases each
      constexpr int NPH = SPLIT ? 6 : 4;
      static_for<NPH>([&](auto PH) {
        constexpr int ph = decltype(PH)::value, r0 = (ph & 1) ? G0 : 0, nr = (ph & 1) ? FM - G0 : G0;
        constexpr int hv = SPLIT ? (ph >= 4 ? 1 : 0) : (ph >> 1);
        constexpr int hp = SPLIT ? ((ph == 2 || ph == 3) ? 1 : 0) : (ph >> 1);
        constexpr bool read_v = SPLIT ? (ph == 0 || ph == 4) : ((ph & 1) == 0);
        // ---- L ----
        if constexpr (ph == 0) static_for<FM>([&](auto R) { sh[decltype(R)::value] = ab_read8<decltype(R)::value * 16 * AB_MAXBLK>(tb); });
        if constexpr (read_v)
          static_for<AB_FN>([&](auto J) { kb[decltype(J)::value] = ab_read128<decltype(J)::value * 2048>(hv ? (b0 ^ 64u) : b0); });
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          qa[r] = ab_read128<(r0 + r) * 2048>(hp ? (a0 ^ 64u) : a0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph < NPH - 1) {
          if constexpr (dma_ph == ph) {
            static_for<DMA_FIRST>([&](auto D) {
              constexpr int d = decltype(D)::value;
              if constexpr (d < AB_A_SLOTS) dma_a(std::integral_constant<int, d>{}, nxt, koff);
              else dma_b(std::integral_constant<int, d - AB_A_SLOTS>{}, nxt, koff);
            });
          } else if constexpr (dma_ph + 1 == ph) {
            static_for<DMA_TOTAL - DMA_FIRST>([&](auto D) {
              constexpr int d = DMA_FIRST + decltype(D)::value;
              if constexpr (d < AB_A_SLOTS) dma_a(std::integral_constant<int, d>{}, nxt, koff);
              else dma_b(std::integral_constant<int, d - AB_A_SLOTS>{}, nxt, koff);
            });
          }
        }
        if constexpr (ph == NPH - 1) {
          if constexpr (wmc != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- C ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph == 0) static_for<FM>([&](auto R) {
          constexpr int r = decltype(R)::value;
          if constexpr (HALFW) {
            // byte e = 25 - min(shift, 25) -> the half 2^-shift in both lanes: a normal number (exponent field e - 10) from e = 11 up,
            // the subnormal 1 << (e - 1) below, zero for e = 0
            const uint32_t e = sh[r];
            const uint32_t hbits = e >= 11u ? (e - 10u) << 10 : ((1u << e) >> 1);
            sh[r] = hbits * 0x00010001u;
          } else {
            sh[r] *= 0x00800080u;   // byte -> both exponent fields
          }
        });
        __builtin_amdgcn_s_setprio(1);
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          const uint32_t s = sh[r0 + r];
#ifdef HVR_DBG_AB_NOADJ   // timing-only build: what the exponent adjustment costs
          const uint4 x = qa[r];
          if (s == 0x12345u) acc[r0 + r][0][0] += 1.f;
#else
          uint4 x;
          if constexpr (HALFW) x = make_uint4(ab_scale_h(qa[r].x, s), ab_scale_h(qa[r].y, s), ab_scale_h(qa[r].z, s), ab_scale_h(qa[r].w, s));
          else x = make_uint4(ab_lower(qa[r].x, s), ab_lower(qa[r].y, s), ab_lower(qa[r].z, s), ab_lower(qa[r].w, s));
#endif
          static_for<AB_FN>([&](auto J) {
            constexpr int j = decltype(J)::value;
            // V^T rows as the MFMA "A" operand: a lane ends up with 4 consecutive output columns of one query row (gemm.hip)
            acc[r0 + r][j] = mfma_half<typename std::conditional<HALFW, f16_t, bf16_t>::type>(kb[j], x, acc[r0 + r][j]);
          });
        });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph == NPH - 1) {
          if constexpr (wmc == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      });
    }
    if constexpr (wmc == 0) __builtin_amdgcn_s_barrier();
    };
    if (wm) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  }

#ifdef HVR_DBG_AB_CLK
  dbg_t[3] = wall_clock64();
#endif
  // ---------------- epilogue: rows x 1 / L, bf16, whole 128-byte row segments out through per-wave LDS staging (bigtile.hip) ----------------
  // lane holds O[m0 + wrow0 + 16 i + frag_row][n0 + wn 64 + 16 j + 4 frag_grp + r] = acc[i][j][r]
  {
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));  // lane-derived values re-derived here: nothing but the accumulators lives across the loop
    const int el = etid & 63, erow = el & 15, egrp = el >> 4;
    constexpr int SPITCH = SPLIT ? AB_WCOLS * 4 + 32 : AB_WCOLS * 2;   // (split half: 256-byte row segments, relation_bt.hip's staging)
    char* stg = smem + wave * (16 * SPITCH);
    const float* rinv = reinterpret_cast<const float*>(smem + AB_RINV);
    char* const Og = (char*)p.O + (long)g * p.gs_o * EB;
    const int wr_lane = erow * SPITCH + (((egrp & 1) ^ (erow >> 3)) << 3);
    const int st_row = el >> 3, st_chunk = el & 7;
    float rs[AB_FM];
#pragma unroll
    for (int i = 0; i < AB_FM; ++i) rs[i] = rinv[wrow0 + i * 16 + erow] * (SPLIT ? 1.f / kSplitProbScale : 1.f);   // (P~ is stored x 2^12)
    __syncthreads();  // every wave is done reading the ring (group 0 leaves the loop a barrier ahead of group 1's last reads)
#pragma unroll
    for (int i = 0; i < AB_FM; ++i) {
#pragma unroll
      for (int j = 0; j < AB_FN; ++j) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = acc[i][j][r] * rs[i];
        if constexpr (SPLIT) {
          uint32_t h0, l0, h1, l1;
          split2(e[0], e[1], h0, l0);
          split2(e[2], e[3], h1, l1);
          char* dst = stg + erow * SPITCH + (int)split_col_bytes(j * 16 + egrp * 4);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + kSplitPlane) = make_uint2(l0, l1);
        } else {
          char* slot = stg + wr_lane + (((2 * j + (egrp >> 1)) ^ (erow & 7)) << 4);
          *reinterpret_cast<uint2*>(slot) = make_uint2(pack2<HT>(e[0], e[1]), pack2<HT>(e[2], e[3]));
        }
      }
      if constexpr (SPLIT) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int row = h * 4 + (el >> 4), m = m0 + wrow0 + i * 16 + row;
          const uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + ((el & 15) << 4));
          if (m < p.Mq) *reinterpret_cast<uint4*>(Og + ((long)m * p.ldo + n0 + wn * AB_WCOLS) * 4 + ((el & 15) << 4)) = v;
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = h * 8 + st_row, m = m0 + wrow0 + i * 16 + row;
          uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + ((st_chunk ^ st_row) << 4));
          if (h) v = make_uint4(v.z, v.w, v.x, v.y);
          if (m < p.Mq) *reinterpret_cast<uint4*>(Og + ((long)m * p.ldo + n0 + wn * AB_WCOLS + st_chunk * 8) * 2) = v;
        }
      }
    }
  }
#ifdef HVR_DBG_AB_CLK
  dbg_t[4] = wall_clock64();
  if (threadIdx.x == 0 && (blockIdx.x & 7) == 0)
    printf("ABCLK wg %d t0 %lld prologue %lld first-dma %lld loop %lld epilogue %lld\n", (int)blockIdx.x, dbg_t[0], dbg_t[1] - dbg_t[0], dbg_t[2] - dbg_t[1],
           dbg_t[3] - dbg_t[2], dbg_t[4] - dbg_t[3]);
#endif
}

bool apply_bt_supported(int Mq, int Mk, int D, long ldp, long ldo, const void* P, const void* Vt, const void* O, int groups, bool split) {
  const uintptr_t al = reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Vt) | reinterpret_cast<uintptr_t>(O);
  const int es = split ? 4 : 2;
  if ((al & (split ? 127 : 15)) || groups < 1 || D % AB_BN || ldp % 128 || ldo % (split ? 32 : 8)) return false;
  if (ldp / 128 > AB_MAXBLK || (long)Mq * ldp * es >= (1L << 31) || (long)D * ldp * es >= (1L << 31)) return false;
  // pays once the 288 x 256 grid over all groups covers most of the chip: four windows of 4 500 rows are 256 tiles, three 192
  const long tiles = (long)groups * ((Mq + AB_BM - 1) / AB_BM) * (D / AB_BN);
  return Mk >= 128 && tiles >= 176 && tiles <= 65535;
}

hipError_t run_apply_bt(const ApplyBTParams& p, hipStream_t stream) {
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_apply_bt_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_apply_bt_kernel<f16s_t>), hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_apply_bt_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
  });
  const int tiles = p.groups * ((p.Mq + AB_BM - 1) / AB_BM) * (p.D / AB_BN);
  if (p.split == 2) hipLaunchKernelGGL(relation_apply_bt_kernel<f16_t>, dim3(tiles), dim3(AB_NT), AB_LDS, stream, p);
  else if (p.split) hipLaunchKernelGGL(relation_apply_bt_kernel<f16s_t>, dim3(tiles), dim3(AB_NT), AB_LDS, stream, p);
  else hipLaunchKernelGGL(relation_apply_bt_kernel<bf16_t>, dim3(tiles), dim3(AB_NT), AB_LDS, stream, p);
  return hipGetLastError();
}

}  // namespace hvr
