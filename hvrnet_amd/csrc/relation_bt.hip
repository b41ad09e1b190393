// Relation scores pass, big-tile form (bf16, gfx950):  P~ = exp2(scale*log2e * Q K^T - blockmax), per (row, 128-key
// block) max / sum, and -- riding in the same launch -- the V^T copy the apply pass reads.
//
// Replaces the bmm / scale / softmax head of one relation stage (mmdet/models/bbox_heads/selsa_bbox_head.py:166-176,
// hrnmp_bbox_head.py:293-330) for the window-sized problem (Mq = Mk = 4 500, D = 1 024).
//
// Why a second scores kernel next to the tile engine (gemm.hip): with 128 x 128 tiles the 4 500 x 4 500 score matrix is
// 1 296 tiles = 2.53 rounds of the chip's 512 resident workgroups, every round ends in an un-overlapped softmax
// epilogue, and the small wave tiles (64 x 64) read 0.5 LDS fragments per MFMA -- measured on this chip, every
// ds_read_b128 returning into a SIMD's registers costs about as much matrix-pipe time as an MFMA, so the loop runs at
// half rate.  Here ONE workgroup per CU owns a 352 x 256 tile: 13 x 18 = 234 tiles = one round on 256 CUs, 148 flop per
// operand byte, one prologue and one epilogue per CU.
//   * 8 waves as 2 x 4: every wave owns 176 rows x 64 keys (11 x 4 fragments, 176 accumulator registers, 0.34 LDS
//     fragment reads per MFMA), two waves on every SIMD;
//   * K-step = 128 bytes of D per row, two LDS stages of (352 + 256) x 128 B = 76 KB filled by global_load_lds (same
//     XOR-swizzled image as gemm.hip); the next K-step's DMA is issued piecewise between this step's MFMAs;
//   * the 11 query fragments of a half K-step stream through a small register ring: fragment t + AHEAD is requested
//     when fragment t is consumed, waits are counted (LDS returns in order); the other half's key fragments are
//     requested in the shadow of the first items;
//   * epilogue: row max over the wave's 64 columns by shuffles, over the 128-key block (2 waves) through LDS, exp2 with
//     the scale folded into one FMA, bf16 pack, row segments staged per wave through LDS so that they leave as whole
//     128-byte lines;
//   * every workgroup also transposes its share of V into V^T[D][ldp] (64 x 64 tiles through LDS) while its first
//     K-step is in flight, which removes the separate transpose launch in front of the apply pass;
//   * the workgroups are PERSISTENT (round 5): a launch takes the score tiles of G independent problems of one shape -- the
//     windows a caller has in flight -- as one list; from its second tile on a workgroup has its first K-step fetched under the
//     previous tile's last K-step, and that tile's P~ stores drain under the new loop instead of in front of a kernel boundary.
#include "common.h"
#include "relation_bt.h"

namespace hvr {

namespace {

constexpr int BT_BM = 352, BT_BN = 256, BT_NT = 512;
constexpr int BT_WN = 4, BT_FM = 11, BT_FN = 4;               // 2 x 4 waves of 176 x 64
constexpr int BT_WROWS = BT_FM * 16, BT_WCOLS = BT_FN * 16;
constexpr int BT_A_BYTES = BT_BM * 128, BT_B_BYTES = BT_BN * 128, BT_STAGE = BT_A_BYTES + BT_B_BYTES;
constexpr int BT_A_SLOTS = (BT_BM * 8 + BT_NT - 1) / BT_NT;  // 6 (the last one: waves 0..3 only)
constexpr int BT_B_SLOTS = BT_BN * 8 / BT_NT;                // 4
constexpr int BT_LDS = 2 * BT_STAGE;                         // 155 648 B
static_assert(BT_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 lds_read128(uint32_t addr) {
  uint4 v;
#ifdef HVR_DBG_BT_NOREADS
  asm volatile("" : "=v"(v) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#endif
  return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> __device__ __forceinline__ void mma(const uint4& keys, const uint4& queries, f32x4& acc) {
  // keys as the MFMA "A" operand: lane ends up with 4 consecutive keys of one query row (see gemm.hip)
#ifdef HVR_DBG_BT_NOMMA
  acc[0] += __uint_as_float(keys.x ^ queries.x);
#else
  acc = mfma_half<typename std::conditional<std::is_same<T, f16s_t>::value, f16_t, T>::type>(keys, queries, acc);
#endif
}

}  // namespace

// HT = bf16_t / f16_t: operands move as raw 16-bit words; the MFMA opcode and the P~ pack differ.
// HT = f16s_t (split half, common.h): 4 bytes per logical element, a row's 128-byte line of a K-step holds the hi plane of 32 logical
// elements in its first 64 bytes and their lo plane in the second -- the SAME LDS image and loader as the two-byte formats (whose second
// 64 bytes are the K-step's second 32 elements); a K-step is D / 32 long and issues three MFMAs per fragment pair, K_hi x Q_hi,
// K_hi x Q_lo and K_lo x Q_hi, in six phases instead of four; P~ leaves x 2^12 in the split layout (kSplitProbScale, as the tile
// engine's EPI_SCORES writes it); the V^T copies move the two planes of a tile as two 16-bit transposes.
template <typename HT>
__global__ __launch_bounds__(BT_NT) void relation_scores_bt_kernel(const ScoresBTParams p) {
  constexpr bool SPLIT = std::is_same<HT, f16s_t>::value;
  constexpr int EB = SPLIT ? 4 : 2;          // bytes per logical element in memory
  constexpr int BKE = SPLIT ? 32 : 64;       // logical elements per K-step (one 128-byte line per row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / BT_WN, wn = wave % BT_WN;
  const int tiles_n = (int)((p.ldp + BT_BN - 1) / BT_BN);
  const int tiles_m = (p.Mq + BT_BM - 1) / BT_BM;
  const int tpg = tiles_m * tiles_n;          // tiles per group
  const int total = tpg * p.groups;
  const int nwg = (int)gridDim.x;
  // PERSISTENT workgroups: the tiles of all groups form one list, walked in rounds of `nwg` (256: one workgroup per CU); round r
  // gives this workgroup tile r * nwg + xcd_remap(blockIdx.x, tiles of the round) -- n fastest inside an XCD's contiguous range, so
  // that neighbouring tiles share their 352-row query panel in one L2.  One group of the 4 500-row window is 234 tiles = one round
  // (every workgroup one tile, the launch of rounds 1-4); G windows in flight are G x 234 tiles, and from the second tile on a
  // workgroup's first K-step is fetched under its previous tile's last K-step and that tile's P~ stores drain under the new loop.
  auto tile_of = [&](int r) -> int {
    const int left = total - r * nwg, here = left < nwg ? left : nwg;
    return (int)blockIdx.x < here ? r * nwg + xcd_remap(blockIdx.x, here) : -1;
  };
#ifdef HVR_DBG_BT_CLK
  long long dbg_t[2 + 3 * 4];
  for (int q = 0; q < 14; ++q) dbg_t[q] = 0;
  dbg_t[0] = wall_clock64();
#endif
  int cur = tile_of(0);
  const bool has_tile = cur >= 0;
  int g_cur = 0, m0 = 0, n0 = 0;
  auto locate = [&](int t, int& g, int& tm0, int& tn0) {
    g = t / tpg;
    const int r = t - g * tpg, pm = r / tiles_n;
    tm0 = pm * BT_BM;
    tn0 = (r - pm * tiles_n) * BT_BN;
  };
  if (has_tile) locate(cur, g_cur, m0, n0);

  // ---- loader: a thread's 16-byte pieces sit 64 rows apart (slot s = i * 512 + tid -> row i * 64 + tid / 8), all in
  // the same swizzled chunk; offsets are rebuilt at issue time (three VALU ops) instead of living in registers.  The tile a
  // piece belongs to is an argument: the last K-step of a tile loads the NEXT tile's first K-step ----
  const int l_row = tid >> 3, l_chunk = ((tid & 7) ^ (l_row & 7)) * 16;
  auto dma_a = [&](auto I, int kt, char* stage, const char* qbase, int tm0) {
    constexpr int i = decltype(I)::value;
    if (i < BT_A_SLOTS - 1 || wave < (BT_BM * 8 - (BT_A_SLOTS - 1) * BT_NT) / 64) {
      int m = tm0 + i * 64 + l_row;
      m = m < p.Mq ? m : p.Mq - 1;
#ifdef HVR_DBG_BT_SAMEROW
      m &= 63;
#endif
      const char* src = qbase + ((long)m * p.ldq * EB + l_chunk + kt * 128);
#ifdef HVR_DBG_BT_NODMA
      if (kt == 0)
#endif
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + (i * BT_NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto dma_b = [&](auto I, int kt, char* stage, const char* kbase, int tn0) {
    constexpr int i = decltype(I)::value;
    int n = tn0 + i * 64 + l_row;
    n = n < p.Mk ? n : p.Mk - 1;
#ifdef HVR_DBG_BT_SAMEROW
    n &= 63;
#endif
    const char* src = kbase + ((long)n * p.ldk * EB + l_chunk + kt * 128);
#ifdef HVR_DBG_BT_NODMA
    if (kt == 0)
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(stage + BT_A_BYTES + (i * BT_NT + wave * 64) * 16), 16, 0, 0);
  };
  const char* q_cur = (const char*)p.Q + (long)g_cur * p.gs_q * EB;
  const char* k_cur = (const char*)p.K + (long)g_cur * p.gs_k * EB;

  // the first tile's first K-step into stage 0; it lands while the workgroup transposes its share of V through stage 1's memory
  if (has_tile) {
    static_for<BT_A_SLOTS>([&](auto I) { dma_a(I, 0, smem, q_cur, m0); });
    static_for<BT_B_SLOTS>([&](auto I) { dma_b(I, 0, smem, k_cur, n0); });
  }

  // ---------------- V^T[g][D][ldp] = V[g][Mk][ldv]^T of every group, zero-filled for keys Mk .. ldp - 1 ----------------
  // 64 x 64 tiles, one per half workgroup (256 threads) and pass, 16-byte global accesses on both sides, the next pass's loads in
  // flight under this pass's LDS round trip (two LDS buffers per half: one barrier per pass).  WHO copies: when the last round of
  // score tiles leaves workgroups without a tile (234 tiles of one 4 500-row window: 22; four windows, 936 tiles: 88 in the fourth
  // round), those workgroups copy V^T beside the others' last tiles -- the copy (7 us per window when every workgroup carried
  // its share in front of its first tile) leaves the critical path; with a full last round every workgroup takes its share first.
  const int rounds = (total + nwg - 1) / nwg;
  const int here_last = total - (rounds - 1) * nwg;   // tiles of the last round
  const int n_idle = nwg - here_last;                 // workgroups without a tile in it
  const bool vt_by_idle = n_idle >= 16;
  auto vt_copy = [&](int first, int step, int tid) {  // 64 x 64 tiles first + half, first + half + step, ... of the list over all groups
#ifndef HVR_DBG_BT_NOTRANSPOSE
    constexpr int PITCH = 64 * 2 + 16, BUF = 64 * PITCH;
    const int half = tid >> 8, ht = tid & 255;
    char* const tb = smem + BT_STAGE + half * (2 * BUF);
    // (split half: the hi and the lo plane of a 64 x 64 logical tile are two independent transposes of 16-bit words -- twice the
    // tiles, plane = tile & 1 -- and only the addressing knows about the [32 hi | 32 lo] groups)
    constexpr int PLANES = SPLIT ? 2 : 1;
    const int tr_c = p.D / 64, ntr = tr_c * (int)(p.ldp / 64) * PLANES, ntr_all = ntr * p.groups;
    auto fetch = [&](int tt_all, uint4 (&v)[2]) {     // (clamped addresses + selects: no predicated load, no per-element round trip)
      const bool live = tt_all < ntr_all;
      const int tc = live ? tt_all : ntr_all - 1;
      const int tg = tc / ntr, tp = tc - tg * ntr, tt = tp / PLANES, plane = (tp - tt * PLANES) * kSplitPlane;
      const int r0 = (tt / tr_c) * 64, c0 = (tt % tr_c) * 64;
      const char* Vg = (const char*)p.V + (long)tg * p.gs_v * EB;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = it * 256 + ht, i = s >> 3, q = s & 7;
        const int r = r0 + i, rc = r < p.Mk ? r : p.Mk - 1;
        const uint4 x = *reinterpret_cast<const uint4*>(Vg + (long)rc * p.ldv * EB + (SPLIT ? split_col_bytes(c0 + q * 8) + plane : (long)(c0 + q * 8) * 2));
        const bool keep = live && r < p.Mk;
        v[it] = make_uint4(keep ? x.x : 0u, keep ? x.y : 0u, keep ? x.z : 0u, keep ? x.w : 0u);
      }
    };
    uint4 cur[2], nxt[2];
    fetch(first + half, cur);
    int buf = 0;
    for (int base = first; base < ntr_all; base += step) {
      const int tt_all = base + half;
      const bool live = tt_all < ntr_all;
      char* const tbuf = tb + buf * BUF;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int s = it * 256 + ht, i = s >> 3, q = s & 7;
        *reinterpret_cast<uint4*>(tbuf + i * PITCH + q * 16) = cur[it];
      }
      fetch(base + step + half, nxt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (live) {
        const int tg = tt_all / ntr, tp = tt_all - tg * ntr, tt = tp / PLANES, plane = (tp - tt * PLANES) * kSplitPlane;
        const int r0 = (tt / tr_c) * 64, c0 = (tt % tr_c) * 64;
        char* Vtg = (char*)p.Vt + (long)tg * p.gs_vt * EB;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int s = it * 256 + ht, q = s >> 6, i = s & 63;
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t lo = *reinterpret_cast<const bf16_t*>(tbuf + (q * 8 + 2 * e) * PITCH + i * 2);
            const uint32_t hi = *reinterpret_cast<const bf16_t*>(tbuf + (q * 8 + 2 * e + 1) * PITCH + i * 2);
            w[e] = lo | (hi << 16);
          }
          *reinterpret_cast<uint4*>(Vtg + (long)(c0 + i) * p.ldp * EB + (SPLIT ? split_col_bytes(r0 + q * 8) + plane : (long)(r0 + q * 8) * 2)) =
              make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      cur[0] = nxt[0]; cur[1] = nxt[1];
      buf ^= 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the buffers are the first tile's stage 1
#else
    (void)first; (void)step; (void)tid;
#endif
  };
  if (!vt_by_idle) vt_copy((int)blockIdx.x * 2, nwg * 2, tid);
#ifdef HVR_DBG_BT_CLK
  dbg_t[1] = wall_clock64();
#endif

  const int nk = p.D / BKE;   // even (scores_bt_supported): every tile starts in stage 0 and ends in stage 1
  if (has_tile) for (int round = 0;; ++round) {
  // the tile after this one (its first K-step is this tile's last prefetch)
  const int nxt_tile = tile_of(round + 1);
  const bool has_next = nxt_tile >= 0;
  int g_nxt = g_cur, m0_nxt = m0, n0_nxt = n0;
  if (has_next) locate(nxt_tile, g_nxt, m0_nxt, n0_nxt);
  const char* q_nxt = (const char*)p.Q + (long)g_nxt * p.gs_q * EB;
  const char* k_nxt = (const char*)p.K + (long)g_nxt * p.gs_k * EB;

  f32x4 acc[BT_FM][BT_FN];
#pragma unroll
  for (int i = 0; i < BT_FM; ++i)
#pragma unroll
    for (int j = 0; j < BT_FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = lds_off(smem) + (wm * BT_WROWS + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = lds_off(smem) + BT_A_BYTES + (wn * BT_WCOLS + frag_row) * 128 + ((frag_grp ^ swz) * 16);

  // the first K-step has landed: the first tile's was issued above (wait for it); a later tile's was waited for by the issuing
  // waves inside the previous tile's last K-step -- no vmcnt wait here, which would also wait for that tile's P~ stores
  if (round == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef HVR_DBG_BT_CLK
  if (round < 4) dbg_t[2 + 3 * round] = wall_clock64();
#endif

  // ---- phase-staggered K loop ----
  // The eight waves are two groups of four (wm = 0 / 1: waves w and w + 4 share SIMD w).  A K-step is four phases --
  // (K half h, row fragments 0..5) and (h, 6..10) for h = 0, 1 -- and a phase is two sections with an s_barrier behind each:
  //   L: request the phase's fragments from the LDS (the half's 4 key fragments + 6 / 5 query fragments) and this wave's
  //      share of the next K-step's DMA;            C: lgkmcnt(0), then nothing but the phase's 24 / 20 MFMAs.
  // Group 1 runs ONE BARRIER behind group 0 (it takes one extra s_barrier before the loop, group 0 one after it), so on
  // every SIMD one wave is in a C section while its partner is in an L section: the matrix pipe always has a pure MFMA
  // stream to run and the LDS / DMA issue of the partner goes down the other ports beside it, instead of both waves of a
  // SIMD asking for the LDS together and for the matrix pipe together (a lock-step loop ran its MFMAs at ~0.6 of the pipe's rate).
  // Sections of K-step kt, numbered by barrier: group 0 reads its stage in sections 0, 2, 4, 6, group 1 in 1, 3, 5, 7 and
  // its last reads have returned at the top of section 8 (= section 0 of kt + 1).  The other stage -- read last during
  // K-step kt - 1 -- is therefore free from section 1 on: group 1 issues its DMA pieces in its L sections of phases 0 and 1
  // (sections 1, 3), group 0 in phases 1 and 2 (sections 2, 4); every wave waits for its own pieces (vmcnt(0)) in front of
  // the barrier that ends section 7, behind which group 0's first read of K-step kt + 1 sits.
  {
    constexpr int G0 = 6;  // row fragments of the first phase of a half (the second takes FM - G0)
    constexpr int DMA_TOTAL = BT_A_SLOTS + BT_B_SLOTS, DMA_FIRST = DMA_TOTAL / 2;
    // The loop exists TWICE, once per wave group, chosen by one branch in front of it: which sections carry a wave's DMA pieces
    // and its vmcnt wait depend on the group, and as run-time tests those were three or four taken branches per K-step in every
    // wave (a taken branch costs a wave ~100 cycles of instruction fetch: profiles/r04_kloop_probe.txt).
    auto kloop = [&](auto WMC) __attribute__((always_inline)) {
    constexpr int wmc = decltype(WMC)::value;
    constexpr int dma_ph = wmc ? 0 : 1;  // first of the two phases whose L sections carry this wave's DMA pieces
    if constexpr (wmc != 0) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t soff = (uint32_t)(kt & 1) * BT_STAGE;
      char* nxt = smem + ((kt + 1) & 1) * BT_STAGE;
      // what this K-step prefetches into the other stage: K-step kt + 1 of this tile; from the last step the NEXT tile's first
      // K-step (scalar selects: one uniform stream); with no tile left the last step re-fetches itself into the idle stage
      const bool last = kt + 1 >= nk, pre = last && has_next;
      const int kn = last ? (has_next ? 0 : kt) : kt + 1;
      const char* lq = pre ? q_nxt : q_cur;
      const char* lk = pre ? k_nxt : k_cur;
      const int lm0 = pre ? m0_nxt : m0, ln0 = pre ? n0_nxt : n0;
      const uint32_t a0 = a_lane + soff, b0 = b_lane + soff;
      uint4 kb[BT_FN], qa[G0];
      // phases of a K-step: (64-byte half of the keys' lines, of the queries' lines, row fragments 0..5 / 6..10).  Two-byte formats:
      // the halves are the two 32-element K halves, (0,0) (0,0) (1,1) (1,1).  Split half: the halves are the hi / lo planes and the
      // three products K_hi Q_hi, K_hi Q_lo, K_lo Q_hi take two phases each, (0,0) (0,0) (0,1) (0,1) (1,0) (1,0) -- one set of key
      // fragments live at a time, the hi query fragments read twice (0.31 LDS fragment reads per MFMA)
      constexpr int NPH = SPLIT ? 6 : 4;
      static_for<NPH>([&](auto PH) {
        constexpr int ph = decltype(PH)::value, r0 = (ph & 1) ? G0 : 0, nr = (ph & 1) ? BT_FM - G0 : G0;
        constexpr int hk = SPLIT ? (ph >= 4 ? 1 : 0) : (ph >> 1);            // half of the key lines this phase multiplies with
        constexpr int hq = SPLIT ? ((ph == 2 || ph == 3) ? 1 : 0) : (ph >> 1);  // half of the query lines
        constexpr bool read_keys = SPLIT ? (ph == 0 || ph == 4) : ((ph & 1) == 0);
        // ---- L ----
        if constexpr (read_keys)
          static_for<BT_FN>([&](auto J) { kb[decltype(J)::value] = lds_read128<decltype(J)::value * 2048>(hk ? (b0 ^ 64u) : b0); });
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          qa[r] = lds_read128<(r0 + r) * 2048>(hq ? (a0 ^ 64u) : a0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph < NPH - 1) {
          if constexpr (dma_ph == ph) {
            static_for<DMA_FIRST>([&](auto D) {
              constexpr int d = decltype(D)::value;
              if constexpr (d < BT_A_SLOTS) dma_a(std::integral_constant<int, d>{}, kn, nxt, lq, lm0);
              else dma_b(std::integral_constant<int, d - BT_A_SLOTS>{}, kn, nxt, lk, ln0);
            });
          } else if constexpr (dma_ph + 1 == ph) {
            static_for<DMA_TOTAL - DMA_FIRST>([&](auto D) {
              constexpr int d = DMA_FIRST + decltype(D)::value;
              if constexpr (d < BT_A_SLOTS) dma_a(std::integral_constant<int, d>{}, kn, nxt, lq, lm0);
              else dma_b(std::integral_constant<int, d - BT_A_SLOTS>{}, kn, nxt, lk, ln0);
            });
          }
        }
        if constexpr (ph == NPH - 1) {
          if constexpr (wmc != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group 1: this barrier is the one in front of K-step kt + 1
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- C ----
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          static_for<BT_FN>([&](auto J) { mma<HT>(kb[decltype(J)::value], qa[r], acc[r0 + r][decltype(J)::value]); });
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
    int wsel = wave;   // (an opaque copy: the test is redone per tile from the scalar wave number instead of living in a vector register)
    asm volatile("" : "+s"(wsel));
    if (wsel >= BT_WN) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  }

#ifdef HVR_DBG_BT_CLK
  if (round < 4) dbg_t[3 + 3 * round] = wall_clock64();
#endif
#ifdef HVR_DBG_BT_NOEPI
  {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < BT_FM; ++i)
#pragma unroll
      for (int j = 0; j < BT_FN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.f) p.mstat[0] = t;
  }
#else
  {
  // ---------------- epilogue: block max / exp2 / pack / sums ----------------
  // (the parameters the epilogue alone needs -- P, the statistics, their strides, the scale -- are re-read from the kernel argument
  // segment here, behind an opaque copy of its address, instead of sitting in scalar registers across the K loop: the kernel ran
  // out of SGPRs and kept uniform values in vector registers and spill lanes)
  const __attribute__((address_space(4))) ScoresBTParams* kp =
      (const __attribute__((address_space(4))) ScoresBTParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  // (the lane-derived values are re-derived from a laundered copy of the thread index: nothing but the accumulators and the
  // loop's own addresses stays live across the K loop, which runs at the 256-register limit)
  // (the lane index from mbcnt, not from threadIdx.x: the launch's v0 would otherwise have to survive the loop)
  int elane;   // (volatile: not hoisted out of the tile loop into a register that lives across the K loop)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
  const int lane = elane, frag_row = lane & 15, frag_grp = lane >> 4;
  char* const Pg = (char*)kp->P + (long)g_cur * kp->gs_p * EB;
  float* const mst = kp->mstat + (long)g_cur * kp->gs_stat;
  float* const lst = kp->lstat + (long)g_cur * kp->gs_stat;
  // LDS scratch in STAGE 1 (the last K-step's stage: every wave is done reading it; stage 0 is receiving the next tile's first
  // K-step): [8][176] maxima, [8][176] sums, then one 16-row x 64-key bf16 staging block per wave
  float* red_max = reinterpret_cast<float*>(smem + BT_STAGE);
  float* red_sum = red_max + 8 * BT_WROWS;
  // staged rows are 128 bytes with no padding; bank-conflict-free on both sides (SQ_LDS_BANK_CONFLICT was 491 k cycles per
  // launch with 144-byte rows): the 16-byte piece index is XOR-ed with (row & 7) -- the 16 lanes of a ds_read_b128 group then
  // cover all 64 banks -- and rows 8..15 swap the two 8-byte halves of a piece, so that the 16 rows of a ds_write_b64 lane
  // group (same fragment column, rows r and r + 8 on the same piece) land on 32 distinct banks; the h = 1 read swaps them back
  // (split half: a staged row is the 256 bytes of its two [32 hi | 32 lo] groups; 288-byte pitch: the 8-byte writes of a lane group,
  // 16 rows x 4 fragment groups, land on 32 distinct banks per half)
  constexpr int SPITCH = SPLIT ? BT_WCOLS * 4 + 32 : BT_WCOLS * 2;
  char* stg = smem + BT_STAGE + 2 * 8 * BT_WROWS * 4 + wave * (16 * SPITCH);
  const int wr_lane = frag_row * SPITCH + (((frag_grp & 1) ^ (frag_row >> 3)) << 3);  // + ((2 j + (g >> 1)) ^ (row & 7)) * 16
  const int blk = wn >> 1;                                          // 128-key block of this wave inside the tile
  const bool blk_live = n0 + blk * 128 < kp->ldp;                     // an odd block count leaves the last tile half empty
  const int ncol0 = n0 + wn * BT_WCOLS + frag_grp * 4;              // first key of this lane's fragment-0 columns
  float tmax[BT_FM];
  if (n0 + BT_BN > kp->Mk) {   // (only the last column tile holds keys past Mk: a scalar branch around 2 x 176 VALU operations)
#pragma unroll
    for (int i = 0; i < BT_FM; ++i)
#pragma unroll
      for (int j = 0; j < BT_FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][j][r] = (ncol0 + j * 16 + r < kp->Mk) ? acc[i][j][r] : -INFINITY;  // keys past Mk never win the max and get weight 0
  }
#pragma unroll
  for (int i = 0; i < BT_FM; ++i) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < BT_FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
    mx = quad_group_max(mx);
    tmax[i] = mx;
    if (frag_grp == 0) red_max[wave * BT_WROWS + i * 16 + frag_row] = mx;
  }
  // (raw barriers: __syncthreads() carries s_waitcnt vmcnt(0), which from the second tile on would wait for global stores)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < BT_FM; ++i) {
    tmax[i] = fmaxf(tmax[i], red_max[(wave ^ 1) * BT_WROWS + i * 16 + frag_row]) * kp->sl2;  // block max, log2 units
    if (kp->int_max) tmax[i] = ceilf(tmax[i]);   // (relation_bt.h: exact power-of-two block weights for relation_apply_bt.hip)
  }
  const int st_row = lane >> 3, st_chunk = lane & 7;  // store phase: lane -> (row, 16-byte piece) of the staged block
#pragma unroll
  for (int i = 0; i < BT_FM; ++i) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < BT_FN; ++j) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], kp->sl2, -tmax[i]));
      sum += (e[0] + e[1]) + (e[2] + e[3]);
      if constexpr (SPLIT) {
        // stored x 2^12 so that the lo halves stay normal (the apply product takes the factor back); the sums are of the unscaled values
        // (values in [0, 2^12]: no saturation needed, the split is split2's otherwise)
        float x[4], hx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = e[r] * kSplitProbScale;
        const uint32_t h0 = pack2h(x[0], x[1]), h1 = pack2h(x[2], x[3]);
        unpack2h(h0, hx[0], hx[1]);
        unpack2h(h1, hx[2], hx[3]);
        const uint32_t l0 = pack2h(x[0] - hx[0], x[1] - hx[1]), l1 = pack2h(x[2] - hx[2], x[3] - hx[3]);
        char* dst = stg + frag_row * SPITCH + (int)split_col_bytes(j * 16 + frag_grp * 4);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + kSplitPlane) = make_uint2(l0, l1);
      } else {
        *reinterpret_cast<uint2*>(stg + wr_lane + (((2 * j + (frag_grp >> 1)) ^ (frag_row & 7)) << 4)) = make_uint2(pack2<HT>(e[0], e[1]), pack2<HT>(e[2], e[3]));
      }
    }
    sum = quad_group_sum(sum);
    if (frag_grp == 0) red_sum[wave * BT_WROWS + i * 16 + frag_row] = sum;
    // the wave's 16 x 64 block leaves as two stores of eight whole 128-byte row segments (wave-private staging: the
    // LDS returns in order, no barrier)
    if constexpr (SPLIT) {
      // four stores of four whole 256-byte row segments (lane -> row lane / 16, 16-byte piece lane % 16)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = h * 4 + (lane >> 4), m = m0 + wm * BT_WROWS + i * 16 + row;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + ((lane & 15) << 4));
        if (m < kp->Mq && blk_live)
          *reinterpret_cast<uint4*>(Pg + ((long)m * kp->ldp + n0 + wn * BT_WCOLS) * 4 + ((lane & 15) << 4)) = v;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 8 + st_row, m = m0 + wm * BT_WROWS + i * 16 + row;
        uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + ((st_chunk ^ st_row) << 4));
        if (h) v = make_uint4(v.z, v.w, v.x, v.y);
        if (m < kp->Mq && blk_live)
          *reinterpret_cast<uint4*>(Pg + ((long)m * kp->ldp + n0 + wn * BT_WCOLS + st_chunk * 8) * 2) = v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if ((wn & 1) == 0 && frag_grp == 0 && blk_live) {
    const int t = n0 / 128 + blk;
#pragma unroll
    for (int i = 0; i < BT_FM; ++i) {
      const int row = i * 16 + frag_row, m = m0 + wm * BT_WROWS + row;
      const float sum = red_sum[wave * BT_WROWS + row] + red_sum[(wave + 1) * BT_WROWS + row];
      if (m < kp->Mq) {
        mst[(long)m * kp->ntile + t] = tmax[i];
        lst[(long)m * kp->ntile + t] = sum;
      }
    }
  }
  }
#endif
#ifdef HVR_DBG_BT_CLK
  if (round < 4) dbg_t[4 + 3 * round] = wall_clock64();
#endif
  if (!has_next) break;
  g_cur = g_nxt; m0 = m0_nxt; n0 = n0_nxt; q_cur = q_nxt; k_cur = k_nxt;
  // the scratch reads above are done (their values went into the stores) before this wave reaches the next tile's first barrier,
  // behind which the first DMA into stage 1 is issued
  }
  if (vt_by_idle && (int)blockIdx.x >= here_last) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the last tile's epilogue scratch shares the copy's LDS buffers
    int vlane;   // (the thread index rebuilt from the wave number and mbcnt: the launch's v0 does not have to survive the tile loop)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(vlane));
    vt_copy(((int)blockIdx.x - here_last) * 2, n_idle * 2, wave * 64 + vlane);
  }
#ifdef HVR_DBG_BT_CLK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const long long t_end = wall_clock64();
    if (threadIdx.x == 0)   // ONE printf per workgroup (several would interleave across workgroups): 0 = this workgroup had no such tile
      printf("BTCLK wg %d xcc %d t0 %lld vt %lld | tile0 loop %lld..%lld epi %lld | tile1 loop %lld..%lld epi %lld | tile2 loop %lld..%lld epi %lld | tile3 loop %lld..%lld epi %lld | drained %lld\n",
             (int)blockIdx.x, (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20), dbg_t[0], dbg_t[1] - dbg_t[0],
             dbg_t[2] ? dbg_t[2] - dbg_t[0] : 0, dbg_t[3] ? dbg_t[3] - dbg_t[0] : 0, dbg_t[4] ? dbg_t[4] - dbg_t[0] : 0,
             dbg_t[5] ? dbg_t[5] - dbg_t[0] : 0, dbg_t[6] ? dbg_t[6] - dbg_t[0] : 0, dbg_t[7] ? dbg_t[7] - dbg_t[0] : 0,
             dbg_t[8] ? dbg_t[8] - dbg_t[0] : 0, dbg_t[9] ? dbg_t[9] - dbg_t[0] : 0, dbg_t[10] ? dbg_t[10] - dbg_t[0] : 0,
             dbg_t[11] ? dbg_t[11] - dbg_t[0] : 0, dbg_t[12] ? dbg_t[12] - dbg_t[0] : 0, dbg_t[13] ? dbg_t[13] - dbg_t[0] : 0, t_end - dbg_t[0]);
  }
#endif
}

bool scores_bt_supported(int Mq, int Mk, int D, long ldq, long ldk, long ldv, long ldp, const void* Q, const void* K,
                         const void* V, const void* P, const void* Vt, int groups, bool split) {
  if (split) {
    // split half: 128-byte lines of [32 hi | 32 lo].  (D % 64: an even number of K-steps, whole 64 x 64 V^T tiles)
    const uintptr_t al = reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(P) |
                         reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(Vt);
    if ((al & 127) || D % 64 || ldq % 32 || ldk % 32 || ldv % 32 || ldp % 128 || groups < 1) return false;
    if ((long)Mq * ldq * 4 >= (1L << 31) || (long)Mk * ldk * 4 >= (1L << 31)) return false;
  } else {
    const uintptr_t al = reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
                         reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Vt);
    if (al & 15) return false;
    // (D % 128: an even number of K-steps, so that every tile of a persistent workgroup starts in LDS stage 0)
    if (D % 128 || ldq % 8 || ldk % 8 || ldv % 8 || ldp % 128 || groups < 1) return false;
    if ((long)Mq * ldq * 2 >= (1L << 31) || (long)Mk * ldk * 2 >= (1L << 31)) return false;
  }
  // the 352 x 256 shape only pays once the tile grid fills most of the chip
  const long tiles = (long)((Mq + BT_BM - 1) / BT_BM) * ((ldp + BT_BN - 1) / BT_BN);
  if (Mk < 128 || tiles * groups > (1L << 20)) return false;
  // several groups: the persistent workgroups take tiles of all of them in turn, any per-group count from 160 up fills the rounds
  if (groups > 1) return tiles >= 160;
  // one group: one round on most of the chip, or two rounds whose second is at least half full (the shipped T = 21 window: 450 tiles)
  return (tiles >= 160 && tiles <= 256) || (tiles >= 384 && tiles <= 512);
}

hipError_t run_scores_bt(const ScoresBTParams& p, hipStream_t stream) {
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_scores_bt_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_scores_bt_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relation_scores_bt_kernel<f16s_t>), hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS);
  });
  // one launch of one workgroup per CU: persistent over the tile list of every group; the workgroups a short list leaves without a
  // tile still carry their share of the V^T copies
  if (p.f16 == 2) hipLaunchKernelGGL(relation_scores_bt_kernel<f16s_t>, dim3(256), dim3(BT_NT), BT_LDS, stream, p);
  else if (p.f16) hipLaunchKernelGGL(relation_scores_bt_kernel<f16_t>, dim3(256), dim3(BT_NT), BT_LDS, stream, p);
  else hipLaunchKernelGGL(relation_scores_bt_kernel<bf16_t>, dim3(256), dim3(BT_NT), BT_LDS, stream, p);
  return hipGetLastError();
}

}  // namespace hvr
