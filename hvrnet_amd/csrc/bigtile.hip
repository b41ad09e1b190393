// Big-tile MFMA kernel for the dense convolutions / linear layers of the stride-16 stages (bf16, gfx950):
//     C[M][N] = act(A[M][K] . B[N][K]^T + bias),   A a plain matrix or an NHWC map gathered per filter tap (implicit GEMM)
// Replaces conv + frozen BN + ReLU of mmdet/models/backbones/resnet.py:224-246 (conv1 / conv2 of the layer-3 and res5
// Bottlenecks, their expand convs from K = 256 on), shared_heads/res_layer.py:67-74 and the RPN's 3x3
// (anchor_heads/rpn_head.py:30-33) wherever a 288 x 256 tile grid still covers the chip.
//
// Why a second shape next to gemm.hip's 144 x 256 tiles: with N = 256 .. 512 output channels every tile streams the whole
// weight matrix, and a 144-row tile -- all that one round of 256 CUs leaves a 35 910-pixel batch -- pays 51 KB of L2 -> LDS
// delivery per K-step of 1 152 MFMA cycles plus 0.61 LDS fragment reads per MFMA: 34 % MFMA-busy (profiles/design_history.md section 10, the
// K-loop ablations).  Here ONE workgroup per CU owns 288 x 256: 70 KB per K-step of 2 304 MFMA cycles (30 B/clk instead of
// 44), 8 waves as 2 x 4 with 144 x 64 wave tiles (0.36 fragment reads per MFMA), and the loop is relation_bt.hip's
// phase-staggered one (two wave groups one barrier apart: on every SIMD one wave runs a pure MFMA section while its partner
// issues LDS reads and DMA).  The grid is HALF as large as the 144-row shapes': with N = 512 (res5, the RPN) it still covers the
// chip and the launch is simply faster (165 -> 136 us); with N = 256 (layer 3: 125 workgroups) it is slower alone and measured
// neutral beside a second window's launches, so the library does not take it there (bigtile_supported).
//
// Loader: buffer-addressed LDS-DMA (resource base + one fixed VGPR offset per slot + one SGPR offset per K-step or filter
// tap; an out-of-image tap is offset 2^31 = hardware zero fill), the XOR-swizzled LDS image of gemm.hip.  The MFMA sequence per
// output element is the tile engine's (K ascending, two 32-wide halves per 128-byte K-step), so the outputs are BIT-identical
// to it (tests/test_kernels_gpu.py).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int BG_BN = 256, BG_NT = 512, BG_WN = 4, BG_FN = 4, BG_WCOLS = BG_FN * 16;
constexpr int BG_B_SLOTS = BG_BN * 8 / BG_NT;  // 4

__device__ __forceinline__ uint32_t bg_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 bg_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <typename T> __device__ __forceinline__ void bg_mma(const uint4& w, const uint4& x, f32x4& acc) {
  // weights as the MFMA "A" operand: a lane ends up with 4 consecutive output channels of one pixel (see gemm.hip)
  if constexpr (std::is_same<T, bf16_t>::value)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  else   // f16_t, and the half planes of f16s_t
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
// 16 bytes per lane, global -> LDS, buffer addressing (see gemm.hip): offsets from 2^31 up read as zeros
__device__ __forceinline__ void bg_load_lds16(const void* base, char* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
#else
  (void)base; (void)lds; (void)voff; (void)soff;
#endif
}

}  // namespace

// FM0 / FM1: 16-row fragments per wave along M of wave group 0 (waves 0..3) / group 1 (waves 4..7): the tile is (FM0 + FM1) x 16
// rows -- 9 + 9 = 288, or 5 + 4 = 144 (the one-round shape of a 35 910-pixel batch with N = 256: every SIMD carries one 5-fragment and
// one 4-fragment wave, so the matrix work per SIMD is balanced; 0.45 / 0.5 LDS fragment reads per MFMA against the tile engine's 0.61
// for the same 144 x 256 tile)
// T: bf16_t, f16_t (the same kernel on the half MFMA) or f16s_t (split half, common.h: the 128-byte line of a row and K-step holds the
// hi and the lo plane of 32 logical k, and a K-step is SIX phases instead of four -- the three terms B_hi x A_hi, B_lo x A_hi,
// B_hi x A_lo, each over the two row-fragment groups -- issued from the one LDS image in the tile engine's order per accumulator, so
// the outputs are bit-identical to gemm_tile.h's split K-step; the fragments of a term are re-read from the LDS, which has the
// bandwidth to spare: 0.36 reads per MFMA as in the two-byte formats, for two thirds of their LDS-DMA per MFMA)
template <typename T, int FM0, int FM1>
__global__ __launch_bounds__(BG_NT) void big_tile_kernel(const GemmParams p) {
  constexpr bool SPLIT = std::is_same<T, f16s_t>::value;
  constexpr int EB = (int)sizeof(T), KSG = 128;
  constexpr int BKE = SPLIT ? 32 : 64;   // logical elements per K-step
  constexpr int BM = (FM0 + FM1) * 16;
  constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BG_BN * 128;
  constexpr int A_SLOTS = (BM * 8 + BG_NT - 1) / BG_NT;
  constexpr int LAST_WAVES = (BM * 8 - (A_SLOTS - 1) * BG_NT) / 64;  // waves that carry a piece of the last A slot
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / BG_WN, wn = wave % BG_WN;
  const int wrow0 = wm ? FM0 * 16 : 0;  // first tile row of this wave
  const int tiles_n = p.N / BG_BN, tiles_m = (p.M + BM - 1) / BM;
  // PERSISTENT workgroups (round 6, relation_bt.hip's scheme): the launch is min(tiles, 256) workgroups, one per CU; the tile list is
  // walked in rounds of gridDim.x, round r gives this workgroup tile r * nwg + xcd_remap(blockIdx.x, tiles of the round) -- n fastest
  // inside an XCD's contiguous range, so that the column tiles of a row panel share its A rows in one L2.  From its second tile on a
  // workgroup has its first K-step fetched under the previous tile's last K-step, its loader state (offsets, out-of-image masks: the
  // integer divisions of the conv geometry) built under the previous tile's epilogue, and that tile's output stores drain under the
  // new loop instead of in front of a kernel boundary (profiles/r06_bigtile_probe.txt: one tile per workgroup spent 3.3 us in the
  // prologue + 4 - 7 us in the epilogue and its drain per 32 - 58 us loop, plus the dispatch of a second wave of workgroups).
  const int total = tiles_m * tiles_n, nwg = (int)gridDim.x;
  auto tile_of = [&](int r) -> int {
    const int left = total - r * nwg, here = left < nwg ? left : nwg;
    return (int)blockIdx.x < here ? r * nwg + xcd_remap(blockIdx.x, here) : -1;
  };
#ifdef HVR_DBG_BG_CLK
  long long dbg_t[1 + 3 * 3];
  for (int q = 0; q < 10; ++q) dbg_t[q] = 0;
  dbg_t[0] = wall_clock64();
#endif

  // ---- loader: a thread's pieces sit 64 rows apart (slot i -> row i * 64 + tid / 8), all in the same swizzled 16-byte chunk ----
  const int l_row = tid >> 3, l_chunk = ((tid & 7) ^ (l_row & 7)) * 16;
  int a_bias = 0;
  if (p.conv) a_bias = (int)(((long)p.pad * p.W + p.pad) * p.Cin * EB);
  const char* const rs_a = (const char*)p.A - a_bias;
  const char* const rs_b = (const char*)p.B;
  // second K segment (p.s2 > 0, plain products: a Bottleneck's projection shortcut folded into its closing 1x1, see gemm.hip):
  // K-steps from K1 / 64 on read the block input, an NHWC map [.][H2][W2][K - K1] sampled at stride s2, through a second resource
  const bool seg2 = p.s2 > 0;
  const int k1_steps = seg2 ? p.K1 / BKE : 0x7fffffff;
  const char* const rs_a2 = (const char*)p.A2;
  // BRANCH-FREE loader state (the form and the reasons: gemm_tile.h's pipelined K-step, profiles/r04_kloop_probe.txt -- a taken
  // branch costs a wave ~100 cycles, and this loop's L sections are only hidden while they are shorter than the partner group's
  // 16-20 MFMAs).  One form for plain products and convs: the plain product is the conv formula with an unreachable channel count;
  // the filter tap advances by scalar selects; a piece's out-of-image test is bit `tap` of a per-slot mask built once per tile.
  const bool is_conv = p.conv != 0;
  const int cCin = is_conv ? p.Cin : 0x40000000, cKW = is_conv ? p.KW : 1, cDil = is_conv ? p.dil : 0;
  const int cRowB = is_conv ? p.W * p.Cin * EB : 0, cPixB = is_conv ? p.Cin * EB : 0;   // bytes per input row / pixel (tensors < 2 GiB)

  // the per-TILE part of the loader state: A / B offsets of this thread's pieces, the second segment's offsets, the out-of-image masks
  // (bit t: filter tap t of slot i's pixel lies outside the image; <= 32 taps: bigtile_supported)
  // (split half has no second segment -- bigtile_supported -- and no registers to spare for its offsets)
  constexpr bool SEG2 = !SPLIT;
  struct TileState { int a_off[A_SLOTS], a_off2[SEG2 ? A_SLOTS : 1], b_off[BG_B_SLOTS]; unsigned oob[A_SLOTS]; int m0, n0; };
  // (what only setup() and the epilogue need -- the conv geometry, the output / residual / bias pointers -- is re-read from the kernel
  // argument segment behind an opaque copy of its address instead of sitting in scalar registers across the K loop: relation_bt.hip)
  auto setup = [&](int tile, TileState& st) {
    const __attribute__((address_space(4))) GemmParams* kq =
        (const __attribute__((address_space(4))) GemmParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const int pid_m = tile / tiles_n, pid_n = tile - pid_m * tiles_n;  // n fastest: the tiles of a row panel share its A rows
    st.m0 = pid_m * BM;
    st.n0 = pid_n * BG_BN;
    int a_yx[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      int m = st.m0 + i * 64 + l_row;
      m = m < kq->M ? m : kq->M - 1;
      if constexpr (SEG2) st.a_off2[i] = 0;
      if (kq->conv || (SEG2 && kq->s2 > 0)) {
        const int ox = m % kq->OW, t = m / kq->OW, oy = t % kq->OH, b = t / kq->OH;
        if (kq->conv) {
          const int iy = oy * kq->stride - kq->pad, ix = ox * kq->stride - kq->pad;
          a_yx[i] = (iy << 16) | (ix & 0xffff);
          st.a_off[i] = (int)((((long)b * kq->H + iy) * kq->W + ix) * (long)kq->Cin * EB) + l_chunk + (int)(((long)kq->pad * kq->W + kq->pad) * kq->Cin * EB);
        } else {
          a_yx[i] = 0;
          st.a_off[i] = (int)((long)m * kq->lda * EB) + l_chunk;
        }
        if constexpr (SEG2) if (kq->s2 > 0) st.a_off2[i] = (int)((((long)b * kq->H2 + oy * kq->s2) * kq->W2 + ox * kq->s2) * (long)(kq->K - kq->K1) * EB) + l_chunk;
      } else {
        a_yx[i] = 0;
        st.a_off[i] = (int)((long)m * kq->lda * EB) + l_chunk;
      }
    }
#pragma unroll
    for (int i = 0; i < BG_B_SLOTS; ++i) st.b_off[i] = (int)((long)(st.n0 + i * 64 + l_row) * kq->ldb * EB) + l_chunk;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) st.oob[i] = 0u;
    if (kq->conv != 0) {
      unsigned colbad[A_SLOTS];
#pragma unroll
      for (int i = 0; i < A_SLOTS; ++i) colbad[i] = 0u;
      for (int kx = 0, dx = 0; kx < kq->KW; ++kx, dx += kq->dil) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) colbad[i] |= (unsigned)((short)a_yx[i] + dx) < (unsigned)kq->W ? 0u : (1u << kx);
      }
      const unsigned all_kw = kq->KW >= 32 ? ~0u : (1u << kq->KW) - 1u;   // (a 1 x 32 filter passes the <= 32 taps gate: no shift by 32)
      for (int ky = 0, dy = 0, sh = 0; ky < kq->KH; ++ky, dy += kq->dil, sh += kq->KW) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) st.oob[i] |= ((unsigned)((a_yx[i] >> 16) + dy) < (unsigned)kq->H ? colbad[i] : all_kw) << sh;
      }
    }
  };
  TileState cs, ns;   // the tile being computed, the tile after it
  int cur_tile = tile_of(0), nxt_tile = tile_of(1);
  setup(cur_tile, cs);          // (gridDim.x <= tiles: every workgroup has a first tile)
  setup(nxt_tile >= 0 ? nxt_tile : cur_tile, ns);

  int a_koff = 0, b_koff = 0, kt_load = 0, t_cin0 = 0, t_kx = 0, t_ky = 0, t_tap = 0;  // of the K-step being loaded (starts at step 0)
  const int nk_real = p.K / BKE;
  // adv = 1: the next K-step; 0: stay (a workgroup's last tile re-fetches its last step into the idle stage: one uniform stream);
  // rst: the NEXT tile's K-step 0 (the last K-step of a tile that has a successor)
  auto tap_next = [&](int adv, bool rst) {
    kt_load += adv;
    b_koff += adv * KSG;
    t_cin0 += adv * BKE;
    const bool w1 = t_cin0 >= cCin;
    t_cin0 = w1 ? 0 : t_cin0;
    t_tap += w1 ? 1 : 0;
    t_kx += w1 ? 1 : 0;
    const bool w2 = t_kx >= cKW;
    t_kx = w2 ? 0 : t_kx;
    t_ky += w2 ? 1 : 0;
    const int wmask = -(int)w1;   // (both arms computed and masked: as a select this comes back as scalar branches)
    a_koff = ((t_ky * cDil * cRowB + t_kx * cDil * cPixB) & wmask) | ((a_koff + adv * KSG) & ~wmask);
    const int keep = rst ? 0 : -1;   // (scalar masks, not a branch)
    kt_load &= keep; b_koff &= keep; t_cin0 &= keep; t_kx &= keep; t_ky &= keep; t_tap &= keep; a_koff &= keep;
  };
  // `nx`: the piece belongs to the next tile (uniform)
  auto dma_a = [&](auto I, char* stage, bool nx) {
    constexpr int i = decltype(I)::value;
    if (i < A_SLOTS - 1 || wave < LAST_WAVES) {   // (scalar: `wave` lives in an SGPR)
      const unsigned ao = (unsigned)(nx ? ns.a_off[i] : cs.a_off[i]), ob = nx ? ns.oob[i] : cs.oob[i];
      const unsigned voff = ao | ((ob >> (t_tap & 31)) << 31);   // offsets from 2^31 up read as zeros
      if constexpr (SEG2) {
        const bool s2 = kt_load >= k1_steps;
        const char* const base = s2 ? rs_a2 : rs_a;
        const unsigned vo = s2 ? (unsigned)(nx ? ns.a_off2[i] : cs.a_off2[i]) : voff;
        const int so = s2 ? (kt_load - k1_steps) * KSG : a_koff;
        bg_load_lds16(base, stage + (i * BG_NT + wave * 64) * 16, vo, __builtin_amdgcn_readfirstlane(so));
      } else {
        bg_load_lds16(rs_a, stage + (i * BG_NT + wave * 64) * 16, voff, __builtin_amdgcn_readfirstlane(a_koff));
      }
    }
  };
  auto dma_b = [&](auto I, char* stage, bool nx) {   // (of the K-step the loader state stands at)
    constexpr int i = decltype(I)::value;
    bg_load_lds16(rs_b, stage + A_BYTES + (i * BG_NT + wave * 64) * 16, (unsigned)(nx ? ns.b_off[i] : cs.b_off[i]), __builtin_amdgcn_readfirstlane(b_koff));
  };

  // the first tile's first K-step into stage 0
  static_for<A_SLOTS>([&](auto I) { dma_a(I, smem, false); });
  static_for<BG_B_SLOTS>([&](auto I) { dma_b(I, smem, false); });

  auto body = [&](auto FMC) {
  constexpr int FM = decltype(FMC)::value;
  int par = 0;   // LDS stage this tile's K-step 0 sits in (an odd K-step count flips it from tile to tile)
  for (int round = 0;; ++round) {
  const bool has_next = nxt_tile >= 0;
  const int m0 = cs.m0, n0 = cs.n0;
  f32x4 acc[FM][BG_FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < BG_FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  const uint32_t a_lane = bg_lds_off(smem) + (wrow0 + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = bg_lds_off(smem) + A_BYTES + (wn * BG_WCOLS + frag_row) * 128 + ((frag_grp ^ swz) * 16);

  const int nk = nk_real;
  // the first K-step has landed: the first tile's was issued above (wait for it); a later tile's was waited for by the issuing waves
  // inside the previous tile's last K-step -- no vmcnt wait here, which would also wait for that tile's output stores
  if (round == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef HVR_DBG_BG_CLK
  if (round < 3) dbg_t[1 + 3 * round] = wall_clock64();
#endif

  // ---- phase-staggered K loop (relation_bt.hip): a K-step is four phases -- (K half h, row fragments 0..G0-1) and (h, G0..FM-1)
  // for h = 0, 1 -- each an L section (fragment reads + this wave's share of the next K-step's DMA) and a C section (nothing but
  // MFMAs) with an s_barrier behind each; wave group 1 (waves 4..7, the SIMD partners of 0..3) runs one barrier behind group 0 ----
  {
    constexpr int G0 = (FM + 1) / 2;
    constexpr int DMA_TOTAL = A_SLOTS + BG_B_SLOTS, DMA_FIRST = DMA_TOTAL / 2;
    // (the loop exists twice, once per wave group, behind one branch: which sections carry a wave's DMA pieces and its vmcnt wait
    // depend on the group -- as run-time tests those were taken branches in every K-step, relation_bt.hip)
    auto kloop = [&](auto WMC) __attribute__((always_inline)) {
    constexpr int wmc = decltype(WMC)::value;
    constexpr int dma_ph = wmc ? 0 : 1;  // first of the two phases whose L sections carry this wave's DMA pieces
    if constexpr (wmc != 0) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t soff = (uint32_t)((kt + par) & 1) * STAGE;
      char* nxt = smem + ((kt + 1 + par) & 1) * STAGE;
      const uint32_t a0 = a_lane + soff, b0 = b_lane + soff;
      uint4 kb[BG_FN], qa[G0];
      // what this K-step prefetches into the other stage: K-step kt + 1 of this tile; from the last step the NEXT tile's first K-step
      // (scalar selects: one uniform stream); a workgroup's last tile re-fetches its last step into the idle stage
      const bool last = kt + 1 >= nk, pre = last && has_next;
      tap_next(last ? 0 : 1, pre);
      // phases: two per term.  Two-byte formats: term = K half (0 / 1) on both operands.  Split half: term 0 = B_hi x A_hi,
      // 1 = B_lo x A_hi, 2 = B_hi x A_lo (the lo plane is the second 64 bytes of the line: the same address with bit 6 flipped)
      constexpr int NPH = SPLIT ? 6 : 4;
      static_for<NPH>([&](auto PH) {
        constexpr int ph = decltype(PH)::value, term = ph >> 1, r0 = (ph & 1) ? G0 : 0, nr = (ph & 1) ? FM - G0 : G0;
        constexpr bool b_lo = term == 1, a_lo = SPLIT ? term == 2 : term == 1;
        // ---- L ----
        if constexpr ((ph & 1) == 0)
          static_for<BG_FN>([&](auto J) { kb[decltype(J)::value] = bg_read128<decltype(J)::value * 2048>(b_lo ? (b0 ^ 64u) : b0); });
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          qa[r] = bg_read128<(r0 + r) * 2048>(a_lo ? (a0 ^ 64u) : a0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph < 3) {
          if constexpr (dma_ph == ph) {
            static_for<DMA_FIRST>([&](auto D) {
              constexpr int d = decltype(D)::value;
              if constexpr (d < A_SLOTS) dma_a(std::integral_constant<int, d>{}, nxt, pre);
              else dma_b(std::integral_constant<int, d - A_SLOTS>{}, nxt, pre);
            });
          } else if constexpr (dma_ph + 1 == ph) {
            static_for<DMA_TOTAL - DMA_FIRST>([&](auto D) {
              constexpr int d = DMA_FIRST + decltype(D)::value;
              if constexpr (d < A_SLOTS) dma_a(std::integral_constant<int, d>{}, nxt, pre);
              else dma_b(std::integral_constant<int, d - A_SLOTS>{}, nxt, pre);
            });
          }
        }
        if constexpr (ph == NPH - 1) {
          if constexpr (wmc != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group 1: this barrier is the one in front of K-step kt + 1
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- C ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        static_for<nr>([&](auto R) {
          constexpr int r = decltype(R)::value;
          static_for<BG_FN>([&](auto J) { bg_mma<T>(kb[decltype(J)::value], qa[r], acc[r0 + r][decltype(J)::value]); });
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
    int wsel = wave;   // (an opaque copy: the test is redone per tile from the scalar wave number)
    asm volatile("" : "+s"(wsel));
    if (wsel >= BG_WN) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
  }
#ifdef HVR_DBG_BG_CLK
  if (round < 3) dbg_t[2 + 3 * round] = wall_clock64();
#endif

  // ---------------- epilogue: bias + ReLU, bf16, whole 128-byte row segments out through per-wave LDS staging ----------------
  // lane holds out[m0 + wrow0 + 16 i + frag_row][n0 + wn 64 + 16 j + 4 frag_grp + r] = acc[i][j][r]
  // The staging blocks sit in the stage the LAST K-step was read from (every wave is done with it behind the barrier below); the other
  // stage is receiving the next tile's first K-step.  Raw barriers: __syncthreads() carries s_waitcnt vmcnt(0), which from the second
  // tile on would wait for the previous tile's stores.
  char* const epi_base = smem + ((nk - 1 + par) & 1) * STAGE;
  const __attribute__((address_space(4))) GemmParams* kp =
      (const __attribute__((address_space(4))) GemmParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  if constexpr (SPLIT) {
    // ---- split half: (alpha acc + beta bias) + resid in f32, ReLU, then the [hi | lo] pair -- the tile engine's order and roundings.
    // A wave's 64 columns are two [32 hi | 32 lo] groups = 256 contiguous bytes of an output row: residual rows come in and output rows
    // leave as whole 16-byte pieces through a per-wave staging image [16 rows][16 pieces], piece position XOR-ed with the row ----
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int el = etid & 63, erow = el & 15, egrp = el >> 4;
    constexpr int SP = 256;
    char* stg = epi_base + wave * (16 * SP);
    float bias[BG_FN][4];
#pragma unroll
    for (int j = 0; j < BG_FN; ++j) {
      const int n = n0 + wn * BG_WCOLS + j * 16 + egrp * 4;
      if (kp->bias) {
        const float4 b = *reinterpret_cast<const float4*>(kp->bias + n);
        bias[j][0] = b.x; bias[j][1] = b.y; bias[j][2] = b.z; bias[j][3] = b.w;
        if (kp->beta != 0.f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bias[j][r] *= kp->beta;
        }
      } else {
        bias[j][0] = bias[j][1] = bias[j][2] = bias[j][3] = 0.f;
      }
    }
    // this lane's 8-byte slots of fragment column j in the staged row `erow`: columns c = 16 j + 4 egrp .. + 4 -> group c / 32, hi plane
    // byte (c % 32) * 2; the lo plane is 64 bytes (four pieces) further
    auto slot = [&](int j, int plane) {
      const int piece = (j >> 1) * 8 + plane * 4 + 2 * (j & 1) + (egrp >> 1);
      return stg + erow * SP + ((piece ^ erow) << 4) + (egrp & 1) * 8;
    };
    // store-phase mapping: piece q = it * 64 + el of the 16 x 16 block -> row q / 16, piece q % 16
    const bool has_res = kp->resid != nullptr;
    const long col_bytes = split_col_bytes(n0 + wn * BG_WCOLS);
    auto load_res = [&](int i, uint4 (&rv)[4]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 64 + el, row = q >> 4, pc = q & 15;
        int m = m0 + wrow0 + i * 16 + row;
        m = m < kp->M ? m : kp->M - 1;
        rv[it] = *reinterpret_cast<const uint4*>((const char*)kp->resid + (long)m * kp->ldr * 4 + col_bytes + pc * 16);
      }
    };
    uint4 rnext[4] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    if (has_res) load_res(0, rnext);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (has_res) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int q = it * 64 + el, row = q >> 4, pc = q & 15;
          *reinterpret_cast<uint4*>(stg + row * SP + ((pc ^ row) << 4)) = rnext[it];
        }
        if (i + 1 < FM) load_res(i + 1, rnext);
      }
#pragma unroll
      for (int j = 0; j < BG_FN; ++j) {
        f32x4 v = acc[i][j];
        if (kp->alpha != 0.f) v *= kp->alpha;
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = v[r] + bias[j][r];
        char* sh = slot(j, 0);
        char* sl = slot(j, 1);
        if (has_res) {
          const uint2 rh = *reinterpret_cast<const uint2*>(sh), rl = *reinterpret_cast<const uint2*>(sl);
          float r0_, r1_, r2_, r3_;
          merge2(rh.x, rl.x, r0_, r1_);
          merge2(rh.y, rl.y, r2_, r3_);
          e[0] += r0_; e[1] += r1_; e[2] += r2_; e[3] += r3_;
        }
        if (kp->relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = fmaxf(e[r], 0.f);
        }
        uint2 oh, ol;
        split2(e[0], e[1], oh.x, ol.x);
        split2(e[2], e[3], oh.y, ol.y);
        *reinterpret_cast<uint2*>(sh) = oh;
        *reinterpret_cast<uint2*>(sl) = ol;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 64 + el, row = q >> 4, pc = q & 15;
        const int m = m0 + wrow0 + i * 16 + row;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * SP + ((pc ^ row) << 4));
        if (m < kp->M) *reinterpret_cast<uint4*>((char*)kp->C + (long)m * kp->ldc * 4 + col_bytes + pc * 16) = v;
      }
    }
  } else
  {
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));  // lane-derived values re-derived here: nothing but the accumulators lives across the loop
    const int el = etid & 63, erow = el & 15, egrp = el >> 4;
    constexpr int SPITCH = BG_WCOLS * 2;  // 128-byte staged rows, bank-conflict-free by the piece swizzle of relation_bt.hip
    char* stg = epi_base + wave * (16 * SPITCH);
    const int wr_lane = erow * SPITCH + (((egrp & 1) ^ (erow >> 3)) << 3);
    float bias[BG_FN][4];
#pragma unroll
    for (int j = 0; j < BG_FN; ++j) {
      const int n = n0 + wn * BG_WCOLS + j * 16 + egrp * 4;
      if (kp->bias) {
        const float4 b = *reinterpret_cast<const float4*>(kp->bias + n);
        bias[j][0] = b.x; bias[j][1] = b.y; bias[j][2] = b.z; bias[j][3] = b.w;
      } else {
        bias[j][0] = bias[j][1] = bias[j][2] = bias[j][3] = 0.f;
      }
    }
    const int st_row = el >> 3, st_chunk = el & 7;
    // Residual (the Bottleneck's identity): fragment i's 16 x 64 block arrives in the store-phase mapping (whole 128-byte row
    // segments, one block ahead), goes through the SAME swizzled staging image the outputs leave through, and every lane picks its
    // four channels from where it is about to write them: (acc + bias) + resid in f32, ReLU, one rounding -- the tile engine's order.
    const bool has_res = kp->resid != nullptr;
    auto load_res = [&](int i, uint4 (&rv)[2]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = m0 + wrow0 + i * 16 + h * 8 + st_row;
        m = m < kp->M ? m : kp->M - 1;
        rv[h] = *reinterpret_cast<const uint4*>((const unsigned short*)kp->resid + (long)m * kp->ldr + n0 + wn * BG_WCOLS + st_chunk * 8);
      }
    };
    uint4 rnext[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    if (has_res) load_res(0, rnext);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring (group 0 leaves the loop a barrier ahead of group 1's last reads)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (has_res) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint4 v = h ? make_uint4(rnext[h].z, rnext[h].w, rnext[h].x, rnext[h].y) : rnext[h];
          *reinterpret_cast<uint4*>(stg + (h * 8 + st_row) * SPITCH + ((st_chunk ^ st_row) << 4)) = v;
        }
        if (i + 1 < FM) load_res(i + 1, rnext);
      }
#pragma unroll
      for (int j = 0; j < BG_FN; ++j) {
        char* slot = stg + wr_lane + (((2 * j + (egrp >> 1)) ^ (erow & 7)) << 4);
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = acc[i][j][r] + bias[j][r];
        if (has_res) {
          const uint2 rr = *reinterpret_cast<const uint2*>(slot);
          float r0_, r1_, r2_, r3_;
          unpack2<T>(rr.x, r0_, r1_);
          unpack2<T>(rr.y, r2_, r3_);
          e[0] += r0_; e[1] += r1_; e[2] += r2_; e[3] += r3_;
        }
        if (kp->relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = fmaxf(e[r], 0.f);
        }
        *reinterpret_cast<uint2*>(slot) = make_uint2(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 8 + st_row, m = m0 + wrow0 + i * 16 + row;
        uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + ((st_chunk ^ st_row) << 4));
        if (h) v = make_uint4(v.z, v.w, v.x, v.y);
        if (m < kp->M) *reinterpret_cast<uint4*>((unsigned short*)kp->C + (long)m * kp->ldc + n0 + wn * BG_WCOLS + st_chunk * 8) = v;
      }
    }
  }
#ifdef HVR_DBG_BG_CLK
  if (round < 3) dbg_t[3 + 3 * round] = wall_clock64();
#endif
  if (!has_next) break;
  // the next tile becomes the current one; the tile after it gets its loader state here, while this tile's stores are on their way
  par = (par + nk) & 1;
  cs = ns;
  cur_tile = nxt_tile;
  nxt_tile = tile_of(round + 2);
  if (nxt_tile >= 0) setup(nxt_tile, ns);
  // (the staging reads above are done -- their values went into the stores -- before this wave reaches the next tile's first barrier,
  // behind which the first DMA into this stage is issued)
  }
#ifdef HVR_DBG_BG_CLK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const long long t_end = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x % 5) == 0)
      printf("BGCLK wg %d t0 %lld | tile0 loop %lld..%lld epi %lld | tile1 loop %lld..%lld epi %lld | tile2 loop %lld..%lld epi %lld | drained %lld\n", (int)blockIdx.x, dbg_t[0],
             dbg_t[1] ? dbg_t[1] - dbg_t[0] : 0, dbg_t[2] ? dbg_t[2] - dbg_t[0] : 0, dbg_t[3] ? dbg_t[3] - dbg_t[0] : 0,
             dbg_t[4] ? dbg_t[4] - dbg_t[0] : 0, dbg_t[5] ? dbg_t[5] - dbg_t[0] : 0, dbg_t[6] ? dbg_t[6] - dbg_t[0] : 0,
             dbg_t[7] ? dbg_t[7] - dbg_t[0] : 0, dbg_t[8] ? dbg_t[8] - dbg_t[0] : 0, dbg_t[9] ? dbg_t[9] - dbg_t[0] : 0, t_end - dbg_t[0]);
  }
#endif
  };  // body
  if constexpr (FM0 == FM1) {
    body(std::integral_constant<int, FM0>{});
  } else {
    if (wm == 0) body(std::integral_constant<int, FM0>{});
    else body(std::integral_constant<int, FM1>{});
  }
}

// The 288 x 256 shape applies to bf16 products with a bias / residual / ReLU epilogue (no f32 output, no split-K), whole
// 256-channel column tiles and whole 128-byte K-steps inside one filter tap.  `throughput`: the caller keeps the rest of the chip
// busy with other launches (tile_hint kBigHint).
bool bigtile_supported(const GemmParams& p, bool throughput) {
  const bool split = p.dtype == DT_F16S;
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16 && !split) || !p.staging || p.out_f32 || p.ksplit_steps > 0) return false;
  const int es = split ? 4 : 2, bke = split ? 32 : 64, row_al = split ? 32 : 8;   // bytes per element, elements per K-step, row pitch granule
  if (p.s2 > 0 && (split || p.conv || p.K1 % 64 || (p.K - p.K1) % 64 || p.K1 < 64 || (reinterpret_cast<uintptr_t>(p.A2) & 15) ||
                   (long)p.M * (p.K - p.K1) * 2 >= (1L << 31))) return false;
  if (p.N % BG_BN || p.K % bke || p.ldc % row_al || p.lda % row_al || p.ldb % row_al) return false;
  if (p.resid && (p.ldr % row_al || (reinterpret_cast<uintptr_t>(p.resid) & (split ? 127 : 15)))) return false;
  if (split && ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C)) & 127)) return false;
  // K >= 256: below that a tile is all prologue and epilogue and the row-panel kernel (expand.hip) is ahead -- layer 2's expand
  // 77 vs 76 us, layer 1's 139 vs 126; layer 3's (K = 256) 40 vs 43 and res5's (K = 512) 117 vs 138 go the other way
  // (with two windows in flight the panel kernel's expand convs, two small workgroups per CU, pack better beside the other
  // window's launches: 160.8 vs 158.1 frames/s; alone on the chip the big tiles win, 139.7 vs 137.5 -- so not under the hint
  // (forcing them there measured -0.7 %, round 4).  Split half has no row-panel kernel for K >= 256: its residual convs take the
  // big tiles in either mode.)
  if (p.tile_hint != kBigForce && (p.K < 256 || (p.resid && throughput && !split))) return false;
  if (p.conv && (p.Cin % bke || p.KH * p.KW > 32)) return false;   // (the loader keeps a 32-bit tap mask per piece)
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.bias);
  if (al & 15) return false;
  if ((long)p.N * p.ldb * es >= (1L << 31)) return false;
  // A grid that fills most of the chip by itself (N = 512: 250 tiles for a 15-frame batch) is faster than the 144-row shapes
  // outright (res5's 3x3 165 -> 136 us, the RPN's 312 -> 260); half a chip's worth (N = 256: 125 tiles, layer 3's 3x3 65 us
  // against 47 on twice the CUs) only pays in CU-time, i.e. for a caller that has other launches for the free half.
  const long tiles = (long)((p.M + 287) / 288) * (p.N / BG_BN);
  static const int on = std::getenv("HVR_BIGTILE") ? std::atoi(std::getenv("HVR_BIGTILE")) : 1;
  constexpr int min_alone = 170;
  // Half-chip grids under the throughput hint -- layer 3's 125 tiles: 63 us on 125 CUs against 43 on 250 is -27 % CU-time, if the
  // other windows in flight have launches for the free half.  With two graph lanes that measured neutral to -1.7 % (round 3); with
  // four (bench.py's default since round 4) it is +0.4 % in bf16 and +0.8 % in split half, twice each on one box
  // (profiles/r04_lanes.txt) -- small, repeatable, so the shared threshold is 96 tiles.
  constexpr int min_shared = 96;
  if (p.tile_hint == kBigForce) return true;
  return on && tiles >= (throughput ? min_shared : min_alone) && tiles <= 4096;
}

template <typename T, int FM0, int FM1>
static hipError_t launch_bigtile(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = (FM0 + FM1) * 16, lds = 2 * (BM + BG_BN) * 128;
  auto kern = big_tile_kernel<T, FM0, FM1>;
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BG_BN);
  // persistent workgroups, one per CU (256 on the MI355X this library is written for): a longer tile list is walked in rounds
  static const int nper = std::getenv("HVR_BIGTILE_WGS") ? std::atoi(std::getenv("HVR_BIGTILE_WGS")) : 256;   // (tuning: 0 = one tile per workgroup)
  const int grid = nper > 0 && tiles > nper ? nper : tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BG_NT), lds, stream, p);
  return hipGetLastError();
}

hipError_t run_bigtile(const GemmParams& p, hipStream_t stream) {
  // (an L2 prefetch of the A rows two K-steps ahead -- one 4-byte load per wave and K-step into an LDS scratch, vmcnt(1) in front of
  // the K-step's closing barrier -- was built in round 6 and measured SLOWER everywhere: layer 3's reducing 1x1 94.2 -> 103.0 us,
  // res5's 2048 -> 512 333 -> 364, fc_new_1 437 -> 475 (profiles/r06_bigtile_probe.txt); removed)
  if (p.dtype == DT_F16S) return launch_bigtile<f16s_t, 9, 9>(p, stream);
  if (p.dtype == DT_F16) return launch_bigtile<f16_t, 9, 9>(p, stream);
  return launch_bigtile<bf16_t, 9, 9>(p, stream);
}
// (the same kernel on 144 x 256 tiles, <5, 4>: bit-identical too, but 61 us against the tile engine's 50 on layer 3's 3x3 -- a
// two-stage ring with 16-20 MFMAs per section does not cover the DMA the way the engine's three-stage ring does; not instantiated)

}  // namespace hvr
