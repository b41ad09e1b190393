// Channel-expanding 1x1 convolution with the block's residual:  out = act(X W^T + shift + R)   (bf16, gfx950)
//   X [M][K] (K = 64 / 128 / 256 / 512), W [N][K] with N = 4 K in a Bottleneck, R / out [M][N].
// Replaces conv3 + bn3 + `out += identity` + ReLU of mmdet/models/backbones/resnet.py:248-264 and of the res5 blocks
// (shared_heads/res_layer.py:67-74) -- 33 launches per frame batch.
//
// Why not the tile engine (gemm.hip): this product is HBM-bound (l3: 18.8 GF over 165 MB; the engine's output tiles
// run 53 us = 3.1 TB/s where an elementwise add over the same tensors streams at 6-7 TB/s on this box).  A K of 1-4
// K-steps leaves an output tile nothing to hide its prologue and its residual / store epilogue under, and every tile
// re-stages the X panel it shares with its row neighbours.  Here a workgroup owns a PANEL of 128 rows and NC chunks of
// 64 output channels:
//   * its X fragments are loaded once, straight from global into registers (the MFMA "B" operand: 16 B per lane),
//     and stay there -- X never touches the LDS;
//   * W streams through a double-buffered LDS chunk of 64 channels x K (global_load_lds, XOR-swizzled image): the next
//     chunk's DMA runs under this chunk's MFMAs and stores; the residual rows are fetched TWO chunks ahead (HBM
//     latency, where W comes from the L2); the shifts sit in the LDS, so a chunk's output stores stay in flight
//     through the whole next chunk (nothing waits on vmcnt behind them but the next top-of-chunk count);
//   * the MFMA row index is permuted (fragment j, row 4g + r <-> channel 16 g + 4 j + r) so that a lane ends a chunk
//     holding 16 CONSECUTIVE channels of one pixel: residual loads and output stores are 16 B per lane without an LDS
//     stage;
//   * two workgroups (4 waves each) per CU: one streams while the other computes;
//   * the chunk loop is FULLY unrolled (NC is a template parameter): every vector-memory operation of the kernel is
//     then in one straight line, the compiler's own vmcnt bookkeeping for the residual registers is exact (inside a
//     loop it merges conservatively and drains the queue in front of every epilogue), and no load needs to be hidden
//     from it.  The only hand-placed vmcnt waits are the ones in front of the barriers (the DMA has no register
//     result the compiler could wait on).
#include <cstdlib>
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

__device__ __forceinline__ uint32_t x_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// the fragment's distance from the lane's base goes into the instruction's 16-bit offset field: with one address
// register per read the unrolled chunks' (identical) address values are kept live across chunks -- 100+ registers
template <int OFF> __device__ __forceinline__ uint4 x_lds_read128(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// s_waitcnt vmcnt(N) through the builtin, not inline asm: the compiler's own wait-count pass reads it and learns that
// everything older has landed.  It treats a pending global_load_lds as "may return out of order" and would otherwise
// put vmcnt(0) in front of the first use of any loaded register while a DMA is in flight (i.e. in every epilogue).
// gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; expcnt / lgkmcnt left at "no wait".
template <int N> __device__ __forceinline__ void x_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

typedef uint32_t xu32x4 __attribute__((ext_vector_type(4)));

constexpr int X_FJ = 4, X_BN = 16 * X_FJ, X_BM = 128, X_NT = 256;

// 16-byte chunk position of global chunk c of W row `row` (row inside the 64-channel LDS chunk): the 16 lanes of a
// ds_read_b128 group read rows {16 a + 4 j + r}: a = 0..3 (two of them per lane-group half), r = 0..3 -- the key
// (2 a + (r >> 1)) makes the 8 lanes of either row parity land on 8 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz_key(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }
// The same for the NEXT conv's weight chunk (NX > 0), whose fragment reads take slot (2 g + v) ^ key -- the lane group g sits in
// bits 1..2 of the slot there, not in bits 0..1.  A ds_read_b128 is served in groups of 16 lanes that hold every q = lane & 15 once,
// with g & 1 = (q in 4..11) (xor the group's parity): the slot is c ^ 2 (q3 ^ q2) ^ key(q), and with swz_key's bits (q3, q2, q1)
// that is (q3, q3, q1) -- four distinct slots for the eight lanes of a row parity, a 2-way bank conflict on every read (round 2's
// SQ_LDS_BANK_CONFLICT: 2.3 - 4.6 M cycles in the NX > 0 kernels, 0 in the plain ones).  Key bits (q3, q3, q1) instead make it
// (q3, q2, q1): eight slots.  q3 = bit 5 of the row (rows are 64 a + 16 (q >> 2) + 4 j + (q & 3)).
__device__ __forceinline__ int nswz_key(int row) { return (((row >> 5) & 1) * 6) | ((row >> 1) & 1); }

}  // namespace

// KF: K / 32 (MFMA K-steps), RES: a residual is added, NC: chunks of 64 output channels per workgroup (blockIdx.y
// selects the range: panels alone leave most CUs with one workgroup when M / 128 is close to the CU count)
// RF: 16-row fragments per wave -- 2 with 4 waves (two workgroups per CU), 1 with 8 waves (K = 512: X alone is 64
// registers per row fragment, and two 64 KB W buffers leave room for one workgroup per CU, so it brings its own 8 waves)
// NX > 0: the workgroup owns ALL N output channels of its 128 pixels (NC = N / 64, gridDim.y = 1) and also computes the next
// block's reducing 1x1 on them, Hn = relu(out Wn^T + bias_n) with Cn = 16 NX output channels (resnet.py:224-232 of block
// i + 1): after a chunk's epilogue the lane's packed bf16 outputs -- 16 consecutive channels of one pixel -- ARE the MFMA
// B fragments of that product (k-slot e of lane group g <-> channel 16 g + e, + 8 for the second MFMA); Wn streams
// through the LDS in 64-input-channel chunks beside W.  The next block then never reads this block's output for its conv1.
// HT: bf16_t or f16_t -- the operands move as raw 16-bit words; only the MFMA opcode and the pack / unpack of outputs and residual differ
template <typename HT, int KF, bool RES, int NC, int RF, int NX>
__global__ __launch_bounds__(64 * (X_BM / 16 / RF), RF == 2 ? 2 : 1) void expand_res_kernel(const GemmParams p) {
  constexpr int K = KF * 32, FJ = X_FJ, BN = X_BN, BM = X_BM, NT = 64 * (X_BM / 16 / RF);
  constexpr int CHUNK = BN * K * 2;               // bytes of one W chunk
  constexpr int SLOTS = CHUNK / 16 / NT;          // DMA pieces per thread per chunk
  static_assert(KF % 2 == 0 && CHUNK % (16 * NT) == 0, "shape");
  constexpr int NV = FJ / 2;                      // 16-byte pieces of a lane's 16 channels (bf16)
  constexpr int CN = NX * 16, NCHUNK = CN * 128;  // next conv: output channels, bytes of one Wn chunk ([Cn][64 inputs])
  constexpr int NSLOTS = NX > 0 ? NCHUNK / 16 / NT : 0;
  static_assert(NX == 0 || (NX % 4 == 0 && NCHUNK % (16 * NT) == 0), "next-conv shape");
  constexpr int NRES = RES ? RF * NV : 0, NST = RF * NV;  // residual loads / output stores per thread and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM + wave * 16 * RF;
  const int cb = blockIdx.y * NC;                 // first chunk of this workgroup

  // ---- W chunk loader: slot s = i * NT + tid -> (ks, row, pos); LDS image [ks][row][128 B], linear destination ----
  auto dma_chunk = [&](int c, char* buf) {
#ifdef HVR_DBG_X_NODMA
    if (c != cb) return;
#endif
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int s = i * NT + tid, ks = s / (BN * 8), rem = s - ks * (BN * 8), row = rem >> 3, pos = rem & 7;
      const int ch = pos ^ swz_key(row);
      const char* src = (const char*)p.B + (((long)(c * BN + row) * p.ldb) * 2 + ks * 128 + ch * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    if constexpr (NX > 0) {
      // Wn chunk c: rows = the Cn output channels, 128 B (this chunk's 64 input channels) each; same row-keyed slot swizzle
      char* nbuf = smem + 2 * CHUNK + NC * BN * 4 + (buf == smem ? 0 : NCHUNK);
#pragma unroll
      for (int i = 0; i < NSLOTS; ++i) {
        const int s = i * NT + tid, row = s >> 3, pos = s & 7;
        const int ch = pos ^ nswz_key(row);
        const char* src = (const char*)p.Wn + ((long)row * p.N * 2 + (long)c * 128 + ch * 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(nbuf + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };
  dma_chunk(cb, smem);

  // this workgroup's shifts -> LDS behind the two W buffers
  float* shl = reinterpret_cast<float*>(smem + 2 * CHUNK);
  for (int n = tid; n < NC * BN; n += NT) shl[n] = p.bias ? p.bias[cb * BN + n] : 0.f;

  // ---- X fragments: rows m0 + 16 i + q, k = 32 kf + 8 g .. + 8 ----
  xu32x4 x[RF][KF];
  int mrow[RF];
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const int m = m0 + i * 16 + q;
    mrow[i] = m < p.M ? m : p.M - 1;
    const char* xr = (const char*)p.A + (long)mrow[i] * p.lda * 2 + g * 16;
    // second K segment (p.s2 > 0: the block's projection shortcut): the row's pixel of the block INPUT, sampled at stride s2
    const char* x2r = xr;
    const int kf1 = p.s2 > 0 ? p.K1 / 32 : KF;
    if (p.s2 > 0) {
      const int ox = mrow[i] % p.OW, t = mrow[i] / p.OW, oy = t % p.OH, b = t / p.OH;
      x2r = (const char*)p.A2 + (((long)b * p.H2 + oy * p.s2) * p.W2 + ox * p.s2) * (long)(K - p.K1) * 2 + g * 16;
    }
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
      x[i][kf] = *reinterpret_cast<const xu32x4*>(kf < kf1 ? xr + kf * 64 : x2r + (kf - kf1) * 64);
  }

  // residual rows of a chunk: lane's channels n = c BN + 16 g .. + 16 of rows mrow[0], mrow[1]
  // ring of RD chunks' residual rows, fetched RD - 1 chunks ahead (K = 256 has registers for one chunk ahead only)
  constexpr int RD = 3, AHEAD = RD - 1;
  xu32x4 res[RD][RF][NV];  // [ring slot][row fragment][piece]
  auto load_res = [&](int c, xu32x4 (&r)[RF][NV]) {
#ifdef HVR_DBG_X_NORES
    if constexpr (false) {
#else
    if constexpr (RES) {
#endif
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const char* rr = (const char*)p.resid + ((long)mrow[i] * p.ldr + c * BN + g * 4 * FJ) * 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) r[i][v] = *reinterpret_cast<const xu32x4*>(rr + v * 16);
      }
    }
  };
  __syncthreads();  // the shifts are in the LDS for every wave (the chunk barriers below are bare s_barrier)
  load_res(cb, res[0]);
  if constexpr (AHEAD > 1 && NC > 1) load_res(cb + 1, res[1]);

  // fragment read address of this lane: W row 16 (q >> 2) + 4 j + (q & 3), chunk (kk * 4 + g) ^ key
  const int key = ((q >> 2) << 1) | ((q >> 1) & 1);
  const uint32_t w_lane = x_lds_off(smem) + ((q >> 2) * 4 * FJ + (q & 3)) * 128 + ((g ^ key) << 4);
  const uint32_t sh_lane = x_lds_off(shl) + g * 4 * FJ * 4;

  f32x4 hacc[NX > 0 ? RF : 1][NX > 0 ? NX : 1];
  if constexpr (NX > 0) {
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) hacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Wn fragment of this lane: row 64 (jo >> 2) + 16 (q >> 2) + 4 (jo & 3) + (q & 3), 16-byte slot (2 g + v) ^ key
  const uint32_t n_lane = x_lds_off(smem) + 2 * CHUNK + NC * BN * 4 + ((q >> 2) * 16 + (q & 3)) * 128;
  const int nkey = ((q >> 3) * 6) | ((q >> 1) & 1);   // nswz_key of that row

  static_for<NC>([&](auto U) {
    constexpr int u = decltype(U)::value;
    const int c = cb + u;
    xu32x4 (&rcur)[RF][NV] = res[u % RD];
    // Top of chunk u.  In flight behind DMA(u), oldest first: res(u + 1) (if there is one) and the previous chunk's NST
    // stores; DMA(u) must have landed.  Vector memory operations retire in order and every wave issues every one of
    // them (rows past M are clamped, not predicated: see the stores), so the count is exact and branch-free -- a
    // conditional wait would leave the compiler's bookkeeping with a path on which nothing was waited for.  Behind the
    // barrier every wave's DMA pieces are in the LDS and every wave is done reading the buffer the next DMA overwrites.
    // (with one chunk of read-ahead res(u) itself sits between DMA(u) and the stores, and is waited for here too: the
    // compiler then knows it has landed and puts no wait of its own -- a vmcnt(0), while a DMA is pending -- in the epilogue)
#if defined(HVR_DBG_X_NORES) || defined(HVR_DBG_X_NOSTORE)
    constexpr int behind = 0;
#else
    constexpr int behind = (AHEAD > 1 && u + 1 < NC ? NRES : 0) + (u > 0 ? NST : 0);
#endif
    x_wait_vm<behind>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (u + 1 < NC) dma_chunk(c + 1, smem + ((u + 1) & 1) * CHUNK);
    if constexpr (u + AHEAD < NC) load_res(c + AHEAD, res[(u + AHEAD) % RD]);
    __builtin_amdgcn_sched_barrier(0);
    // two base addresses per buffer (the 32-channel half kk = 1 is the same address with bit 6 flipped); everything else
    // is an immediate offset
    const uint32_t base0 = w_lane + (uint32_t)(u & 1) * CHUNK, base1 = base0 ^ 64u;
    const int nb = c * BN + g * 4 * FJ;
    f32x4 acc[RF][FJ];
    // software pipeline over the KF MFMA K-steps in half-steps of FJ / 2 fragments: while one half's MFMAs run, the
    // other half's fragments (and then the next step's) are on their way -- one step's worth of fragment registers
    constexpr int H = FJ / 2;
    uint4 wf[2][H];  // [half][fragment]
    auto read_half = [&](auto T, auto HF) {
      constexpr int t = decltype(T)::value, hf = decltype(HF)::value;
      static_for<H>([&](auto J) {
        constexpr int j = decltype(J)::value;
        wf[hf][j] = x_lds_read128<(t >> 1) * (BN * 128) + (hf * H + j) * 4 * 128>((t & 1) ? base1 : base0);
      });
    };
    read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    static_for<KF>([&](auto T) {
      constexpr int t = decltype(T)::value;
      static_for<2>([&](auto HF) {
        constexpr int hf = decltype(HF)::value;
        // outstanding here: this half, then the other half (of this step for hf = 0, of the next for hf = 1)
#ifdef HVR_DBG_X_NOLDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        if constexpr (t + 1 < KF || hf == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(H) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < H; ++j)
#pragma unroll
          for (int i = 0; i < RF; ++i) {
            const f32x4 cin = t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][hf * H + j];
#ifdef HVR_DBG_X_NOMMA
            if (t == 0) acc[i][hf * H + j] = __builtin_bit_cast(f32x4, wf[hf][j]); else acc[i][hf * H + j][0] += __uint_as_float(wf[hf][j].x ^ x[i][t][0]);
#else
            acc[i][hf * H + j] = mfma_half<HT>(wf[hf][j], __builtin_bit_cast(uint4, x[i][t]), cin);
#endif
          }
        __builtin_amdgcn_sched_barrier(0);
#ifdef HVR_DBG_X_NOLDS
        if constexpr (false) read_half(std::integral_constant<int, t + 1>{}, HF);
#else
        if constexpr (t + 1 < KF) read_half(std::integral_constant<int, t + 1>{}, HF);
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // ---- epilogue: lane holds out[m0 + 16 i + q][c BN + 16 g + 4 j + r] = acc[i][j][r] ----
    uint4 yv[NX > 0 ? RF : 1][NV];  // (NX > 0) the chunk's packed outputs: the next conv's B fragments
    uint4 sh[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) sh[j] = x_lds_read128<0>(sh_lane + (u * BN + 4 * j) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      // a lane whose row is past M was given row M - 1's X and residual: it holds row M - 1's outputs bit for bit and
      // stores them where row M - 1 goes (same bytes from several lanes), so no store is ever skipped
      char* dst = (char*)p.C + ((long)mrow[i] * p.ldc + nb) * 2;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        uint32_t o[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {  // two channels per 32-bit word: channel 8 v + 2 h (+1) = fragment 2 v + h / 2, r = 2 (h & 1) (+1)
          const int j = 2 * v + (h >> 1), r = 2 * (h & 1);
          const uint32_t shw[4] = {sh[j].x, sh[j].y, sh[j].z, sh[j].w};
          float lo = acc[i][j][r] + __uint_as_float(shw[r]);
          float hi = acc[i][j][r + 1] + __uint_as_float(shw[r + 1]);
          if constexpr (RES) {
            float rl, rh;
            unpack2<HT>(rcur[i][v][h], rl, rh);
            lo += rl;
            hi += rh;
          }
          if (p.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          o[h] = pack2<HT>(lo, hi);
        }
#ifdef HVR_DBG_X_NOSTORE
        if (o[0] == 0x12345678u && o[3] == 0x9abcdef0u)
#endif
        *reinterpret_cast<uint4*>(dst + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        if constexpr (NX > 0) yv[i][v] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NX > 0) {
      // next block's conv1 on this chunk: 2 K = 32 MFMAs (v = 0 / 1) per (row fragment, output fragment).  The Wn fragment pairs stream
      // through a ring of ND pairs requested ND output fragments ahead (round 6: with one read pair waited for per two MFMAs the loop
      // was a chain of LDS latencies -- 16 x ~150 cycles per chunk in the NX = 16 instance); LDS returns in order, the waits are counted
      const uint32_t nb0 = n_lane + (uint32_t)(u & 1) * NCHUNK;
      constexpr int ND = NX < 4 ? NX : 4;
      uint4 wq[ND][2];
      auto nread = [&](auto JO) {
        constexpr int jo = decltype(JO)::value;
        constexpr int roff = (64 * (jo >> 2) + 4 * (jo & 3)) * 128;
        wq[jo % ND][0] = x_lds_read128<roff>(nb0 + ((((g << 1) | 0) ^ nkey) << 4));
        wq[jo % ND][1] = x_lds_read128<roff>(nb0 + ((((g << 1) | 1) ^ nkey) << 4));
      };
      static_for<ND>([&](auto JO) { nread(JO); });
      static_for<NX>([&](auto JO) {
        constexpr int jo = decltype(JO)::value;
        constexpr int ahead = (NX - 1 - jo) < (ND - 1) ? (NX - 1 - jo) : (ND - 1);   // pairs requested behind this one
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ahead) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          hacc[i][jo] = mfma_half<HT>(wq[jo % ND][0], __builtin_bit_cast(uint4, yv[i][0]), hacc[i][jo]);
          hacc[i][jo] = mfma_half<HT>(wq[jo % ND][1], __builtin_bit_cast(uint4, yv[i][1]), hacc[i][jo]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (jo + ND < NX) nread(std::integral_constant<int, jo + ND>{});
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  if constexpr (NX > 0) {
    // ---- next conv epilogue: lane holds Hn[m0 + 16 i + q][64 hg + 16 g + 4 j + r] = hacc[i][4 hg + j][r] ----
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      char* dst = (char*)p.Hn + ((long)mrow[i] * CN + g * 16) * 2;
#pragma unroll
      for (int hg = 0; hg < NX / 4; ++hg) {
        const float* bn = p.bias_n + hg * 64 + g * 16;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          uint32_t o[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int j = 2 * v + (h >> 1), r = 2 * (h & 1);
            const float lo = fmaxf(hacc[i][4 * hg + j][r] + bn[4 * j + r], 0.f);
            const float hi = fmaxf(hacc[i][4 * hg + j][r + 1] + bn[4 * j + r + 1], 0.f);
            o[h] = pack2<HT>(lo, hi);
          }
          *reinterpret_cast<uint4*>(dst + hg * 128 + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
}

// chunks per workgroup: 8 when that still gives the chip two workgroups per CU, else 4, else 2
static int expand_nc(int M, int N) {
  const int panels = (M + X_BM - 1) / X_BM, nchunks = N / X_BN;
  // (16 = one workgroup per panel at N = 1024: measured SLOWER at layer 3 / 15 frames, 51 us against 42 -- 281 workgroups do
  // not keep enough loads in flight)
  if (nchunks % 8 == 0 && (long)panels * (nchunks / 8) >= 512) return 8;
  // (round 6: few panels -- one 600 x 1000 frame at stride 16 is 19 -- and four chunks per workgroup leave 76 workgroups walking a chain of
  // four chunks each; two chunks per workgroup halve the chain and double the workgroups: 13.3 -> 11.1 us per layer-3 call, the one-frame
  // graph 1.39 - 1.42 -> 1.34 ms; the pipelined stream loop is bound by the chip's total work and does not move.  Same MFMA order per
  // output: bit-identical)
  if (nchunks % 4 == 0 && (long)panels * (nchunks / 4) >= 256) return 4;
  if (nchunks % 2 == 0) return 2;
  return 0;
}

// The panel kernel applies when the product is a plain bf16 GEMM with a short K, whole 16-byte rows, and an output of
// an even number of 64-channel chunks (the expand convs of layers 1-3 and res5: K = 64 / 128 / 256 / 512, N = 4 K), or the
// first block of a stage with its projection shortcut as a second K segment (K = 64 + 64, 128 + 256: hvr_bottleneck_tail).
bool expand_supported(const GemmParams& p) {
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16) || p.conv || p.out_f32 || p.ksplit_steps > 0) return false;
  if (!(p.K == 64 || p.K == 128 || p.K == 256 || p.K == 512 || (p.K == 384 && p.s2 > 0))) return false;
  if (p.s2 > 0 && (p.K1 % 32 || p.K1 <= 0 || p.K1 >= p.K || (p.K - p.K1) % 8 || (reinterpret_cast<uintptr_t>(p.A2) & 15) || p.resid)) return false;
  if (p.N % X_BN || p.M < X_BM || expand_nc(p.M, p.N) == 0) return false;
  if (p.lda % 8 || p.ldb % 8 || p.ldc % 8 || (p.resid && p.ldr % 8)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.resid);
  if (al & 15) return false;
  if ((long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  return true;
}

template <typename T, int KF, bool RES, int NC, int NX = 0>
static hipError_t launch_expand_nc(const GemmParams& p, hipStream_t stream) {
  // (NX = 16 -- layer 3's closing 1x1 with the next block's 1024 -> 256 conv1: 64 next-conv accumulator registers per row fragment -- takes
  // the 8-wave form too: one row fragment per wave, one workgroup per CU)
  constexpr int RF = (KF > 8 || NX >= 16) ? 1 : 2;
  constexpr int lds = 2 * X_BN * KF * 32 * 2 + NC * X_BN * 4 + 2 * NX * 16 * 128;  // two W chunks + shifts (+ two Wn chunks)
  static_assert(lds <= (RF == 2 ? 80 : 160) * 1024, "two workgroups per CU (one with 8 waves at K = 512)");
  auto kern = expand_res_kernel<T, KF, RES, NC, RF, NX>;
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(kern, dim3((p.M + X_BM - 1) / X_BM, p.N / X_BN / NC), dim3(64 * (X_BM / 16 / RF)), lds, stream, p);
  return hipGetLastError();
}

template <typename T, int KF, bool RES>
static hipError_t launch_expand_res(const GemmParams& p, hipStream_t stream) {
  switch (expand_nc(p.M, p.N)) {
    case 16:
      if constexpr (KF == 8) return launch_expand_nc<T, KF, RES, 16>(p, stream);
      else return hipErrorInvalidValue;
    case 8: return launch_expand_nc<T, KF, RES, 8>(p, stream);
    case 4: return launch_expand_nc<T, KF, RES, 4>(p, stream);
    case 2: return launch_expand_nc<T, KF, RES, 2>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

template <typename T, int KF>
static hipError_t launch_expand(const GemmParams& p, hipStream_t stream) {
  return p.resid ? launch_expand_res<T, KF, true>(p, stream) : launch_expand_res<T, KF, false>(p, stream);
}

// with the next block's conv1: the workgroup owns all N channels (NC = N / 64); stage 1 (N = 256, Cn = 64) and stage 2
// (N = 512, Cn = 128) of the R-101, with or without the projection-shortcut segment
template <typename T>
static hipError_t run_expand_next(const GemmParams& p, hipStream_t stream) {
  if (p.N == 256 && p.Cn == 64) {
    if (p.K == 64 && p.resid) return launch_expand_nc<T, 2, true, 4, 4>(p, stream);
    if (p.K == 128 && !p.resid) return launch_expand_nc<T, 4, false, 4, 4>(p, stream);
  }
  if (p.N == 512 && p.Cn == 128) {
    if (p.K == 128 && p.resid) return launch_expand_nc<T, 4, true, 8, 8>(p, stream);
    if (p.K == 384 && !p.resid) return launch_expand_nc<T, 12, false, 8, 8>(p, stream);
  }
  // layer 3's identity blocks (round 6): the 1024-channel block output is written once and never read back for the next block's conv1
  // (92 MB of the block pair's 257 MB per window: resnet.py:224-232 of block i + 1 on the registers of resnet.py:248-264 of block i)
  // (a WAVE-PAIR form -- 4 row groups x 2 channel halves, 32 pixels per wave, every W / Wn fragment read feeding two MFMAs, the halves'
  // output pieces exchanged through the LDS: half the fragment reads, one more barrier per chunk -- was built and measured SLOWER, 272 us
  // against 232 - 240 for a 60-frame block: the kernel is bound by the 1.15 GB of W3 / Wn chunks every call re-streams from the L2s into
  // 1 123 panels' LDS (4.9 TB/s of LDS-DMA beside 3.2 TB/s of HBM traffic), not by its LDS reads; profiles/r06_l3_fused.txt; removed)
  if (p.N == 1024 && p.Cn == 256 && p.K == 256 && p.resid) return launch_expand_nc<T, 8, true, 16, 16>(p, stream);
  return hipErrorInvalidValue;
}

bool expand_next_supported(const GemmParams& p) {
  if (!p.Wn || !p.Hn || !p.bias_n || !expand_supported(p)) return false;
  if ((reinterpret_cast<uintptr_t>(p.Wn) | reinterpret_cast<uintptr_t>(p.Hn) | reinterpret_cast<uintptr_t>(p.bias_n)) & 15) return false;
  if (p.N == 256 && p.Cn == 64) return (p.K == 64 && p.resid) || (p.K == 128 && !p.resid && p.s2 > 0);
  if (p.N == 512 && p.Cn == 128) return (p.K == 128 && p.resid) || (p.K == 384 && !p.resid && p.s2 > 0);
  // (one 8-wave workgroup per CU and 128-pixel panels: it needs at least two rounds of panels to fill the chip -- 60 frames of 38 x 63:
  // 240 us against 260 - 298 for the two launches; 15 frames = 281 panels: 86 against 67 - 78, profiles/r06_l3_fused.txt)
  if (p.N == 1024 && p.Cn == 256) return p.K == 256 && p.resid && p.s2 == 0 && p.M >= 512 * X_BM;
  return false;
}

template <typename T>
static hipError_t run_expand_t(const GemmParams& p, hipStream_t stream) {
  if (p.Wn) return run_expand_next<T>(p, stream);
  switch (p.K) {
    case 64: return launch_expand<T, 2>(p, stream);
    case 128: return launch_expand<T, 4>(p, stream);
    case 256: return launch_expand<T, 8>(p, stream);
    case 384: return launch_expand<T, 12>(p, stream);
    case 512: return launch_expand<T, 16>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t run_expand(const GemmParams& p, hipStream_t stream) {
  return p.dtype == DT_F16 ? run_expand_t<f16_t>(p, stream) : run_expand_t<bf16_t>(p, stream);
}

}  // namespace hvr
