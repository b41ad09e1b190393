// Channel-expanding 1x1 convolution with the block's residual on SPLIT-HALF operands (HVR_F16S, common.h: f16s_t), gfx950:
//     out = act(alpha X W^T + beta shift + R)           X [M][K] (K = 64 / 128), W [N][K] with N = 4 K, R / out [M][N]
// and optionally, in the same pass, the next block's reducing 1x1 on the output while it is still in registers:
//     hn  = relu(alpha out Wn^T + beta shift_n)         Wn [Cn][N], Cn = 64
// Replaces conv3 + bn3 + `out += identity` + ReLU of mmdet/models/backbones/resnet.py:248-264 (and conv1 + bn1 + ReLU of the next
// Bottleneck, :224-232) in layers 1 and 2 when the detector computes in the split-half mode -- the mode whose detections carry the
// reference's f32 tolerance.
//
// expand.hip's row-panel design (read its header first), with the format's consequences:
//   * a logical element is 4 bytes: a row of K elements is K / 32 groups of [32 hi halves | 32 lo halves] (128 bytes), so a
//     workgroup's X fragments are TWO register sets (the hi and the lo plane of every 32-wide K-step), still loaded once, straight
//     from global memory, and a W chunk of 64 output channels is 64 x K x 4 bytes in the LDS (16 / 32 KB, double-buffered);
//   * a K-step is three half MFMAs per fragment pair -- W_hi X_hi, W_hi X_lo when the chunk's hi fragments have landed, W_lo X_hi
//     when the lo ones have (lo x lo is 2^-22 of the product and dropped) -- from ONE LDS image: the W fragments of a plane are read
//     once and meet both X planes;
//   * the lane that ends a chunk with 16 consecutive channels of one pixel holds half of a 32-channel group: 32 bytes of the hi plane
//     and the 32 bytes of the lo plane 64 bytes further -- four 16-byte residual loads and output stores per row fragment and chunk;
//   * the epilogue is the tile engine's (gemm_tile.h): (alpha acc + beta shift) + merged f32 residual, ReLU, one rounding into the
//     [hi | lo] pair (split2: saturating, see common.h);
//   * NX > 0: a chunk's packed hi / lo outputs ARE the next product's B fragments of both planes (k-slot e of lane group g <->
//     channel 16 g + e of the chunk, + 8 for the second MFMA), three MFMAs per fragment pair again; Wn streams through the LDS in
//     64-input-channel chunks of 256-byte rows beside W.
#include <cstdlib>
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

__device__ __forceinline__ uint32_t xs_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 xs_lds_read128(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// s_waitcnt vmcnt(N) through the builtin (the compiler's wait-count pass reads it: see expand.hip)
template <int N> __device__ __forceinline__ void xs_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ f32x4 xs_mma(const uint4& w, const uint4& x, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
}

typedef uint32_t xsu32x4 __attribute__((ext_vector_type(4)));

constexpr int XS_FJ = 4, XS_BN = 16 * XS_FJ, XS_BM = 128, XS_NT = 256, XS_RF = 2;

// W chunk image [k-step][row][128 B]: the 16-byte piece at position pos of row `row` holds global chunk pos ^ key(row) -- expand.hip's
// key: the 16 lanes of a ds_read_b128 group read rows {16 a + 4 j + r}, (2 a + (r >> 1)) spreads them over 8 slots per row parity
__device__ __forceinline__ int xs_key(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }
// Wn chunk image [row][256 B] (two [hi | lo] groups = 64 input channels): piece position pos holds global piece pos ^ key(row), key =
// the q of the lane that reads the row (rows are 64 a + 16 (q >> 2) + 4 b + (q & 3)): a ds_read_b128 lane group holds every q once,
// with the lane-group bit that separates its two halves in bit 1 of the piece index -- q ^ {0, 2} is a bijection on 0..15
__device__ __forceinline__ int xs_nkey(int row) { return (((row >> 4) & 3) << 2) | (row & 3); }

}  // namespace

// KF: K / 32; RES: a residual is added; NC: chunks of 64 output channels per workgroup (blockIdx.y selects the range);
// NX: 16-channel fragments of the next conv's output (0 = off; then NC = N / 64 and gridDim.y = 1)
template <int KF, bool RES, int NC, int NX>
__global__ __launch_bounds__(XS_NT, 2) void expand_split_kernel(const GemmParams p) {
  constexpr int K = KF * 32, FJ = XS_FJ, BN = XS_BN, BM = XS_BM, NT = XS_NT, RF = XS_RF;
  constexpr int CHUNK = BN * K * 4;               // bytes of one W chunk: [KF][64 rows][128 B]
  constexpr int SLOTS = CHUNK / 16 / NT;          // DMA pieces per thread per chunk
  constexpr int NP = 4;                           // 16-byte pieces of a lane's 16 channels: hi 0, hi 1, lo 0, lo 1
  constexpr int CN = NX * 16, NCHUNK = CN * 256;  // next conv: output channels, bytes of one Wn chunk ([Cn][64 inputs] = 256-byte rows)
  constexpr int NSLOTS = NX > 0 ? NCHUNK / 16 / NT : 0;
  static_assert(CHUNK % (16 * NT) == 0 && (NX == 0 || (NX % 4 == 0 && NCHUNK % (16 * NT) == 0)), "shape");
  constexpr int NRES = RES ? RF * NP : 0, NST = RF * NP;
  // residual ring: RD chunks' rows, fetched RD - 1 chunks ahead (K = 128's X planes and the fused next conv need the registers: one
  // chunk ahead there)
  constexpr int RD = (NX > 0 || KF >= 4) ? 2 : 3, AHEAD = RD - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM + wave * 16 * RF;
  const int cb = blockIdx.y * NC;
  char* const nbase = smem + 2 * CHUNK + NC * BN * 4;   // the two Wn chunk buffers

  auto dma_chunk = [&](int c, int par) {
    char* buf = smem + par * CHUNK;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int s = i * NT + tid, kf = s / (BN * 8), rem = s - kf * (BN * 8), row = rem >> 3, pos = rem & 7;
      const int ch = pos ^ xs_key(row);
      const char* src = (const char*)p.B + (((long)(c * BN + row) * p.ldb) * 4 + kf * 128 + ch * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    if constexpr (NX > 0) {
      char* nbuf = nbase + par * NCHUNK;
#pragma unroll
      for (int i = 0; i < NSLOTS; ++i) {
        const int s = i * NT + tid, row = s >> 4, pos = s & 15;
        const int ch = pos ^ xs_nkey(row);
        const char* src = (const char*)p.Wn + ((long)row * p.N * 4 + (long)c * 256 + ch * 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(nbuf + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };
  dma_chunk(cb, 0);

  // this workgroup's shifts (x beta) -> LDS behind the two W buffers
  float* shl = reinterpret_cast<float*>(smem + 2 * CHUNK);
  const float beta = p.beta != 0.f ? p.beta : 1.f;
  for (int n = tid; n < NC * BN; n += NT) shl[n] = p.bias ? p.bias[cb * BN + n] * beta : 0.f;

  // ---- X fragments, both planes: rows m0 + 16 i + q, k = 32 kf + 8 g .. + 8 ----
  xsu32x4 xh[RF][KF], xl[RF][KF];
  int mrow[RF];
#pragma unroll
  for (int i = 0; i < RF; ++i) {
    const int m = m0 + i * 16 + q;
    mrow[i] = m < p.M ? m : p.M - 1;
    const char* xr = (const char*)p.A + (long)mrow[i] * p.lda * 4 + g * 16;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) {
      xh[i][kf] = *reinterpret_cast<const xsu32x4*>(xr + kf * 128);
      xl[i][kf] = *reinterpret_cast<const xsu32x4*>(xr + kf * 128 + kSplitPlane);
    }
  }

  // byte offset of this lane's 16 channels inside an output / residual row, chunk 0: group (g >> 1), hi plane bytes 32 (g & 1)
  const int lane_col = (g >> 1) * 128 + (g & 1) * 32;
  xsu32x4 res[RD][RF][NP];
  auto load_res = [&](int c, xsu32x4 (&r)[RF][NP]) {
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const char* rr = (const char*)p.resid + (long)mrow[i] * p.ldr * 4 + c * 256 + lane_col;
        r[i][0] = *reinterpret_cast<const xsu32x4*>(rr);
        r[i][1] = *reinterpret_cast<const xsu32x4*>(rr + 16);
        r[i][2] = *reinterpret_cast<const xsu32x4*>(rr + kSplitPlane);
        r[i][3] = *reinterpret_cast<const xsu32x4*>(rr + kSplitPlane + 16);
      }
    }
  };
  __syncthreads();  // the shifts are in the LDS for every wave (the chunk barriers below are bare s_barrier)
  load_res(cb, res[0]);
  if constexpr (AHEAD > 1 && NC > 1) load_res(cb + 1, res[1]);

  // fragment read address of this lane: W row 16 (q >> 2) + 4 j + (q & 3), piece (plane * 4 + g) ^ key
  const int key = ((q >> 2) << 1) | ((q >> 1) & 1);
  const uint32_t w_lane = xs_lds_off(smem) + ((q >> 2) * 4 * FJ + (q & 3)) * 128 + ((g ^ key) << 4);
  const uint32_t sh_lane = xs_lds_off(shl) + g * 4 * FJ * 4;
  const float alpha = p.alpha != 0.f ? p.alpha : 1.f;

  f32x4 hacc[NX > 0 ? RF : 1][NX > 0 ? NX : 1];
  if constexpr (NX > 0) {
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) hacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Wn fragment of this lane: row 64 (jo >> 2) + 16 (q >> 2) + 4 (jo & 3) + (q & 3); piece (g >> 1) * 8 + plane * 4 + 2 (g & 1) + v, ^ q
  const uint32_t n_lane = xs_lds_off(nbase) + ((q >> 2) * 16 + (q & 3)) * 256;
  const int npiece = (g >> 1) * 8 + 2 * (g & 1);

  static_for<NC>([&](auto U) {
    constexpr int u = decltype(U)::value;
    const int c = cb + u;
    xsu32x4 (&rcur)[RF][NP] = res[u % RD];
    // top of chunk u: DMA(u) has landed; behind it, oldest first, may stay in flight: res(u + 1) (two chunks of read-ahead) and
    // the previous chunk's stores (expand.hip: exact, branch-free counts -- every wave issues every memory operation)
    constexpr int behind = (AHEAD > 1 && u + 1 < NC ? NRES : 0) + (u > 0 ? NST : 0);
    xs_wait_vm<behind>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (u + 1 < NC) dma_chunk(c + 1, (u + 1) & 1);
    if constexpr (u + AHEAD < NC) load_res(c + AHEAD, res[(u + AHEAD) % RD]);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t base0 = w_lane + (uint32_t)(u & 1) * CHUNK, base1 = base0 ^ 64u;
    f32x4 acc[RF][FJ];
    // pseudo-steps t = 2 kf + plane, each in two halves of FJ / 2 fragments: while one half's MFMAs run, the other half's fragments
    // (and then the next step's) are on their way
    constexpr int H = FJ / 2, TS = 2 * KF;
    uint4 wf[2][H];
    auto read_half = [&](auto T, auto HF) {
      constexpr int t = decltype(T)::value, hf = decltype(HF)::value;
      static_for<H>([&](auto J) {
        constexpr int j = decltype(J)::value;
        wf[hf][j] = xs_lds_read128<(t >> 1) * (BN * 128) + (hf * H + j) * 4 * 128>((t & 1) ? base1 : base0);
      });
    };
    read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    static_for<TS>([&](auto T) {
      constexpr int t = decltype(T)::value, kf = t >> 1, plane = t & 1;
      static_for<2>([&](auto HF) {
        constexpr int hf = decltype(HF)::value;
        if constexpr (t + 1 < TS || hf == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(H) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < H; ++j)
#pragma unroll
          for (int i = 0; i < RF; ++i) {
            const uint4 xhi = __builtin_bit_cast(uint4, xh[i][kf]);
            if constexpr (plane == 0) {   // W_hi: meets both X planes
              const f32x4 cin = t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][hf * H + j];
              acc[i][hf * H + j] = xs_mma(wf[hf][j], xhi, cin);
              acc[i][hf * H + j] = xs_mma(wf[hf][j], __builtin_bit_cast(uint4, xl[i][kf]), acc[i][hf * H + j]);
            } else {                      // W_lo x X_hi
              acc[i][hf * H + j] = xs_mma(wf[hf][j], xhi, acc[i][hf * H + j]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < TS) read_half(std::integral_constant<int, t + 1>{}, HF);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // ---- epilogue: lane holds out[m0 + 16 i + q][c 64 + 16 g + 4 j + r] = acc[i][j][r] ----
    uint4 yh[NX > 0 ? RF : 1][2], yl[NX > 0 ? RF : 1][2];   // (NX > 0) the chunk's packed outputs: the next conv's B fragments
    uint4 sh[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) sh[j] = xs_lds_read128<0>(sh_lane + (u * BN + 4 * j) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RF; ++i) {
      // (a lane whose row is past M holds row M - 1's outputs bit for bit and stores them where row M - 1 goes: no store is skipped)
      char* dst = (char*)p.C + (long)mrow[i] * p.ldc * 4 + c * 256 + lane_col;
      uint32_t oh[8], ol[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {   // word w: channels 2 w, 2 w + 1 of the lane's 16 = fragment w / 2, r = 2 (w & 1)
        const int j = w >> 1, r = 2 * (w & 1);
        const uint32_t shw[4] = {sh[j].x, sh[j].y, sh[j].z, sh[j].w};
        float a = acc[i][j][r] * alpha + __uint_as_float(shw[r]);
        float b = acc[i][j][r + 1] * alpha + __uint_as_float(shw[r + 1]);
        if constexpr (RES) {
          float ra, rb;
          merge2(rcur[i][w >> 2][w & 3], rcur[i][2 + (w >> 2)][w & 3], ra, rb);
          a += ra;
          b += rb;
        }
        if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        split2(a, b, oh[w], ol[w]);
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(oh[4], oh[5], oh[6], oh[7]);
      *reinterpret_cast<uint4*>(dst + kSplitPlane) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
      *reinterpret_cast<uint4*>(dst + kSplitPlane + 16) = make_uint4(ol[4], ol[5], ol[6], ol[7]);
      if constexpr (NX > 0) {
        yh[i][0] = make_uint4(oh[0], oh[1], oh[2], oh[3]); yh[i][1] = make_uint4(oh[4], oh[5], oh[6], oh[7]);
        yl[i][0] = make_uint4(ol[0], ol[1], ol[2], ol[3]); yl[i][1] = make_uint4(ol[4], ol[5], ol[6], ol[7]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NX > 0) {
      // next block's conv1 on this chunk's 64 channels: per (row fragment, output fragment) 2 x (hi hi, lo hi, hi lo) MFMAs
      const uint32_t nb0 = n_lane + (uint32_t)(u & 1) * NCHUNK;
      static_for<NX>([&](auto JO) {
        constexpr int jo = decltype(JO)::value;
        constexpr int roff = (64 * (jo >> 2) + 4 * (jo & 3)) * 256;
        const uint4 wh0 = xs_lds_read128<roff>(nb0 + (((npiece + 0) ^ q) << 4));
        const uint4 wh1 = xs_lds_read128<roff>(nb0 + (((npiece + 1) ^ q) << 4));
        const uint4 wl0 = xs_lds_read128<roff>(nb0 + (((npiece + 4) ^ q) << 4));
        const uint4 wl1 = xs_lds_read128<roff>(nb0 + (((npiece + 5) ^ q) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          hacc[i][jo] = xs_mma(wh0, yh[i][0], hacc[i][jo]);
          hacc[i][jo] = xs_mma(wh1, yh[i][1], hacc[i][jo]);
          hacc[i][jo] = xs_mma(wl0, yh[i][0], hacc[i][jo]);
          hacc[i][jo] = xs_mma(wl1, yh[i][1], hacc[i][jo]);
          hacc[i][jo] = xs_mma(wh0, yl[i][0], hacc[i][jo]);
          hacc[i][jo] = xs_mma(wh1, yl[i][1], hacc[i][jo]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  if constexpr (NX > 0) {
    // ---- next conv epilogue: lane holds hn[m0 + 16 i + q][64 hg + 16 g + 4 j + r] = hacc[i][4 hg + j][r] ----
#pragma unroll
    for (int i = 0; i < RF; ++i) {
#pragma unroll
      for (int hg = 0; hg < NX / 4; ++hg) {
        char* dst = (char*)p.Hn + (long)mrow[i] * CN * 4 + hg * 256 + lane_col;
        const float* bn = p.bias_n + hg * 64 + g * 16;
        uint32_t oh[8], ol[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const int j = w >> 1, r = 2 * (w & 1);
          const float a = fmaxf(hacc[i][4 * hg + j][r] * alpha + bn[4 * j + r] * beta, 0.f);
          const float b = fmaxf(hacc[i][4 * hg + j][r + 1] * alpha + bn[4 * j + r + 1] * beta, 0.f);
          split2(a, b, oh[w], ol[w]);
        }
        *reinterpret_cast<uint4*>(dst) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        *reinterpret_cast<uint4*>(dst + 16) = make_uint4(oh[4], oh[5], oh[6], oh[7]);
        *reinterpret_cast<uint4*>(dst + kSplitPlane) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
        *reinterpret_cast<uint4*>(dst + kSplitPlane + 16) = make_uint4(ol[4], ol[5], ol[6], ol[7]);
      }
    }
  }
}

// chunks per workgroup (expand.hip's rule): 8 when that still gives the chip two workgroups per CU, else 4, else 2
static int expand_split_nc(int M, int N) {
  const int panels = (M + XS_BM - 1) / XS_BM, nchunks = N / XS_BN;
  if (nchunks % 8 == 0 && (long)panels * (nchunks / 8) >= 512) return 8;
  if (nchunks % 4 == 0) return 4;
  if (nchunks % 2 == 0) return 2;
  return 0;
}

// split-half operands, plain 1x1 product with K = 64 / 128, whole [hi | lo] groups everywhere, 128-byte aligned bases
bool expand_split_supported(const GemmParams& p) {
  if (p.dtype != DT_F16S || p.conv || p.out_f32 || p.ksplit_steps > 0 || p.s2 > 0) return false;
  if (!(p.K == 64 || p.K == 128)) return false;
  if (p.N % XS_BN || p.M < XS_BM || expand_split_nc(p.M, p.N) == 0) return false;
  if (p.lda % 32 || p.ldb % 32 || p.ldc % 32 || (p.resid && p.ldr % 32)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.resid);
  if (al & 127) return false;
  if (reinterpret_cast<uintptr_t>(p.bias) & 3) return false;
  if ((long)p.N * p.ldb * 4 >= (1L << 31)) return false;
  return true;
}

// with the next block's conv1: stage 1 of the R-101 (N = 256, Cn = 64, K = 64, identity residual or the projection as the residual)
bool expand_split_next_supported(const GemmParams& p) {
  if (!p.Wn || !p.Hn || !p.bias_n || !p.resid || !expand_split_supported(p)) return false;
  if ((reinterpret_cast<uintptr_t>(p.Wn) | reinterpret_cast<uintptr_t>(p.Hn)) & 127) return false;
  if (reinterpret_cast<uintptr_t>(p.bias_n) & 3) return false;
  return p.N == 256 && p.Cn == 64 && p.K == 64;
}

template <int KF, bool RES, int NC, int NX = 0>
static hipError_t launch_expand_split(const GemmParams& p, hipStream_t stream) {
  constexpr int lds = 2 * XS_BN * KF * 32 * 4 + NC * XS_BN * 4 + 2 * NX * 16 * 256;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  auto kern = expand_split_kernel<KF, RES, NC, NX>;
  static std::atomic<unsigned> attr_set{0};   // (the attribute is per device)
  per_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(kern, dim3((p.M + XS_BM - 1) / XS_BM, p.N / XS_BN / NC), dim3(XS_NT), lds, stream, p);
  return hipGetLastError();
}

template <int KF, bool RES>
static hipError_t launch_expand_split_nc(const GemmParams& p, hipStream_t stream) {
  switch (expand_split_nc(p.M, p.N)) {
    case 8: return launch_expand_split<KF, RES, 8>(p, stream);
    case 4: return launch_expand_split<KF, RES, 4>(p, stream);
    case 2: return launch_expand_split<KF, RES, 2>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t run_expand_split(const GemmParams& p, hipStream_t stream) {
  if (p.Wn) {
    if (expand_split_next_supported(p)) return launch_expand_split<2, true, 4, 4>(p, stream);
    return hipErrorInvalidValue;
  }
  if (p.K == 64) return p.resid ? launch_expand_split_nc<2, true>(p, stream) : launch_expand_split_nc<2, false>(p, stream);
  if (p.K == 128) return p.resid ? launch_expand_split_nc<4, true>(p, stream) : launch_expand_split_nc<4, false>(p, stream);
  return hipErrorInvalidValue;
}

}  // namespace hvr
