// Channel-expanding 1x1 convolution with the block's residual:  out = act(X W^T + shift + R)   (bf16, gfx950)
//   X [M][K] (K = 64 .. 512), W [N][K] with N = 4 K in a Bottleneck, R / out [M][N].
// Replaces conv3 + bn3 + `out += identity` + ReLU of mmdet/models/backbones/resnet.py:248-264 (and the same tail of the
// res5 blocks, shared_heads/res_layer.py:67-74) -- 33 launches per frame batch.
//
// Why not the tile engine (gemm.hip): this product is HBM-bound (l3: 18.8 GF over 165 MB; the engine's output tiles
// run 53 us = 3.1 TB/s where an elementwise add over the same tensors streams at 6-7 TB/s on this box).  A K of 1-8
// K-steps leaves an output tile nothing to hide its prologue and its residual / store epilogue under, and every tile
// re-stages the X panel it shares with its row neighbours.  Here a workgroup owns a PANEL of 128 rows for all N:
//   * its X fragments are loaded once, straight from global into registers (the MFMA "B" operand: 16 B per lane),
//     and stay there -- X never touches the LDS;
//   * W streams through a double-buffered LDS chunk of BN output channels x K (global_load_lds, XOR-swizzled image),
//     the next chunk's DMA and the next chunk's residual rows are in flight under this chunk's MFMAs and stores;
//   * the MFMA row index is permuted (fragment j, row 4g + r <-> channel 16g' ...) so that a lane ends a chunk holding
//     4 FJ CONSECUTIVE channels of one pixel: residual loads and output stores are 16 B per lane without an LDS stage;
//   * two workgroups (4 waves each) per CU: one streams while the other computes.
#include <cstdlib>
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

__device__ __forceinline__ uint32_t x_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ uint4 x_lds_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

// 16-byte chunk position of global chunk c of W row `row` (row inside the BN-channel LDS chunk): the 16 lanes of a
// ds_read_b128 group read rows {a * 4FJ + 4j + r}: a = 0..3 (two of them per lane-group half), r = 0..3 -- the key
// (2a + (r >> 1)) makes the 8 lanes of either row parity land on 8 distinct 16-byte slots of the 256-byte bank row.
template <int FJ> __device__ __forceinline__ int swz_key(int row) { return (((row / (4 * FJ)) & 3) << 1) | ((row >> 1) & 1); }

}  // namespace

// Loads the compiler does not track (it would drain the whole DMA queue in front of their first use): the kernel waits
// for them with hand-counted s_waitcnt and marks the point with x_landed().  Vector memory operations retire in order.
typedef uint32_t xu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ xu32x4 x_load128_untracked(const void* p) {
  xu32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void x_landed(xu32x4& v) { asm volatile("" : "+v"(v)); }
template <int N> __device__ __forceinline__ void x_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KF: K / 32 (MFMA K-steps), FJ: 16-channel fragments per chunk (BN = 16 FJ), RES: a residual is added
template <int KF, int FJ, bool RES>
__global__ __launch_bounds__(256, 2) void expand_res_kernel(const GemmParams p) {
  constexpr int K = KF * 32, BN = 16 * FJ, BM = 128, NT = 256;
  constexpr int CHUNK = BN * K * 2;               // bytes of one W chunk
  constexpr int SLOTS = CHUNK / 16 / NT;          // DMA pieces per thread per chunk
  static_assert(KF % 2 == 0 && CHUNK % (16 * NT) == 0, "shape");
  constexpr int NV = FJ / 2;                      // 16-byte pieces of a lane's 4 FJ channels (bf16)
  static_assert(FJ == 2 || FJ == 4, "a lane's channels must be whole 16-byte pieces");
  constexpr int NRES = RES ? 2 * NV : 0, NST = 2 * NV;  // residual loads / output stores per thread and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM + wave * 32;
  // blockIdx.y splits the chunk loop (panels alone leave most CUs with one workgroup when M / 128 is close to the CU
  // count: nothing to overlap a workgroup's waits with); X is then read once per split, from L2
  const int nchunks_all = p.N / BN;
  const int cb = (int)((long)blockIdx.y * nchunks_all / gridDim.y), ce = (int)((long)(blockIdx.y + 1) * nchunks_all / gridDim.y);
  const bool wave_full = m0 + 32 <= p.M;  // every store of this wave issues: the counted wait at a chunk's top relies on it

  // ---- W chunk loader: slot s = i * NT + tid -> (ks, row, pos); LDS image [ks][row][128 B], linear destination ----
  auto dma_chunk = [&](int c, char* buf) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int s = i * NT + tid, ks = s / (BN * 8), rem = s - ks * (BN * 8), row = rem >> 3, pos = rem & 7;
      const int ch = pos ^ swz_key<FJ>(row);
      const char* src = (const char*)p.B + (((long)(c * BN + row) * p.ldb) * 2 + ks * 128 + ch * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  dma_chunk(cb, smem);

  // ---- X fragments: rows m0 + 16 i + q, k = 32 kf + 8 g .. + 8 ----
  xu32x4 x[2][KF];
  int mrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 16 + q;
    mrow[i] = m < p.M ? m : p.M - 1;
    const char* xr = (const char*)p.A + (long)mrow[i] * p.lda * 2 + g * 16;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) x[i][kf] = *reinterpret_cast<const xu32x4*>(xr + kf * 64);
  }
  // a compiler-visible use: its own wait for the X loads sits here, once, and not in front of every MFMA of the loop
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) asm volatile("" : "+v"(x[i][kf]));

  // residual rows of a chunk: lane's channels n = c BN + 4 FJ g .. + 4 FJ of rows mrow[0], mrow[1]
  xu32x4 res[2][2][NV];  // [buffer][row fragment][piece]
  auto load_res = [&](int c, xu32x4 (&r)[2][NV]) {
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* rr = (const char*)p.resid + ((long)mrow[i] * p.ldr + c * BN + g * 4 * FJ) * 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) r[i][v] = x_load128_untracked(rr + v * 16);
      }
    }
  };
  load_res(cb, res[0]);

  // fragment read address of this lane: W row (q >> 2) * 4FJ + 4 j + (q & 3), chunk (kk * 4 + g) ^ key
  const int key = ((q >> 2) << 1) | ((q >> 1) & 1);
  const uint32_t w_lane = x_lds_off(smem) + ((q >> 2) * 4 * FJ + (q & 3)) * 128 + ((g ^ key) << 4);
  const float* zero4 = reinterpret_cast<const float*>(p.zero);

  auto do_chunk = [&](int c, xu32x4 (&rcur)[2][NV], xu32x4 (&rnext)[2][NV]) {
    // Top of chunk c.  In flight, oldest first: res(c), DMA(c), the previous chunk's NST stores.  Everything but the
    // stores must have landed (a wave whose tail rows skip stores waits for everything); then every wave's DMA pieces
    // are in the LDS and every wave is done reading the buffer the next DMA overwrites.
    if (wave_full && c != cb) x_wait_vm<NST>();  // (the first chunk has no stores behind its loads)
    else x_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v) x_landed(rcur[i][v]);
    }
    // this chunk's shift (BN floats, L2-resident) first, then the next chunk's prefetch: the epilogue's wait leaves
    // exactly the prefetch in flight
    const int nb = c * BN + g * 4 * FJ;
    xu32x4 sh[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) sh[j] = x_load128_untracked(p.bias ? p.bias + nb + 4 * j : zero4);
    const bool more = c + 1 < ce;
    if (more) {
      load_res(c + 1, rnext);
      dma_chunk(c + 1, smem + ((c + 1 - cb) & 1) * CHUNK);
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t base = w_lane + (uint32_t)((c - cb) & 1) * CHUNK;
    f32x4 acc[2][FJ];
    // software pipeline over the KF MFMA K-steps: step t + 1's FJ fragments are requested before step t's MFMAs
    uint4 wf[2][FJ];
    auto read_step = [&](int t, uint4 (&dst)[FJ]) {
      const uint32_t a = (base + (uint32_t)(t >> 1) * (BN * 128)) ^ ((t & 1) ? 64u : 0u);
#pragma unroll
      for (int j = 0; j < FJ; ++j) dst[j] = x_lds_read128(a + j * 4 * 128);
    };
    read_step(0, wf[0]);
#pragma unroll
    for (int t = 0; t < KF; ++t) {
      if (t + 1 < KF) {
        read_step(t + 1, wf[(t + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FJ) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 cin = t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t & 1][j]),
                                                               __builtin_bit_cast(bf16x8, x[i][t]), cin, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: lane holds out[m0 + 16 i + q][c BN + 4 FJ g + 4 j + r] = acc[i][j][r] ----
    if (more) x_wait_vm<NRES + SLOTS>();
    else x_wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FJ; ++j) x_landed(sh[j]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * 16 + q;
      char* dst = (char*)p.C + ((long)m * p.ldc + nb) * 2;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        uint32_t o[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {  // two channels per 32-bit word: channel 8 v + 2 h (+1) = fragment 2 v + h / 2, r = 2 (h & 1) (+1)
          const int j = 2 * v + (h >> 1), r = 2 * (h & 1);
          float lo = acc[i][j][r] + __uint_as_float(sh[j][r]);
          float hi = acc[i][j][r + 1] + __uint_as_float(sh[j][r + 1]);
          if constexpr (RES) {
            lo += __uint_as_float(rcur[i][v][h] << 16);
            hi += __uint_as_float(rcur[i][v][h] & 0xffff0000u);
          }
          if (p.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          o[h] = pack2bf(lo, hi);
        }
        if (m < p.M) *reinterpret_cast<uint4*>(dst + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int c = cb; c < ce; c += 2) {
    do_chunk(c, res[0], res[1]);
    if (c + 1 < ce) do_chunk(c + 1, res[1], res[0]);
  }
}

// The panel kernel applies when the product is a plain bf16 GEMM with a short K, whole 16-byte rows, and an output wide
// enough for the chunk loop (the expand convs: K = 64 / 128 / 256 / 512, N = 4 K).
bool expand_supported(const GemmParams& p) {
  if (p.dtype != DT_BF16 || p.conv || p.out_f32 || p.ksplit_steps > 0) return false;
  if (!(p.K == 64 || p.K == 128 || p.K == 256 || p.K == 512)) return false;
  const int bn = p.K == 512 ? 32 : 64;
  if (p.N % bn || p.N < 2 * bn || p.M < 128) return false;
  if (p.lda % 8 || p.ldb % 8 || p.ldc % 8 || (p.resid && p.ldr % 8)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.resid) | reinterpret_cast<uintptr_t>(p.bias);
  if (al & 15) return false;
  if ((long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  if (!p.bias && !p.zero) return false;  // a missing shift reads the zero page
  return true;
}

template <int KF, int FJ, bool RES>
static hipError_t launch_expand_impl(const GemmParams& p, hipStream_t stream) {
  constexpr int lds = 2 * 16 * FJ * KF * 32 * 2;
  static bool attr_set = false;
  auto kern = expand_res_kernel<KF, FJ, RES>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  // enough workgroups for two per CU (512 slots), never fewer than two chunks per workgroup
  static const int env_split = std::getenv("HVR_EXPAND_SPLIT") ? std::atoi(std::getenv("HVR_EXPAND_SPLIT")) : 0;
  const int panels = (p.M + 127) / 128, nchunks = p.N / (16 * FJ);
  int split = env_split > 0 ? env_split : (panels >= 512 ? 1 : (512 + panels - 1) / panels);
  if (split > nchunks / 2) split = nchunks / 2;
  if (split < 1) split = 1;
  hipLaunchKernelGGL(kern, dim3(panels, split), dim3(256), lds, stream, p);
  return hipGetLastError();
}

template <int KF, int FJ>
static hipError_t launch_expand(const GemmParams& p, hipStream_t stream) {
  return p.resid ? launch_expand_impl<KF, FJ, true>(p, stream) : launch_expand_impl<KF, FJ, false>(p, stream);
}

hipError_t run_expand(const GemmParams& p, hipStream_t stream) {
  switch (p.K) {
    case 64: return launch_expand<2, 4>(p, stream);
    case 128: return launch_expand<4, 4>(p, stream);
    case 256: return launch_expand<8, 4>(p, stream);
    case 512: return launch_expand<16, 2>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace hvr
