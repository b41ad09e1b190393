// 3x3 convolution, 64 -> 64 channels, stride 1, pad 1, + shift + ReLU (bf16, gfx950): conv2 of the layer-1 Bottlenecks
// (mmdet/models/backbones/resnet.py:236-241 with planes = 64; 3 launches per frame batch, M = 574 560 pixels at 15 frames).
//
// Why not the tile engine's implicit GEMM (gemm.hip): with Cin = 64 a K-step is one filter tap, so an output tile of
// 128 pixels re-stages its input rows nine times and its 74 KB of weights once per tile -- 237 KB through the L2 for
// 9.4 MF, i.e. 1.06 GB per launch: the kernel sits on the L2 -> LDS bandwidth (10.7 TB/s, 99 us) although the
// convolution only needs 147 MB from HBM and 42 GF.  Here:
//   * a PERSISTENT workgroup (8 waves) keeps all nine taps' weights in the LDS for its whole life (73.7 KB, loaded once);
//   * it walks 16 x 16-pixel output tiles; a tile's 18 x 18 x 64 input halo (41 KB) arrives by global_load_lds into one
//     of two buffers while the previous tile computes (out-of-image pixels come from the zero page), and every tap reads
//     its MFMA operand from that halo at a shifted address: 1.27 input bytes staged per output byte instead of 9;
//   * fragment order and MFMA sequence per output element are the tile engine's (tap-major, two 32-channel halves per
//     tap), so the results are bit-identical to it;
//   * the MFMA row index is permuted as in expand.hip, so a lane ends a tile with 16 consecutive channels of a pixel:
//     32-byte runs per lane in the stores;
//   * waits: one s_waitcnt vmcnt(0) (through the builtin: the compiler's wait-count pass sees it) per tile, placed AFTER
//     the tile's MFMAs and BEFORE its stores -- the next halo has had the whole MFMA phase to land, and the stores stay
//     in flight under the next tile's MFMAs; one bare s_barrier per tile.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int C3_T = 16;                          // output tile edge (pixels)
constexpr int C3_HW = C3_T + 2;                   // halo edge
constexpr int C3_HPIX = C3_HW * C3_HW;            // 324 halo pixels
constexpr int C3_NT = 512;
constexpr int C3_HPIECES = (C3_HPIX * 8 + 63) / 64 * 64;  // 16-byte pieces, whole wave instructions: 2 624 (328 pixels)
constexpr int C3_HALO_BYTES = C3_HPIECES * 16;    // 41 984
constexpr int C3_W_BYTES = 9 * 64 * 128;          // 73 728
constexpr int C3_LDS = C3_W_BYTES + 2 * C3_HALO_BYTES;  // 157 696
static_assert(C3_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ uint32_t c3_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ uint4 c3_lds_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// weight rows inside a 64-channel block: same row order / swizzle key as expand.hip (lane q reads row 16 (q >> 2) + 4 j + (q & 3))
__device__ __forceinline__ int c3_wkey(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }

}  // namespace

struct Conv3x3Params {
  const bf16_t* x;    // [B][H][W][64]
  const bf16_t* w;    // [64][3][3][64]
  const float* bias;  // [64] or null
  bf16_t* y;          // [B][H][W][64]
  const void* zero;   // >= 16 zero bytes
  int B, H, W, relu;
  int tiles_x, tiles_y, ntiles;
};

template <typename T>   // bf16_t / f16_t (raw 16-bit words everywhere but the MFMA and the output pack)
__global__ __launch_bounds__(C3_NT, 1) void conv3x3_c64_kernel(const Conv3x3Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  char* halo0 = smem + C3_W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;

  // ---- weights -> LDS once: image [tap][n][128 B], chunk position pos holds global chunk pos ^ key(n) ----
#pragma unroll
  for (int i = 0; i < C3_W_BYTES / 16 / C3_NT; ++i) {
    const int s = i * C3_NT + tid, tap = s >> 9, rem = s & 511, n = rem >> 3, pos = rem & 7;
    const int ch = pos ^ c3_wkey(n);
    const char* src = (const char*)p.w + ((long)(n * 9 + tap) * 64) * 2 + ch * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(wl + (i * C3_NT + wave * 64) * 16), 16, 0, 0);
  }

  // ---- halo loader: piece s -> halo pixel hp = s >> 3 (row-major 18 x 18), position s & 7 holds chunk (s & 7) ^ (hp & 7) ----
  auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
    const int per_img = p.tiles_x * p.tiles_y;
    b = t / per_img;
    const int r = t - b * per_img, ty = r / p.tiles_x;
    ty0 = ty * C3_T;
    tx0 = (r - ty * p.tiles_x) * C3_T;
  };
  auto dma_halo = [&](int t, char* buf) {
    int b, ty0, tx0;
    tile_origin(t, b, ty0, tx0);
#pragma unroll
    for (int i = 0; i < (C3_HPIECES + C3_NT - 1) / C3_NT; ++i) {
      if (i * C3_NT + wave * 64 < C3_HPIECES) {  // (the last slot: the first waves only; wave-uniform)
        const int s = i * C3_NT + tid, hp = s >> 3, pos = s & 7;
        const int hy = hp / C3_HW, hx = hp - hy * C3_HW;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = hp < C3_HPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const char* src = (const char*)p.x + ((((long)b * p.H + iy) * p.W + ix) * 64) * 2 + ((pos ^ (hp & 7)) * 16);
        src = ok ? src : (const char*)p.zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + (i * C3_NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };

  int t = blockIdx.x;
  if (t < p.ntiles) dma_halo(t, halo0);

  // shifts of this lane's 16 channels (n = 16 g + 4 j + r)
  float sh[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sh[e] = p.bias ? p.bias[g * 16 + e] : 0.f;

  // weight fragment address: row 16 (q >> 2) + 4 j + (q & 3), chunk (kk * 4 + g) ^ key
  const int wk = ((q >> 2) << 1) | ((q >> 1) & 1);
  const uint32_t w_lane = c3_lds_off(wl) + ((q >> 2) * 16 + (q & 3)) * 128 + ((g ^ wk) << 4);
  // this wave's output rows inside the tile: 2 wave, 2 wave + 1; pixel x = q
  const uint32_t h_base = c3_lds_off(halo0);

  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // vmcnt(0): weights, first halo, shifts
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  for (int it = 0; t < p.ntiles; t += gridDim.x, ++it) {
    const uint32_t hb = h_base + (uint32_t)(it & 1) * C3_HALO_BYTES;
    const int tn = t + gridDim.x;
    if (tn < p.ntiles) dma_halo(tn, halo0 + ((it + 1) & 1) * C3_HALO_BYTES);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[2][4];
    // 18 steps (tap, 32-channel half); step s + 1's six fragments are requested before step s's eight MFMAs
    uint4 xf[2][2], wf[2][4];
    auto read_step = [&](int s, uint4 (&xd)[2], uint4 (&wd)[4]) {
      const int tap = s >> 1, kk = s & 1, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hp = (2 * wave + i + ky) * C3_HW + kx + q;  // halo pixel of output (row 2 wave + i, x = q) under this tap
        xd[i] = c3_lds_read128(hb + hp * 128 + (((kk * 4 + g) ^ (hp & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) wd[j] = c3_lds_read128((w_lane + tap * (64 * 128) + j * 4 * 128) ^ (kk ? 64u : 0u));
    };
    read_step(0, xf[0], wf[0]);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s + 1 < 18) {
        read_step(s + 1, xf[(s + 1) & 1], wf[(s + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 cin = s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
          acc[i][j] = mfma_half<T>(wf[s & 1][j], xf[s & 1][i], cin);
        }
      __builtin_amdgcn_sched_barrier(0);
    }

    // the next halo has landed (it had the MFMA phase); the previous tile's stores are long gone
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    __builtin_amdgcn_sched_barrier(0);

    int b, ty0, tx0;
    tile_origin(t, b, ty0, tx0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = ty0 + 2 * wave + i, ox = tx0 + q;
      uint32_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {  // channels 16 g + 2 e, + 1: fragment j = e / 2, r = 2 (e & 1)
        const int j = e >> 1, r = 2 * (e & 1);
        float lo = acc[i][j][r] + sh[2 * e], hi = acc[i][j][r + 1] + sh[2 * e + 1];
        if (p.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
        o[e] = pack2<T>(lo, hi);
      }
      if (oy < p.H && ox < p.W) {
        char* dst = (char*)p.y + ((((long)b * p.H + oy) * p.W + ox) * 64 + g * 16) * 2;
        *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(dst + 16) = make_uint4(o[4], o[5], o[6], o[7]);
      }
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of the next halo are in; every wave is done with this one
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bf16, 3x3, stride 1, pad 1, no dilation, 64 -> 64 channels, no residual, 16-byte aligned operands
bool conv3x3_c64_supported(const GemmParams& p) {
  if ((p.dtype != DT_BF16 && p.dtype != DT_F16) || !p.conv || p.out_f32 || p.resid) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.Cin != 64 || p.N != 64) return false;
  if (p.OH != p.H || p.OW != p.W || !p.zero) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C);
  if (al & 15) return false;
  return (long)p.M >= 4 * C3_T * C3_T;  // a handful of tiles at least: below that the one-time weight load does not pay
}

hipError_t run_conv3x3_c64(const GemmParams& g, hipStream_t stream) {
  Conv3x3Params p;
  p.x = (const bf16_t*)g.A; p.w = (const bf16_t*)g.B; p.bias = g.bias; p.y = (bf16_t*)g.C; p.zero = g.zero;
  p.H = g.H; p.W = g.W; p.B = g.M / (g.H * g.W); p.relu = g.relu;
  p.tiles_x = (g.W + C3_T - 1) / C3_T; p.tiles_y = (g.H + C3_T - 1) / C3_T;
  p.ntiles = p.B * p.tiles_x * p.tiles_y;
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS);
  });
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  if (g.dtype == DT_F16) hipLaunchKernelGGL(conv3x3_c64_kernel<f16_t>, dim3(grid), dim3(C3_NT), C3_LDS, stream, p);
  else hipLaunchKernelGGL(conv3x3_c64_kernel<bf16_t>, dim3(grid), dim3(C3_NT), C3_LDS, stream, p);
  return hipGetLastError();
}

}  // namespace hvr
