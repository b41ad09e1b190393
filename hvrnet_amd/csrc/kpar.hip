// Few-row products (stream mode: ONE frame through layer 3 / res5 / the RPN conv -- 2 394 output pixels, so 152 tiles of 64 x 64 for a
// 256-channel conv on 256 CUs), bf16, gfx950:  C = act(A W^T + bias [+ residual]),  A plain [M][lda] or the implicit-GEMM gather of an
// NHWC map (1x1 / 3x3, stride, dilation).  Replaces backbones/resnet.py:220-266's conv1 / conv2 of a Bottleneck and shared_heads/
// res_layer.py's for a one-frame batch (tools/test.py:214-250, the reference's steady-state loop).
//
// Why a kernel of its own: with so few rows the tile engine fills the chip by slicing K across WORKGROUPS (capi.hip: run_fewrow_split) --
// every slice writes a 128 x 64 f32 partial to memory and a second launch reads them back, sums, and applies the epilogue: 10 MB of
// partials each way and two launches per conv, for 2.8 GF of work (profiles/r05_frame_breakdown.txt: 23 / 17 us per conv of which the
// reduce launch is 4.3 and the partial stores ~3).  Here K is sliced across the WAVES of one workgroup instead:
//   * one workgroup (4 waves) per 64 x 64 output tile; a macro-step is four K-steps of 64 -- one per wave -- staged as four 16 KB
//     LDS images (64 A lines + 64 W lines of 128 bytes, the XOR-swizzled image of gemm.hip) by global_load_lds from all 256 threads;
//   * wave w multiplies K-steps w, w + 4, w + 8, ... : the full 64 x 64 tile (4 x 4 fragments, 64 accumulator registers, 16 LDS
//     fragment reads per 32 MFMAs) of ITS K-steps;
//   * two stages of 64 KB: macro-step s + 1 is requested before the wait for macro-step s, so two are in flight most of the time --
//     the tile is a latency chain of K / 256 macro-steps (9 for layer 3's 3x3), not of K / 64 K-steps;
//   * the four partial tiles meet in the LDS (the stages' memory), and the epilogue -- bias, residual, ReLU, bf16 -- runs on the sum:
//     no partials in memory, no second launch.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

namespace {

constexpr int KP_BM = 64, KP_BN = 64, KP_W = 4, KP_NT = KP_W * 64;
constexpr int KP_IMG = (KP_BM + KP_BN) * 128;   // one K-step: 64 A lines + 64 W lines
constexpr int KP_STAGE = KP_W * KP_IMG;         // a macro-step: one K-step per wave
constexpr int KP_LDS = 2 * KP_STAGE;            // 131 072 B
constexpr int KP_PIECES = 4 * KP_W;             // global_load_lds per thread and macro-step (2 A + 2 W rows per K-step)
constexpr int KP_RP = KP_BN + 4;                // floats per row of a partial tile in the LDS (272 B: 16-byte aligned, rows on different banks)
static_assert(KP_W * KP_BM * KP_RP * 4 <= KP_LDS, "the partial tiles fit in the stages' memory");

__device__ __forceinline__ uint32_t kp_lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ uint4 kp_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

}  // namespace

template <bool CONV>
__global__ __launch_bounds__(KP_NT) void kpar_tile_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / KP_BN, tiles_m = (p.M + KP_BM - 1) / KP_BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int pid_m = tile / tiles_n, pid_n = tile - pid_m * tiles_n;   // n fastest: the tiles of one A panel sit in one XCD's L2
  const int m0 = pid_m * KP_BM, n0 = pid_n * KP_BN;

  // ---- loader: thread -> rows (tid >> 3) and (tid >> 3) + 32 of both operands, 16-byte piece (tid & 7) ^ (row & 7) of their lines ----
  const int l_row = tid >> 3, l_chunk = ((tid & 7) ^ (l_row & 7)) * 16;
  const char* a_row[2];   // plain product: the row's first byte + piece offset; conv: unused
  int a_base[2], a_iy[2], a_ix[2];   // conv: pixel index of the image's first row, top-left input coordinate of the window
  const char* w_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + i * 32 + l_row;
    m = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
      a_base[i] = b * p.H * p.W;
      a_iy[i] = oy * p.stride - p.pad;
      a_ix[i] = ox * p.stride - p.pad;
      a_row[i] = nullptr;
    } else {
      a_row[i] = (const char*)p.A + (long)m * p.lda * 2 + l_chunk;
      a_base[i] = a_iy[i] = a_ix[i] = 0;
    }
    w_row[i] = (const char*)p.B + (long)(n0 + i * 32 + l_row) * p.ldb * 2 + l_chunk;
  }
  const int nk = p.K / 64, nms = nk / KP_W;

  auto issue = [&](int ms, char* stage) {
#pragma unroll
    for (int j = 0; j < KP_W; ++j) {
      const int kt = ms * KP_W + j;
      char* img = stage + j * KP_IMG;
      int ky = 0, kx = 0, c0 = 0;
      if constexpr (CONV) {   // (uniform: the K-step's tap and channel offset)
        const int c = kt * 64, tap = c / p.Cin;
        c0 = c - tap * p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* src;
        if constexpr (CONV) {
          const int iy = a_iy[i] + ky * p.dil, ix = a_ix[i] + kx * p.dil;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          // (the address is formed for every tap -- a padding tap's from clamped coordinates -- and selected, not branched around)
          const int cy = ok ? iy : 0, cx = ok ? ix : 0;
          const char* in = (const char*)p.A + ((long)(a_base[i] + cy * p.W + cx) * p.Cin + c0) * 2 + l_chunk;
          src = ok ? in : (const char*)p.zero;
        } else {
          src = a_row[i] + kt * 128;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(img + (i * 32 + wave * 8) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_row[i] + kt * 128),
                                         (__attribute__((address_space(3))) void*)(img + KP_BM * 128 + (i * 32 + wave * 8) * 128), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_grp = lane >> 4;
  const uint32_t a_lane = kp_lds_off(smem) + wave * KP_IMG + frag_row * 128 + ((frag_grp ^ (frag_row & 7)) * 16);
  const uint32_t w_lane = a_lane + KP_BM * 128;

  issue(0, smem);
  for (int ms = 0; ms < nms; ++ms) {
    const uint32_t soff = (uint32_t)(ms & 1) * KP_STAGE;
    // macro-step ms + 1 into the other stage: every wave left it at the barrier that closed macro-step ms - 1
    if (ms + 1 < nms) {
      issue(ms + 1, smem + ((ms + 1) & 1) * KP_STAGE);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KP_PIECES) : "memory");   // this thread's pieces of macro-step ms have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                        // ... and everybody else's
    const uint32_t a0 = a_lane + soff, w0 = w_lane + soff;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 wf[4], af[4];
      static_for<4>([&](auto J) { wf[decltype(J)::value] = kp_read128<decltype(J)::value * 2048>(h ? (w0 ^ 64u) : w0); });
      static_for<4>([&](auto I) { af[decltype(I)::value] = kp_read128<decltype(I)::value * 2048>(h ? (a0 ^ 64u) : a0); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // weights as the MFMA "A" operand: a lane ends up with 4 consecutive output channels of one output row
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_half<bf16_t>(wf[j], af[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();   // the stage is free for macro-step ms + 2
  }

  // ---- the four partial tiles meet in the LDS; epilogue on their sum ----
  float* part = reinterpret_cast<float*>(smem) + wave * (KP_BM * KP_RP);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(part + (i * 16 + frag_row) * KP_RP + j * 16 + frag_grp * 4) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  __syncthreads();
  {
    const int row = tid >> 2, c0 = (tid & 3) * 16, m = m0 + row;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 s = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem) + row * KP_RP + c0 + q * 4);
#pragma unroll
      for (int w = 1; w < KP_W; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem) + w * (KP_BM * KP_RP) + row * KP_RP + c0 + q * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      v[q * 4 + 0] = s.x; v[q * 4 + 1] = s.y; v[q * 4 + 2] = s.z; v[q * 4 + 3] = s.w;
    }
    if (m < p.M) {
      if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n0 + c0 + q * 4);
          v[q * 4 + 0] += b.x; v[q * 4 + 1] += b.y; v[q * 4 + 2] += b.z; v[q * 4 + 3] += b.w;
        }
      }
      if (p.resid) {
        const bf16_t* r = (const bf16_t*)p.resid + (long)m * p.ldr + n0 + c0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint4 u = *reinterpret_cast<const uint4*>(r + q * 8);
          const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lo, hi;
            unpack2<bf16_t>(wds[e], lo, hi);
            v[q * 8 + 2 * e] += lo;
            v[q * 8 + 2 * e + 1] += hi;
          }
        }
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      bf16_t* out = (bf16_t*)p.C + (long)m * p.ldc + n0 + c0;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *reinterpret_cast<uint4*>(out + q * 8) = make_uint4(pack2<bf16_t>(v[q * 8 + 0], v[q * 8 + 1]), pack2<bf16_t>(v[q * 8 + 2], v[q * 8 + 3]),
                                                            pack2<bf16_t>(v[q * 8 + 4], v[q * 8 + 5]), pack2<bf16_t>(v[q * 8 + 6], v[q * 8 + 7]));
    }
  }
}

// bf16 products with staged operands whose 64 x 64 tile grid leaves room on the chip and whose K loop is whole macro-steps
bool kpar_supported(const GemmParams& p) {
  if (p.dtype != DT_BF16 || !p.staging || p.out_f32 || p.ksplit_steps > 0 || p.s2 > 0 || p.Wn) return false;
  if (p.N % KP_BN || p.K % (64 * KP_W) || p.ldc % 8 || p.ldb % 8) return false;
  if (p.conv ? (p.Cin % 64 || !p.zero) : (p.lda % 8 != 0)) return false;
  if (p.resid && p.ldr % 8) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
                       reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.resid) | reinterpret_cast<uintptr_t>(p.zero);
  if (al & 15) return false;
  if ((long)p.N * p.ldb * 2 >= (1L << 31)) return false;
  // Where it pays (one 600 x 1000 frame, profiles/r05_frame_breakdown.txt): a tile is a latency chain of K / 256 macro-steps of ~1.7 us
  // (two 64 KB stages in flight per CU is all the LDS holds), so the gain over slicing K across workgroups -- no partials, no reduce
  // launch: layer 3's 3x3 22.8 -> 17.3 us, its reducing 1x1 16.7 -> 10.2 -- is gone once the chain is long: res5's dilated 3x3
  // (K = 4 608) 37.9 -> 50.7, the RPN conv (K = 9 216) 62 -> 95, fc_new_1 (K = 12 544) 31 -> 50 stay on the sliced form.
  const long tiles = (long)((p.M + KP_BM - 1) / KP_BM) * (p.N / KP_BN);
  return tiles <= 320 && p.K <= 2560;
}

hipError_t run_kpar(const GemmParams& p, hipStream_t stream) {
  static std::atomic<unsigned> attr_set_dev{0};   // (the attribute is per device)
  per_device_once(attr_set_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kpar_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, KP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kpar_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, KP_LDS);
  });
  const int tiles = ((p.M + KP_BM - 1) / KP_BM) * (p.N / KP_BN);
  if (p.conv) hipLaunchKernelGGL(kpar_tile_kernel<true>, dim3(tiles), dim3(KP_NT), KP_LDS, stream, p);
  else hipLaunchKernelGGL(kpar_tile_kernel<false>, dim3(tiles), dim3(KP_NT), KP_LDS, stream, p);
  return hipGetLastError();
}

}  // namespace hvr
