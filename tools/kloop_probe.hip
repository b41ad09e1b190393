// K-loop probe: what caps an MFMA loop that takes its fragments from the LDS?  No global memory in the loop, no DMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kloop_probe.hip -o /tmp/kloop_probe && /tmp/kloop_probe
// One workgroup per CU, WAVES waves, every wave an FM x FN tile of 16x16x32 bf16 MFMAs.  A "half" = FM*FN MFMAs fed by FM+FN fragment
// reads (ds_read_b128, the tile engine's swizzled addressing), one read behind each of the first MFMAs, into the fragment set that
// will be used S-1 halves later (S = 2: the tile engine's pipeline; 3 / 4: one / two more halves of lead).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename F, int... Is> __device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF> __device__ __forceinline__ uint4 rd(uint32_t a) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}

__device__ __forceinline__ void dma16(const void* base, char* lds, unsigned voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// S: fragment sets; READS: 0 = fragments read once before the loop (MFMA only), 1 = read in the loop; BAR: s_barrier per two halves
template <int WAVES, int FM, int FN, int S, int READS, int BAR, int U = 1, int ND = 0, int PW = 0>
__global__ __launch_bounds__((WAVES + PW) * 64) void kloop(const uint4* __restrict__ src, float* __restrict__ out, long long* __restrict__ clk, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NR = FM + FN, NM = FM * FN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 147456 / 16; i += (WAVES + PW) * 64) reinterpret_cast<uint4*>(smem)[i] = src[i];
  __syncthreads();
  char* const dma_dst = smem + 147456 - 16384;   // a 16 KB landing strip behind the fragment area (the reads never touch it)
  const unsigned dma_voff = (unsigned)(lane * 16);
  if constexpr (PW > 0) {
    if (wave >= WAVES) {
      // producer waves: per two halves (one "K-step") all of the compute waves' DMA pieces, then the step's barrier
      constexpr int PIECES = 2 * ND * WAVES / PW;
      for (int st = 0; st < steps * S / 2; ++st) {
        sfor<PIECES>([&](auto I) { dma16(src, dma_dst + ((decltype(I)::value + wave) & 15) * 1024, dma_voff, ((st + decltype(I)::value) & 63) * 1024); });
        wait_vm<PIECES>();
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
      }
      return;
    }
  }
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const uint32_t a_lane = base + ((lane & 15) * 128) + ((((lane >> 4) ^ (lane & 7))) * 16);
  const uint32_t b_lane = a_lane + 73728 + (wave % 8) * (FN * 2048 > 4096 ? 4096 : FN * 2048);
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 fa[S][FM], fb[S][FN];
  auto read_set = [&](auto SS, uint32_t off) {
    constexpr int s = decltype(SS)::value;
    sfor<FM>([&](auto I) { fa[s][decltype(I)::value] = rd<decltype(I)::value * 2048>(a_lane + off); });
    sfor<FN>([&](auto J) { fb[s][decltype(J)::value] = rd<decltype(J)::value * 2048>(b_lane + off); });
  };
  sfor<S>([&](auto SS) { read_set(SS, (uint32_t)decltype(SS)::value * 64u); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const long long t0 = __builtin_readcyclecounter();
  uint32_t off = 0;
  for (int st = 0; st < steps; st += U) {
   sfor<U>([&](auto UU) {
    sfor<S>([&](auto HH) {   // S halves per iteration, set index = half index
      constexpr int h = decltype(HH)::value, tgt = (h + S - 1) % S;   // the set freed by the previous half
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (READS) {
        // set h was requested S - 1 halves ago; younger requests: (S - 2) halves' worth
        if constexpr (S == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((S - 2) * NR > 15 ? 15 : (S - 2) * NR) : "memory");
      }
      if constexpr (BAR && (h & 1)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      sfor<NM>([&](auto Q) {
        constexpr int q = decltype(Q)::value, i = q / FN, j = q % FN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[h][j]), __builtin_bit_cast(bf16x8, fa[h][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (READS && q < NR) {
          // (S == 2: into the other set, as the tile engine does; the set being computed from is never a target)
          constexpr int t2 = S == 2 ? (h ^ 1) : tgt;
          if constexpr (q < FM) fa[t2][q] = rd<(q % 9) * 2048>(a_lane + off);
          else fb[t2][q - FM] = rd<((q - FM) % 2) * 2048>(b_lane + off);
        }
        if constexpr (PW == 0 && ND > 0 && q >= NR && q < NR + ND) {
          dma16(src, dma_dst + ((q + wave) & 15) * 1024, dma_voff, ((int)(off >> 6) & 63) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (PW == 0 && ND > 0) wait_vm<ND>();
      off = (off + 64u) & 0x3fffu;
    });
   });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (lane == 0) clk[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, int FM, int FN, int S, int READS, int BAR, int U = 1, int ND = 0, int PW = 0>
static void run(const char* name, const uint4* src, float* out, long long* clk, int steps) {
  auto k = kloop<WAVES, FM, FN, S, READS, BAR, U, ND, PW>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3((WAVES + PW) * 64), 147456, 0, src, out, clk, steps);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3((WAVES + PW) * 64), 147456, 0, src, out, clk, steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(256 * WAVES);
  hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : c) avg += (double)v;
  avg /= c.size();
  const double mfma_per_wave = (double)steps * S * FM * FN;
  const double flops = mfma_per_wave * 16384.0 * 256 * WAVES;
  // s_memtime / readcyclecounter ticks at 100 MHz on this chip: report wall-clock-derived figures
  printf("%-46s %7.3f ms  %7.0f TF/s  (%.2f of 2.5 PF)  reads/MFMA %.2f  ticks/half %.1f\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0,
         READS ? (double)(FM + FN) / (FM * FN) : 0.0, avg / (steps * S));
}

int main() {
  uint4* src; float* out; long long* clk;
  hipMalloc(&src, 147456); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
  std::vector<unsigned short> h(147456 / 2);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; __builtin_memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(src, h.data(), 147456, hipMemcpyHostToDevice);
  const int steps = 2000;
  run<8, 9, 2, 2, 0, 0>("8 waves 9x2, MFMA only", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 0>("8 waves 9x2, reads, S=2 (engine), no barrier", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1>("8 waves 9x2, reads, S=2 (engine), barrier", src, out, clk, steps);
  run<8, 9, 2, 2, 0, 0, 2>("8 waves 9x2, MFMA only, loop unrolled x2", src, out, clk, steps);
  run<8, 9, 2, 2, 0, 0, 4>("8 waves 9x2, MFMA only, loop unrolled x4", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 2>("8 waves 9x2, reads, S=2, barrier, unrolled x2", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 4>("8 waves 9x2, reads, S=2, barrier, unrolled x4", src, out, clk, steps);
  run<8, 9, 4, 2, 1, 1, 2>("8 waves 9x4, reads, S=2, barrier, unrolled x2", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 2, 3>("8 waves 9x2, reads, barrier, x2, + 3 DMA pieces / half / wave", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 2, 1>("8 waves 9x2, reads, barrier, x2, + 1 DMA piece / half / wave", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 2, 6>("8 waves 9x2, reads, barrier, x2, + 6 DMA pieces / half / wave", src, out, clk, steps);
  run<8, 9, 2, 2, 0, 1, 2, 3>("8 waves 9x2, NO reads, barrier, x2, + 3 DMA pieces", src, out, clk, steps);
  run<8, 9, 2, 2, 1, 1, 2, 3, 4>("8 waves 9x2, reads, barrier, x2, DMA by 4 producer waves", src, out, clk, steps);
  run<4, 9, 4, 2, 1, 1, 2, 6, 4>("4 waves 9x4, reads, barrier, x2, DMA by 4 producer waves", src, out, clk, steps);
  run<4, 9, 4, 2, 1, 1, 2, 6, 0>("4 waves 9x4, reads, barrier, x2, + 6 DMA pieces / half / wave", src, out, clk, steps);
  run<8, 9, 4, 2, 1, 1, 1, 4>("8 waves 9x4 (288x256), reads, barrier, + 4 DMA pieces", src, out, clk, steps);
  run<8, 9, 2, 3, 1, 0>("8 waves 9x2, reads, S=3, no barrier", src, out, clk, steps);
  run<8, 9, 2, 3, 1, 1>("8 waves 9x2, reads, S=3, barrier", src, out, clk, steps);
  run<8, 9, 2, 4, 1, 1>("8 waves 9x2, reads, S=4, barrier", src, out, clk, steps);
  run<4, 9, 4, 2, 0, 0>("4 waves 9x4, MFMA only", src, out, clk, steps);
  run<4, 9, 4, 2, 1, 1>("4 waves 9x4, reads, S=2, barrier", src, out, clk, steps);
  run<4, 9, 4, 3, 1, 1>("4 waves 9x4, reads, S=3, barrier", src, out, clk, steps);
  run<8, 9, 4, 2, 0, 0>("8 waves 9x4 (288x256), MFMA only", src, out, clk, steps);
  run<8, 9, 4, 2, 1, 1>("8 waves 9x4 (288x256), reads, S=2, barrier", src, out, clk, steps);
  run<8, 4, 4, 2, 0, 0>("8 waves 4x4, MFMA only", src, out, clk, steps);
  run<8, 4, 4, 2, 1, 1>("8 waves 4x4, reads, S=2, barrier", src, out, clk, steps);
  run<8, 4, 4, 3, 1, 1>("8 waves 4x4, reads, S=3, barrier", src, out, clk, steps);
  return 0;
}
