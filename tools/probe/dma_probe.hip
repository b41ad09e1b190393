// L2 -> LDS streaming ceiling of one CU (tuning probe, not product code): every wave of a 1-workgroup-per-CU grid issues
// `global_load_lds_dwordx4` pieces (1 KiB per wave-instruction, the tile kernels' swizzled row-panel pattern) from an
// L2-resident panel into an LDS ring as fast as counted vmcnt allows; nothing else runs.  Prints bytes / clk / CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); } } while (0)

template <int INFLIGHT, bool PLAIN, int DRAIN = 0>
__global__ __launch_bounds__(1024) void dma_stream(const char* src, long panel_bytes, int rows_ld, int iters, float* sink, int shared_panel) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  // this CU's panel: `panel_bytes` of rows of rows_ld bytes; a wave-instruction covers 8 rows x 128 B, chunks XOR-swizzled
  const char* panel = src + (shared_panel ? 0 : (long)blockIdx.x * panel_bytes);
  const int row = lane >> 3, c = (lane & 7) ^ (row & 7);
  const long lane_off = (long)row * rows_ld + c * 16;
  const long nrows = panel_bytes / rows_ld;       // rows in the panel
  char* ring = smem + wave * (INFLIGHT * 2 * 1024);
  float acc = 0.f;
  // piece p of the panel = 8 rows x 128 B at byte (p / kc_n) * 8 * rows_ld + (p % kc_n) * 128; panel sizes and rows_ld are
  // powers of two, so the walk is shifts and masks (no division in the issue loop)
  const int kc_n = rows_ld / 128, kc_sh = __builtin_ctz(kc_n);
  const unsigned pmask = (unsigned)(panel_bytes / 1024) - 1u;  // pieces in the panel - 1
  unsigned piece = wave;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const unsigned pr = piece & pmask;
      const char* g = panel + (long)(pr >> kc_sh) * 8 * rows_ld + (pr & (kc_n - 1)) * 128 + lane_off;
      if constexpr (PLAIN) {
        const uint4 v = *reinterpret_cast<const uint4*>(g);
        acc += __uint_as_float(v.x);
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(ring + ((it & 1) * INFLIGHT + u) * 1024), 16, 0, 0);
      }
      piece += nw;
    }
    if constexpr (DRAIN == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // 2-stage pipeline: batch landed before the next is issued
    else if constexpr (DRAIN == 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory"); __builtin_amdgcn_s_barrier(); }  // one batch stays in flight across the barrier
    else if constexpr (!PLAIN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");  // the previous batch has landed
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.f) sink[0] = acc;
}

template <int INFLIGHT, bool PLAIN, int DRAIN = 0>
static void run(const char* name, int waves, long panel_bytes, int rows_ld, int shared_panel, const char* d, float* sink) {
  const int iters = 400;
  const size_t lds = (size_t)waves * INFLIGHT * 2 * 1024;
  auto k = dma_stream<INFLIGHT, PLAIN, DRAIN>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), lds, 0, d, panel_bytes, rows_ld, iters, sink, shared_panel);
  CK(hipEventRecord(a));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), lds, 0, d, panel_bytes, rows_ld, iters, sink, shared_panel);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  const double bytes = 256.0 * waves * iters * INFLIGHT * 1024.0;
  printf("%-28s waves %d in-flight/wave %2d panel %5ld KB %s: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, waves, INFLIGHT, panel_bytes / 1024,
         shared_panel ? "shared by all CUs " : "one per CU        ", ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
  const long total = 256L * 512 * 1024;  // 128 MB: one 512 KB panel per CU at most
  char* d; CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total));
  float* sink; CK(hipMalloc(&sink, 4));
  // panels that stay L2-resident (64 KB per CU = 16 MB over the chip's 32 MB of L2), a panel everybody shares (L2 / L1 hits),
  // and panels that stream from the Infinity Cache / HBM (512 KB per CU = 128 MB)
  for (int shared = 0; shared < 2; ++shared) {
    run<4, false>("LDS-DMA", 4, 64 * 1024, 2048, shared, d, sink);
    run<8, false>("LDS-DMA", 4, 64 * 1024, 2048, shared, d, sink);
    run<4, false>("LDS-DMA", 8, 64 * 1024, 2048, shared, d, sink);
    run<8, false>("LDS-DMA", 8, 64 * 1024, 2048, shared, d, sink);
    run<8, false>("LDS-DMA", 16, 64 * 1024, 2048, shared, d, sink);
    run<8, true>("global_load_dwordx4 -> VGPR", 8, 64 * 1024, 2048, shared, d, sink);
  }
  run<10, false, 1>("LDS-DMA drain/barrier per 10", 8, 64 * 1024, 2048, 0, d, sink);
  run<10, false, 2>("LDS-DMA 1 batch across barrier", 8, 64 * 1024, 2048, 0, d, sink);
  run<5, false, 1>("LDS-DMA drain/barrier per 5", 8, 64 * 1024, 2048, 0, d, sink);
  run<5, false, 2>("LDS-DMA 1 batch(5) across barrier", 8, 64 * 1024, 2048, 0, d, sink);
  run<10, false, 1>("LDS-DMA drain/barrier per 10", 8, 512 * 1024, 2048, 0, d, sink);
  run<10, false, 2>("LDS-DMA 1 batch across barrier", 8, 512 * 1024, 2048, 0, d, sink);
  run<8, false>("LDS-DMA", 8, 512 * 1024, 2048, 0, d, sink);
  run<8, false>("LDS-DMA", 16, 512 * 1024, 2048, 0, d, sink);
  run<8, true>("global_load_dwordx4 -> VGPR", 8, 512 * 1024, 2048, 0, d, sink);
  return 0;
}
