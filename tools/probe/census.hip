// Census: where do the blocks of a 2-workgroups-per-CU grid land?  (tuning probe, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
  extern __shared__ char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_readcyclecounter();
  // keep the block resident for a while so that the whole grid is co-resident
  volatile float* s = (volatile float*)smem;
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) { s[threadIdx.x] = acc; acc += s[(threadIdx.x + 1) & 255]; }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffffu);
    out[blockIdx.x * 4 + 3] = (unsigned)acc;
  }
}
int main() {
  const int nb = 768;
  unsigned* d; hipMalloc(&d, nb * 16);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(census, dim3(nb), dim3(256), 65536, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
  unsigned tmin = ~0u; for (int b = 0; b < nb; ++b) tmin = h[b*4+2] < tmin ? h[b*4+2] : tmin;
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[b*4], x = h[b*4+1];
    printf("b %3d xcc %u hw %08x wave %u simd %u pipe %u cu %u sh %u se %u tg %u t0 %u\n", b, x & 0xf, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3,
           (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15, h[b*4+2] - tmin);
  }
  return 0;
}
