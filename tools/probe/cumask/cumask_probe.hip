// Where do the workgroups of a launch on a CU-masked HIP stream run (XCD id, CU id), directly and through a captured graph?
//   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void report(const char* what, const std::vector<uint32_t>& h, int n) {
  std::map<int, int> per_xcc;
  std::map<int, int> cus;
  for (int i = 0; i < n; ++i) {
    const int xcc = h[2 * i] & 0xf;
    const uint32_t hw = h[2 * i + 1];
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    per_xcc[xcc]++;
    cus[(xcc << 8) | (se << 5) | (sh << 4) | cu]++;
  }
  printf("%-44s distinct (xcc, se, sh, cu): %3zu   workgroups per XCC:", what, cus.size());
  for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
}

int main() {
  const int n = 2048;
  uint32_t* d;
  CK(hipMalloc(&d, n * 8));
  std::vector<uint32_t> h(2 * n);
  struct M { const char* name; uint32_t w[8]; } masks[] = {
      {"all 256", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
      {"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
      {"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
      {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
      {"bits with (i % 8) < 4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}},
      {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
  };
  for (auto& m : masks) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m.w));
    CK(hipMemsetAsync(d, 0xff, n * 8, s));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    char buf[96];
    snprintf(buf, sizeof buf, "mask %-22s direct launch", m.name);
    report(buf, h, n);
    // the same launch captured into a graph on that stream and replayed on it
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d, 20000);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemsetAsync(d, 0xff, n * 8, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    snprintf(buf, sizeof buf, "mask %-22s graph replay", m.name);
    report(buf, h, n);
    // and replayed on a plain stream (is the mask a property of the capture or of the launch stream?)
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    CK(hipMemsetAsync(d, 0xff, n * 8, plain));
    CK(hipGraphLaunch(ge, plain));
    CK(hipStreamSynchronize(plain));
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    snprintf(buf, sizeof buf, "mask %-22s graph on a plain stream", m.name);
    report(buf, h, n);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(plain));
    CK(hipStreamDestroy(s));
  }
  return 0;
}
