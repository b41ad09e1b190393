// Does the ACCESS PATTERN of the row-panel expand kernel limit it?  out = max(R + X-less bias, 0) over [M][1024] bf16 (73.5 MB in, 73.5 MB out at
// M = 35 910) with (a) linear 16-byte-per-lane addressing, (b) expand.hip's mapping: a workgroup of 4 waves owns 128 rows x 512 channels and walks
// eight 64-channel chunks; a wave instruction touches 16 rows x 64 bytes (2 KB apart), (c) the same with 128-byte (d) 256-byte runs per row.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/spp tools/stream_pattern_probe.hip && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint4 relu_add(uint4 a) {   // cheap stand-in for the epilogue arithmetic
  auto f = [](uint32_t u) { return (u & 0x80008000u) ? (u & 0x7fff7fffu) : u; };
  return make_uint4(f(a.x), f(a.y), f(a.z), f(a.w));
}

__global__ __launch_bounds__(256) void linear_kernel(const uint4* __restrict__ r, uint4* __restrict__ o, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) o[i] = relu_add(r[i]);
}

// RUN = contiguous bytes a 4-lane.. group covers per row: 64 (expand.hip today: lane = (row q = lane & 15, piece g = lane >> 4), two instructions v = 0, 1
// 32 bytes apart... modelled as 16 rows x 64 B per instruction), 128, 256
template <int RUN>
__global__ __launch_bounds__(256, 2) void panel_kernel(const char* __restrict__ r, char* __restrict__ o, int M, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int LPR = RUN / 16;           // lanes per row
  constexpr int ROWS = 64 / LPR;          // rows per instruction
  const int row_in = lane / LPR, piece = lane % LPR;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int cb = blockIdx.y * 512;        // channel range of this workgroup (bytes = 2 * channels)
  for (int c = 0; c < 8; ++c) {           // 64-channel chunks = 128 bytes per row
    // a wave covers its 32 rows x 128 bytes with 32 * 128 / 1024 = 4 instructions
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int unit = it * 64 + lane;    // 256 units of 16 B = 32 rows x 8 pieces
      int row, off;
      if (RUN >= 128) { row = unit >> 3; off = (unit & 7) * 16; }                       // 8 rows x 128 B per instruction
      else { row = (it >> 1) * 16 + row_in; off = ((it & 1) * 4 + piece) * 16; }          // 16 rows x 64 B per instruction
      int m = m0 + row;
      m = m < M ? m : M - 1;
      const long a = (long)m * N * 2 + (cb + c * 64) * 2 + off;
      *reinterpret_cast<uint4*>(o + a) = relu_add(*reinterpret_cast<const uint4*>(r + a));
    }
    __builtin_amdgcn_s_barrier();         // the chunk barrier of the real kernel
  }
  (void)ROWS;
}

int main() {
  const int M = 35910, N = 1024;
  const size_t bytes = (size_t)M * N * 2;
  char *r, *o;
  hipMalloc(&r, bytes); hipMalloc(&o, bytes);
  hipMemset(r, 0x3c, bytes);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(s);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-44s %7.1f us  %6.0f GB/s\n", name, ms / 20 * 1e3, 2.0 * bytes / (ms / 20 * 1e-3) / 1e9);
  };
  time("linear, 16 B per lane, grid 2048", [&] { hipLaunchKernelGGL(linear_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)r, (uint4*)o, (long)(bytes / 16)); });
  time("panel 128 x 512, 16 rows x 64 B per instruction", [&] { hipLaunchKernelGGL(panel_kernel<64>, dim3((M + 127) / 128, 2), dim3(256), 0, 0, r, o, M, N); });
  time("panel 128 x 512, 8 rows x 128 B per instruction", [&] { hipLaunchKernelGGL(panel_kernel<128>, dim3((M + 127) / 128, 2), dim3(256), 0, 0, r, o, M, N); });
  return 0;
}
