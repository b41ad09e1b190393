// Which LANE -> ADDRESS mapping does the row-panel expand kernel's residual / store traffic want?  out = relu(R) over [M][1024] bf16 at
// M = 143 640 (60 frames of 38 x 63: 294 MB in, 294 MB out -- past the 256 MB Infinity Cache), a workgroup of 4 waves owns 128 rows x 512
// channels and walks eight 64-channel chunks (128 bytes per row) with a barrier per chunk, as expand.hip does.  Per wave and chunk: 32 rows x
// 128 B = four 16-byte-per-lane instructions.  Mappings (q = lane & 15, g = lane >> 4):
//   mfma32   expand.hip today: row = q (+16 for the second fragment), byte = 32 g + 16 v        (a lane's 16 consecutive channels)
//   mfma64   row = q, byte = 16 g + 64 v   (an instruction covers 64 contiguous bytes of each of its 16 rows: channel permutation 8 g + 32 v)
//   run64    row = lane / 4, byte = 16 (lane % 4) + 64 v   (adjacent lanes adjacent pieces; r05's probe "16 rows x 64 B")
//   run128   row = lane / 8, byte = 16 (lane % 8)          (what an LDS-staged epilogue issues: 8 rows x 128 B)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lmp tools/lane_map_probe.hip && /tmp/lmp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint4 relu4(uint4 a) {
  auto f = [](uint32_t u) { return (u & 0x80008000u) ? (u & 0x7fff7fffu) : u; };
  return make_uint4(f(a.x), f(a.y), f(a.z), f(a.w));
}

template <int MAP, bool LOAD, bool STORE>
__global__ __launch_bounds__(256, 2) void panel_kernel(const char* __restrict__ r, char* __restrict__ o, int M, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int cb = blockIdx.y * 512;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int row, off;
      if (MAP == 0) { row = (it >> 1) * 16 + q; off = g * 32 + (it & 1) * 16; }
      else if (MAP == 1) { row = (it >> 1) * 16 + q; off = g * 16 + (it & 1) * 64; }
      else if (MAP == 2) { row = (it >> 1) * 16 + (lane >> 2); off = (lane & 3) * 16 + (it & 1) * 64; }
      else { row = it * 8 + (lane >> 3); off = (lane & 7) * 16; }
      int m = m0 + row;
      m = m < M ? m : M - 1;
      const long a = (long)m * N * 2 + (cb + c * 64) * 2 + off;
      uint4 v = make_uint4(lane, c, it, 0x3c003c00u);
      if (LOAD) v = *reinterpret_cast<const uint4*>(r + a);
      if (STORE) *reinterpret_cast<uint4*>(o + a) = relu4(v);
      else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    __builtin_amdgcn_s_barrier();
  }
  if (!STORE && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *reinterpret_cast<uint4*>(o) = acc;
}

int main() {
  const int M = 143640, N = 1024;
  const size_t bytes = (size_t)M * N * 2;
  char *r, *o;
  hipMalloc(&r, bytes); hipMalloc(&o, bytes);
  hipMemset(r, 0x3c, bytes);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  auto time = [&](const char* name, double moved, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(s);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-60s %7.1f us  %6.0f GB/s\n", name, ms / 20 * 1e3, moved / (ms / 20 * 1e-3) / 1e9);
  };
  const dim3 grid((M + 127) / 128, 2), blk(256);
#define RUN3(MAPID, NAME) \
  time(NAME " load + store", 2.0 * bytes, [&] { hipLaunchKernelGGL((panel_kernel<MAPID, true, true>), grid, blk, 0, 0, r, o, M, N); }); \
  time(NAME " load only", 1.0 * bytes, [&] { hipLaunchKernelGGL((panel_kernel<MAPID, true, false>), grid, blk, 0, 0, r, o, M, N); }); \
  time(NAME " store only", 1.0 * bytes, [&] { hipLaunchKernelGGL((panel_kernel<MAPID, false, true>), grid, blk, 0, 0, r, o, M, N); });
  RUN3(0, "mfma32 (expand.hip today: 16 B at 32 B stride)")
  RUN3(1, "mfma64 (64 contiguous B per row and instruction)")
  RUN3(2, "run64  (adjacent lanes adjacent pieces, 64 B runs)")
  RUN3(3, "run128 (8 rows x 128 B: an LDS-staged epilogue)")
  return 0;
}
