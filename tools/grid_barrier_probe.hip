// Cost of a grid-wide barrier among persistent workgroups on gfx950 (one workgroup per CU), with and without the agent-scope
// fences that make one workgroup's global stores visible to the others across XCDs, and with each workgroup dirtying `bytes` of
// its own output before every barrier (what a producer phase leaves in its L2).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, bool fence) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fence) __threadfence();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000) break;   // never hang the box
    }
    if (fence) __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void probe(unsigned* counter, float* buf, int rounds, int fence, int dirty_floats, float* sink) {
  float acc = 0.f;
  float* mine = buf + (long)blockIdx.x * dirty_floats;
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < dirty_floats; i += 256) mine[i] = (float)(r + i);
    grid_barrier(counter, (unsigned)(r + 1) * gridDim.x, fence != 0);
    // read a neighbour's data (checks visibility when fenced)
    const float* other = buf + (long)((blockIdx.x + 37) % gridDim.x) * dirty_floats;
    if (dirty_floats) acc += other[threadIdx.x % dirty_floats] - (float)(r + threadIdx.x % dirty_floats);
    grid_barrier(counter + 32, (unsigned)(r + 1) * gridDim.x, fence != 0);
  }
  if (acc != 0.f) sink[blockIdx.x] = acc;
}

int main() {
  unsigned* counter; float *buf, *sink;
  const int G = 256;
  hipMalloc(&counter, 4096); hipMalloc(&buf, (size_t)G * 65536 * 4); hipMalloc(&sink, G * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int fence = 0; fence < 2; ++fence)
    for (int dirty : {0, 1024, 16384}) {
      hipMemset(sink, 0, G * 4);
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4096);
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, counter, buf, 200, fence, dirty, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      std::vector<float> h(G); hipMemcpy(h.data(), sink, G * 4, hipMemcpyDeviceToHost);
      int bad = 0; for (float v : h) bad += v != 0.f;
      printf("fence %d dirty %6d B per workgroup: %.2f us per barrier pair (incl. the writes), %d workgroups saw stale data\n", fence, dirty * 4, best * 1e3 / 200, bad);
    }
  return 0;
}
