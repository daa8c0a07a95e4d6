// lab: what a phase boundary INSIDE a persistent kernel costs on this GPU, next to the life of a dependent launch
// (would a whole ViS-MP layer as ONE persistent launch with counter barriers between its
// phases beat the four / six dependent launches it replaces at small-shard sizes?).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/grid_barrier.hip -o tools/lab/grid_barrier && tools/lab/grid_barrier
// Prints, per configuration, microseconds per barrier: all workgroups of the grid (8 XCDs), and the workgroups of ONE
// XCD only (blockIdx % 8 == 0: the dispatcher deals consecutive workgroups round-robin over the XCDs).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k_barriers(unsigned* counter, int iters, int stride, int members, float* sink) {
  if (blockIdx.x % stride) return;  // (not a member: leaves at once)
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    acc += (float)it * 1e-9f;  // (a phase would do its work here)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned want = (unsigned)members * (unsigned)it;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (acc < 0.f) *sink = acc;
}

__global__ void k_empty(float* sink) {
  if (threadIdx.x == 12345) *sink = 1.f;
}

int main() {
  unsigned* counter;
  float* sink;
  hipMalloc(&counter, 4);
  hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 2000;
  struct Cfg {
    int grid, stride, threads;
    const char* what;
  };
  std::vector<Cfg> cfgs = {{256, 1, 256, "256 WGs x 256 thr, all 8 XCDs"}, {512, 1, 256, "512 WGs x 256 thr, all 8 XCDs"},
                           {256, 8, 256, "32 WGs of ONE XCD (x 256 thr)"}, {256, 1, 1024, "256 WGs x 1024 thr, all 8 XCDs"},
                           {64, 1, 256, "64 WGs x 256 thr (8 per XCD)"}};
  for (auto& c : cfgs) {
    const int members = (c.grid + c.stride - 1) / c.stride;
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(counter, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k_barriers, dim3(c.grid), dim3(c.threads), 0, 0, counter, iters, c.stride, members, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      if (rep) printf("%-36s %7.2f us per barrier\n", c.what, 1e3f * ms / iters);
    }
  }
  // the alternative: a chain of dependent launches (each a few hundred workgroups that do nothing)
  for (int g : {64, 256, 1024}) {
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("chain of empty launches, %4d WGs      %7.2f us per launch\n", g, 1e3f * ms / iters);
  }
  return 0;
}
