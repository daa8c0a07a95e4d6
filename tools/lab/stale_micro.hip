// lab: does a kernel ever read STALE data that the kernel before it (same stream) wrote, while a neighbour process runs
// the split-3 GEMM?  P writes buf[i] = f(rep, i); Q (next launch) checks buf[(i + n/2) % n] - the element another
// workgroup, most likely on another XCD, wrote - against f(rep, .), counting mismatches.  The values change EVERY rep, so
// a stale line (from rep - 1) shows; in the engine only the buffers that are reused across layers (m, A) can show it.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/stale_micro.hip -o tools/lab/stale_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float f(int rep, size_t i) { return (float)rep + 1e-3f * (float)(i % 1000); }
__global__ void k_write(float4* buf, size_t n4, int rep) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = make_float4(f(rep, 4 * i), f(rep, 4 * i + 1), f(rep, 4 * i + 2), f(rep, 4 * i + 3));
}
__global__ void k_check(const float4* buf, size_t n4, int rep, unsigned long long* bad, int* first) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t j = (i + n4 / 2 + 12345) % n4;
    const float4 v = buf[j];
    if (v.x != f(rep, 4 * j) || v.y != f(rep, 4 * j + 1) || v.z != f(rep, 4 * j + 2) || v.w != f(rep, 4 * j + 3)) {
      if (atomicAdd(bad, 1ull) == 0) {
        first[0] = rep;
        first[1] = (int)v.x;
      }
    }
  }
}
int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t mb = argc > 2 ? atoi(argv[2]) : 256;
  const size_t n4 = mb * 1024 * 1024 / 16;
  float4* buf;
  unsigned long long* bad;
  int* first;
  hipMalloc(&buf, n4 * 16); hipMalloc(&bad, 8); hipMalloc(&first, 8);
  hipMemset(bad, 0, 8); hipMemset(first, 0, 8);
  for (int rep = 1; rep <= R; ++rep) {
    hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, buf, n4, rep);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, buf, n4, rep, bad, first);
  }
  unsigned long long hb = 0; int hf[2] = {0, 0};
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
  printf("stale_micro: %d reps of write -> check over %zu MB: %llu mismatching 16-byte elements", R, mb, hb);
  if (hb) printf(" (first in rep %d: saw the value of rep %d)", hf[0], hf[1]);
  printf("\n");
  return 0;
}
