// Lab: cost of a chain of dependent tiny kernels - plain stream launches vs a captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/launch_floor.hip -o tools/lab/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(float* p) {
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}
__global__ void k_wide(float* p, int n) {  // 391 workgroups of 512 threads touching 8 KB rows each
  const size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x * 4;
  float4 v = *reinterpret_cast<float4*>(p + i);
  v.x += 1.f;
  *reinterpret_cast<float4*>(p + i) = v;
}
__global__ void k_spin(float* p, int cycles) {  // ~ a latency-bound gather kernel: every wave idles `cycles`
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {
  }
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* p;
  hipMalloc(&p, 64 << 20);
  hipMemset(p, 0, 64 << 20);
  hipStream_t st;
  hipStreamCreate(&st);
  const int N = 150, REP = 50;
  for (int wide = 0; wide < 2; ++wide) {
    auto enqueue = [&]() {
      for (int i = 0; i < N; ++i) {
        if (wide) hipLaunchKernelGGL(k_wide, dim3(391), dim3(512), 0, st, p, 0);
        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, p);
      }
    };
    enqueue();
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) enqueue();
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%s kernels, stream launches : %.2f us per kernel\n", wide ? "391x512" : "1x64", us / (REP * N));
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    enqueue();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%s kernels, hipGraph replay  : %.2f us per kernel\n", wide ? "391x512" : "1x64", us / (REP * N));
  }
  // fork/join pattern of the engine: main chain of 10-us kernels, every 4th one forks a side-stream kernel that is
  // joined two kernels later (events), stream launches vs graph replay
  {
    hipStream_t side;
    hipStreamCreate(&side);
    hipEvent_t ef, ej;
    hipEventCreateWithFlags(&ef, hipEventDisableTiming);
    hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    for (int forks = 0; forks < 2; ++forks) {
      auto enqueue = [&]() {
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(k_spin, dim3(391), dim3(512), 0, st, p, 4000);
          if (forks && (i & 3) == 0) {
            hipEventRecord(ef, st);
            hipStreamWaitEvent(side, ef, 0);
            hipLaunchKernelGGL(k_spin, dim3(391), dim3(512), 0, side, p + (32 << 18), 4000);
            hipEventRecord(ej, side);
          }
          if (forks && ((i & 3) == 2 || i == N - 1)) hipStreamWaitEvent(st, ej, 0);
        }
      };
      enqueue();
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; ++r) enqueue();
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("spin kernels %s, stream launches : %.2f us per main-chain kernel\n", forks ? "with fork/join" : "no forks", us / (REP * N));
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      enqueue();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st));
      CK(hipDeviceSynchronize());
      t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, st);
      CK(hipDeviceSynchronize());
      us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("spin kernels %s, hipGraph replay  : %.2f us per main-chain kernel\n", forks ? "with fork/join" : "no forks", us / (REP * N));
    }
  }
  return 0;
}
