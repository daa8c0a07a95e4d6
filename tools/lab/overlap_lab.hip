// Lab: can an MFMA-bound kernel and an HBM-bound kernel on two streams really overlap on MI355X, or does one starve
// the other (CU slots, power)?  Times (a) the production fp32 MFMA GEMM alone, (b) a streaming float4 copy alone,
// (c) both launched back to back on two streams.  Perfect overlap: t(c) = max(a, b); none: a + b.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value tools/lab/overlap_lab.hip -o tools/lab/overlap_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ai2bmd_amd/csrc/gemm.hip"

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float4 v = src[i];
    v.x += 1.0f;
    dst[i] = v;
  }
}
// gather-like: every wave reads 1-KiB rows at pseudo-random row indices of a table that fits L2/MALL poorly
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, const int* __restrict__ idx,
                                                float4* __restrict__ dst, int nrows_out, int per) {
  const int lane = threadIdx.x & 63;
  const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  for (int r = w; r < nrows_out; r += nw) {
    float4 acc = {0, 0, 0, 0};
    for (int k = 0; k < per; ++k) {
      const int j = idx[r * per + k];
      const float4 v = tab[(size_t)j * 64 + lane];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    dst[(size_t)r * 64 + lane] = acc;
  }
}

int main() {
  const int M = 700000, Nc = 768, K = 256;
  const size_t na = (size_t)M * K, nb = (size_t)Nc * K, nc = (size_t)M * Nc;
  float *A, *B, *C, *bias;
  hipMalloc(&A, na * 4);
  hipMalloc(&B, nb * 4);
  hipMalloc(&C, nc * 4);
  hipMalloc(&bias, Nc * 4);
  std::vector<float> h((size_t)1 << 24);
  srand(1);
  for (auto& v : h) v = (float)((rand() % 20001) - 10000) * 1e-4f;
  for (size_t off = 0; off < na; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), na - off) * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), Nc * 4, hipMemcpyHostToDevice);
  const size_t ncopy = (size_t)3 << 28;  // 3 GiB of float4 = 12 GiB read + 12 GiB written?  no: elements of 16 B
  float4 *S, *Dst;
  const size_t n4 = (size_t)600 << 20 >> 4;  // 600 MiB per array
  (void)ncopy;
  hipMalloc(&S, n4 * 16);
  hipMalloc(&Dst, n4 * 16);
  hipMemset(S, 0, n4 * 16);
  // gather set-up: table of 40k rows x 1 KiB (40 MB), 700k output rows x 9 gathered rows each
  const int trows = 40000, orows = 700000, per = 9;
  float4* T;
  int* idx;
  float4* G;
  hipMalloc(&T, (size_t)trows * 1024);
  hipMalloc(&idx, (size_t)orows * per * 4);
  hipMalloc(&G, (size_t)orows * 1024);
  hipMemset(T, 0, (size_t)trows * 1024);
  {
    std::vector<int> hi((size_t)orows * per);
    for (size_t i = 0; i < hi.size(); ++i) hi[i] = (int)((i / per / 17 * 20 + (rand() % 20)) % trows);  // fragment-local
    hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  }
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto gemm = [&](hipStream_t st) { vsn::launch_gemm(st, A, K, B, K, C, Nc, bias, M, nullptr, Nc, K, 0); };
  auto copy = [&](hipStream_t st) { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, st, S, Dst, n4); };
  auto gath = [&](hipStream_t st) {
    hipLaunchKernelGGL(k_gather, dim3(256 * 8), dim3(256), 0, st, T, idx, G, orows, per);
  };
  auto wall = [&](const char* name, int reps, auto&& f) {
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(s1, e0, 0);
    hipStreamWaitEvent(s2, e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEvent_t j1, j2;
    hipEventCreate(&j1);
    hipEventCreate(&j2);
    hipEventRecord(j1, s1);
    hipEventRecord(j2, s2);
    hipStreamWaitEvent(0, j1, 0);
    hipStreamWaitEvent(0, j2, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms per rep\n", name, ms / reps);
    hipEventDestroy(j1);
    hipEventDestroy(j2);
    return ms / reps;
  };
  const int R = 6;
  const double a = wall("GEMM 700k x 768 x 256 alone (stream 1)", R, [&] { gemm(s1); });
  const double b = wall("copy 600 MiB -> 600 MiB alone (stream 2)", R, [&] { copy(s2); });
  const double g = wall("gather 700k x 9 rows alone (stream 2)", R, [&] { gath(s2); });
  const double c = wall("GEMM (s1) || copy (s2)", R, [&] { gemm(s1); copy(s2); });
  const double d = wall("GEMM (s1) || gather (s2)", R, [&] { gemm(s1); gath(s2); });
  const double c2 = wall("GEMM (s1) || 2 x copy (s2)", R, [&] { gemm(s1); copy(s2); copy(s2); });
  for (int n : {4, 8, 12}) {
    char nm[64];
    snprintf(nm, sizeof nm, "GEMM (s1) || %d x copy (s2)", n);
    const double t = wall(nm, R, [&] { gemm(s1); for (int i = 0; i < n; ++i) copy(s2); });
    printf("   -> %.3f vs sum %.3f, max %.3f\n", t, a + n * b, std::max(a, n * b));
  }
  for (int n : {2, 4}) {
    char nm[64];
    snprintf(nm, sizeof nm, "GEMM (s1) || %d x gather (s2)", n);
    const double t = wall(nm, R, [&] { gemm(s1); for (int i = 0; i < n; ++i) gath(s2); });
    printf("   -> %.3f vs sum %.3f, max %.3f\n", t, a + n * g, std::max(a, n * g));
  }
  {
    const double t = wall("2 x GEMM (s1) || 4 x gather + 8 x copy (s2)", R, [&] { gemm(s1); gemm(s1); for (int i = 0; i < 4; ++i) gath(s2); for (int i = 0; i < 8; ++i) copy(s2); });
    printf("   -> %.3f vs sum %.3f, max %.3f\n", t, 2 * a + 4 * g + 8 * b, std::max(2 * a, 4 * g + 8 * b));
  }
  printf("GEMM %.1f TFLOP/s alone; copy %.2f TB/s alone; gather %.2f TB/s (L2 side) alone\n", 2.0 * M * Nc * K / a / 1e9,
         2.0 * n4 * 16 / b / 1e9, (double)orows * (per + 1) * 1024 / g / 1e9);
  printf("overlap GEMM||copy: %.3f vs sum %.3f, max %.3f  -> hidden %.0f %% of the shorter\n", c, a + b, std::max(a, b),
         100.0 * (a + b - c) / std::min(a, b));
  printf("overlap GEMM||gather: %.3f vs sum %.3f, max %.3f -> hidden %.0f %% of the shorter\n", d, a + g, std::max(a, g),
         100.0 * (a + g - d) / std::min(a, g));
  printf("overlap GEMM||2copy: %.3f vs sum %.3f, max %.3f\n", c2, a + 2 * b, std::max(a, 2 * b));
  return 0;
}
