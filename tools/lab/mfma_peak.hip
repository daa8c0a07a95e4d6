// Micro-benchmark: achievable v_mfma_f32_32x32x2_f32 / 16x16x4 rate on this GPU, by accumulator-chain shape and
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o tools/lab/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

// the same stream with REAL operands: eight full-mantissa values per lane, rotated - operand toggling costs power, and
// the sustained clock (so the reachable peak of a real GEMM) follows
template <int NACC>
__global__ __launch_bounds__(256) void k32r(float* out, int iters, const float* __restrict__ rnd) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    a[k] = rnd[(threadIdx.x * 16 + k) & 4095];
    b[k] = rnd[(threadIdx.x * 16 + 8 + k) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * NACC + i) & 7], b[(u * NACC + i + 3) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <typename K>
static void run_r(const char* name, K kern, int wg_per_cu, double flop_per_mfma) {
  float *out, *rnd;
  hipMalloc(&out, 4096);
  hipMalloc(&rnd, 4096 * 4);
  float h[4096];
  unsigned x = 12345u;
  for (int i = 0; i < 4096; ++i) {
    x = x * 1664525u + 1013904223u;
    h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 1e-3f;  // full mantissa, small (no overflow in the sum)
  }
  hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 40000, grid = 256 * wg_per_cu;  // ~10+ ms: long enough for the power manager to settle
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, rnd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, rnd);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 16 * flop_per_mfma;
  printf("%-34s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, wg_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
  hipFree(rnd);
}

template <typename K>
static void run(const char* name, K kern, int wg_per_cu, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 16 * flop_per_mfma;
  printf("%-34s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, wg_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run("32x32x2 1 accumulator (dependent)", k32<1>, w, 4096);
    run("32x32x2 2 accumulators", k32<2>, w, 4096);
    run("32x32x2 4 accumulators", k32<4>, w, 4096);
    run("16x16x4 1 accumulator (dependent)", k16<1>, w, 2048);
    run("16x16x4 4 accumulators", k16<4>, w, 2048);
  }
  for (int w : {1, 2, 4}) {
    run_r("32x32x2 1 acc, random operands", k32r<1>, w, 4096);
    run_r("32x32x2 4 acc, random operands", k32r<4>, w, 4096);
  }
  return 0;
}
