// Lab: does an fp32 MFMA GEMM WITHOUT LDS staging and WITHOUT k-loop barriers beat the production 64x64 /
// 4-wave / one-barrier-per-k-tile kernel on the single-protein ViSNet shapes?
//
// k_direct: a 64x64 output tile per 256-thread workgroup; the four waves split K (wave w takes the 8-wide k-steps
// w, w+4, ...), every wave accumulates the WHOLE 64x64 tile for its k-steps in four 32x32 accumulators, MFMA operands
// come straight from global memory into registers (16 B per lane = the 32x32x2 operand layout: lanes 0-31 k0..k0+3,
// lanes 32-63 k0+4..k0+7 of a row), PD steps in flight.  One LDS reduce-scatter at the end (wave q finishes quadrant q).
// Operand bytes per MFMA are half those of the LDS kernel (each fragment feeds two MFMAs per k) and there is no
// barrier until the final exchange.
//
// The production kernels are compiled in from ai2bmd_amd/csrc/gemm.hip (build variants with -DVSN_LAB_PRIO=1/2 to
// test s_setprio placements):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value tools/lab/gemm_direct.hip -o tools/lab/gemm_direct
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ai2bmd_amd/csrc/gemm.hip"
#include "gemm_streamk.hip"


template <int PD, bool XCD>
__global__ __launch_bounds__(256) void k_direct(const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                                int ldb, float* __restrict__ C, int ldc,
                                                const float* __restrict__ bias, int M, int Nc, int K) {
  __shared__ __attribute__((aligned(16))) float red[4 * 3 * 16 * 64];  // 48 KB: [dst wave][src slot][reg][lane]
  const int tiles_n = Nc / 64;
  const int live = ((M + 63) / 64) * tiles_n;
  const int bid = XCD ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int row0 = tm * 64, col0 = tn * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int r0 = row0 + l31, r1 = row0 + 32 + l31;
  r0 = r0 < M ? r0 : M - 1;
  r1 = r1 < M ? r1 : M - 1;
  const float* pa0 = A + (size_t)r0 * lda + hi * 4;
  const float* pa1 = A + (size_t)r1 * lda + hi * 4;
  const float* pb0 = Bt + (size_t)(col0 + l31) * ldb + hi * 4;
  const float* pb1 = Bt + (size_t)(col0 + 32 + l31) * ldb + hi * 4;
  const int nmy = K / 32;  // k-steps of 8 owned by this wave: steps wave, wave+4, ...
  f32x4 ra0[PD], ra1[PD], rb0[PD], rb1[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    const int it = p < nmy ? p : nmy - 1;
    const int k0 = (it * 4 + wave) * 8;
    ra0[p] = *reinterpret_cast<const f32x4*>(pa0 + k0);
    ra1[p] = *reinterpret_cast<const f32x4*>(pa1 + k0);
    rb0[p] = *reinterpret_cast<const f32x4*>(pb0 + k0);
    rb1[p] = *reinterpret_cast<const f32x4*>(pb1 + k0);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int base = 0; base < nmy; base += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int it = base + p;
      if (it < nmy) {
        const f32x4 a0 = ra0[p], a1 = ra1[p], b0 = rb0[p], b1 = rb1[p];
        const int nx = it + PD < nmy ? it + PD : nmy - 1;
        const int k0 = (nx * 4 + wave) * 8;
        ra0[p] = *reinterpret_cast<const f32x4*>(pa0 + k0);
        ra1[p] = *reinterpret_cast<const f32x4*>(pa1 + k0);
        rb0[p] = *reinterpret_cast<const f32x4*>(pb0 + k0);
        rb1[p] = *reinterpret_cast<const f32x4*>(pb1 + k0);
        __builtin_amdgcn_sched_barrier(0);
#define VSN_STEP(T)                                                                    \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.T, b0.T, acc[0][0], 0, 0, 0);    \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.T, b1.T, acc[0][1], 0, 0, 0);    \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.T, b0.T, acc[1][0], 0, 0, 0);    \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.T, b1.T, acc[1][1], 0, 0, 0);
        VSN_STEP(x)
        VSN_STEP(y)
        VSN_STEP(z)
        VSN_STEP(w)
#undef VSN_STEP
      }
    }
  }
  // reduce-scatter: wave q ends up with the full sum of quadrant q = (i, j) = (q >> 1, q & 1)
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (q != wave) {
      const int slot = wave < q ? wave : wave - 1;
      float* dst = red + ((size_t)(q * 3 + slot) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[q >> 1][q & 1][r];
    }
  __syncthreads();
  f32x16 mine;
#pragma unroll
  for (int r = 0; r < 16; ++r) mine[r] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (q == wave) mine = acc[q >> 1][q & 1];
  // fixed order: contributions of waves 0..3 (own one in its place)
  f32x16 tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) tot[r] = 0.f;
  for (int w = 0; w < 4; ++w) {
    if (w == wave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[r] += mine[r];
    } else {
      const int slot = w < wave ? w : w - 1;
      const float* src = red + ((size_t)(wave * 3 + slot) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[r] += src[r * 64];
    }
  }
  const int qi = wave >> 1, qj = wave & 1;
  const int col = col0 + qj * 32 + l31;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < M) C[(size_t)row * ldc + col] = tot[r] + bv;
  }
}

__global__ void k_naive(const float* A, const float* Bt, float* C, const float* bias, int M, int Nc, int K) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)M * Nc) return;
  const int r = (int)(gid / Nc), c = (int)(gid % Nc);
  double s = bias ? bias[c] : 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)r * K + k] * Bt[(size_t)c * K + k];
  C[gid] = (float)s;
}

struct Shape {
  int M, Nc, K;
};

template <typename F>
static double time_us(F f, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / reps;
}

static double max_err(const float* dC, const float* dR, size_t n) {
  std::vector<float> c(n), r(n);
  hipMemcpy(c.data(), dC, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), dR, n * 4, hipMemcpyDeviceToHost);
  double m = 0;
  for (size_t i = 0; i < n; ++i) m = std::max(m, (double)fabsf(c[i] - r[i]));
  return m;
}

int main() {
  const Shape shapes[] = {{6687, 768, 256}, {3128, 1280, 256}, {6687, 512, 256}, {6687, 256, 512},
                          {6687, 256, 768}, {3128, 256, 1280}, {391, 768, 256}, {26624, 768, 256}, {262144, 768, 256}, {26624, 768, 2048}, {8192, 1024, 4096}};
  printf("%7s %5s %5s | %-24s %9s %8s %9s\n", "M", "Nc", "K", "kernel", "us", "TFLOP/s", "max|err|");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.Nc * s.K, nc = (size_t)s.M * s.Nc;
    std::vector<float> ha(na), hb(nb), hbias(s.Nc);
    srand(1);
    auto gauss = [] {  // full-mantissa data: MFMA power (and so the sustained clock) depends on operand toggling
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    for (auto& v : ha) v = gauss();
    for (auto& v : hb) v = gauss() / sqrtf((float)s.K);
    for (auto& v : hbias) v = (rand() % 2001 - 1000) * 1e-3f;
    float *A, *B, *C, *R, *bias;
    hipMalloc(&A, na * 4);
    hipMalloc(&B, nb * 4);
    hipMalloc(&C, nc * 4);
    hipMalloc(&R, nc * 4);
    hipMalloc(&bias, s.Nc * 4);
    hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hbias.data(), s.Nc * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_naive, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, A, B, R, bias, s.M, s.Nc, s.K);
    const double fl = 2.0 * s.M * s.Nc * s.K;
    auto report = [&](const char* name, double us) {
      printf("%7d %5d %5d | %-24s %9.1f %8.1f %9.2e\n", s.M, s.Nc, s.K, name, us, fl / us / 1e6, max_err(C, R, nc));
    };
    const int grid = ((s.M + 63) / 64) * (s.Nc / 64);
    hipMemset(C, 0, nc * 4);
    report("prod 64x64 db", time_us([&] {
             hipLaunchKernelGGL((vsn::k_gemm<64, 64, 2, 2, true>), dim3(grid), dim3(256), 0, 0, A, s.K, B, s.K, C, s.Nc,
                                bias, s.M, nullptr, s.Nc, s.K, 0, 1, nullptr);
           }));
#if VSN_LAB_TRACE
    {  // one traced launch: per-wave phase stamps -> gpurun_out/gemm_trace_<M>.csv  (block, wave, hw_id, t0, stamps...)
      unsigned* tr;
      hipMalloc(&tr, (size_t)grid * 4 * 64 * 4);
      hipMemset(tr, 0, (size_t)grid * 4 * 64 * 4);
      hipLaunchKernelGGL((vsn::k_gemm<64, 64, 2, 2, true>), dim3(grid), dim3(256), 0, 0, A, s.K, B, s.K, C, s.Nc, bias,
                         s.M, nullptr, s.Nc, s.K, 0, 1, reinterpret_cast<float*>(tr));
      hipDeviceSynchronize();
      std::vector<unsigned> h((size_t)grid * 4 * 64);
      hipMemcpy(h.data(), tr, h.size() * 4, hipMemcpyDeviceToHost);
      char fn[128];
      snprintf(fn, sizeof fn, "gpurun_out/gemm_trace_%d_%d_%d.csv", s.M, s.Nc, s.K);
      if (FILE* f = fopen(fn, "w")) {
        for (int b = 0; b < grid; ++b)
          for (int w = 0; w < 4; ++w) {
            fprintf(f, "%d,%d", b, w);
            for (int k = 0; k < 64; ++k) fprintf(f, ",%u", h[((size_t)b * 4 + w) * 64 + k]);
            fprintf(f, "\n");
          }
        fclose(f);
      }
      hipFree(tr);
    }
    if (s.M >= 20000) {  // the batch tile (128x128, single LDS buffer) on the large grids
      const int g128 = ((s.M + 127) / 128) * (s.Nc / 128);
      unsigned* tr;
      hipMalloc(&tr, (size_t)g128 * 4 * 64 * 4);
      hipMemset(tr, 0, (size_t)g128 * 4 * 64 * 4);
      hipLaunchKernelGGL((vsn::k_gemm<128, 128, 2, 2, false>), dim3(g128), dim3(256), 0, 0, A, s.K, B, s.K, C, s.Nc,
                         bias, s.M, nullptr, s.Nc, s.K, 0, 1, reinterpret_cast<float*>(tr));
      hipDeviceSynchronize();
      std::vector<unsigned> h((size_t)g128 * 4 * 64);
      hipMemcpy(h.data(), tr, h.size() * 4, hipMemcpyDeviceToHost);
      char fn[128];
      snprintf(fn, sizeof fn, "gpurun_out/gemm_trace128_%d_%d_%d.csv", s.M, s.Nc, s.K);
      if (FILE* f = fopen(fn, "w")) {
        for (int b = 0; b < g128; ++b)
          for (int w = 0; w < 4; ++w) {
            fprintf(f, "%d,%d", b, w);
            for (int k = 0; k < 64; ++k) fprintf(f, ",%u", h[((size_t)b * 4 + w) * 64 + k]);
            fprintf(f, "\n");
          }
        fclose(f);
      }
      hipFree(tr);
    }
#endif
    {  // persistent stream-K form of the same product (one member), G = 1024 workgroups
      static int* cnt = nullptr;
      static float* ws = nullptr;
      if (!cnt) {
        hipMalloc(&cnt, (1 << 20) * 4);
        hipMemset(cnt, 0, (1 << 20) * 4);
        hipMalloc(&ws, vsn::gemm_sk_ws_floats(1024) * 4);
      }
      vsn::set_gemm_sk_workspace(cnt, 1 << 20, ws, 1024);
      hipMemset(C, 0, nc * 4);
      vsn::GemmDesc d = vsn::gemm_desc(A, s.K, B, s.K, C, s.Nc, bias, s.M, nullptr, s.Nc, s.K, 0);
      report("stream-K persistent", time_us([&] { vsn::launch_gemm_sk(0, &d, 1); }));
    }
#define RUN_PROD(NAME, ...)                                                                                       \
  hipMemset(C, 0, nc * 4);                                                                                        \
  report(NAME, time_us([&] {                                                                                      \
           hipLaunchKernelGGL((vsn::k_gemm<__VA_ARGS__>), dim3(grid), dim3(256), 0, 0, A, s.K, B, s.K, C, s.Nc, bias, \
                              s.M, nullptr, s.Nc, s.K, 0, 1, nullptr);                                            \
         }));
#undef RUN_PROD
#define RUN_PROD(NAME, BM_, BN_, ...)                                                                              \
  {                                                                                                               \
    const int grid2 = ((s.M + BM_ - 1) / BM_) * (s.Nc / BN_);                                                      \
    hipMemset(C, 0, nc * 4);                                                                                      \
    report(NAME, time_us([&] {                                                                                    \
             hipLaunchKernelGGL((vsn::k_gemm<BM_, BN_, __VA_ARGS__>), dim3(grid2), dim3(256), 0, 0, A, s.K, B, s.K, C, \
                                s.Nc, bias, s.M, nullptr, s.Nc, s.K, 0, 1, nullptr);                              \
           }));                                                                                                   \
  }
#define RUN_PROD8(NAME, BM_, BN_, WM_, WN_)                                                                        \
  {                                                                                                               \
    const int grid2 = ((s.M + BM_ - 1) / BM_) * (s.Nc / BN_);                                                      \
    hipMemset(C, 0, nc * 4);                                                                                      \
    report(NAME, time_us([&] {                                                                                    \
             hipLaunchKernelGGL((vsn::k_gemm<BM_, BN_, WM_, WN_, true>), dim3(grid2), dim3(WM_* WN_ * 64), 0, 0, A, s.K, B, \
                                s.K, C, s.Nc, bias, s.M, nullptr, s.Nc, s.K, 0, 1, nullptr);                       \
           }));                                                                                                   \
  }
    RUN_PROD("128x128 sb (batch tile)", 128, 128, 2, 2, false)
    RUN_PROD8("128x64 db 8 waves", 128, 64, 4, 2)
    RUN_PROD8("64x128 db 8 waves", 64, 128, 2, 4)
    RUN_PROD8("128x128 db 16 waves", 128, 128, 4, 4)
#define RUN_DIRECT(PD, X)                                                                                        \
  hipMemset(C, 0, nc * 4);                                                                                       \
  report("direct pd" #PD " xcd" #X, time_us([&] {                                                                 \
           hipLaunchKernelGGL((k_direct<PD, X>), dim3(grid), dim3(256), 0, 0, A, s.K, B, s.K, C, s.Nc, bias, s.M, \
                              s.Nc, s.K);                                                                         \
         }));
    hipFree(A);
    hipFree(B);
    hipFree(C);
    hipFree(R);
    hipFree(bias);
  }
  return 0;
}
