// Lab: "wave-private tile" fp32 MFMA GEMM - every wave owns a TM x TN output tile, stages its own A/B k-tiles
// through a private LDS region and never meets a workgroup barrier.  Compared against the production-style
// 64x64 / 4-wave / one-barrier-per-k-tile kernel shape on the ViSNet GEMM shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_wave.hip -o tools/lab/gemm_wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[M,Nc] = A[M,K] * Bt[Nc,K]^T + bias
template <int TM, int TN, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_wave(const float* __restrict__ A, const float* __restrict__ Bt,
                                                   float* __restrict__ C, const float* __restrict__ bias, int M,
                                                   int Nc, int K) {
  constexpr int BK = 32, LS = BK + 4;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int LA = TM * 8 / 64, LB = TN * 8 / 64;  // float4 loads per lane per k-tile
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* As = smem_all + wave * (TM + TN) * LS;
  float* Bs = As + TM * LS;
  const int tiles_n = Nc / TN, tiles_m = (M + TM - 1) / TM;
  const int tile = blockIdx.x * WPB + wave;
  if (tile >= tiles_m * tiles_n) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * TM, col0 = tn * TN;
  const int l31 = lane & 31, hi = lane >> 5;

  f32x4 ra[LA], rb[LB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int f = lane + it * 64, r = f >> 3, c4 = f & 7;
      int gr = row0 + r;
      gr = gr < M ? gr : M - 1;
      ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)gr * K + k0 + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int f = lane + it * 64, r = f >> 3, c4 = f & 7;
      rb[it] = *reinterpret_cast<const f32x4*>(Bt + (size_t)(col0 + r) * K + k0 + c4 * 4);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int f = lane + it * 64, r = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(As + r * LS + c4 * 4) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int f = lane + it * 64, r = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(Bs + r * LS + c4 * 4) = rb[it];
    }
  };
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = K / BK;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    sstore();  // LDS ops of one wave execute in order: no barrier, no wait needed before the reads below
    gload((kt + 1 < nkt ? kt + 1 : kt) * BK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + (i * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + (j * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + j * 32 + l31;
      const float bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < M) C[(size_t)row * Nc + col] = acc[i][j][r] + bv;
      }
    }
}


// general workgroup tile: BM x BN per workgroup of WM x WN waves, each wave (BM/WM) x (BN/WN), double-buffered LDS,
// one barrier per k-tile (the production scheme with a free number of waves)
template <int BM, int BN, int WM, int WN, bool DB>
__global__ __launch_bounds__(64 * WM * WN) void k_wgt(const float* __restrict__ A, const float* __restrict__ Bt,
                                                      float* __restrict__ C, const float* __restrict__ bias, int M,
                                                      int Nc, int K) {
  constexpr int NT = 64 * WM * WN, BK = 32, LS = 36, STAGE = (BM + BN) * LS;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;
  static_assert(LA >= 1 && LB >= 1, "tile too small for this many threads");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tiles_n = Nc / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n, row0 = tm * BM, col0 = tn * BN;
  f32x4 ra[LA], rb[LB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int f = tid + it * NT, r = f >> 3, c4 = f & 7;
      int gr = row0 + r;
      gr = gr < M ? gr : M - 1;
      ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)gr * K + k0 + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int f = tid + it * NT, r = f >> 3, c4 = f & 7;
      rb[it] = *reinterpret_cast<const f32x4*>(Bt + (size_t)(col0 + r) * K + k0 + c4 * 4);
    }
  };
  auto sstore = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * LS;
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int f = tid + it * NT, r = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(As + r * LS + c4 * 4) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int f = tid + it * NT, r = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(Bs + r * LS + c4 * 4) = rb[it];
    }
  };
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nkt = K / BK;
  gload(0);
  if (DB) {
    sstore(0);
    gload((1 < nkt ? 1 : 0) * BK);
    __syncthreads();
  }
  for (int kt = 0; kt < nkt; ++kt) {
    if (!DB) {
      sstore(0);
      __syncthreads();
      gload((kt + 1 < nkt ? kt + 1 : kt) * BK);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float* As = smem + (DB ? (kt & 1) : 0) * STAGE;
    const float* Bs = As + BM * LS;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + (wm * TM + i * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * TN + j * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    if (DB && kt + 1 < nkt) {
      sstore((kt + 1) & 1);
      gload((kt + 2 < nkt ? kt + 2 : kt + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + wn * TN + j * 32 + l31;
      const float bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < M) C[(size_t)row * Nc + col] = acc[i][j][r] + bv;
      }
    }
}

// production-shaped reference point: 64x64 tile, 4 waves (2x2), double-buffered LDS, one barrier per k-tile
__global__ __launch_bounds__(256) void k_wg64(const float* __restrict__ A, const float* __restrict__ Bt,
                                              float* __restrict__ C, const float* __restrict__ bias, int M, int Nc,
                                              int K) {
  constexpr int BK = 32, LS = 36, BM = 64, BN = 64, STAGE = (BM + BN) * LS;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tiles_n = Nc / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n, row0 = tm * BM, col0 = tn * BN;
  f32x4 ra[2], rb[2];
  auto gload = [&](int k0) {
    for (int it = 0; it < 2; ++it) {
      const int f = tid + it * 256, r = f >> 3, c4 = f & 7;
      int gr = row0 + r;
      gr = gr < M ? gr : M - 1;
      ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)gr * K + k0 + c4 * 4);
      rb[it] = *reinterpret_cast<const f32x4*>(Bt + (size_t)(col0 + r) * K + k0 + c4 * 4);
    }
  };
  auto sstore = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * LS;
    for (int it = 0; it < 2; ++it) {
      const int f = tid + it * 256, r = f >> 3, c4 = f & 7;
      *reinterpret_cast<f32x4*>(As + r * LS + c4 * 4) = ra[it];
      *reinterpret_cast<f32x4*>(Bs + r * LS + c4 * 4) = rb[it];
    }
  };
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nkt = K / BK;
  gload(0);
  sstore(0);
  gload((1 < nkt ? 1 : 0) * BK);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const float* As = smem + (kt & 1) * STAGE;
    const float* Bs = As + BM * LS;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(As + (wm * 32 + l31) * LS + kk * 8 + hi * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      sstore((kt + 1) & 1);
      gload((kt + 2 < nkt ? kt + 2 : kt + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  const int col = col0 + wn * 32 + l31;
  const float bv = bias[col];
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (row < M) C[(size_t)row * Nc + col] = acc[r] + bv;
  }
}

__global__ void k_naive(const float* A, const float* Bt, float* C, const float* bias, int M, int Nc, int K) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)M * Nc) return;
  const int r = (int)(g / Nc), c = (int)(g % Nc);
  double s = bias[c];
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)r * K + k] * (double)Bt[(size_t)c * K + k];
  C[g] = (float)s;
}

struct Shape {
  int M, Nc, K;
};

template <typename L>
static double time_us(L launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  const int n = 20;
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / n;
}

static double max_err(const float* d_c, const float* d_ref, size_t n) {
  std::vector<float> a(n), b(n);
  hipMemcpy(a.data(), d_c, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_ref, n * 4, hipMemcpyDeviceToHost);
  double e = 0;
  for (size_t i = 0; i < n; ++i) e = fmax(e, fabs((double)a[i] - b[i]));
  return e;
}

int main() {
  const Shape all[] = {{26624, 768, 256}, {26624, 256, 768}, {6651, 768, 256}, {6651, 512, 256}, {6651, 256, 512},
                       {6651, 256, 768},  {3128, 1280, 256}, {3128, 256, 1280}, {391, 768, 256}, {170000, 768, 256}};
  const int nshape = getenv("LAB_NSHAPES") ? atoi(getenv("LAB_NSHAPES")) : 10;
  std::vector<Shape> shapes(all, all + nshape);
  printf("%7s %5s %5s | %-22s %9s %8s %9s\n", "M", "Nc", "K", "kernel", "us", "TFLOP/s", "max|err|");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.Nc * s.K, nc = (size_t)s.M * s.Nc;
    std::vector<float> ha(na), hb(nb), hbias(s.Nc);
    srand(1);
    auto gauss = [] {  // full-mantissa data: MFMA power (and so the sustained clock) depends on operand toggling
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    const bool lowbits = getenv("LAB_LOWBITS") != nullptr;
    for (auto& v : ha) v = lowbits ? (rand() % 2001 - 1000) * 1e-3f : gauss();
    for (auto& v : hb) v = (lowbits ? (rand() % 2001 - 1000) * 1e-3f : gauss()) / sqrtf((float)s.K);
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
    const bool check = s.M <= 30000;
    if (check) hipLaunchKernelGGL(k_naive, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, A, B, R, bias, s.M, s.Nc, s.K);
    const double fl = 2.0 * s.M * s.Nc * s.K;
    auto report = [&](const char* name, double us) {
      printf("%7d %5d %5d | %-22s %9.1f %8.1f %9.2e\n", s.M, s.Nc, s.K, name, us, fl / us / 1e6,
             check ? max_err(C, R, nc) : -1.0);
    };
    {
      const int grid = ((s.M + 63) / 64) * (s.Nc / 64);
      hipMemset(C, 0, nc * 4);
      report("wg64 4 waves barrier", time_us([&] { hipLaunchKernelGGL(k_wg64, dim3(grid), dim3(256), 0, 0, A, B, C, bias, s.M, s.Nc, s.K); }));
    }

#define RUN_WGT(BM, BN, WM, WN, DB)                                                                                  \
  if (s.Nc % BN == 0) {                                                                                              \
    const int grid = ((s.M + BM - 1) / BM) * (s.Nc / BN);                                                            \
    const size_t lds = (size_t)(DB ? 2 : 1) * (BM + BN) * 36 * 4;                                                    \
    hipFuncSetAttribute((const void*)k_wgt<BM, BN, WM, WN, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipMemset(C, 0, nc * 4);                                                                                         \
    report("wgt " #BM "x" #BN " w" #WM "x" #WN " db" #DB, time_us([&] {                                              \
             hipLaunchKernelGGL((k_wgt<BM, BN, WM, WN, DB>), dim3(grid), dim3(64 * WM * WN), lds, 0, A, B, C, bias, s.M, s.Nc, s.K); \
           }));                                                                                                      \
  }
    if (getenv("LAB_WGT")) {
      RUN_WGT(64, 64, 2, 2, true)
      RUN_WGT(64, 64, 2, 1, true)
      RUN_WGT(64, 64, 1, 2, true)
      RUN_WGT(64, 64, 1, 1, true)
      RUN_WGT(128, 64, 4, 2, true)
      RUN_WGT(128, 64, 4, 1, true)
      RUN_WGT(128, 128, 4, 2, true)
      RUN_WGT(128, 128, 4, 2, false)
      RUN_WGT(128, 128, 4, 4, false)
      RUN_WGT(128, 128, 2, 2, false)
      RUN_WGT(64, 128, 2, 2, true)
      RUN_WGT(64, 128, 2, 4, true)
    } else {
#define RUN_WAVE(TM, TN, WPB)                                                                                     \
  {                                                                                                               \
    const int tiles = ((s.M + TM - 1) / TM) * (s.Nc / TN);                                                        \
    const int grid = (tiles + WPB - 1) / WPB;                                                                     \
    const size_t lds = (size_t)WPB * (TM + TN) * 36 * 4;                                                          \
    hipMemset(C, 0, nc * 4);                                                                                      \
    report("wave " #TM "x" #TN " wpb" #WPB, time_us([&] {                                                         \
             hipLaunchKernelGGL((k_wave<TM, TN, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, A, B, C, bias, s.M, s.Nc, s.K); \
           }));                                                                                                   \
  }
    RUN_WAVE(32, 64, 4)
    RUN_WAVE(64, 64, 4)
    RUN_WAVE(32, 64, 1)
    RUN_WAVE(64, 64, 1)
    RUN_WAVE(64, 32, 4)
    RUN_WAVE(32, 32, 4)
    }
    hipFree(A);
    hipFree(B);
    hipFree(C);
    hipFree(R);
    hipFree(bias);
  }
  return 0;
}
