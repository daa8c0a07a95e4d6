// Lab: a 256 x 256 macro tile for the plain fp32 products of a fragment batch.  The GEMM library (Tensile,
// MT256x256x32, 4 waves, one workgroup per CU) runs these shapes at 120-146 TFLOP/s where the production 128 x 128 /
// four-workgroups-per-CU kernel of gemm.hip reaches 102-112 (tools/lab/blas_fp32_names.py); this kernel asks how much
// of that a plain-HIP kernel of the same geometry gets:
//   256 x 256 tile, BK = 32 (128-byte LDS rows), 8 waves as 2 (M) x 4 (N) = 128 x 64 per wave (4 x 2 accumulators of
//   v_mfma_f32_32x32x2_f32), two LDS stages of 64 KiB filled by LDS-DMA (global_load_lds_dwordx4 issued from inline asm:
//   one instruction = 8 swizzled rows, no staging VGPRs, no ds_write pass, and the compiler does not serialise the
//   other stage's ds_reads behind it), ONE barrier per k-tile, same k order as gemm.hip => bitwise equal results.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value tools/lab/big_lab.hip -o tools/lab/big_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../ai2bmd_amd/csrc/gemm.hip"

namespace lab {



// LDS float offset of logical 16-byte chunk c of tile row r (rows of 32 floats) - the swizzle of gemm.hip
__device__ __forceinline__ int lds_at(int r, int c) { return r * 32 + ((c ^ ((r >> 1) & 7)) << 2); }

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// EPI: 0 store, 1 store + bias, 2 accumulate into C
template <int EPI>
__global__ __launch_bounds__(512) void k_big(const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb,
                                             float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int Nc,
                                             int K) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];  // 2 stages x 512 rows x 32 floats
  constexpr int STAGE = 512 * 32;
  const int tiles_n = Nc >> 8;
  const int live = ((M + 255) >> 8) * tiles_n;
  const int tile = xcd_block((int)blockIdx.x, live);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm << 8, col0 = tn << 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  // LDS-DMA role: waves 0-3 fill the A rows, waves 4-7 the B rows; instruction q of wave w = rows 8 g .. 8 g + 7 of
  // the [A ; B] image, g = 8 w + q; lane l -> row 8 g + (l >> 3), physical chunk l & 7
  const float* gp[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = ((wave & 3) * 8 + q) * 8 + (lane >> 3);  // row within the A (or B) tile
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    if (wave < 4) {
      int gr = row0 + r;
      gr = gr < M ? gr : M - 1;  // rows past the end are clamped (computed, never stored)
      gp[q] = A + (size_t)gr * lda + c * 4;
    } else {
      gp[q] = Bt + (size_t)(col0 + r) * ldb + c * 4;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned dma0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 8 * 1024);
  auto dma = [&](int stage, int k0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) glds16(gp[q] + k0, dma0 + (unsigned)stage * STAGE * 4 + q * 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment offsets (floats) within a stage: A rows wm * 128 + i * 32 + l31, B rows 256 + wn * 64 + j * 32 + l31
  int fa[4], fb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = (wm * 128 + i * 32 + l31) * 32;
#pragma unroll
  for (int j = 0; j < 2; ++j) fb[j] = (256 + wn * 64 + j * 32 + l31) * 32;
  // (rows of one fragment share (r >> 1) & 7 = (l31 >> 1) & 7: tile-row bases are multiples of 32)
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = ((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 2;

  const int nkt = K >> 5;
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const float* st = smem + (kt & 1) * STAGE;
    if (kt + 1 < nkt) dma((kt + 1) & 1, (kt + 1) * 32);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + fa[i] + fo[kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + fb[j] + fo[kk]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    // the next tile has landed (this wave's part) and every wave is done reading this one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + l31;
      const float bv = EPI == 1 ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 128 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
        if (row < M) {
          float* cp = C + (size_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (EPI == 2) v += *cp;
          *cp = v;
        }
      }
    }
}

// The library's own decomposition: 4 waves (one per SIMD), 128 x 128 per wave = 4 x 4 accumulators (256 registers).
template <int EPI>
__global__ __launch_bounds__(256) void k_big4(const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb,
                                              float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int Nc,
                                              int K) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];  // 2 stages x 512 rows x 32 floats
  constexpr int STAGE = 512 * 32;
  const int tiles_n = Nc >> 8;
  const int live = ((M + 255) >> 8) * tiles_n;
  const int tile = xcd_block((int)blockIdx.x, live);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm << 8, col0 = tn << 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  // LDS-DMA: waves 0-1 fill the A rows, waves 2-3 the B rows; 16 instructions each per stage
  const float* gp[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = ((wave & 1) * 16 + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    if (wave < 2) {
      int gr = row0 + r;
      gr = gr < M ? gr : M - 1;
      gp[q] = A + (size_t)gr * lda + c * 4;
    } else {
      gp[q] = Bt + (size_t)(col0 + r) * ldb + c * 4;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned dma0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 16 * 1024);
  auto dma = [&](int stage, int k0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) glds16(gp[q] + k0, dma0 + (unsigned)stage * STAGE * 4 + q * 1024);
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = (wm * 128 + i * 32 + l31) * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = (256 + wn * 128 + j * 32 + l31) * 32;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = ((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 2;
  const int nkt = K >> 5;
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const float* st = smem + (kt & 1) * STAGE;
    if (kt + 1 < nkt) dma((kt + 1) & 1, (kt + 1) * 32);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + fa[i] + fo[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + fb[j] + fo[kk]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 128 + j * 32 + l31;
      const float bv = EPI == 1 ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 128 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
        if (row < M) {
          float* cp = C + (size_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (EPI == 2) v += *cp;
          *cp = v;
        }
      }
    }
}

template <int EPI>
static void launch4(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, const float* bias, int M, int Nc,
                    int K) {
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute((const void*)k_big4<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    set = true;
  }
  const int grid = ((M + 255) / 256) * (Nc / 256);
  hipLaunchKernelGGL(k_big4<EPI>, dim3(grid), dim3(256), 128 * 1024, 0, A, lda, Bt, ldb, C, ldc, bias, M, Nc, K);
}

template <int EPI>
static void launch(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, const float* bias, int M, int Nc,
                   int K) {
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute((const void*)k_big<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    set = true;
  }
  const int grid = ((M + 255) / 256) * (Nc / 256);
  hipLaunchKernelGGL(k_big<EPI>, dim3(grid), dim3(512), 128 * 1024, 0, A, lda, Bt, ldb, C, ldc, bias, M, Nc, K);
}
}  // namespace lab

struct Shape {
  int M, Nc, K, acc;
};

static double time_us(const std::function<void()>& f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return 1e3 * ms / reps;
}

static double max_diff(const float* dC, const float* dR, size_t n, size_t* nbad) {
  const size_t W = (size_t)1 << 25;
  double m = 0;
  size_t bad = 0;
  for (int part = 0; part < 2; ++part) {
    const size_t off = part == 0 ? 0 : (n > 2 * W ? n - W : W);
    if (off >= n) break;
    const size_t cnt = std::min(W, n - off);
    std::vector<float> c(cnt), r(cnt);
    hipMemcpy(c.data(), dC + off, cnt * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dR + off, cnt * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < cnt; ++i) {
      const double d = fabs((double)c[i] - (double)r[i]);
      if (!(d == 0)) ++bad;
      if (d > m || d != d) m = d != d ? 1e30 : d;
    }
  }
  *nbad = bad;
  return m;
}

int main(int argc, char** argv) {
  std::vector<Shape> shapes = {{1000000, 768, 256, 0}, {1000000, 512, 256, 0}, {490000, 1280, 256, 0},
                               {61000, 768, 256, 0},   {1000000, 256, 512, 0}, {1000000, 256, 768, 1},
                               {490000, 256, 1280, 1}, {999937, 256, 768, 0}};
  printf("%8s %5s %5s %3s | %-28s %9s %8s %9s %8s\n", "M", "Nc", "K", "acc", "kernel", "us", "TFLOP/s", "max|diff|",
         "n_diff");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.Nc * s.K, nc = (size_t)s.M * s.Nc;
    std::vector<float> hb(nb), hbias(s.Nc);
    srand(1);
    auto gauss = [] {
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    for (auto& v : hb) v = gauss() / sqrtf((float)s.K);
    for (auto& v : hbias) v = (rand() % 2001 - 1000) * 1e-3f;
    float *A, *B, *C, *R, *C0, *bias;
    hipMalloc(&A, na * 4);
    hipMalloc(&B, nb * 4);
    hipMalloc(&C, nc * 4);
    hipMalloc(&R, nc * 4);
    hipMalloc(&C0, nc * 4);
    hipMalloc(&bias, s.Nc * 4);
    {
      std::vector<float> ha(std::min<size_t>(na, (size_t)1 << 24));
      for (auto& v : ha) v = gauss();
      for (size_t off = 0; off < na; off += ha.size())
        hipMemcpy(A + off, ha.data(), std::min(ha.size(), na - off) * 4, hipMemcpyHostToDevice);
      std::vector<float> hc(std::min<size_t>(nc, (size_t)1 << 24));
      for (auto& v : hc) v = gauss();
      for (size_t off = 0; off < nc; off += hc.size())
        hipMemcpy(C0 + off, hc.data(), std::min(hc.size(), nc - off) * 4, hipMemcpyHostToDevice);
    }
    hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hbias.data(), s.Nc * 4, hipMemcpyHostToDevice);
    const float* bptr = s.acc ? nullptr : bias;
    const double fl = 2.0 * s.M * s.Nc * s.K;
    const int reps = s.M > 100000 ? 5 : 50;
    auto reset = [&](float* dst) { hipMemcpy(dst, C0, nc * 4, hipMemcpyDeviceToDevice); };
    reset(R);
    vsn::launch_gemm(0, A, s.K, B, s.K, R, s.Nc, bptr, s.M, nullptr, s.Nc, s.K, s.acc);
    hipDeviceSynchronize();
    auto report = [&](const char* name, const std::function<void()>& f) {
      const double us = time_us(f, reps);
      reset(C);
      f();
      hipDeviceSynchronize();
      hipError_t e = hipGetLastError();
      size_t bad = 0;
      const double md = max_diff(C, R, nc, &bad);
      printf("%8d %5d %5d %3d | %-28s %9.1f %8.1f %9.2e %8zu%s\n", s.M, s.Nc, s.K, s.acc, name, us, fl / us * 1e-6, md, bad,
             e == hipSuccess ? "" : "  HIP ERROR");
    };
    report("production (launch_gemm)",
           [&] { vsn::launch_gemm(0, A, s.K, B, s.K, C, s.Nc, bptr, s.M, nullptr, s.Nc, s.K, s.acc); });
    report("256x256, 8 waves, LDS-DMA x2", [&] {
      if (s.acc) lab::launch<2>(A, s.K, B, s.K, C, s.Nc, nullptr, s.M, s.Nc, s.K);
      else lab::launch<1>(A, s.K, B, s.K, C, s.Nc, bias, s.M, s.Nc, s.K);
    });
    report("256x256, 4 waves (128x128 each)", [&] {
      if (s.acc) lab::launch4<2>(A, s.K, B, s.K, C, s.Nc, nullptr, s.M, s.Nc, s.K);
      else lab::launch4<1>(A, s.K, B, s.K, C, s.Nc, bias, s.M, s.Nc, s.K);
    });
    hipFree(A);
    hipFree(B);
    hipFree(C);
    hipFree(R);
    hipFree(C0);
    hipFree(bias);
  }
  return 0;
}
