// Lab (NOT the product path: the contract of the path is the reference's fp32 arithmetic): a dense linear
//   C[M,Nc] = A[M,K] * W[Nc,K]^T
// with both fp32 operands split into THREE bf16 terms (x = hi + mid + lo, 8 + 8 + 8 mantissa bits) and six
// v_mfma_f32_32x32x16_bf16 products per k-block (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the three dropped terms
// are below 2^-24 relative), fp32 accumulation.  Question: how close to the fp32 result is it (against an fp64
// product), and how fast against the production fp32-MFMA kernels of gemm.hip, on the fragment-batch products and on
// the six Chignolin ones?  The weights are split once on the host (they are constants of a checkpoint); the
// activations are split by the kernel on their way into LDS (ablation "A pre-split": planes read from global).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value tools/lab/bf16x3_lab.hip -o tools/lab/bf16x3_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../ai2bmd_amd/csrc/gemm.hip"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16l __attribute__((ext_vector_type(16)));
typedef float f32x4l __attribute__((ext_vector_type(4)));

// ---- host-side split (round-to-nearest-even, like v_cvt_pk_bf16_f32) ----
static uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(r >> 16);
}
static float bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---- device-side split of 8 consecutive k-values into the three planes ----
__device__ __forceinline__ void split8(const f32x4l x0, const f32x4l x1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = t < 4 ? x0[t] : x1[t - 4];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[t] = h;
    mid[t] = m;
    lo[t] = (__bf16)r2;
  }
}

// LDS plane: [128 rows][4 chunks of 16 B = 8 bf16], chunk c of row r at r * 64 + ((c ^ ((r >> 1) & 3)) << 4):
// the 32 rows a half-wave reads at one chunk index land in distinct 16-byte slots of every 128-byte bank row
__device__ __forceinline__ int plane_at(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 3)) << 4); }

// 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA tiles), BK = 32, single LDS stage + register prefetch.
// PRESPLIT: the A planes come from global memory (ablation: the cost of splitting in the kernel).
template <bool PRESPLIT>
__global__ __launch_bounds__(256) void k_bf16x3(const float* __restrict__ A, const uint16_t* __restrict__ Ah,
                                                const uint16_t* __restrict__ Am, const uint16_t* __restrict__ Al,
                                                const uint16_t* __restrict__ Wh, const uint16_t* __restrict__ Wm,
                                                const uint16_t* __restrict__ Wl, float* __restrict__ C, int M, int Nc,
                                                int K, int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 8192];
  unsigned char* const lA = lds;
  unsigned char* const lB = lds + 3 * 8192;
  const int tiles_n = Nc / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int row0 = tm * 128, col0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi5 = lane >> 5;
  // staging role: row lr of the tile, k-half lh (16 k-values = chunks 2 lh, 2 lh + 1)
  const int lr = tid >> 1, lh = tid & 1;
  const int ar = row0 + lr < M ? row0 + lr : M - 1;  // rows past M repeat the last row; never stored
  const float* ag = A + (size_t)ar * K + lh * 16;
  const uint16_t* agp[3] = {Ah + (size_t)ar * K + lh * 16, Am + (size_t)ar * K + lh * 16, Al + (size_t)ar * K + lh * 16};
  const uint16_t* wg[3] = {Wh + (size_t)(col0 + lr) * K + lh * 16, Wm + (size_t)(col0 + lr) * K + lh * 16,
                           Wl + (size_t)(col0 + lr) * K + lh * 16};
  f32x4l ra[4];
  bf16x8 rap[3][2];
  bf16x8 rb[3][2];
  auto gload = [&](int k0) {
    if (PRESPLIT) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c = 0; c < 2; ++c) rap[p][c] = *reinterpret_cast<const bf16x8*>(agp[p] + k0 + c * 8);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const f32x4l*>(ag + k0 + q * 4);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < 2; ++c) rb[p][c] = *reinterpret_cast<const bf16x8*>(wg[p] + k0 + c * 8);
  };
  auto sstore = [&]() {
    if (!PRESPLIT) {
      split8(ra[0], ra[1], rap[0][0], rap[1][0], rap[2][0]);
      split8(ra[2], ra[3], rap[0][1], rap[1][1], rap[2][1]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        *reinterpret_cast<bf16x8*>(lA + p * 8192 + plane_at(lr, 2 * lh + c)) = rap[p][c];
        *reinterpret_cast<bf16x8*>(lB + p * 8192 + plane_at(lr, 2 * lh + c)) = rb[p][c];
      }
  };
  f32x16l acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nkt = K / 32;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();  // the previous stage's fragments are consumed
    sstore();
    __syncthreads();
    if (kt + 1 < nkt) gload((kt + 1) * 32);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int c = kc * 2 + hi5;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(lA + p * 8192 + plane_at(r, c));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(lB + p * 8192 + plane_at(r, c));
      }
      // smallest terms first
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + i * 32 + 4 * hi5 + (r & 3) + 8 * (r >> 2);
        const int col = col0 + wn * 64 + j * 32 + l31;
        if (row < M) {
          float* cp = C + (size_t)row * Nc + col;
          *cp = accumulate ? *cp + acc[i][j][r] : acc[i][j][r];
        }
      }
}

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

int main(int argc, char** argv) {
  const bool small = argc > 1 && !strcmp(argv[1], "small");
  std::vector<Shape> shapes;
  if (!small)
    shapes = {{1000000, 768, 256, 0}, {1000000, 512, 256, 0}, {490000, 1280, 256, 0},
              {1000000, 256, 512, 0}, {1000000, 256, 768, 0}, {490000, 256, 1280, 0}};
  else
    shapes = {{6687, 768, 256, 0}, {6687, 512, 256, 0}, {3128, 1280, 256, 0},
              {6687, 256, 512, 0}, {6687, 256, 768, 0}, {3128, 256, 1280, 0}};
  printf("%8s %5s %5s | %-34s %9s %8s | error against the fp64 product on 256 sample rows: max|d| rms(d) / rms(C)\n", "M",
         "Nc", "K", "kernel", "us", "TFLOP/s");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.Nc * s.K, nc = (size_t)s.M * s.Nc;
    srand(1);
    auto gauss = [] {
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    std::vector<float> hb(nb);
    for (auto& v : hb) v = gauss() / sqrtf((float)s.K);
    std::vector<uint16_t> hw[3];
    for (int p = 0; p < 3; ++p) hw[p].resize(nb);
    for (size_t i = 0; i < nb; ++i) {
      const uint16_t h = bf16_rne(hb[i]);
      const float r1 = hb[i] - bf16_to_f(h);
      const uint16_t m = bf16_rne(r1);
      const float r2 = r1 - bf16_to_f(m);
      hw[0][i] = h;
      hw[1][i] = m;
      hw[2][i] = bf16_rne(r2);
    }
    // A: one block of 2^22 gaussian values, tiled (full-mantissa data; the sample rows below lie in the first block)
    const size_t blk = std::min<size_t>(na, (size_t)1 << 22);
    std::vector<float> ha(blk);
    for (auto& v : ha) v = gauss();
    std::vector<uint16_t> hap[3];
    for (int p = 0; p < 3; ++p) hap[p].resize(blk);
    for (size_t i = 0; i < blk; ++i) {
      const uint16_t h = bf16_rne(ha[i]);
      const float r1 = ha[i] - bf16_to_f(h);
      const uint16_t m = bf16_rne(r1);
      const float r2 = r1 - bf16_to_f(m);
      hap[0][i] = h;
      hap[1][i] = m;
      hap[2][i] = bf16_rne(r2);
    }
    float *A, *B, *C, *R;
    uint16_t *Ap[3], *Wp[3];
    hipMalloc(&A, na * 4);
    hipMalloc(&B, nb * 4);
    hipMalloc(&C, nc * 4);
    hipMalloc(&R, nc * 4);
    for (int p = 0; p < 3; ++p) {
      hipMalloc(&Ap[p], na * 2);
      hipMalloc(&Wp[p], nb * 2);
      hipMemcpy(Wp[p], hw[p].data(), nb * 2, hipMemcpyHostToDevice);
    }
    for (size_t off = 0; off < na; off += blk) {
      const size_t cnt = std::min(blk, na - off);
      hipMemcpy(A + off, ha.data(), cnt * 4, hipMemcpyHostToDevice);
      for (int p = 0; p < 3; ++p) hipMemcpy(Ap[p] + off, hap[p].data(), cnt * 2, hipMemcpyHostToDevice);
    }
    hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemset(C, 0, nc * 4);
    hipMemset(R, 0, nc * 4);
    const double fl = 2.0 * s.M * s.Nc * s.K;
    const int reps = s.M > 100000 ? 5 : 50;
    // fp64 product of the first 256 rows
    const int SR = std::min(256, s.M);
    std::vector<double> ref((size_t)SR * s.Nc);
    double rms_c = 0;
    for (int r = 0; r < SR; ++r)
      for (int n = 0; n < s.Nc; ++n) {
        double acc = 0;
        for (int k = 0; k < s.K; ++k) acc += (double)ha[(size_t)r * s.K + k] * (double)hb[(size_t)n * s.K + k];
        ref[(size_t)r * s.Nc + n] = acc;
        rms_c += acc * acc;
      }
    rms_c = sqrt(rms_c / ((double)SR * s.Nc));
    auto report = [&](const char* name, float* out, const std::function<void()>& f) {
      hipMemset(out, 0, nc * 4);
      const double us = time_us(f, reps);
      hipMemset(out, 0, nc * 4);
      f();
      hipDeviceSynchronize();
      hipError_t e = hipGetLastError();
      std::vector<float> hc((size_t)SR * s.Nc);
      hipMemcpy(hc.data(), out, hc.size() * 4, hipMemcpyDeviceToHost);
      double mx = 0, sq = 0;
      for (size_t i = 0; i < hc.size(); ++i) {
        const double d = fabs((double)hc[i] - ref[i]);
        mx = d > mx || d != d ? (d != d ? 1e30 : d) : mx;
        sq += d * d;
      }
      printf("%8d %5d %5d | %-34s %9.1f %8.1f | %.3e  %.3e  (rms C %.3f)%s\n", s.M, s.Nc, s.K, name, us, fl / us * 1e-6,
             mx, sqrt(sq / hc.size()) / rms_c, rms_c, e == hipSuccess ? "" : "  HIP ERROR");
    };
    report("production fp32 MFMA (gemm.hip)", R,
           [&] { vsn::launch_gemm(0, A, s.K, B, s.K, R, s.Nc, nullptr, s.M, nullptr, s.Nc, s.K, 0); });
    const dim3 grid(((s.M + 127) / 128) * (s.Nc / 128));
    report("3 x bf16 split, 6 products", C, [&] {
      hipLaunchKernelGGL(k_bf16x3<false>, grid, dim3(256), 0, 0, A, Ap[0], Ap[1], Ap[2], Wp[0], Wp[1], Wp[2], C, s.M, s.Nc,
                         s.K, 0);
    });
    report("  ablation: A planes pre-split", C, [&] {
      hipLaunchKernelGGL(k_bf16x3<true>, grid, dim3(256), 0, 0, A, Ap[0], Ap[1], Ap[2], Wp[0], Wp[1], Wp[2], C, s.M, s.Nc,
                         s.K, 0);
    });
    hipFree(A);
    hipFree(B);
    hipFree(C);
    hipFree(R);
    for (int p = 0; p < 3; ++p) {
      hipFree(Ap[p]);
      hipFree(Wp[p]);
    }
  }
  return 0;
}
