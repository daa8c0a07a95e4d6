// Lab: panel GEMM (ai2bmd_amd/csrc/pgemm.h: A panel stationary in LDS via LDS-DMA, packed weights streamed from L2
// into registers, no barrier in the k-loop) against the production LDS-tiled kernels of gemm.hip on the ViSNet
// product shapes of a fragment batch (and of one protein).  Results must be BITWISE equal to the production kernel
// (same k order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value tools/lab/pgemm_lab.hip -o tools/lab/pgemm_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../ai2bmd_amd/csrc/gemm.hip"
#include "../../ai2bmd_amd/csrc/pgemm.h"

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
  // big outputs: the first and the last 32M elements (the tail panel is among them)
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
  const bool small = argc > 1 && !strcmp(argv[1], "small");
  // "const": constant operands (A = 1, weights = 0.01) instead of gaussian ones - the matrix pipes draw less power on
  // data that does not toggle, so the difference to the default run is the DVFS share of the fp32-MFMA "ceiling"
  const bool constant = argc > 1 && !strcmp(argv[1], "const");
  std::vector<Shape> shapes;
  if (constant) {
    shapes = {{1000000, 768, 256, 0}, {490000, 1280, 256, 0}};
  } else if (!small) {
    // one chunk of a 4096-fragment batch: ~1M edge rows, ~61k nodes, 8 vector components
    shapes = {{1000000, 768, 256, 0}, {1000000, 512, 256, 0}, {490000, 1280, 256, 0}, {61000, 768, 256, 0},
              {1000000, 256, 512, 0}, {1000000, 256, 768, 1}, {490000, 256, 1280, 1}, {61000, 256, 768, 0}};
  } else {
    shapes = {{6687, 768, 256, 0}, {6687, 512, 256, 0}, {3128, 1280, 256, 0}, {391, 768, 256, 0},
              {6687, 256, 512, 0}, {6687, 256, 768, 1}, {3128, 256, 1280, 1}, {391, 256, 768, 0},
              {26624, 768, 256, 0}, {1000, 768, 256, 0}};
  }
  printf("%8s %5s %5s %3s | %-28s %9s %8s %9s %8s\n", "M", "Nc", "K", "acc", "kernel", "us", "TFLOP/s", "max|diff|",
         "n_diff");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.Nc * s.K, nc = (size_t)s.M * s.Nc;
    std::vector<float> hb(nb), hbp(nb + VSN_PGEMM_PAD_FLOATS, 0.f), hbias(s.Nc);
    srand(1);
    auto gauss = [] {
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    for (auto& v : hb) v = constant ? 0.01f : gauss() / sqrtf((float)s.K);
    for (int n = 0; n < s.Nc; ++n)
      for (int k = 0; k < s.K; ++k) hbp[vsn::pgemm_pack_index(n, k, s.K)] = hb[(size_t)n * s.K + k];
    for (auto& v : hbias) v = (rand() % 2001 - 1000) * 1e-3f;
    float *A, *B, *Bp, *C, *R, *C0, *bias;
    hipMalloc(&A, na * 4);
    hipMalloc(&B, nb * 4);
    hipMalloc(&Bp, hbp.size() * 4);
    hipMalloc(&C, nc * 4);
    hipMalloc(&R, nc * 4);
    hipMalloc(&C0, nc * 4);
    hipMalloc(&bias, s.Nc * 4);
    {  // full-mantissa gaussian A, generated on the host in pieces
      std::vector<float> ha(std::min<size_t>(na, (size_t)1 << 24));
      for (auto& v : ha) v = constant ? 1.0f : gauss();
      for (size_t off = 0; off < na; off += ha.size())
        hipMemcpy(A + off, ha.data(), std::min(ha.size(), na - off) * 4, hipMemcpyHostToDevice);
      std::vector<float> hc(std::min<size_t>(nc, (size_t)1 << 24));
      for (auto& v : hc) v = gauss();
      for (size_t off = 0; off < nc; off += hc.size())
        hipMemcpy(C0 + off, hc.data(), std::min(hc.size(), nc - off) * 4, hipMemcpyHostToDevice);
    }
    hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(Bp, hbp.data(), hbp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hbias.data(), s.Nc * 4, hipMemcpyHostToDevice);
    const float* bptr = (s.acc || s.Nc == 256) ? nullptr : bias;  // (the engine's dX products carry no bias)
    const double fl = 2.0 * s.M * s.Nc * s.K;
    const int reps = s.M > 100000 ? 5 : 50;
    // reference = production kernel, the variant the engine would pick
    auto reset = [&](float* dst) { hipMemcpy(dst, C0, nc * 4, hipMemcpyDeviceToDevice); };
    reset(R);
    vsn::launch_gemm(0, A, s.K, B, s.K, R, s.Nc, bptr, s.M, nullptr, s.Nc, s.K, s.acc);
    hipDeviceSynchronize();
    auto report = [&](const char* name, const std::function<void()>& f) {
      // timing on a scratch output, then one clean run for the comparison
      const double us = time_us(f, reps);
      reset(C);
      f();
      hipDeviceSynchronize();
      size_t bad = 0;
      const double md = max_diff(C, R, nc, &bad);
      printf("%8d %5d %5d %3d | %-28s %9.1f %8.1f %9.2e %8zu\n", s.M, s.Nc, s.K, s.acc, name, us, fl / us / 1e6, md, bad);
      fflush(stdout);
    };
    report("production (launch_gemm)",
           [&] { vsn::launch_gemm(0, A, s.K, B, s.K, C, s.Nc, bptr, s.M, nullptr, s.Nc, s.K, s.acc); });
    if (s.K == 256) {
#define RUN_FWD(NAME, MI_, WN_, D_, NSPLIT, MINW_, ABL_)                                                            \
  if (s.Nc % (32 * WN_) == 0) {                                                                                     \
    typedef vsn::PgemmFwd<MI_, WN_, D_, ABL_> G;                                                                    \
    auto kern = vsn::k_pgemm_fwd<MI_, WN_, D_, 1, MINW_, ABL_>;                                                     \
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_FLOATS * 4);         \
    const int grid = ((s.M + G::BM - 1) / G::BM) * (NSPLIT);                                                        \
    report(NAME, [&] {                                                                                              \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::LDS_FLOATS * 4, 0, A, s.K, Bp, C, s.Nc, bptr, s.M,       \
                         (const int*)nullptr, s.Nc, NSPLIT);                                                        \
    });                                                                                                             \
  }
      RUN_FWD("panel fwd 64r x4w D8", 2, 4, 8, 1, 2, 0)
      RUN_FWD("panel fwd 64r x8w D8", 2, 8, 8, 1, 2, 0)
      RUN_FWD("panel fwd 128r x4w D8", 4, 4, 8, 1, 1, 0)
      RUN_FWD("panel fwd 128r x8w D8", 4, 8, 8, 1, 2, 0)
      RUN_FWD("panel fwd 128r x8w D4", 4, 8, 4, 1, 2, 0)
      RUN_FWD("panel fwd 32r x4w D8", 1, 4, 8, 1, 4, 0)
      RUN_FWD("  abl: no B refill", 2, 4, 8, 1, 2, 1)
      RUN_FWD("  abl: no stores", 2, 4, 8, 1, 2, 2)
      RUN_FWD("  abl: MFMA only", 2, 4, 8, 1, 2, 7)
      if (s.M < 100000) {
        RUN_FWD("panel fwd 64r x4w D8 split2", 2, 4, 8, 2, 2, 0)
        if ((s.Nc / 128) % 3 == 0) RUN_FWD("panel fwd 64r x4w D8 split3", 2, 4, 8, 3, 2, 0)
        if ((s.Nc / 128) % 6 == 0) RUN_FWD("panel fwd 64r x4w D8 split6", 2, 4, 8, 6, 2, 0)
        RUN_FWD("panel fwd 32r x4w D8 splitmax", 1, 4, 8, (s.Nc / 128), 4, 0)
      }
    }
    if (s.Nc == 256) {
#define RUN_BWD(NAME, D_)                                                                                           \
  {                                                                                                                 \
    typedef vsn::PgemmBwd<D_> G;                                                                                    \
    auto kern = s.acc ? vsn::k_pgemm_bwd<D_, 2> : vsn::k_pgemm_bwd<D_, 0>;                                          \
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_FLOATS * 4);         \
    const int grid = (s.M + G::BM - 1) / G::BM;                                                                     \
    report(NAME, [&] {                                                                                              \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::LDS_FLOATS * 4, 0, A, s.K, Bp, C, s.Nc, s.M,             \
                         (const int*)nullptr, s.K);                                                                 \
    });                                                                                                             \
  }
      RUN_BWD("panel bwd 64r D4", 4)
      RUN_BWD("panel bwd 64r D8", 8)
    }
    hipFree(A);
    hipFree(B);
    hipFree(Bp);
    hipFree(C);
    hipFree(R);
    hipFree(C0);
    hipFree(bias);
  }
  return 0;
}
