// The small per-atom pieces at the two ends of an MD step - the fragment-geometry gather in front of the force
// evaluation and the combine behind it - as device functions, so that the stand-alone kernels (head.hip) and the
// launches fused with the integrator halves (md.hip) run the very same arithmetic in the same order.
#pragma once
#include "common.h"

namespace vsn {

// device views of the two plans (engine.hip owns the handles)
struct CombineView {
  int device, n_prot, n_e;
  const int *off, *rows, *e_idx;
  const float *sign, *e_sign;
};
struct FragView {
  int device, n;
  const int *src, *acc, *tow;
  const float* len;
};

// f_prot[a] = sum_{k in [off[a], off[a+1])} sign[k] * f_frag[rows[k]]   (fixed order; combiner.py:24-41)
__device__ __forceinline__ void combine_atom(int a, const int* __restrict__ off, const int* __restrict__ rows,
                                             const float* __restrict__ sign, const float* __restrict__ f_frag,
                                             float& fx, float& fy, float& fz) {
#pragma clang fp contract(off)  // (explicit fmaf only: every kernel that inlines this rounds the same way)
  fx = fy = fz = 0.f;
  for (int k = off[a]; k < off[a + 1]; ++k) {
    const float s = sign[k];
    const size_t r = (size_t)rows[k] * 3;
    fx = fmaf(s, f_frag[r + 0], fx);
    fy = fmaf(s, f_frag[r + 1], fy);
    fz = fmaf(s, f_frag[r + 2], fz);
  }
}

// E = sum_k e_sign[k] * buf[e_idx[k]] (combiner.py:12-22), one wave, fixed order; valid in lane 0
__device__ __forceinline__ float combine_energy_wave(int lane, int n_e, const int* __restrict__ e_idx,
                                                     const float* __restrict__ e_sign, const float* __restrict__ buf) {
#pragma clang fp contract(off)
  float s = 0.f;
  for (int k = lane; k < n_e; k += 64) s = fmaf(e_sign[k], buf[e_idx[k]], s);
  return wave_sum(s);
}

// row k of the fragment batch: a copy of protein atom src[k], or (src[k] < 0) a cap hydrogen at
// acceptor + len * unit(toward - acceptor)   (distancefrag.py:35-54)
__device__ __forceinline__ void build_row(int k, const int* __restrict__ src, const int* __restrict__ acc,
                                          const int* __restrict__ tow, const float* __restrict__ len,
                                          const float* __restrict__ prot, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int s = src[k];
  float x, y, z;
  if (s >= 0) {
    x = prot[3 * (size_t)s + 0];
    y = prot[3 * (size_t)s + 1];
    z = prot[3 * (size_t)s + 2];
  } else {
    const int a = acc[k], t = tow[k];
    const float ax = prot[3 * (size_t)a + 0], ay = prot[3 * (size_t)a + 1], az = prot[3 * (size_t)a + 2];
    float dx = prot[3 * (size_t)t + 0] - ax, dy = prot[3 * (size_t)t + 1] - ay, dz = prot[3 * (size_t)t + 2] - az;
    const float sc = len[k] / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    x = fmaf(dx, sc, ax);
    y = fmaf(dy, sc, ay);
    z = fmaf(dz, sc, az);
  }
  out[3 * (size_t)k + 0] = x;
  out[3 * (size_t)k + 1] = y;
  out[3 * (size_t)k + 2] = z;
}

}  // namespace vsn

// plan views for the fused integrator launches (defined next to the handles, engine.hip)
int vsn_combine_plan_view(struct vsn_combine_plan* p, vsn::CombineView* out);
int vsn_fragplan_view(struct vsn_fragplan* p, vsn::FragView* out);
