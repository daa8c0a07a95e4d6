// MM non-bonded term between atoms that do not share a dipeptide (SURVEY.md 8f "next #2"):
// all ordered pairs (src j -> dst i), Lennard-Jones + Coulomb with analytic forces, restating
// /root/reference/src/Calculators/nonbonded.py:33-63 with the pair list of
// /root/reference/src/AIMD/protein.py:133-151 (every i != j not in `exclude_pair`,
// distancefrag.py:355-363 = pairs inside one dipeptide).
//
//   sigma_ij = (sigma_i + sigma_j)/2 * nm ; eps_ij = sqrt(eps_i eps_j)
//   c6 = (sigma_ij^2 / d^2)^3 ; E_lj = 4 eps (c6^2 - c6) ; F_lj(on i) = 24 eps (2 c6^2 - c6)/d^2 * (x_i - x_j)
//   E_c = k q_i q_j / d ; F_c(on i) = E_c / d^2 * (x_i - x_j)
//   E = (sum over ordered pairs)/2 * (kJ/mol) ; F *= (kJ/mol)
// The exclusion test replaces the reference's materialised O(N^2) pair list: every atom carries the ids of
// the (<= 4) dipeptides it belongs to; a pair is excluded iff the two id sets intersect.
// One wave per destination atom, lanes stride over sources, fixed-order wave reduction (deterministic).
#include <cmath>
#include <vector>

#include "../../include/vsn.h"
#include "common.h"

namespace vsn {

__global__ __launch_bounds__(256) void k_mm_nonbonded(int n, const float* __restrict__ pos,
                                                      const float* __restrict__ q, const float* __restrict__ sig,
                                                      const float* __restrict__ eps, const int4* __restrict__ grp,
                                                      float kcoul, float nm, float kjmol, int accumulate,
                                                      float* __restrict__ f, float* __restrict__ e_atom) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  const float qi = q[i], si = sig[i], ei = eps[i];
  const int4 gi = grp[i];
  float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
  for (int j = lane; j < n; j += 64) {
    if (j == i) continue;
    const int4 gj = grp[j];
    const int a[4] = {gi.x, gi.y, gi.z, gi.w}, b[4] = {gj.x, gj.y, gj.z, gj.w};
    bool excl = false;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) excl |= (a[u] >= 0 && a[u] == b[v]);
    if (excl) continue;
    const float vx = xi - pos[3 * j], vy = yi - pos[3 * j + 1], vz = zi - pos[3 * j + 2];
    const float d2 = vx * vx + vy * vy + vz * vz;
    const float d = sqrtf(d2);
    const float sij = 0.5f * (si + sig[j]) * nm;
    const float eij = sqrtf(ei * eps[j]);
    const float s2 = sij * sij / d2;
    const float c6 = s2 * s2 * s2;
    const float c12 = c6 * c6;
    const float ec = kcoul * qi * q[j] / d;
    en += 4.0f * eij * (c12 - c6) + ec;
    const float fs = 24.0f * eij * (2.0f * c12 - c6) / d2 + ec / d2;
    fx += fs * vx;
    fy += fs * vy;
    fz += fs * vz;
  }
  fx = wave_sum(fx);
  fy = wave_sum(fy);
  fz = wave_sum(fz);
  en = wave_sum(en);
  if (lane == 0) {
    float* o = f + 3 * (size_t)i;
    if (accumulate) {
      o[0] += fx * kjmol;
      o[1] += fy * kjmol;
      o[2] += fz * kjmol;
    } else {
      o[0] = fx * kjmol;
      o[1] = fy * kjmol;
      o[2] = fz * kjmol;
    }
    e_atom[i] = en;
  }
}

__global__ __launch_bounds__(1024) void k_mm_energy(int n, const float* __restrict__ e_atom, float scale,
                                                    float* __restrict__ e_out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += e_atom[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    e_out[0] = t * scale;
  }
}

}  // namespace vsn

struct vsn_mm {
  int device = 0, n = 0;
  float *q = nullptr, *sig = nullptr, *eps = nullptr, *e_atom = nullptr;
  int4* grp = nullptr;
  float kcoul = 0, nm = 10.0f, kjmol = 0;
};

extern "C" int vsn_mm_create(vsn_mm_handle* out, int device_id, int64_t n, const float* host_charge,
                             const float* host_sigma, const float* host_epsilon, const int32_t* host_groups4) {
  if (!out || n <= 0 || !host_charge || !host_sigma || !host_epsilon || !host_groups4) return -22;
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  vsn_mm* p = new vsn_mm();
  p->device = device_id;
  p->n = (int)n;
  // ase.units (CODATA 2014): _e, _Nav, _eps0 ; C = 1/_e ; kJ = 1000/_e ; mol = _Nav ; nm = 10 Angstrom
  const double e_ = 1.6021766208e-19, Nav = 6.022140857e23, eps0 = 8.854187817620389e-12, pi = 3.14159265358979323846;
  const double Cc = 1.0 / e_, kJ = 1000.0 / e_, mol = Nav;
  p->kcoul = (float)(1.0 / (4.0 * pi * eps0) * 10e6 * mol / (Cc * Cc));  // nonbonded.py:18
  p->kjmol = (float)(kJ / mol);
  const size_t nb = (size_t)n * sizeof(float);
  bool ok = hipMalloc((void**)&p->q, nb) == hipSuccess && hipMalloc((void**)&p->sig, nb) == hipSuccess &&
            hipMalloc((void**)&p->eps, nb) == hipSuccess && hipMalloc((void**)&p->e_atom, nb) == hipSuccess &&
            hipMalloc((void**)&p->grp, (size_t)n * sizeof(int4)) == hipSuccess;
  if (!ok) {
    delete p;
    return -12;
  }
  hipMemcpy(p->q, host_charge, nb, hipMemcpyHostToDevice);
  hipMemcpy(p->sig, host_sigma, nb, hipMemcpyHostToDevice);
  hipMemcpy(p->eps, host_epsilon, nb, hipMemcpyHostToDevice);
  hipMemcpy(p->grp, host_groups4, (size_t)n * sizeof(int4), hipMemcpyHostToDevice);
  *out = p;
  return 0;
}

extern "C" void vsn_mm_destroy(vsn_mm_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipFree(p->q);
  hipFree(p->sig);
  hipFree(p->eps);
  hipFree(p->e_atom);
  hipFree(p->grp);
  delete p;
}

extern "C" int vsn_mm_forces(vsn_mm_handle p, const float* dev_pos, float* dev_e, float* dev_f, int accumulate,
                             void* stream) {
  if (!p || !dev_pos || !dev_e || !dev_f) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vsn::k_mm_nonbonded, dim3((p->n + 3) / 4), dim3(256), 0, st, p->n, dev_pos, p->q, p->sig, p->eps,
                     p->grp, p->kcoul, p->nm, p->kjmol, accumulate, dev_f, p->e_atom);
  hipLaunchKernelGGL(vsn::k_mm_energy, dim3(1), dim3(1024), 0, st, p->n, p->e_atom, 0.5f * p->kjmol, dev_e);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
