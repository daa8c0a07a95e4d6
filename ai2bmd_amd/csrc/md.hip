// Langevin integrator step on the device (SURVEY.md 8f "next #4"): the ASE 3.22 `Langevin.step`
// update that /root/reference/src/AIMD/simulator.py:96-116 configures (fixcm=True, two normal
// draws per atom per step - RNGPool(count=2), simulator.py:108), restated as two kernels around the
// force evaluation so positions / velocities / forces never leave HBM and one MD step costs two
// launches instead of ~25 elementwise ones.
//
//   half1:  xi, eta ~ N(0,1)            (Philox4x32-10, counter = (step, atom), key = seed)
//           rnd_pos = c5 eta ; rnd_vel = c3 xi - c4 eta ; both made centre-of-mass free
//           v += c1 F/m - c2 v + rnd_vel ; x += dt v + rnd_pos
//   (forces at the new x)
//   half2:  v += c1 F/m - c2 v + rnd_vel
// Restraints (the reference attaches ASE `Hookean` constraints: position restraints of all QM atoms during the
// pre-equilibration stages, simulator.py:139-166, and X-H bond restraints with --hydrogen-constraints, :168-180):
// a per-atom list of springs (towards a fixed point, or towards another atom) applied when the distance exceeds
// its threshold rt:  F += k (r - rt) * unit(towards),  E += k (r - rt)^2 / 2 per spring.  half2 evaluates them at the
// new positions, ADDS them into the force array it was handed (so the array the caller sees and the next half1
// reads is model + restraints, like atoms.get_forces() in ASE) and leaves the per-atom restraint energy for
// vsn_md_observe.  Per-atom lists, fixed order: no atomics, bit-reproducible.  A `tether_k` at creation is the
// same thing for every atom towards its start position with rt = 0.
// PARITY UNPINNED: ASE is not installed here, so the Langevin coefficients and the Hookean force law restate
// ASE 3.22 from its published algorithm; tests check them against an independent torch restatement and analytic
// properties only.
#include <cmath>
#include <vector>

#include "../../include/vsn.h"
#include "common.h"
#include "kernels.h"
#include "tail.h"
#include "md_body.h"

namespace vsn {

__global__ __launch_bounds__(1024) void k_md_half1(int n, const float* __restrict__ mass,
                                                   const float* __restrict__ c3, const float* __restrict__ c4,
                                                   const float* __restrict__ c5, float c1, float c2, float dt,
                                                   unsigned long long seed, unsigned step, float* __restrict__ x,
                                                   float* __restrict__ v, const float* __restrict__ F,
                                                   float* __restrict__ rnd_vel, const float* __restrict__ ext_xi,
                                                   const float* __restrict__ ext_eta) {
  md_half1_body(n, mass, c3, c4, c5, c1, c2, dt, seed, step, x, v, F, rnd_vel, ext_xi, ext_eta);
}

// half1 + the fragment-geometry gather of the NEW positions (the first kernel of the force evaluation that follows:
// vsn_build_fragments), one launch: the workgroup's own position writes are visible to it after the barrier
__global__ __launch_bounds__(1024) void k_md_half1_build(int n, const float* __restrict__ mass,
                                                         const float* __restrict__ c3, const float* __restrict__ c4,
                                                         const float* __restrict__ c5, float c1, float c2, float dt,
                                                         unsigned long long seed, unsigned step,
                                                         float* __restrict__ x, float* __restrict__ v,
                                                         const float* __restrict__ F, float* __restrict__ rnd_vel,
                                                         const float* __restrict__ ext_xi,
                                                         const float* __restrict__ ext_eta, FragView fp,
                                                         float* __restrict__ frag_pos) {
  md_half1_body(n, mass, c3, c4, c5, c1, c2, dt, seed, step, x, v, F, rnd_vel, ext_xi, ext_eta);
  __syncthreads();
  const float* xn = x;  // (x is written above: no __restrict__ promise on this read)
  for (int k = threadIdx.x; k < fp.n; k += blockDim.x) build_row(k, fp.src, fp.acc, fp.tow, fp.len, xn, frag_pos);
}

// One spring of an atom's list: towards a fixed point (partner < 0) or towards atom `partner`.
struct Spring {
  int partner;
  float ox, oy, oz, k, rt;
};

// restraint force on atom i at positions x, and its share of the restraint energy (a pair spring is listed at both
// ends and gives half its energy to each)
__device__ __forceinline__ void restraint_of(int i, const float* __restrict__ x, const int* __restrict__ ptr,
                                             const Spring* __restrict__ sp, float (&f)[3], float& e) {
#pragma clang fp contract(off)
  f[0] = f[1] = f[2] = 0.f;
  e = 0.f;
  const float xi = x[3 * (size_t)i], yi = x[3 * (size_t)i + 1], zi = x[3 * (size_t)i + 2];
  for (int t = ptr[i]; t < ptr[i + 1]; ++t) {
    const Spring s = sp[t];
    float dx, dy, dz;
    if (s.partner >= 0) {
      dx = x[3 * (size_t)s.partner] - xi, dy = x[3 * (size_t)s.partner + 1] - yi, dz = x[3 * (size_t)s.partner + 2] - zi;
    } else {
      dx = s.ox - xi, dy = s.oy - yi, dz = s.oz - zi;
    }
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    if (r > s.rt && r > 0.f) {
      const float mag = s.k * (r - s.rt) / r;
      f[0] += mag * dx, f[1] += mag * dy, f[2] += mag * dz;
      e += (s.partner >= 0 ? 0.25f : 0.5f) * s.k * (r - s.rt) * (r - s.rt);
    }
  }
}

// F += restraints(x); e_r[i] = restraint energy share of atom i   (used once, for the forces of the start geometry)
__global__ void k_md_restrain(int n, const float* __restrict__ x, float* __restrict__ F, const int* __restrict__ ptr,
                              const Spring* __restrict__ sp, float* __restrict__ e_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float f[3], e;
  restraint_of(i, x, ptr, sp, f, e);
#pragma unroll
  for (int k = 0; k < 3; ++k) F[3 * (size_t)i + k] += f[k];
  e_r[i] = e;
}

// second half of the step for atom i, model force fm: restraints at the new positions are added (the sum is what the
// caller stores as F), then v += c1 F/m - c2 v + rnd_vel.  One statement of the arithmetic for the stand-alone and
// the fused kernel, contraction pinned (explicit fmaf only), so that both round identically.
__device__ __forceinline__ void md_half2_atom(int i, const float (&fm)[3], const float* __restrict__ mass, float c1,
                                              float c2, const float* __restrict__ x, float* __restrict__ v,
                                              float* __restrict__ F, bool store_f, const float* __restrict__ rnd_vel,
                                              const int* __restrict__ ptr, const Spring* __restrict__ sp,
                                              float* __restrict__ e_r) {
#pragma clang fp contract(off)
  float fr[3] = {0.f, 0.f, 0.f}, e = 0.f;
  if (ptr) restraint_of(i, x, ptr, sp, fr, e);
  const float m = mass[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const size_t a = 3 * (size_t)i + k;
    const float f = fm[k] + fr[k];
    if (store_f) F[a] = f;
    const float vo = v[a];
    v[a] = vo + (fmaf(-c2, vo, c1 * f / m) + rnd_vel[a]);
  }
  if (ptr) e_r[i] = e;
}

__global__ void k_md_half2(int n, const float* __restrict__ mass, float c1, float c2, const float* __restrict__ x,
                           float* __restrict__ v, float* __restrict__ F, const float* __restrict__ rnd_vel,
                           const int* __restrict__ ptr, const Spring* __restrict__ sp, float* __restrict__ e_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float fm[3] = {F[3 * (size_t)i], F[3 * (size_t)i + 1], F[3 * (size_t)i + 2]};
  md_half2_atom(i, fm, mass, c1, c2, x, v, F, ptr != nullptr, rnd_vel, ptr, sp, e_r);
}

// the combine of the force evaluation (fragment rows -> protein atoms, total energy; vsn_combine_with_energy) + half2,
// one launch: atom a's thread sums its fragment rows, writes F (model) and goes on as k_md_half2 does
__global__ void k_md_combine_half2(int n, const float* __restrict__ mass, float c1, float c2,
                                   const float* __restrict__ x, float* __restrict__ v, float* __restrict__ F,
                                   const float* __restrict__ rnd_vel, const int* __restrict__ ptr,
                                   const Spring* __restrict__ sp, float* __restrict__ e_r, CombineView cp,
                                   const float* __restrict__ buf, float* __restrict__ e_out) {
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const float s = combine_energy_wave((int)threadIdx.x, cp.n_e, cp.e_idx, cp.e_sign, buf);
    if (threadIdx.x == 0) *e_out = s;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float fm[3];
  combine_atom(i, cp.off, cp.rows, cp.sign, buf, fm[0], fm[1], fm[2]);
  md_half2_atom(i, fm, mass, c1, c2, x, v, F, true, rnd_vel, ptr, sp, e_r);
}

// observables without leaving HBM: out[0] = kinetic energy sum m v^2 / 2, out[1] = restraint energy,
// out[2] = temperature 2 Ekin / (3 n kB)  (ASE Atoms.get_temperature with 3 n degrees of freedom)
__global__ __launch_bounds__(1024) void k_md_observe(int n, const float* __restrict__ mass, const float* __restrict__ v,
                                                     const float* __restrict__ e_r, float kB, float* __restrict__ out) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ek = 0.f, er = 0.f;
  for (int i = tid; i < n; i += blockDim.x) {
    const float vx = v[3 * (size_t)i], vy = v[3 * (size_t)i + 1], vz = v[3 * (size_t)i + 2];
    ek += 0.5f * mass[i] * (vx * vx + vy * vy + vz * vz);
    if (e_r) er += e_r[i];
  }
  ek = wave_sum(ek);
  er = wave_sum(er);
  if (lane == 0) red[0][wave] = ek, red[1][wave] = er;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += red[0][w], b += red[1][w];
    out[0] = a;
    out[1] = b;
    out[2] = 2.0f * a / (3.0f * (float)n * kB);
  }
}

// The stand-alone forms of the two step ends live HERE, next to the launches that fuse them with the integrator
// halves: this translation unit is the one compiled with -ffp-contract=off (ai2bmd_amd/build.py, PER_FILE_FLAGS), so
// every kernel that shares per-atom arithmetic with another launch rounds the same way by construction, not by a
// per-function pragma.
static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

// ---- overlap-force recombination (Calculators/combiner.py:38-39) ----------------
// f_prot[a] = sum_{k in [off[a], off[a+1])} sign[k] * f_frag[rows[k]]   (fixed order)
// optionally also E = sum_k e_sign[k] * buf[e_idx[k]] (combiner.py:19), reduced by the first wave in a fixed order
__global__ void k_combine(int n_prot, const int* __restrict__ off, const int* __restrict__ rows,
                          const float* __restrict__ sign, const float* __restrict__ f_frag,
                          float* __restrict__ f_prot, int n_e, const int* __restrict__ e_idx,
                          const float* __restrict__ e_sign, float* __restrict__ e_out) {
  if (e_out && blockIdx.x == 0 && threadIdx.x < 64) {
    const float s = combine_energy_wave((int)threadIdx.x, n_e, e_idx, e_sign, f_frag);
    if (threadIdx.x == 0) *e_out = s;
  }
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_prot) return;
  float fx, fy, fz;
  combine_atom(a, off, rows, sign, f_frag, fx, fy, fz);
  f_prot[3 * (size_t)a + 0] = fx;
  f_prot[3 * (size_t)a + 1] = fy;
  f_prot[3 * (size_t)a + 2] = fz;
}

int launch_combine(hipStream_t st, int n_prot, const int* off, const int* rows, const float* sign,
                   const float* f_frag, float* f_prot, int n_e, const int* e_idx, const float* e_sign, float* e_out) {
  if (n_prot <= 0) return 0;
  hipLaunchKernelGGL(k_combine, dim3(nblk(n_prot)), dim3(256), 0, st, n_prot, off, rows, sign, f_frag, f_prot, n_e,
                     e_idx, e_sign, e_out);
  return 0;
}

// ---- per-step fragment geometry (distancefrag.py:35-54): gather + cap hydrogens ---------
__global__ void k_build_fragments(int n, const int* __restrict__ src, const int* __restrict__ acc,
                                  const int* __restrict__ tow, const float* __restrict__ len,
                                  const float* __restrict__ prot, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  build_row(k, src, acc, tow, len, prot, out);
}

int launch_build_fragments(hipStream_t st, int n, const int* src, const int* acc, const int* tow, const float* len,
                           const float* prot, float* out) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_build_fragments, dim3(nblk(n)), dim3(256), 0, st, n, src, acc, tow, len, prot, out);
  return 0;
}

}  // namespace vsn


static int set_springs(vsn_md* p, const std::vector<std::vector<vsn::Spring>>& per_atom) {
  hipFree(p->sp_ptr);
  hipFree(p->sp);
  p->sp_ptr = nullptr;
  p->sp = nullptr;
  size_t tot = 0;
  for (auto& v : per_atom) tot += v.size();
  hipMemset(p->e_r, 0, (size_t)p->n * sizeof(float));
  if (tot == 0) return 0;
  std::vector<int> ptr((size_t)p->n + 1, 0);
  std::vector<vsn::Spring> flat;
  flat.reserve(tot);
  for (int i = 0; i < p->n; ++i) {
    ptr[(size_t)i + 1] = ptr[(size_t)i] + (int)per_atom[(size_t)i].size();
    flat.insert(flat.end(), per_atom[(size_t)i].begin(), per_atom[(size_t)i].end());
  }
  if (hipMalloc((void**)&p->sp_ptr, ptr.size() * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&p->sp, flat.size() * sizeof(vsn::Spring)) != hipSuccess)
    return -12;
  hipMemcpy(p->sp_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice);
  hipMemcpy(p->sp, flat.data(), flat.size() * sizeof(vsn::Spring), hipMemcpyHostToDevice);
  return 0;
}

extern "C" int vsn_md_create(vsn_md_handle* out, int device_id, int64_t n, const float* host_mass, float dt,
                             float kT, float friction, uint64_t seed, float tether_k, const float* host_x0) {
  if (!out || n <= 0 || !host_mass || dt <= 0.f || (tether_k != 0.f && !host_x0)) return -22;
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  vsn_md* p = new vsn_md();
  p->device = device_id;
  p->n = (int)n;
  p->dt = dt;
  p->tether_k = tether_k;
  p->seed = seed;
  const double fr = friction, d = dt;
  p->c1 = (float)(d / 2.0 - d * d * fr / 8.0);
  p->c2 = (float)(d * fr / 2.0 - d * d * fr * fr / 8.0);
  std::vector<float> c3((size_t)n), c4((size_t)n), c5((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const double sigma = std::sqrt(2.0 * kT * fr / host_mass[i]);
    c3[(size_t)i] = (float)(std::sqrt(d) * sigma / 2.0 - std::pow(d, 1.5) * fr * sigma / 8.0);
    const double c5d = std::pow(d, 1.5) * sigma / (2.0 * std::sqrt(3.0));
    c5[(size_t)i] = (float)c5d;
    c4[(size_t)i] = (float)(fr / 2.0 * c5d);
  }
  const size_t nb = (size_t)n * sizeof(float);
  bool ok = hipMalloc((void**)&p->mass, nb) == hipSuccess && hipMalloc((void**)&p->c3, nb) == hipSuccess &&
            hipMalloc((void**)&p->c4, nb) == hipSuccess && hipMalloc((void**)&p->c5, nb) == hipSuccess &&
            hipMalloc((void**)&p->rnd_vel, 3 * nb) == hipSuccess && hipMalloc((void**)&p->e_r, nb) == hipSuccess &&
            hipMalloc((void**)&p->obs, 4 * sizeof(float)) == hipSuccess;
  if (!ok) {
    delete p;
    return -12;
  }
  hipMemcpy(p->mass, host_mass, nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c3, c3.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c4, c4.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c5, c5.data(), nb, hipMemcpyHostToDevice);
  hipMemset(p->rnd_vel, 0, 3 * nb);
  hipMemset(p->e_r, 0, nb);
  hipMemset(p->obs, 0, 4 * sizeof(float));
  if (tether_k != 0.f) {  // every atom on a spring to its start position, rt = 0
    std::vector<std::vector<vsn::Spring>> per((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      per[(size_t)i].push_back(vsn::Spring{-1, host_x0[3 * i], host_x0[3 * i + 1], host_x0[3 * i + 2], tether_k, 0.f});
    int rc = set_springs(p, per);
    if (rc) {
      vsn_md_destroy(p);
      return rc;
    }
  }
  *out = p;
  return 0;
}

extern "C" void vsn_md_destroy(vsn_md_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipFree(p->mass);
  hipFree(p->c3);
  hipFree(p->c4);
  hipFree(p->c5);
  hipFree(p->rnd_vel);
  hipFree(p->sp_ptr);
  hipFree(p->sp);
  hipFree(p->e_r);
  hipFree(p->obs);
  delete p;
}

extern "C" int vsn_md_half1(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_half1, dim3(1), dim3(1024), 0, (hipStream_t)stream, p->n, p->mass, p->c3, p->c4, p->c5,
                     p->c1, p->c2, p->dt, p->seed, p->step, dev_x, dev_v, dev_F, p->rnd_vel, p->ext_xi, p->ext_eta);
  p->step++;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_set_noise(vsn_md_handle p, const float* dev_xi, const float* dev_eta) {
  if (!p || ((dev_xi == nullptr) != (dev_eta == nullptr))) return -22;
  p->ext_xi = dev_xi;   // borrowed: [n, 3] floats each, read by every following first half until reset with (NULL, NULL)
  p->ext_eta = dev_eta;
  return 0;
}

extern "C" int vsn_md_half2(vsn_md_handle p, const float* dev_x, float* dev_v, float* dev_F, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_half2, dim3((p->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p->n, p->mass,
                     p->c1, p->c2, dev_x, dev_v, dev_F, p->rnd_vel, p->sp_ptr, p->sp, p->e_r);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_half1_build(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F,
                                  vsn_fragplan_handle plan, float* dev_frag_pos, void* stream) {
  if (!p || !plan || !dev_frag_pos) return -22;
  vsn::FragView fv;
  if (vsn_fragplan_view(plan, &fv) || fv.device != p->device) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_half1_build, dim3(1), dim3(1024), 0, (hipStream_t)stream, p->n, p->mass, p->c3, p->c4,
                     p->c5, p->c1, p->c2, p->dt, p->seed, p->step, dev_x, dev_v, dev_F, p->rnd_vel, p->ext_xi, p->ext_eta,
                     fv, dev_frag_pos);
  p->step++;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_combine_half2(vsn_md_handle p, vsn_combine_handle plan, const float* dev_buf, float* dev_F,
                                    float* dev_e_out, const float* dev_x, float* dev_v, void* stream) {
  if (!p || !plan || !dev_buf || !dev_F || !dev_e_out) return -22;
  vsn::CombineView cv;
  if (vsn_combine_plan_view(plan, &cv) || cv.device != p->device || cv.n_prot != p->n || cv.n_e <= 0) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_combine_half2, dim3((p->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p->n,
                     p->mass, p->c1, p->c2, dev_x, dev_v, dev_F, p->rnd_vel, p->sp_ptr, p->sp, p->e_r, cv, dev_buf,
                     dev_e_out);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_set_restraints(vsn_md_handle p, int64_t n_point, const int64_t* atom, const float* origin3,
                                     const float* k_point, const float* rt_point, int64_t n_pair, const int64_t* a1,
                                     const int64_t* a2, const float* k_pair, const float* rt_pair) {
  if (!p || n_point < 0 || n_pair < 0) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  std::vector<std::vector<vsn::Spring>> per((size_t)p->n);
  for (int64_t t = 0; t < n_point; ++t) {
    if (atom[t] < 0 || atom[t] >= p->n) return -22;
    per[(size_t)atom[t]].push_back(
        vsn::Spring{-1, origin3[3 * t], origin3[3 * t + 1], origin3[3 * t + 2], k_point[t], rt_point[t]});
  }
  for (int64_t t = 0; t < n_pair; ++t) {
    if (a1[t] < 0 || a1[t] >= p->n || a2[t] < 0 || a2[t] >= p->n || a1[t] == a2[t]) return -22;
    per[(size_t)a1[t]].push_back(vsn::Spring{(int)a2[t], 0.f, 0.f, 0.f, k_pair[t], rt_pair[t]});
    per[(size_t)a2[t]].push_back(vsn::Spring{(int)a1[t], 0.f, 0.f, 0.f, k_pair[t], rt_pair[t]});
  }
  hipDeviceSynchronize();  // the previous lists may still be read by a queued step
  return set_springs(p, per);
}

extern "C" int vsn_md_restrain(vsn_md_handle p, const float* dev_x, float* dev_F, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  if (!p->sp_ptr) return 0;
  hipLaunchKernelGGL(vsn::k_md_restrain, dim3((p->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p->n, dev_x,
                     dev_F, p->sp_ptr, p->sp, p->e_r);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_observe(vsn_md_handle p, const float* dev_v, float kB, float* dev_out3, void* stream) {
  if (!p || !dev_v || !dev_out3) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_observe, dim3(1), dim3(1024), 0, (hipStream_t)stream, p->n, p->mass, dev_v,
                     p->sp_ptr ? p->e_r : nullptr, kB, dev_out3);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
