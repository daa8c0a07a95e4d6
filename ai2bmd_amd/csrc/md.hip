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
// with F = model force - k (x - x0) when a harmonic tether is configured (the reference's restrained
// pre-equilibration uses Hookean restraints, simulator.py:139-166).
#include <cmath>
#include <vector>

#include "../../include/vsn.h"
#include "common.h"

namespace vsn {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
  const float u2 = (float)b * 2.3283064365386963e-10f;           // [0,1)
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.283185307179586f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// six standard normals for (step, atom)
__device__ __forceinline__ void normals6(unsigned long long seed, unsigned step, unsigned atom, float (&z)[6]) {
  unsigned r[4];
  philox4x32_10(atom, step, 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
  box_muller(r[0], r[1], z[0], z[1]);
  box_muller(r[2], r[3], z[2], z[3]);
  philox4x32_10(atom, step, 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
  box_muller(r[0], r[1], z[4], z[5]);
}

// single workgroup (proteins here have a few hundred to a few thousand atoms)
__global__ __launch_bounds__(1024) void k_md_half1(int n, const float* __restrict__ mass,
                                                   const float* __restrict__ c3, const float* __restrict__ c4,
                                                   const float* __restrict__ c5, float c1, float c2, float dt,
                                                   float tether_k, const float* __restrict__ x0,
                                                   unsigned long long seed, unsigned step, float* __restrict__ x,
                                                   float* __restrict__ v, const float* __restrict__ F,
                                                   float* __restrict__ rnd_vel) {
  __shared__ float red[6][16];
  __shared__ float tot[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < n; i += blockDim.x) {
    float z[6];
    normals6(seed, step, (unsigned)i, z);
    const float m = mass[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float xi = z[k], eta = z[3 + k];
      acc[k] += c5[i] * eta;                            // rnd_pos
      acc[3 + k] += (c3[i] * xi - c4[i] * eta) * m;      // rnd_vel * m
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[k][wave] = s;
  }
  __syncthreads();
  if (tid < 6) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[tid][w];
    tot[tid] = s;
  }
  __syncthreads();
  const float invn = 1.0f / (float)n;
  for (int i = tid; i < n; i += blockDim.x) {
    float z[6];
    normals6(seed, step, (unsigned)i, z);  // counter-based: regenerated, not stored
    const float m = mass[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float xi = z[k], eta = z[3 + k];
      const float rp = c5[i] * eta - tot[k] * invn;
      const float rv = (c3[i] * xi - c4[i] * eta) - tot[3 + k] * invn / m;
      const size_t a = 3 * (size_t)i + k;
      float f = F[a];
      if (tether_k != 0.f) f -= tether_k * (x[a] - x0[a]);
      const float vn = v[a] + (c1 * f / m - c2 * v[a] + rv);
      v[a] = vn;
      x[a] = x[a] + dt * vn + rp;
      rnd_vel[a] = rv;
    }
  }
}

__global__ void k_md_half2(int n, const float* __restrict__ mass, float c1, float c2, float tether_k,
                           const float* __restrict__ x0, const float* __restrict__ x, float* __restrict__ v,
                           const float* __restrict__ F, const float* __restrict__ rnd_vel) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= 3 * n) return;
  const float m = mass[a / 3];
  float f = F[a];
  if (tether_k != 0.f) f -= tether_k * (x[a] - x0[a]);
  v[a] = v[a] + (c1 * f / m - c2 * v[a] + rnd_vel[a]);
}

}  // namespace vsn

struct vsn_md {
  int device = 0, n = 0;
  float c1 = 0, c2 = 0, dt = 0, tether_k = 0;
  unsigned long long seed = 0;
  unsigned step = 0;
  float *mass = nullptr, *c3 = nullptr, *c4 = nullptr, *c5 = nullptr, *x0 = nullptr, *rnd_vel = nullptr;
};

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
            hipMalloc((void**)&p->x0, 3 * nb) == hipSuccess && hipMalloc((void**)&p->rnd_vel, 3 * nb) == hipSuccess;
  if (!ok) {
    delete p;
    return -12;
  }
  hipMemcpy(p->mass, host_mass, nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c3, c3.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c4, c4.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(p->c5, c5.data(), nb, hipMemcpyHostToDevice);
  if (host_x0) hipMemcpy(p->x0, host_x0, 3 * nb, hipMemcpyHostToDevice);
  hipMemset(p->rnd_vel, 0, 3 * nb);
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
  hipFree(p->x0);
  hipFree(p->rnd_vel);
  delete p;
}

extern "C" int vsn_md_half1(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(vsn::k_md_half1, dim3(1), dim3(1024), 0, (hipStream_t)stream, p->n, p->mass, p->c3, p->c4, p->c5,
                     p->c1, p->c2, p->dt, p->tether_k, p->x0, p->seed, p->step, dev_x, dev_v, dev_F, p->rnd_vel);
  p->step++;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_half2(vsn_md_handle p, const float* dev_x, float* dev_v, const float* dev_F, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  const int n3 = 3 * p->n;
  hipLaunchKernelGGL(vsn::k_md_half2, dim3((n3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, p->n, p->mass,
                     p->c1, p->c2, p->tether_k, p->x0, dev_x, dev_v, dev_F, p->rnd_vel);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
