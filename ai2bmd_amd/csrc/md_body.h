// The first half of the Langevin step as a device function + the integrator handle, shared by the two translation
// units that launch it: md.hip (stand-alone and with the fragment gather) and hydrogen.hip (with the fragment gather AND
// the cap-hydrogen relaxation behind it: one launch where the step used to start with two).  Every function body
// switches contraction off itself, so the arithmetic does not depend on the flags of the including file (md.hip is
// compiled with -ffp-contract=off, hydrogen.hip is not): all launches give the same bits.
#pragma once
#include "common.h"
#include "tail.h"

namespace vsn {

struct Spring;

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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

// the step's (xi, eta) of atom i: the caller's draws when it supplied them (vsn_md_set_noise: the reference feeds ASE's
// Langevin from utils/utils.py RNGPool, numpy normals - a trajectory can only be compared with ASE's on the SAME draws),
// else the counter-based generator above
__device__ __forceinline__ void noise6(unsigned long long seed, unsigned step, unsigned atom,
                                       const float* __restrict__ ext_xi, const float* __restrict__ ext_eta,
                                       float (&z)[6]) {
  if (ext_xi) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      z[k] = ext_xi[3 * (size_t)atom + k];
      z[3 + k] = ext_eta[3 * (size_t)atom + k];
    }
  } else {
    normals6(seed, step, atom, z);
  }
}

// single workgroup (proteins here have a few hundred to a few thousand atoms)
__device__ __forceinline__ void md_half1_body(int n, const float* __restrict__ mass, const float* __restrict__ c3,
                                              const float* __restrict__ c4, const float* __restrict__ c5, float c1,
                                              float c2, float dt, unsigned long long seed, unsigned step,
                                              float* __restrict__ x, float* __restrict__ v,
                                              const float* __restrict__ F, float* __restrict__ rnd_vel,
                                              const float* __restrict__ ext_xi, const float* __restrict__ ext_eta) {
#pragma clang fp contract(off)
  __shared__ float red[6][16];
  __shared__ float tot[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < n; i += blockDim.x) {
    float z[6];
    noise6(seed, step, (unsigned)i, ext_xi, ext_eta, z);
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
    noise6(seed, step, (unsigned)i, ext_xi, ext_eta, z);  // counter-based (or caller-supplied): regenerated, not stored
    const float m = mass[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float xi = z[k], eta = z[3 + k];
      const float rp = c5[i] * eta - tot[k] * invn;
      const float rv = (c3[i] * xi - c4[i] * eta) - tot[3 + k] * invn / m;
      const size_t a = 3 * (size_t)i + k;
      const float f = F[a];  // model + restraint forces at the current positions (half2 / vsn_md_restrain added them)
      const float vn = v[a] + (c1 * f / m - c2 * v[a] + rv);
      v[a] = vn;
      x[a] = x[a] + dt * vn + rp;
      rnd_vel[a] = rv;
    }
  }
}

}  // namespace vsn

struct vsn_md {
  int device = 0, n = 0;
  float c1 = 0, c2 = 0, dt = 0, tether_k = 0;
  unsigned long long seed = 0;
  unsigned step = 0;
  float *mass = nullptr, *c3 = nullptr, *c4 = nullptr, *c5 = nullptr, *rnd_vel = nullptr;
  int* sp_ptr = nullptr;          // [n+1] CSR over atoms of the restraint springs, nullptr = none
  vsn::Spring* sp = nullptr;
  float* e_r = nullptr;           // [n] per-atom restraint energy of the last evaluation
  float* obs = nullptr;           // [4] observables
  const float *ext_xi = nullptr, *ext_eta = nullptr;  // caller-supplied normal draws (vsn_md_set_noise), borrowed
};
