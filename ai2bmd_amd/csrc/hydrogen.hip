// Cap-hydrogen relaxation on the device (SURVEY.md 8f "next #1").
//
// Reference: every MD step `DistanceFragment.get_fragments` relaxes the hydrogens added at the cut bonds with one
// torch.optim.LBFGS.step (lr 0.1, max_iter 10, tolerance_grad 0.1, tolerance_change 0.01, no line search) on an AMBER
// force field restricted to the terms touching those hydrogens
// (/root/reference/src/Fragmentation/distancefrag.py:30-32,56-92, hydrogen/energies.py:8-61,211-242).
//
// Here: ONE kernel launch per step, one 1024-thread workgroup running the whole L-BFGS loop.  A cap hydrogen is always
// an END atom of its bond / angle / dihedral terms, so its gradient is a closed form per term; the host resolves every
// (cap, term) occurrence into a 32-byte record with the cap as first atom (all five energies are invariant under
// reversing the atom order).  Sixteen lanes gather the occurrences of one cap and reduce with xor shuffles; every dot product / norm is a fixed-order workgroup reduction in fp64 - no atomics, bit-reproducible.
// The joint problem couples all dipeptides through the scalar step length and stop tests, so it is not sharded:
// under multi-GPU every rank relaxes all caps redundantly (3*ncap unknowns, two energy evaluations in the usual case).
#include <math.h>

#include <vector>

#include "../../include/vsn.h"
#include "common.h"
#include "md_body.h"

#ifndef HOPT_THREADS
#define HOPT_THREADS 1024  // (A/B builds: -DHOPT_THREADS=512)
#endif
#define HOPT_WAVES (HOPT_THREADS / VSN_WAVE)
#define HOPT_MAX_ITER 16

struct HoptDev {
  int ncap, n_alias, max_iter;
  float lr, tol_grad, tol_change;
  const int* cap_row;    // [ncap]
  const int* occ_ptr;    // [ncap+1]
  const int4* occ_i;     // {type, a1, a2, a3}
  const float4* occ_f;   // {p0, p1, p2, weight}
  const int2* alias;     // {dst row, src row}
  float* ws;             // g, g_prev, d, Y[max_iter], S[max_iter]   (each 3*ncap)
  int ws_in_lds;         // 1: the optimiser state lives in LDS (it fits for every protein seen so far), `ws` is unused
  int* stats;            // n_iter, func_evals
  double* estats;        // loss at entry, last evaluated loss
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < HOPT_WAVES; ++w) s += sh[w];
  __syncthreads();
  return s;
}
__device__ __forceinline__ float block_max(float v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (double)v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < HOPT_WAVES; ++w) s = fmaxf(s, (float)sh[w]);
  __syncthreads();
  return s;
}
__device__ __forceinline__ float dot_ws(const float* a, const float* b, int n, double* sh) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += HOPT_THREADS) v += (double)a[i] * (double)b[i];
  return (float)block_sum(v, sh);
}

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float* p, int row) { return {p[3 * row], p[3 * row + 1], p[3 * row + 2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// energy (weighted share) and gradient w.r.t. the cap atom of one occurrence; energies.py:8-61
__device__ __forceinline__ void occurrence(const int4 I, const float4 P, const V3 x, const float* pos, float& e,
                                           V3& g) {
  if (I.x == 0) {  // bond  0.5 k (r - r0)^2
    const V3 dv = sub(x, ld3(pos, I.y));
    const float r = sqrtf(dot(dv, dv)), dr = r - P.y;
    e = 0.5f * P.x * dr * dr;
    const float s = P.x * dr / r;
    g = {s * dv.x, s * dv.y, s * dv.z};
  } else if (I.x == 1) {  // angle  0.5 k (theta - theta0)^2, theta = atan2(|v0 x v1|, v0.v1)
    const V3 pj = ld3(pos, I.y);
    const V3 v0 = sub(x, pj), v1 = sub(ld3(pos, I.z), pj);
    const V3 n = cross(v0, v1);
    const float nn = sqrtf(dot(n, n));
    const float th = atan2f(nn, dot(v0, v1)), dt = th - P.y;
    e = 0.5f * P.x * dt * dt;
    const V3 c = cross(v0, n);  // d theta / d x_i = v0 x n / (|v0|^2 |n|)
    const float s = P.x * dt / (dot(v0, v0) * nn);
    g = {s * c.x, s * c.y, s * c.z};
  } else if (I.x == 2) {  // dihedral  0.5 k (1 + cos(n phi - phase))
    const V3 p1 = ld3(pos, I.y), p2 = ld3(pos, I.z), p3 = ld3(pos, I.w);
    const V3 w0 = sub(p1, p2), w1 = sub(p1, x), w2 = sub(p3, p2);
    const V3 a1 = cross(w1, w0), a2 = cross(w0, w2);
    const float a1sq = dot(a1, a1), a2sq = dot(a2, a2), w0n = sqrtf(dot(w0, w0));
    const float i1 = 1.0f / fmaxf(sqrtf(a1sq), 1e-12f), i2 = 1.0f / fmaxf(sqrtf(a2sq), 1e-12f);  // F.normalize eps
    const V3 n1 = {a1.x * i1, a1.y * i1, a1.z * i1}, n2 = {a2.x * i2, a2.y * i2, a2.z * i2};
    const float iw = 1.0f / fmaxf(w0n, 1e-12f);
    const V3 m1 = cross(n1, V3{w0.x * iw, w0.y * iw, w0.z * iw});
    const float phi = atan2f(dot(m1, n2), dot(n1, n2));
    const float arg = P.y * phi - P.z;
    e = 0.5f * P.x * (1.0f + cosf(arg));
    const float dE = -0.5f * P.x * P.y * sinf(arg);  // dE/dphi ; dphi/dx_0 = +|w0| a1 / |a1|^2 (a1 = w1 x w0)
    const float s = dE * w0n / a1sq;
    g = {s * a1.x, s * a1.y, s * a1.z};
  } else {  // non-bonded pair: A/r^12 - B/r^6 (already / scnb) + qq / r (already / scee)
    const V3 dv = sub(x, ld3(pos, I.y));
    const float r2 = dot(dv, dv), ir2 = 1.0f / r2, ir = sqrtf(ir2);
    const float ir6 = ir2 * ir2 * ir2;
    e = (P.x * ir6 - P.y) * ir6 + P.z * ir;
    const float s = ((-12.0f * P.x * ir6 + 6.0f * P.y) * ir6 - P.z * ir) * ir2;  // (dE/dr) / r
    g = {s * dv.x, s * dv.y, s * dv.z};
  }
  e *= P.w;
}

// loss and gradient at the current cap positions (held in `pos`); returns the loss to every thread.
// 16 lanes per cap hydrogen (a cap has ~40 term occurrences): 64 caps per pass of the 1024-thread workgroup.
__device__ double evaluate(const HoptDev& a, const float* pos, float* g, double* sh) {
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
  double e_acc = 0.0;
  for (int c0 = 0; c0 < a.ncap; c0 += HOPT_THREADS / 16) {  // uniform trip count: the shuffles below need every lane
    const int c = c0 + grp;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (c < a.ncap) {
      const V3 x = ld3(pos, a.cap_row[c]);
      const int o0 = a.occ_ptr[c], o1 = a.occ_ptr[c + 1];
      for (int o = o0 + l; o < o1; o += 16) {
        float e;
        V3 gv;
        occurrence(a.occ_i[o], a.occ_f[o], x, pos, e, gv);
        e_acc += (double)e;
        gx += gv.x;
        gy += gv.y;
        gz += gv.z;
      }
    }
    gx = group_sum(gx, 16);
    gy = group_sum(gy, 16);
    gz = group_sum(gz, 16);
    if (c < a.ncap && l == 0) {
      g[3 * c] = gx;
      g[3 * c + 1] = gy;
      g[3 * c + 2] = gz;
    }
  }
  return block_sum(e_acc, sh);  // its barriers also publish g
}

__device__ __forceinline__ void hopt_body(const HoptDev& a, float* pos, float* lws) {
  __shared__ double sh[HOPT_WAVES];
  // the optimiser state (a few KB: 3 n_cap floats per vector, 3 + 2 max_iter vectors) lives in LDS: every phase of the
  // loop is a write - barrier - read of these vectors by other threads (Chignolin: 24.6 -> 23.5 us per step; what is
  // left are the two energy evaluations and ~20 fp64 workgroup reductions of a 16-wave workgroup)
  const int n = 3 * a.ncap, tid = threadIdx.x;
  float* g = a.ws_in_lds ? lws : a.ws;
  float* gp = g + n;
  float* d = gp + n;
  float* Y = d + n;
  float* S = Y + (size_t)a.max_iter * n;
  float al[HOPT_MAX_ITER], ro[HOPT_MAX_ITER];
  int nh = 0, n_iter = 0, evals = 1;
  float H = 1.0f, t = 0.f;
  const int max_eval = a.max_iter * 5 / 4;  // torch default

  double loss = evaluate(a, pos, g, sh);
  const double loss0 = loss;
  float gmax = 0.f;
  for (int i = tid; i < n; i += HOPT_THREADS) gmax = fmaxf(gmax, fabsf(g[i]));
  gmax = block_max(gmax, sh);
  // torch/optim/lbfgs.py LBFGS.step with fresh state (a new optimiser is built every call, energies.py:232-238)
  if (gmax > a.tol_grad) {
    while (n_iter < a.max_iter) {
      ++n_iter;
      if (n_iter == 1) {
        for (int i = tid; i < n; i += HOPT_THREADS) d[i] = -g[i];
      } else {
        float* y = Y + (size_t)nh * n;
        float* s = S + (size_t)nh * n;
        double ys_ = 0.0, yy_ = 0.0;
        for (int i = tid; i < n; i += HOPT_THREADS) {
          const float yi = g[i] - gp[i], si = d[i] * t;
          y[i] = yi;
          s[i] = si;
          ys_ += (double)yi * si;
          yy_ += (double)yi * yi;
        }
        const float ys = (float)block_sum(ys_, sh), yy = (float)block_sum(yy_, sh);
        if (ys > 1e-10f) {
          ro[nh] = 1.0f / ys;
          H = ys / yy;
          ++nh;
        }
        for (int i = tid; i < n; i += HOPT_THREADS) d[i] = -g[i];
        __syncthreads();
#pragma unroll 1
        for (int h = nh - 1; h >= 0; --h) {
          al[h] = dot_ws(S + (size_t)h * n, d, n, sh) * ro[h];
          const float* yh = Y + (size_t)h * n;
          for (int i = tid; i < n; i += HOPT_THREADS) d[i] -= al[h] * yh[i];
          __syncthreads();
        }
        for (int i = tid; i < n; i += HOPT_THREADS) d[i] *= H;
        __syncthreads();
#pragma unroll 1
        for (int h = 0; h < nh; ++h) {
          const float be = dot_ws(Y + (size_t)h * n, d, n, sh) * ro[h];
          const float* sv = S + (size_t)h * n;
          for (int i = tid; i < n; i += HOPT_THREADS) d[i] += sv[i] * (al[h] - be);
          __syncthreads();
        }
      }
      double gsum_ = 0.0, gtd_ = 0.0;
      for (int i = tid; i < n; i += HOPT_THREADS) {
        const float gi = g[i];
        gp[i] = gi;
        gsum_ += (double)fabsf(gi);
        gtd_ += (double)gi * d[i];
      }
      const double prev_loss = loss;
      const float gsum = (float)block_sum(gsum_, sh), gtd = (float)block_sum(gtd_, sh);
      t = (n_iter == 1) ? fminf(1.0f, 1.0f / gsum) * a.lr : a.lr;
      if (gtd > -a.tol_change) break;
      float dmax = 0.f;
      for (int i = tid; i < n; i += HOPT_THREADS) {
        const float st = d[i] * t;
        dmax = fmaxf(dmax, fabsf(st));
        pos[3 * a.cap_row[i / 3] + i % 3] += st;
      }
      __syncthreads();
      bool opt_cond = false;
      if (n_iter != a.max_iter) {
        loss = evaluate(a, pos, g, sh);
        ++evals;
        gmax = 0.f;
        for (int i = tid; i < n; i += HOPT_THREADS) gmax = fmaxf(gmax, fabsf(g[i]));
        gmax = block_max(gmax, sh);
        opt_cond = gmax <= a.tol_grad;
      }
      if (n_iter == a.max_iter) break;
      if (evals >= max_eval) break;
      if (opt_cond) break;
      dmax = block_max(dmax, sh);
      if (dmax <= a.tol_change) break;
      if (fabs((double)(float)loss - (double)(float)prev_loss) < (double)a.tol_change) break;
    }
  }
  __syncthreads();
  // ACE-NME fragments take their atoms from the relaxed dipeptides (distancefrag.py:82 `positions[fragments_index]`)
  for (int i = tid; i < a.n_alias; i += HOPT_THREADS) {
    const int2 p = a.alias[i];
    pos[3 * p.x] = pos[3 * p.y];
    pos[3 * p.x + 1] = pos[3 * p.y + 1];
    pos[3 * p.x + 2] = pos[3 * p.y + 2];
  }
  if (tid == 0) {
    a.stats[0] = n_iter;
    a.stats[1] = evals;
    a.estats[0] = loss0;
    a.estats[1] = loss;
  }
}

__global__ __launch_bounds__(HOPT_THREADS) void k_hopt(HoptDev a, float* pos) {
  extern __shared__ __attribute__((aligned(16))) float lws[];
  hopt_body(a, pos, lws);
}

// The START of an MD step in one launch (round 6): first Langevin half + fragment-geometry gather of the new positions
// (= k_md_half1_build, md.hip) + the cap-hydrogen relaxation of those fragments (= k_hopt).  All three are single
// 1024-thread workgroups and strictly serial, so as two launches the step began with 9 + 24 us of which ~6 us were a
// second launch's life.  Same device functions, contraction fixed per function body: the same bits as the two
// launches (tests/test_gpu_pipeline.py::test_fused_integrator_ends_...).
__global__ __launch_bounds__(HOPT_THREADS) void k_md_half1_build_hopt(
    int n, const float* __restrict__ mass, const float* __restrict__ c3, const float* __restrict__ c4,
    const float* __restrict__ c5, float c1, float c2, float dt, unsigned long long seed, unsigned step, float* x,
    float* __restrict__ v, const float* __restrict__ F, float* __restrict__ rnd_vel, const float* __restrict__ ext_xi,
    const float* __restrict__ ext_eta, vsn::FragView fp, float* frag_pos, HoptDev a) {
  extern __shared__ __attribute__((aligned(16))) float lws[];
  vsn::md_half1_body(n, mass, c3, c4, c5, c1, c2, dt, seed, step, x, v, F, rnd_vel, ext_xi, ext_eta);
  __syncthreads();
  const float* xn = x;
  for (int k = threadIdx.x; k < fp.n; k += blockDim.x) vsn::build_row(k, fp.src, fp.acc, fp.tow, fp.len, xn, frag_pos);
  __syncthreads();
  hopt_body(a, frag_pos, lws);
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
struct vsn_hopt {
  int device = 0;
  size_t lds_bytes = 0;
  HoptDev dev{};
  std::vector<void*> allocs;
};

template <typename T>
static T* upload(vsn_hopt* p, const std::vector<T>& v, bool& ok) {
  void* d = nullptr;
  const size_t nb = (v.empty() ? 1 : v.size()) * sizeof(T);
  if (hipMalloc(&d, nb) != hipSuccess) {
    ok = false;
    return nullptr;
  }
  p->allocs.push_back(d);
  if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
  return (T*)d;
}

extern "C" int vsn_hopt_create(vsn_hopt_handle* out, int device_id, const vsn_hopt_terms* t) {
  if (!out || !t || t->n_cap <= 0 || !t->cap_rows || !t->occ_ptr || t->max_iter < 1 || t->max_iter > HOPT_MAX_ITER)
    return -22;
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  const int ncap = t->n_cap, nocc = t->occ_ptr[ncap];
  std::vector<int> cap_row(ncap), occ_ptr(t->occ_ptr, t->occ_ptr + ncap + 1);
  for (int c = 0; c < ncap; ++c) {
    if (t->cap_rows[c] < 0 || t->cap_rows[c] >= t->n_rows) return -22;
    cap_row[c] = (int)t->cap_rows[c];
  }
  std::vector<int4> oi(nocc);
  std::vector<float4> of(nocc);
  for (int c = 0; c < ncap; ++c) {
    for (int o = occ_ptr[c]; o < occ_ptr[c + 1]; ++o) {
      const int ty = t->occ_type[o], k = t->occ_term[o];
      const bool last = t->occ_end[o] != 0;
      int4 I = {ty, -1, -1, -1};
      float4 P = {0.f, 0.f, 0.f, t->occ_w[o]};
      int self = -1;
      if (ty == 0) {
        if (k < 0 || k >= t->n_bond) return -22;
        self = last ? t->bond_j[k] : t->bond_i[k];
        I.y = last ? t->bond_i[k] : t->bond_j[k];
        P.x = t->bond_k[k];
        P.y = t->bond_r0[k];
      } else if (ty == 1) {
        if (k < 0 || k >= t->n_angle) return -22;
        self = last ? t->angle_k[k] : t->angle_i[k];
        I.y = t->angle_j[k];
        I.z = last ? t->angle_i[k] : t->angle_k[k];
        P.x = t->angle_kf[k];
        P.y = t->angle_th0[k];
      } else if (ty == 2) {
        if (k < 0 || k >= t->n_dihedral) return -22;
        self = last ? t->dih_l[k] : t->dih_i[k];
        I.y = last ? t->dih_k[k] : t->dih_j[k];
        I.z = last ? t->dih_j[k] : t->dih_k[k];
        I.w = last ? t->dih_i[k] : t->dih_l[k];
        P.x = t->dih_kf[k];
        P.y = t->dih_per[k];
        P.z = t->dih_phase[k];
      } else if (ty == 3) {
        if (k < 0 || k >= t->n_pair) return -22;
        self = last ? t->pair_j[k] : t->pair_i[k];
        I.y = last ? t->pair_i[k] : t->pair_j[k];
        P.x = t->pair_a[k] / t->scnb;
        P.y = t->pair_b[k] / t->scnb;
        P.z = t->pair_qq[k] / t->scee;
      } else {
        return -22;
      }
      if (self != cap_row[c]) return -22;  // the occurrence list must name the cap atom as an END atom of the term
      for (int v : {I.y, ty >= 1 && ty <= 2 ? I.z : 0, ty == 2 ? I.w : 0})
        if (v < 0 || v >= t->n_rows) return -22;
      oi[o] = I;
      of[o] = P;
    }
  }
  std::vector<int2> alias;
  if (t->alias)
    for (int64_t r = 0; r < t->n_rows; ++r)
      if (t->alias[r] >= 0) {
        if (t->alias[r] >= t->n_rows) return -22;
        alias.push_back({(int)r, (int)t->alias[r]});
      }
  vsn_hopt* p = new vsn_hopt();
  p->device = device_id;
  bool ok = true;
  HoptDev& d = p->dev;
  d.ncap = ncap;
  d.n_alias = (int)alias.size();
  d.max_iter = t->max_iter;
  d.lr = t->lr;
  d.tol_grad = t->tolerance_grad;
  d.tol_change = t->tolerance_change;
  d.cap_row = upload(p, cap_row, ok);
  d.occ_ptr = upload(p, occ_ptr, ok);
  d.occ_i = upload(p, oi, ok);
  d.occ_f = upload(p, of, ok);
  d.alias = upload(p, alias, ok);
  const size_t ws_floats = (size_t)3 * ncap * (3 + 2 * (size_t)t->max_iter);
  d.ws = upload(p, std::vector<float>(ws_floats, 0.f), ok);
  d.ws_in_lds = ws_floats * sizeof(float) <= 60 * 1024 ? 1 : 0;
  p->lds_bytes = d.ws_in_lds ? ws_floats * sizeof(float) : 0;
  d.stats = upload(p, std::vector<int>(2, 0), ok);
  d.estats = upload(p, std::vector<double>(2, 0.0), ok);
  if (!ok) {
    vsn_hopt_destroy(p);
    return -12;
  }
  *out = p;
  return 0;
}

extern "C" void vsn_hopt_destroy(vsn_hopt_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  for (void* a : p->allocs) hipFree(a);
  delete p;
}

extern "C" int vsn_hopt_run(vsn_hopt_handle p, float* dev_frag_pos, void* stream) {
  if (!p || !dev_frag_pos) return -22;
  hipLaunchKernelGGL(k_hopt, dim3(1), dim3(HOPT_THREADS), p->lds_bytes, (hipStream_t)stream, p->dev, dev_frag_pos);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_md_half1_build_relax(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F,
                                         vsn_fragplan_handle plan, float* dev_frag_pos, vsn_hopt_handle hopt,
                                         void* stream) {
  if (!p || !plan || !dev_frag_pos || !hopt) return -22;
  vsn::FragView fv;
  if (vsn_fragplan_view(plan, &fv) || fv.device != p->device || hopt->device != p->device) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipLaunchKernelGGL(k_md_half1_build_hopt, dim3(1), dim3(HOPT_THREADS), hopt->lds_bytes, (hipStream_t)stream, p->n,
                     p->mass, p->c3, p->c4, p->c5, p->c1, p->c2, p->dt, p->seed, p->step, dev_x, dev_v, dev_F,
                     p->rnd_vel, p->ext_xi, p->ext_eta, fv, dev_frag_pos, hopt->dev);
  p->step++;
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_hopt_stats(vsn_hopt_handle p, int32_t* host_iters_evals, double* host_loss_first_last,
                              void* stream) {
  if (!p) return -22;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -5;
  if (host_iters_evals && hipMemcpy(host_iters_evals, p->dev.stats, 2 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return -5;
  if (host_loss_first_last &&
      hipMemcpy(host_loss_first_last, p->dev.estats, 2 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return -5;
  return 0;
}
