// Read-out head: two GatedEquivariantBlocks + per-fragment energy sum, and its
// reverse pass.  Reference: ViSNet/model/output_modules.py:52-62 (block forward),
// :136-140 (EquivariantScalar.pre_reduce), ViSNet/model/visnet.py:139-149
// (x*std, Atomref prior priors.py:86-87, scatter-sum per molecule, +mean).
//
// Block 1's vec2_proj / gate only feed `v.sum() * 0` (output_modules.py:140) and
// are therefore not evaluated.  The dense products run on the MFMA GEMM; the
// elementwise glue here is N-sized (negligible next to the E-sized layer work).
#include "common.h"
#include "kernels.h"

namespace vsn {

// out[i*ldo + off + c] = || p[(i*S+s)*ldp + c] ||_2 over s
__global__ void hk_norm_s(int N, int S, int C, const float* __restrict__ p, int ldp, float* __restrict__ out,
                          int ldo, int off) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)N * C) return;
  const int i = (int)(gid / C), c = (int)(gid % C);
  float s2 = 0.f;
  for (int s = 0; s < S; ++s) {
    float v = p[((size_t)i * S + s) * ldp + c];
    s2 += v * v;
  }
  out[(size_t)i * ldo + off + c] = sqrtf(s2);
}

// x1 = silu(xs) -> cat1[:, :h2] ; vec1o[s] = gate * v2[s]
template <bool GEN>
__global__ void hk_mid(int N, int S, int H, int act, const float* __restrict__ u0, const float* __restrict__ pv0,
                       int ldp, float* __restrict__ cat1, float* __restrict__ vec1o) {
  const int h2 = H / 2;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)N * h2) return;
  const int i = (int)(gid / h2), c = (int)(gid % h2);
  cat1[(size_t)i * H + c] = act_f(GEN ? act : VSN_ACT_SILU, u0[(size_t)i * H + c]);
  const float gate = u0[(size_t)i * H + h2 + c];
  for (int s = 0; s < S; ++s)
    vec1o[((size_t)i * S + s) * h2 + c] = gate * pv0[((size_t)i * S + s) * ldp + H + c];
}

// y_i = std * (wb1 . silu(a1b_i) + bb1) + atomref[z_i]   (one wave per node)
template <bool GEN>
__global__ void hk_final(int N, int h2, int act, const float* __restrict__ a1b, const float* __restrict__ wb1, float bb1,
                         float stdv, const float* __restrict__ atomref, const int* __restrict__ zi,
                         float* __restrict__ y, float* __restrict__ g_a1) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  float acc = 0.f;
  for (int c = lane; c < h2; c += 64) {
    float av, dav;
    act_both(GEN ? act : VSN_ACT_SILU, a1b[(size_t)i * h2 + c], av, dav);
    acc += av * wb1[c];
    // first step of the reverse pass rides along (dE/dy = 1): g_a1 = std * wb1 * act'(a1b)
    g_a1[(size_t)i * h2 + c] = stdv * wb1[c] * dav;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = (acc + bb1) * stdv;
    if (atomref) v += atomref[zi[i]];
    y[i] = v;
  }
}

__global__ void hk_energy(int B, const int* __restrict__ fstart, const int* __restrict__ fend,
                          const float* __restrict__ y, float mean, float* __restrict__ e_out,
                          const int* __restrict__ status, int epoch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int i = fstart[b]; i < fend[b]; ++i) acc += y[i];
  e_out[b] = (status && *status == epoch) ? __builtin_nanf("") : acc + mean;
}

// ---- reverse ----
// g_p[s] = g_v * p[s] / v   (adjoint of the 2-norm over s; 0 where v == 0 like torch.norm)
__global__ void hk_b_norm_s(int N, int S, int C, const float* __restrict__ g_cat, int ldg, int goff,
                            const float* __restrict__ cat, int ldc, int coff, const float* __restrict__ p, int ldp,
                            float* __restrict__ g_p, int ldgp) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)N * C) return;
  const int i = (int)(gid / C), c = (int)(gid % C);
  const float v = cat[(size_t)i * ldc + coff + c];
  const float g = g_cat[(size_t)i * ldg + goff + c];
  const float sc = v > 0.f ? g / v : 0.f;
  for (int s = 0; s < S; ++s) g_p[((size_t)i * S + s) * ldgp + c] = sc * p[((size_t)i * S + s) * ldp + c];
}

// g_gate = sum_s g_vec1o[s] v2[s] ; g_v2[s] = g_vec1o[s] gate ; g_xs = g_x1 silu'(xs)
template <bool GEN>
__global__ void hk_b_mid(int N, int S, int H, int act, const float* __restrict__ u0, const float* __restrict__ pv0, int ldp,
                         const float* __restrict__ g_vec1o, const float* __restrict__ g_cat1,
                         float* __restrict__ g_u0, float* __restrict__ g_pv0) {
  const int h2 = H / 2;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)N * h2) return;
  const int i = (int)(gid / h2), c = (int)(gid % h2);
  const float gate = u0[(size_t)i * H + h2 + c];
  float gg = 0.f;
  for (int s = 0; s < S; ++s) {
    const float gv = g_vec1o[((size_t)i * S + s) * h2 + c];
    gg += gv * pv0[((size_t)i * S + s) * ldp + H + c];
    g_pv0[((size_t)i * S + s) * ldp + H + c] = gv * gate;
  }
  g_u0[(size_t)i * H + c] = g_cat1[(size_t)i * H + c] * dact_f(GEN ? act : VSN_ACT_SILU, u0[(size_t)i * H + c]);
  g_u0[(size_t)i * H + h2 + c] = gg;
}

template <bool GEN>
__global__ void hk_b_mul_dsilu(long long n, int act, const float* __restrict__ a, float* __restrict__ g) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  g[gid] *= dact_f(GEN ? act : VSN_ACT_SILU, a[gid]);
}

// activation-using head kernels: <false> = all-silu network (folds to the branch-free form), <true> = kind from D.act
#define VSN_HK(K) hipLaunchKernelGGL((D.act == VSN_ACT_SILU ? K<false> : K<true>)
static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

bool head_defers_energy(const Dims& D, const HeadW& W) {
  return W.defer_energy && W.fuse && D.N > 0 && D.N < 4096 && head_fused_supported(D);
}

int launch_head_forward(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf, const float* vo,
                        const int* fstart, const int* fend, int B, float* e_out) {
  const int N = D.N, S = D.S, H = D.H, h2 = H / 2, ldp = H + h2;
  if (N <= 0) {
    if (B > 0) hipLaunchKernelGGL(hk_energy, dim3(nblk(B)), dim3(256), 0, st, B, fstart, fend, Bf.y, W.mean, e_out,
                                  W.status, W.epoch);
    return 0;
  }
  int rc = 0;
  rc |= launch_gemm(st, vo, H, W.Wpv0, H, Bf.pv0, ldp, nullptr, N * S, nullptr, ldp, H, 0);
  if (W.fuse && head_fused_supported(D)) {
    // single-protein sizes: the node-local rest of the head, forward and reverse, is one launch
    rc |= launch_head_fused(st, D, W, Bf);
    if (!head_defers_energy(D, W))
      hipLaunchKernelGGL(hk_energy, dim3(nblk(B)), dim3(256), 0, st, B, fstart, fend, Bf.y, W.mean, e_out,
                         W.status, W.epoch);
    return rc;
  }
  hipLaunchKernelGGL(hk_norm_s, dim3(nblk((long long)N * H)), dim3(256), 0, st, N, S, H, Bf.pv0, ldp, Bf.cat0,
                     2 * H, H);
  rc |= launch_gemm(st, Bf.cat0, 2 * H, W.Wa0, 2 * H, Bf.a0, H, W.ba0, N, nullptr, H, 2 * H, 0);
  rc |= launch_gemm(st, Bf.a0, H, W.Wb0, H, Bf.u0, H, W.bb0, N, nullptr, H, H, 2 | (D.act << 8));
  VSN_HK(hk_mid), dim3(nblk((long long)N * h2)), dim3(256), 0, st, N, S, H, D.act, Bf.u0, Bf.pv0, ldp,
                     Bf.cat1, Bf.vec1o);
  rc |= launch_gemm(st, Bf.vec1o, h2, W.W11, h2, Bf.p1, h2, nullptr, N * S, nullptr, h2, h2, 0);
  hipLaunchKernelGGL(hk_norm_s, dim3(nblk((long long)N * h2)), dim3(256), 0, st, N, S, h2, Bf.p1, h2, Bf.cat1, H,
                     h2);
  rc |= launch_gemm(st, Bf.cat1, H, W.Wa1, H, Bf.a1b, h2, W.ba1, N, nullptr, h2, H, 0);
  VSN_HK(hk_final), dim3((N + 3) / 4), dim3(256), 0, st, N, h2, D.act, Bf.a1b, W.wb1, W.bb1, W.stdv, W.atomref,
                     D.zi, Bf.y, Bf.g_a1);
  hipLaunchKernelGGL(hk_energy, dim3(nblk(B)), dim3(256), 0, st, B, fstart, fend, Bf.y, W.mean, e_out,
                                  W.status, W.epoch);
  return rc;
}

// leaves dE/d(out_norm(x)) in Bf.g_cat0[:, :H] (row stride 2H) and dE/d(vec_out_norm(vec)) in g_vo
int launch_head_backward(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf, float* g_vo) {
  const int N = D.N, S = D.S, H = D.H, h2 = H / 2, ldp = H + h2;
  if (N <= 0) return 0;
  int rc = 0;
  if (W.fuse && head_fused_supported(D))  // everything up to g_pv0 was done by the fused forward launch
    return launch_gemm(st, Bf.g_pv0, ldp, W.Wpv0T, ldp, g_vo, H, nullptr, N * S, nullptr, H, ldp, 0);
  rc |= launch_gemm(st, Bf.g_a1, h2, W.Wa1T, h2, Bf.g_cat1, H, nullptr, N, nullptr, H, h2, 0);
  hipLaunchKernelGGL(hk_b_norm_s, dim3(nblk((long long)N * h2)), dim3(256), 0, st, N, S, h2, Bf.g_cat1, H, h2,
                     Bf.cat1, H, h2, Bf.p1, h2, Bf.g_p1, h2);
  rc |= launch_gemm(st, Bf.g_p1, h2, W.W11T, h2, Bf.g_vec1o, h2, nullptr, N * S, nullptr, h2, h2, 0);
  VSN_HK(hk_b_mid), dim3(nblk((long long)N * h2)), dim3(256), 0, st, N, S, H, D.act, Bf.u0, Bf.pv0, ldp,
                     Bf.g_vec1o, Bf.g_cat1, Bf.g_u0, Bf.g_pv0);
  rc |= launch_gemm(st, Bf.g_u0, H, W.Wb0T, H, Bf.g_h0, H, nullptr, N, nullptr, H, H, 0);
  VSN_HK(hk_b_mul_dsilu), dim3(nblk((long long)N * H)), dim3(256), 0, st, (long long)N * H, D.act, Bf.a0,
                     Bf.g_h0);
  rc |= launch_gemm(st, Bf.g_h0, H, W.Wa0T, H, Bf.g_cat0, 2 * H, nullptr, N, nullptr, 2 * H, H, 0);
  hipLaunchKernelGGL(hk_b_norm_s, dim3(nblk((long long)N * H)), dim3(256), 0, st, N, S, H, Bf.g_cat0, 2 * H, H,
                     Bf.cat0, 2 * H, H, Bf.pv0, ldp, Bf.g_pv0, ldp);
  rc |= launch_gemm(st, Bf.g_pv0, ldp, W.Wpv0T, ldp, g_vo, H, nullptr, N * S, nullptr, H, ldp, 0);
  return rc;
}

}  // namespace vsn
