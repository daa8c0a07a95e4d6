// Reverse-pass (input-gradient only) kernels of the ViS-MP layers and embeddings.
//
// The reference obtains forces with torch.autograd.grad(E, pos)
// (ViSNet/model/visnet.py:153-165) with every parameter frozen (visnet.py:89-90),
// so only input gradients are ever needed.  These kernels are the hand-derived
// adjoints of layer_fwd.hip (derivation checked against autograd in
// oracle/visnet_oracle.py::energy_forces_analytic, same staging and names).
//
// Positions enter the network only through per-edge r (rbf, cutoff) and the
// spherical harmonics d[S]; every kernel here adds its share of dE/dd and
// dE/dC into g_geo[E,16] (d at 0..S-1, C at 8); graph.hip::k_bwd_geom folds them
// into forces once at the end.
//
// Sums over a node's IN-edges ("T" kernels, CSR by target) and over its
// OUT-edges ("S" kernels, perm/colptr by source) are register accumulations by
// one wave per node in a fixed order - no atomics, bit-reproducible.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace vsn {

// template dispatch on V = H/64 (1,2,4), S (3,8) and WPN (1 or VSN_WPN_SMALL)
#ifndef VSN_WPN_SMALL
#define VSN_WPN_SMALL 8  // waves cooperating on one node at single-protein sizes (A/B builds: -DVSN_WPN_SMALL=4)
#endif
#define VSN_DISPATCH3(V_, S_, W_, FN, ...)                              \
  do {                                                                   \
    if ((W_) == 1) FN<V_, S_, 1> __VA_ARGS__;                            \
    else FN<V_, S_, VSN_WPN_SMALL> __VA_ARGS__;                          \
  } while (0)
// kernels that evaluate activations take a 4th parameter GEN: false = the all-silu network (the reference default;
// the activation folds to the branch-free silu), true = kinds read from Dims at run time (utils.py:93-116 table)
#define VSN_DISPATCH3A(V_, S_, W_, G_, FN, ...)                                \
  do {                                                                         \
    if ((W_) == 1) {                                                           \
      if (G_) FN<V_, S_, 1, true> __VA_ARGS__;                                 \
      else FN<V_, S_, 1, false> __VA_ARGS__;                                   \
    } else {                                                                   \
      if (G_) FN<V_, S_, VSN_WPN_SMALL, true> __VA_ARGS__;                     \
      else FN<V_, S_, VSN_WPN_SMALL, false> __VA_ARGS__;                       \
    }                                                                          \
  } while (0)
#define VSN_DISPATCH_VA(V_, S_, W_, G_, FN, ...)                               \
  do {                                                                         \
    if ((S_) == 8) VSN_DISPATCH3A(V_, 8, W_, G_, FN, __VA_ARGS__);             \
    else if ((S_) == 3) VSN_DISPATCH3A(V_, 3, W_, G_, FN, __VA_ARGS__);        \
    else return -22;                                                           \
  } while (0)
#define VSN_DISPATCH_VSA(H_, S_, W_, G_, FN, ...)                              \
  do {                                                                         \
    switch ((H_) / 64) {                                                       \
      case 1: VSN_DISPATCH_VA(1, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 2: VSN_DISPATCH_VA(2, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 3: VSN_DISPATCH_VA(3, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 4: VSN_DISPATCH_VA(4, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 5: VSN_DISPATCH_VA(5, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 6: VSN_DISPATCH_VA(6, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 7: VSN_DISPATCH_VA(7, S_, W_, G_, FN, __VA_ARGS__); break;          \
      case 8: VSN_DISPATCH_VA(8, S_, W_, G_, FN, __VA_ARGS__); break;          \
      default: return -22;                                                     \
    }                                                                          \
  } while (0)
#define VSN_DISPATCH_V(V_, S_, W_, FN, ...)                              \
  do {                                                                   \
    if ((S_) == 8) VSN_DISPATCH3(V_, 8, W_, FN, __VA_ARGS__);            \
    else if ((S_) == 3) VSN_DISPATCH3(V_, 3, W_, FN, __VA_ARGS__);       \
    else return -22;                                                     \
  } while (0)
#define VSN_DISPATCH_VS(H_, S_, W_, FN, ...)                             \
  do {                                                                   \
    switch ((H_) / 64) { /* hidden = 64 V, V = 1..8 */                   \
      case 1: VSN_DISPATCH_V(1, S_, W_, FN, __VA_ARGS__); break;         \
      case 2: VSN_DISPATCH_V(2, S_, W_, FN, __VA_ARGS__); break;         \
      case 3: VSN_DISPATCH_V(3, S_, W_, FN, __VA_ARGS__); break;         \
      case 4: VSN_DISPATCH_V(4, S_, W_, FN, __VA_ARGS__); break;         \
      case 5: VSN_DISPATCH_V(5, S_, W_, FN, __VA_ARGS__); break;         \
      case 6: VSN_DISPATCH_V(6, S_, W_, FN, __VA_ARGS__); break;         \
      case 7: VSN_DISPATCH_V(7, S_, W_, FN, __VA_ARGS__); break;         \
      case 8: VSN_DISPATCH_V(8, S_, W_, FN, __VA_ARGS__); break;         \
      default: return -22;                                               \
    }                                                                    \
  } while (0)

int g_fuse_side = 2;  // single-protein sizes, reverse pass (env VSN_FUSE_SIDE): 2 = no second stream, the side kernels ride in
                      // main-chain launches (k_bwd_hf1/2); 1 = one fused side-stream launch per layer; 0 = three
int g_part_layout = 1;  // k_bwd_hf1 / k_bwd_hf2: how their parts map onto workgroups (see part_of_block; env VSN_PART_LAYOUT).
                        // Measured (Chignolin, us per launch hf1 / hf2): 0: 27.7 / 18.8, 1: 27.6 / 19.3, 2: 33.0 / 24.6
int g_split_channels = 1;  // k_bwd_edge_update_T: two waves per node, half the channels each (env VSN_SPLIT_CH=0 disables)
static const bool g_bwd_env_read = [] {  // A/B switches, read once when the library is loaded
  if (const char* e = getenv("VSN_SPLIT_CH")) g_split_channels = atoi(e);
  if (const char* e = getenv("VSN_FUSE_SIDE")) g_fuse_side = atoi(e);
  if (const char* e = getenv("VSN_PART_LAYOUT")) g_part_layout = atoi(e);
  return true;
}();
// small batches (one protein per MD step): several waves per node
static const int g_wpn_n = [] {  // several waves per node below this many nodes (env VSN_WPN_N, tuning aid)
  const char* e = getenv("VSN_WPN_N");
  return e ? atoi(e) : 4096;
}();
static inline int pick_wpn(int N) { return N < g_wpn_n ? VSN_WPN_SMALL : 1; }
static inline int node_grid(int N, int wpn) {
  int g = wpn == 1 ? (N + 3) / 4 : N;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return g;
}
static inline int node_block(int wpn) { return wpn == 1 ? 256 : 64 * wpn; }
// LDS for node_reduce of K*V*64 floats per extra wave
static inline size_t node_lds(int wpn, int K, int V) {
  const int rmax = (32 / V) < 8 ? (32 / V) : 8;  // node_reduce: RMAX rows per pass
  const int kc = K < rmax ? K : rmax;
  // (env VSN_BATCH_LDS_PAD: lab knob - unused dynamic LDS on the one-wave-per-node launches caps the workgroups a CU
  //  holds, i.e. the nodes in flight per XCD whose gathered rows compete for its 4 MB of L2)
  static const size_t pad = [] {
    const char* e = getenv("VSN_BATCH_LDS_PAD");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  return wpn == 1 ? pad : (size_t)(wpn - 1) * kc * V * 64 * 4;
}

// ---- adjoint of the node update (visnet_block.py:271-274) ------------------------
// g_o = [sum_s g_vec*vec3 | g_x*vec_dot | g_x] ; g_vp = [g_vdot*vec2 | g_vdot*vec1 | g_vec*o1]
template <int V, int S, int WPN>
__global__ __launch_bounds__(256) void k_bwd_node_update(Dims D, const float* __restrict__ g_x,
                                                         const float* __restrict__ g_vec,
                                                         const float* __restrict__ vp, const float* __restrict__ o,
                                                         float* __restrict__ g_o, float* __restrict__ g_vp) {
  const int H = D.H;
  VSN_NODE_LOOP(i, D.N, 1) {
    (void)sub;
    float gx[V], o1[V], o2[V], gvd[V], vd[V], go1[V];
    ldrow<V>(g_x + (size_t)i * H, lane, gx);
    ldrow<V>(o + (size_t)i * 3 * H, lane, o1);
    ldrow<V>(o + (size_t)i * 3 * H + H, lane, o2);
#pragma unroll
    for (int c = 0; c < V; ++c) {
      gvd[c] = gx[c] * o2[c];
      vd[c] = 0.f;
      go1[c] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* row = vp + ((size_t)i * S + s) * 5 * H;
      float* grow = g_vp + ((size_t)i * S + s) * 5 * H;
      float v1[V], v2[V], v3[V], gv[V], t1[V], t2[V], t3[V];
      ldrow<V>(row, lane, v1);
      ldrow<V>(row + H, lane, v2);
      ldrow<V>(row + 2 * H, lane, v3);
      ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, gv);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        vd[c] += v1[c] * v2[c];
        go1[c] += gv[c] * v3[c];
        t1[c] = gvd[c] * v2[c];
        t2[c] = gvd[c] * v1[c];
        t3[c] = gv[c] * o1[c];
      }
      strow<V>(grow, lane, t1);
      strow<V>(grow + H, lane, t2);
      strow<V>(grow + 2 * H, lane, t3);
    }
    float go2[V];
#pragma unroll
    for (int c = 0; c < V; ++c) go2[c] = gx[c] * vd[c];
    strow<V>(g_o + (size_t)i * 3 * H, lane, go1);
    strow<V>(g_o + (size_t)i * 3 * H + H, lane, go2);
    strow<V>(g_o + (size_t)i * 3 * H + 2 * H, lane, gx);
  }
}

// ---- adjoint of the edge update, target side --------------------------------------
// wd = u1.u2 + a1 a2 cc ; df = silu(pf) wd
// g_pf = g_f wd silu'(pf) ; g_wd = g_f silu(pf)
// g_wt_i = sum_e g_wd (u2 + a2 cc d) ; g_d += sum_c g_wd (cc (a2 u1 + a1 u2) + 2 a1 a2 d)
// CS = 2: TWO waves per node, each owning half of the channels (V = H / 128 per lane).  The kernel holds three
// [S][V] register blocks (wt, gwt, u2): at V = 4, S = 8 that is 164-202 VGPRs, two or three waves per SIMD, and it ran
// 3.4x longer than its source-side twin (86 VGPRs).  The channels are independent except for the S per-edge
// dE/dd sums, which each half adds into its own eight slots of the g_geo row (16..23 and 24..31; k_bwd_geom adds them).
// PART: 0 = all of it; 1 = the per-edge outputs only (g_pf, dE/dd); 2 = the per-node sum g_wt only.  At single-protein
// sizes the two halves ride in different launches of the main chain (k_bwd_hf1 / k_bwd_hf2) - each re-reads u2.
template <int V, int S, int WPN, bool GEN, int CS = 1, int PART = 0>
__device__ __forceinline__ void bwd_edge_update_T_body(const Dims& D, const float* __restrict__ vp,
                                                       const float* __restrict__ pe, const float* __restrict__ g_f,
                                                       float* __restrict__ g_pe, float* __restrict__ g_vp,
                                                       float* __restrict__ g_geo, float* __restrict__ smem,
                                                       const int bid, const int nblk) {
  const int H = D.H;
  const int half = CS == 1 ? 0 : (int)blockIdx.y;
  const int co = half * 64 * V;  // first channel of this wave's share
  VSN_NODE_LOOP_B(i, D.N, WPN, bid, nblk) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float wt[S][V], gwt[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if constexpr (PART != 2) {
        ldrow<V>(vp + ((size_t)i * S + s) * 5 * H + 3 * H + co, lane, wt[s]);
      } else {
#pragma unroll
        for (int c = 0; c < V; ++c) wt[s][c] = 0.f;  // unused (dead code below)
      }
#pragma unroll
      for (int c = 0; c < V; ++c) gwt[s][c] = 0.f;
    }
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float geo_old = 0.f;
      if constexpr (PART != 2)  // fetched early, see k_bwd_vecmsg_T
        geo_old = lane < S ? g_geo[(size_t)e * VSN_GEO_W + 16 + 8 * half + lane] : 0.f;
      float u2[S][V], dd[S];
      float dot[V], a1[V], a2[V];
#pragma unroll
      for (int c = 0; c < V; ++c) dot[c] = a1[c] = a2[c] = 0.f;
      float cc = -2.0f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        ldrow<V>(vp + ((size_t)j * S + s) * 5 * H + 4 * H + co, lane, u2[s]);
        dd[s] = D.d[(size_t)e * 8 + s];
        cc += dd[s] * dd[s];
#pragma unroll
        for (int c = 0; c < V; ++c) {
          dot[c] += wt[s][c] * u2[s][c];
          a1[c] += wt[s][c] * dd[s];
          a2[c] += u2[s][c] * dd[s];
        }
      }
      float pf[V], gf[V], gpf[V], gwd[V];
      ldrow<V>(pe + (size_t)e * 3 * H + 2 * H + co, lane, pf);
      ldrow<V>(g_f + (size_t)e * H + co, lane, gf);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        float sp, dsp;
        act_both((GEN ? D.act : VSN_ACT_SILU), pf[c], sp, dsp);
        const float wd = dot[c] + a1[c] * a2[c] * cc;
        gpf[c] = gf[c] * wd * dsp;
        gwd[c] = gf[c] * sp;
      }
      if constexpr (PART != 2) strow<V>(g_pe + (size_t)e * 3 * H + 2 * H + co, lane, gpf);
      constexpr int P = S <= 4 ? 4 : 8;
      float pp[P];
#pragma unroll
      for (int s = 0; s < P; ++s) pp[s] = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          if constexpr (PART != 1) gwt[s][c] += gwd[c] * (u2[s][c] + a2[c] * cc * dd[s]);
          if constexpr (PART != 2)
            p += gwd[c] * (cc * (a2[c] * wt[s][c] + a1[c] * u2[s][c]) + 2.0f * a1[c] * a2[c] * dd[s]);
        }
        pp[s] = p;
      }
      if constexpr (PART != 2) {
        // the S per-component sums over the wave in ONE multi-value butterfly (was: S separate 6-step reductions)
        const float mine = wave_multi_sum<P>(pp, lane);
        if (lane < S) g_geo[(size_t)e * VSN_GEO_W + 16 + 8 * half + lane] = geo_old + mine;  // own slots: may run next to vecmsg_T
      }
    }
    if constexpr (PART == 1) continue;
    if constexpr (WPN > 1 && S <= WPN && V <= 4) {
      // reduce-SCATTER: wave s ends up with the node total of component s and stores its own row (wave 0 summing
      // and storing all S rows alone while seven waves idle was ~20 % of k_bwd_hf1 at single-protein sizes)
      float tot[V];
      node_reduce_scatter<V, S, WPN>(gwt, tot, smem, lane, sub);
      if (sub < S) strow<V>(g_vp + ((size_t)i * S + sub) * 5 * H + 3 * H + co, lane, tot);
    } else {
      node_reduce<V, S, WPN>(gwt, smem, lane, sub);
      if (sub == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) strow<V>(g_vp + ((size_t)i * S + s) * 5 * H + 3 * H + co, lane, gwt[s]);
      }
    }
  }
}

template <int V, int S, int WPN, bool GEN, int CS = 1>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_bwd_edge_update_T(
    Dims D, const float* __restrict__ vp, const float* __restrict__ pe, const float* __restrict__ g_f,
    float* __restrict__ g_pe, float* __restrict__ g_vp, float* __restrict__ g_geo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_edge_update_T_body<V, S, WPN, GEN, CS>(D, vp, pe, g_f, g_pe, g_vp, g_geo, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ---- adjoint of the edge update, source side: g_ws_j = sum_{e: src=j} g_wd (u1 + a1 cc d) ----
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void bwd_edge_update_S_body(const Dims& D, const float* __restrict__ vp,
                                                       const float* __restrict__ pe, const float* __restrict__ g_f,
                                                       float* __restrict__ g_vp, float* __restrict__ smem,
                                                       const int bid, const int nblk) {
  const int H = D.H;
  VSN_NODE_LOOP_B(j, D.N, WPN, bid, nblk) {
    const int t0 = uni(D.colptr[j]), t1 = uni(D.colptr[j + 1]);
    const int permc = edge_cache_load(D.perm, t0, t1, lane);
    const int tgtc = (t0 + lane < t1) ? D.tgt[permc] : 0;
    float gws[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < V; ++c) gws[s][c] = 0.f;
    for (int t = t0 + sub; t < t1; t += WPN) {
      const int e = edge_cache_get(permc, D.perm, t, t0);
      const int i = uni(t - t0) < 64 ? __builtin_amdgcn_readlane(tgtc, uni(t - t0)) : uni(D.tgt[e]);
      float u1[S][V], dd[S], a1[V];
#pragma unroll
      for (int c = 0; c < V; ++c) a1[c] = 0.f;
      float cc = -2.0f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        ldrow<V>(vp + ((size_t)i * S + s) * 5 * H + 3 * H, lane, u1[s]);
        dd[s] = D.d[(size_t)e * 8 + s];
        cc += dd[s] * dd[s];
#pragma unroll
        for (int c = 0; c < V; ++c) a1[c] += u1[s][c] * dd[s];
      }
      float pf[V], gf[V], gwd[V];
      ldrow<V>(pe + (size_t)e * 3 * H + 2 * H, lane, pf);
      ldrow<V>(g_f + (size_t)e * H, lane, gf);
#pragma unroll
      for (int c = 0; c < V; ++c) gwd[c] = gf[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pf[c]);
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < V; ++c) gws[s][c] += gwd[c] * (u1[s][c] + a1[c] * cc * dd[s]);
    }
    if constexpr (WPN > 1 && S <= WPN && V <= 4) {
      float tot[V];
      node_reduce_scatter<V, S, WPN>(gws, tot, smem, lane, sub);
      if (sub < S) strow<V>(g_vp + ((size_t)j * S + sub) * 5 * H + 4 * H, lane, tot);
    } else {
      node_reduce<V, S, WPN>(gws, smem, lane, sub);
      if (sub == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) strow<V>(g_vp + ((size_t)j * S + s) * 5 * H + 4 * H, lane, gws[s]);
      }
    }
  }
}

template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_bwd_edge_update_S(
    Dims D, const float* __restrict__ vp, const float* __restrict__ pe, const float* __restrict__ g_f,
    float* __restrict__ g_vp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_edge_update_S_body<V, S, WPN, GEN>(D, vp, pe, g_f, g_vp, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ---- adjoint of the vector messages, target side -----------------------------------
// mv_e[s] = vh_j[s] s1 + d_s s2 ; g_s1 = sum_s g_vec_i[s] vh_j[s] ; g_s2 = sum_s g_vec_i[s] d_s
// g_t = [g_s1 silu'(t1) | g_s2 silu'(t2)] ; g_d[s] += sum_c g_vec_i[s] s2
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void bwd_vecmsg_T_body(const Dims& D, const float* __restrict__ g_vec,
                                                  const float* __restrict__ vh, const float* __restrict__ tpre,
                                                  float* __restrict__ g_t, float* __restrict__ g_geo, const int bid,
                                                  const int nblk) {
  const int H = D.H;
  VSN_NODE_LOOP_B(i, D.N, WPN, bid, nblk) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float gv[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s) ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, gv[s]);
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      // the running dE/dd of this edge: fetched with the other operands, not after the reductions (a load that is
      // issued only when the sum is ready stalls the wave for a full memory round trip per edge)
      const float geo_old = lane < S ? g_geo[(size_t)e * VSN_GEO_W + lane] : 0.f;
      float t1[V], t2[V], s2[V], d1[V], d2[V];
      ldrow<V>(tpre + (size_t)e * 2 * H, lane, t1);
      ldrow<V>(tpre + (size_t)e * 2 * H + H, lane, t2);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        d1[c] = dact_f((GEN ? D.act : VSN_ACT_SILU), t1[c]);
        act_both((GEN ? D.act : VSN_ACT_SILU), t2[c], s2[c], d2[c]);
      }
      float gs1[V], gs2[V];
#pragma unroll
      for (int c = 0; c < V; ++c) gs1[c] = gs2[c] = 0.f;
      constexpr int P = S <= 4 ? 4 : 8;
      float pp[P];
#pragma unroll
      for (int s = 0; s < P; ++s) pp[s] = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float vj[V];
        ldrow<V>(vh + ((size_t)j * S + s) * H, lane, vj);
        const float ds = D.d[(size_t)e * 8 + s];
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          gs1[c] += gv[s][c] * vj[c];
          gs2[c] += gv[s][c] * ds;
          p += gv[s][c] * s2[c];
        }
        pp[s] = p;
      }
      const float mine = wave_multi_sum<P>(pp, lane);  // lane s < S: sum over the wave of component s
      if (lane < S) g_geo[(size_t)e * VSN_GEO_W + lane] = geo_old + mine;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        gs1[c] *= d1[c];
        gs2[c] *= d2[c];
      }
      strow<V>(g_t + (size_t)e * 2 * H, lane, gs1);
      strow<V>(g_t + (size_t)e * 2 * H + H, lane, gs2);
    }
  }
}

template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_vecmsg_T(
    Dims D, const float* __restrict__ g_vec, const float* __restrict__ vh, const float* __restrict__ tpre,
    float* __restrict__ g_t, float* __restrict__ g_geo) {
  bwd_vecmsg_T_body<V, S, WPN, GEN>(D, g_vec, vh, tpre, g_t, g_geo, (int)blockIdx.x, (int)gridDim.x);
}

// ---- adjoint of the vector messages, source side: g_vh_j[s] = sum_{e: src=j} g_vec_tgt[s] s1_e ----
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void bwd_vecmsg_S_body(const Dims& D, const float* __restrict__ g_vec,
                                                  const float* __restrict__ tpre, float* __restrict__ g_vh,
                                                  float* __restrict__ smem, const int bid, const int nblk) {
  const int H = D.H;
  VSN_NODE_LOOP_B(j, D.N, WPN, bid, nblk) {
    const int t0 = uni(D.colptr[j]), t1 = uni(D.colptr[j + 1]);
    const int permc = edge_cache_load(D.perm, t0, t1, lane);
    const int tgtc = (t0 + lane < t1) ? D.tgt[permc] : 0;
    float acc[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < V; ++c) acc[s][c] = 0.f;
    for (int t = t0 + sub; t < t1; t += WPN) {
      const int e = edge_cache_get(permc, D.perm, t, t0);
      const int i = uni(t - t0) < 64 ? __builtin_amdgcn_readlane(tgtc, uni(t - t0)) : uni(D.tgt[e]);
      float s1[V];
      ldrow<V>(tpre + (size_t)e * 2 * H, lane, s1);
#pragma unroll
      for (int c = 0; c < V; ++c) s1[c] = act_f((GEN ? D.act : VSN_ACT_SILU), s1[c]);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float gv[V];
        ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, gv);
#pragma unroll
        for (int c = 0; c < V; ++c) acc[s][c] += gv[c] * s1[c];
      }
    }
    if constexpr (WPN > 1 && S <= WPN && V <= 4) {
      float tot[V];
      node_reduce_scatter<V, S, WPN>(acc, tot, smem, lane, sub);
      if (sub < S) strow<V>(g_vh + ((size_t)j * S + sub) * H, lane, tot);
    } else {
      node_reduce<V, S, WPN>(acc, smem, lane, sub);
      if (sub == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) strow<V>(g_vh + ((size_t)j * S + s) * H, lane, acc[s]);
      }
    }
  }
}

template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_bwd_vecmsg_S(
    Dims D, const float* __restrict__ g_vec, const float* __restrict__ tpre, float* __restrict__ g_vh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_vecmsg_S_body<V, S, WPN, GEN>(D, g_vec, tpre, g_vh, smem, (int)blockIdx.x, (int)gridDim.x);
}

// The three reverse-pass kernels that do not depend on a layer's main chain (edge-update adjoint, both sides, and the
// source side of the vector messages), as ONE launch: blocks [0,G) target side, [G,2G) source side, [2G,3G) vector
// messages.  They are independent of each other (own outputs, own g_geo slots) and latency-bound at single-protein
// sizes: side by side they take as long as the longest, and the step has two launches less per layer.
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_side(
    Dims D, const float* __restrict__ vp, const float* __restrict__ pe, const float* __restrict__ g_f,
    float* __restrict__ g_pe, float* __restrict__ g_vp, float* __restrict__ g_geo, const float* __restrict__ g_vec,
    const float* __restrict__ tpre, float* __restrict__ g_vh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = (int)gridDim.x / 3;
  const int b = (int)blockIdx.x;
  if (b < G) bwd_edge_update_T_body<V, S, WPN, GEN, 1>(D, vp, pe, g_f, g_pe, g_vp, g_geo, smem, b, G);
  else if (b < 2 * G) bwd_edge_update_S_body<V, S, WPN, GEN>(D, vp, pe, g_f, g_vp, smem, b - G, G);
  else bwd_vecmsg_S_body<V, S, WPN, GEN>(D, g_vec, tpre, g_vh, smem, b - 2 * G, G);
}

// ---- adjoint of attention / scalar message, target side ---------------------------
// gm = g_m_e + g_A_i (overwrites g_m) ; recompute sat, a
// g_a[h] = sum_{c in h} gm v_j dv ; g_sat = g_a silu'(sat) C ; g_C += sum_h g_a silu(sat)
// g_pk = g_sat q_i k_j silu'(pk) ; g_pv = gm v_j a silu'(pv) ; g_q_i = sum_e g_sat k_j dk
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void bwd_attn_T_body(const Dims& D, const float* __restrict__ qkv,
                                                const float* __restrict__ pe, const float* __restrict__ g_A,
                                                float* __restrict__ g_m, float* __restrict__ g_pe,
                                                float* __restrict__ g_qkv, float* __restrict__ sat_tmp,
                                                float* __restrict__ g_geo, const Parts mp, const Parts ap,
                                                float* __restrict__ smem, const int bid, const int nblk) {
  const int H = D.H;
  const int nh = D.nh;
  const int lph = 64 / nh;
  VSN_NODE_LOOP_B(i, D.N, WPN, bid, nblk) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float q[V], gA[V], gq[1][V];
    ldrow<V>(qkv + (size_t)i * 3 * H, lane, q);
    if (ap.n == 3) {  // dE/dA left as 3 K-slices by the GEMM: all loads first, then the sum in a fixed order
      float t1[V], t2[V];
      ldrow<V>(ap.p + (size_t)i * H, lane, gA);
      ldrow<V>(ap.p + ap.stride + (size_t)i * H, lane, t1);
      ldrow<V>(ap.p + 2 * ap.stride + (size_t)i * H, lane, t2);
#pragma unroll
      for (int c = 0; c < V; ++c) gA[c] = (gA[c] + t1[c]) + t2[c];
    } else if (ap.n > 0) {
      ldrow<V>(ap.p + (size_t)i * H, lane, gA);
      for (int k = 1; k < ap.n; ++k) {
        float t[V];
        ldrow<V>(ap.p + (size_t)k * ap.stride + (size_t)i * H, lane, t);
#pragma unroll
        for (int c = 0; c < V; ++c) gA[c] += t[c];
      }
    } else {
      ldrow<V>(g_A + (size_t)i * H, lane, gA);
    }
#pragma unroll
    for (int c = 0; c < V; ++c) gq[0][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      const float C = D.geo[(size_t)e * 8 + 1];
      const float gC_old = lane == 0 ? g_geo[(size_t)e * VSN_GEO_W + 8] : 0.f;  // fetched early, see k_bwd_vecmsg_T
      float k[V], v[V], pk[V], pv[V], gm[V];
      ldrow<V>(qkv + (size_t)j * 3 * H + H, lane, k);
      ldrow<V>(qkv + (size_t)j * 3 * H + 2 * H, lane, v);
      ldrow<V>(pe + (size_t)e * 3 * H, lane, pk);
      ldrow<V>(pe + (size_t)e * 3 * H + H, lane, pv);
      if (mp.n == 2) {  // the common split: both slices fetched together with the other operands of the edge
        float t[V];
        ldrow<V>(mp.p + (size_t)e * H, lane, gm);
        ldrow<V>(mp.p + mp.stride + (size_t)e * H, lane, t);
#pragma unroll
        for (int c = 0; c < V; ++c) gm[c] += t[c];
      } else if (mp.n > 0) {
        ldrow<V>(mp.p + (size_t)e * H, lane, gm);
        for (int k = 1; k < mp.n; ++k) {
          float t[V];
          ldrow<V>(mp.p + (size_t)k * mp.stride + (size_t)e * H, lane, t);
#pragma unroll
          for (int c = 0; c < V; ++c) gm[c] += t[c];
        }
      } else {
        ldrow<V>(g_m + (size_t)e * H, lane, gm);
      }
      float dk[V], ddk[V], dv[V], ddv[V];
      float part = 0.f, gpart = 0.f;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        gm[c] += gA[c];
        act_both((GEN ? D.act : VSN_ACT_SILU), pk[c], dk[c], ddk[c]);
        act_both((GEN ? D.act : VSN_ACT_SILU), pv[c], dv[c], ddv[c]);
        part += q[c] * k[c] * dk[c];
        gpart += gm[c] * v[c] * dv[c];
      }
      strow<V>(g_m + (size_t)e * H, lane, gm);
      float av[V], gsv[V];  // a and g_sat of the head of each of the lane's channels
      bool any_heads = false;
      if constexpr (GEN) any_heads = D.hgen != 0;
      if (any_heads) {  // head count that does not divide 64 (see head_sums_any)
        float pc[V], gc[V], sc[V], gac[V];
#pragma unroll
        for (int c = 0; c < V; ++c) {
          pc[c] = q[c] * k[c] * dk[c];
          gc[c] = gm[c] * v[c] * dv[c];
        }
        head_sums_any<V>(pc, lane, D.hd, nh, sc);
        head_sums_any<V>(gc, lane, D.hd, nh, gac);
        float gcp = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          float ssat, dssat;
          act_both(D.attn_act, sc[c], ssat, dssat);
          av[c] = ssat * C;
          gsv[c] = gac[c] * dssat * C;
          const int ch = lane * V + c;
          if (ch % D.hd == 0) {  // the first channel of a head writes the head's pair and carries its dE/dC term
            gcp += gac[c] * ssat;
            sat_tmp[(size_t)e * 2 * nh + ch / D.hd] = gsv[c];
            sat_tmp[(size_t)e * 2 * nh + nh + ch / D.hd] = av[c];
          }
        }
        const float gC = wave_sum(gcp);
        if (lane == 0) g_geo[(size_t)e * VSN_GEO_W + 8] = gC_old + gC;
      } else {
        const float sat = group_sum(part, lph);
        const float ga = group_sum(gpart, lph);
        float ssat, dssat;
        act_both((GEN ? D.attn_act : VSN_ACT_SILU), sat, ssat, dssat);
        const float a = ssat * C;
        const float gsat = ga * dssat * C;
        const bool head_lead = (lane & (lph - 1)) == 0;
        const float gC = wave_sum(head_lead ? ga * ssat : 0.f);
        if (lane == 0) g_geo[(size_t)e * VSN_GEO_W + 8] = gC_old + gC;
        if (head_lead) {
          sat_tmp[(size_t)e * 2 * nh + lane / lph] = gsat;
          sat_tmp[(size_t)e * 2 * nh + nh + lane / lph] = a;
        }
#pragma unroll
        for (int c = 0; c < V; ++c) av[c] = a, gsv[c] = gsat;
      }
      float gpk[V], gpv[V];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        gpk[c] = gsv[c] * q[c] * k[c] * ddk[c];
        gpv[c] = gm[c] * v[c] * av[c] * ddv[c];
        gq[0][c] += gsv[c] * k[c] * dk[c];
      }
      strow<V>(g_pe + (size_t)e * 3 * H, lane, gpk);
      strow<V>(g_pe + (size_t)e * 3 * H + H, lane, gpv);
    }
    node_reduce<V, 1, WPN>(gq, smem, lane, sub);
    if (sub == 0) strow<V>(g_qkv + (size_t)i * 3 * H, lane, gq[0]);
  }
}

template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_attn_T(
    Dims D, const float* __restrict__ qkv, const float* __restrict__ pe, const float* __restrict__ g_A,
    float* __restrict__ g_m, float* __restrict__ g_pe, float* __restrict__ g_qkv, float* __restrict__ sat_tmp,
    float* __restrict__ g_geo, Parts mp, Parts ap) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_attn_T_body<V, S, WPN, GEN>(D, qkv, pe, g_A, g_m, g_pe, g_qkv, sat_tmp, g_geo, mp, ap, smem, (int)blockIdx.x,
                                  (int)gridDim.x);
}

// Single-protein sizes, no second stream: the reverse kernels of a layer that do not feed each other share launches
// of the MAIN chain (a fork/join through events costs ~7 us on the main stream at each end, every layer).
//   k_bwd_hf1: [0,G) vector messages target side | [G,2G) edge update, per-edge half | [2G,3G) edge update source
//              side | [3G,4G) vector messages source side          (before the g_m / g_A products)
//   k_bwd_hf2: [0,G) attention target side | [G,2G) edge update, per-node half (after them)
// with_eu = 0 (the last layer has no edge update): hf1 = {vector messages, both sides}; hf2 is not used.
// Occupancy: the straight kernel needs 140 VGPRs = three waves per SIMD, i.e. ONE eight-wave workgroup per CU, and its
// 4 N workgroups ran in six rounds on a Chignolin step.  Asking for four waves per SIMD makes hipcc schedule the row
// loads less eagerly: 128 VGPRs, no spills, two workgroups per CU (Chignolin 431 -> 438 steps/s).  [Cutting the
// registers by walking the channels in two chunks also gave two workgroups per CU but doubled the dependent memory
// phases per edge and gained nothing.]
#ifndef VSN_HF1_MINWAVES
#define VSN_HF1_MINWAVES 4
#endif
// Which part a workgroup runs and which node block it is (g_part_layout, env VSN_PART_LAYOUT):
//   0  parts one after the other, G = gridDim / P blocks each: [0,G) part 0, [G,2G) part 1, ...  With G = N not a
//      multiple of 8 the parts start on different XCDs (the dispatcher places workgroup b on XCD b % 8), so the
//      xcd_block() slab of a node - and with it the fragment's rows, the per-edge streams and the neighbour rows all
//      four parts read - is pulled into up to four different L2s: measured 67.7 MB fetched per k_bwd_hf1 launch
//      against 41 MB of distinct arrays (profiles/r04_chig_md_pmc.csv).
//   1  the same order with G rounded up to a multiple of 8 (blocks past N idle): every part of a node on ONE XCD.
//   2  interleaved: workgroups 8 (P k + part) + xcd, i.e. the P parts of the k-th node of every XCD slab are
//      dispatched back to back on that XCD - what one part pulled into the L2 is still there for the others.
template <int P>
__device__ __forceinline__ void part_of_block(int il, int& part, int& bid, int& G) {
  const int b = (int)blockIdx.x;
  G = (int)gridDim.x / P;
  if (il) {
    const int t = b >> 3;
    part = t % P;
    bid = ((t / P) << 3) + (b & 7);
  } else {
    part = b / G;
    bid = b - part * G;
  }
}
template <int V, int S, int WPN, bool GEN>
__global__ __launch_bounds__(64 * (WPN == 1 ? 4 : WPN), ((WPN > 1 && V <= 4 && !GEN) ? VSN_HF1_MINWAVES : 1)) void k_bwd_hf1(
    Dims D, const float* __restrict__ g_vec, const float* __restrict__ vh, const float* __restrict__ tpre,
    float* __restrict__ g_t, float* __restrict__ g_geo, const float* __restrict__ vp, const float* __restrict__ pe,
    const float* __restrict__ g_f, float* __restrict__ g_pe, float* __restrict__ g_vp, float* __restrict__ g_vh,
    int with_eu, int il) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int part, b, G;
  if (with_eu) part_of_block<4>(il, part, b, G);
  else part_of_block<2>(il, part, b, G);
  if (part == 0) {
    bwd_vecmsg_T_body<V, S, WPN, GEN>(D, g_vec, vh, tpre, g_t, g_geo, b, G);
  } else if (!with_eu || part == 3) {
    bwd_vecmsg_S_body<V, S, WPN, GEN>(D, g_vec, tpre, g_vh, smem, b, G);
  } else if (part == 1) {
    bwd_edge_update_T_body<V, S, WPN, GEN, 1, 1>(D, vp, pe, g_f, g_pe, g_vp, g_geo, smem, b, G);
  } else {
    bwd_edge_update_S_body<V, S, WPN, GEN>(D, vp, pe, g_f, g_vp, smem, b, G);
  }
}
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_hf2(
    Dims D, const float* __restrict__ qkv, const float* __restrict__ pe, const float* __restrict__ g_A,
    float* __restrict__ g_m, float* __restrict__ g_pe, float* __restrict__ g_qkv, float* __restrict__ sat_tmp,
    float* __restrict__ g_geo, Parts mp, Parts ap, const float* __restrict__ vp, const float* __restrict__ g_f,
    float* __restrict__ g_vp, int il) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int part, b, G;
  part_of_block<2>(il, part, b, G);
  if (part == 0) bwd_attn_T_body<V, S, WPN, GEN>(D, qkv, pe, g_A, g_m, g_pe, g_qkv, sat_tmp, g_geo, mp, ap, smem, b, G);
  else bwd_edge_update_T_body<V, S, WPN, GEN, 1, 2>(D, vp, pe, g_f, g_pe, g_vp, g_geo, smem, b, G);
}

// ---- source side: g_k_j = sum g_sat q_i dk ; g_v_j = sum gm dv a ---------------------
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_bwd_attn_S(
    Dims D, const float* __restrict__ qkv, const float* __restrict__ pe, const float* __restrict__ g_m,
    const float* __restrict__ sat_tmp, float* __restrict__ g_qkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = D.H;
  const int nh = D.nh;
  const int lph = 64 / nh;
  VSN_NODE_LOOP(j, D.N, WPN) {
    const int t0 = uni(D.colptr[j]), t1 = uni(D.colptr[j + 1]);
    const int permc = edge_cache_load(D.perm, t0, t1, lane);
    const int tgtc = (t0 + lane < t1) ? D.tgt[permc] : 0;
    float g2[2][V];
#pragma unroll
    for (int c = 0; c < V; ++c) g2[0][c] = g2[1][c] = 0.f;
    for (int t = t0 + sub; t < t1; t += WPN) {
      const int e = edge_cache_get(permc, D.perm, t, t0);
      const int i = uni(t - t0) < 64 ? __builtin_amdgcn_readlane(tgtc, uni(t - t0)) : uni(D.tgt[e]);
      float q[V], pk[V], pv[V], gm[V];
      ldrow<V>(qkv + (size_t)i * 3 * H, lane, q);
      ldrow<V>(pe + (size_t)e * 3 * H, lane, pk);
      ldrow<V>(pe + (size_t)e * 3 * H + H, lane, pv);
      ldrow<V>(g_m + (size_t)e * H, lane, gm);
      float gsv[V], av[V];
      bool any_heads = false;
      if constexpr (GEN) any_heads = D.hgen != 0;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const int h = any_heads ? (lane * V + c) / D.hd : lane / lph;
        gsv[c] = sat_tmp[(size_t)e * 2 * nh + h];
        av[c] = sat_tmp[(size_t)e * 2 * nh + nh + h];
      }
#pragma unroll
      for (int c = 0; c < V; ++c) {
        g2[0][c] += gsv[c] * q[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pk[c]);
        g2[1][c] += gm[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pv[c]) * av[c];
      }
    }
    node_reduce<V, 2, WPN>(g2, smem, lane, sub);
    if (sub == 0) {
      strow<V>(g_qkv + (size_t)j * 3 * H + H, lane, g2[0]);
      strow<V>(g_qkv + (size_t)j * 3 * H + 2 * H, lane, g2[1]);
    }
  }
}

// ---- target side dE/dq alone: g_q_i = sum_{e -> i} g_sat_e k_j act(pk_e) ------------------------------
// (fragment batches: the rest of the target-side attention adjoint is the prologue of the fused g_f product,
//  fused.hip::k_bwd_gf_fused, which leaves g_sat in sat_tmp; a per-node sum cannot live in a per-edge-panel kernel)
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_attn_Q(
    Dims D, const float* __restrict__ qkv, const float* __restrict__ pe, const float* __restrict__ sat_tmp,
    float* __restrict__ g_qkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = D.H;
  const int nh = D.nh;
  const int lph = 64 / nh;
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float gq[1][V];
#pragma unroll
    for (int c = 0; c < V; ++c) gq[0][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float k[V], pk[V];
      ldrow<V>(qkv + (size_t)j * 3 * H + H, lane, k);
      ldrow<V>(pe + (size_t)e * 3 * H, lane, pk);
      bool any_heads = false;
      if constexpr (GEN) any_heads = D.hgen != 0;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const float gsat = sat_tmp[(size_t)e * 2 * nh + (any_heads ? (lane * V + c) / D.hd : lane / lph)];
        gq[0][c] += gsat * k[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pk[c]);
      }
    }
    node_reduce<V, 1, WPN>(gq, smem, lane, sub);
    if (sub == 0) strow<V>(g_qkv + (size_t)i * 3 * H, lane, gq[0]);
  }
}

// ---- adjoint of LayerNorm + VecLayerNorm("none") ------------------------------------
template <int V, int S, int WPN>
__global__ __launch_bounds__(256) void k_bwd_node_norm(Dims D, const float* __restrict__ g_xh, int ldg,
                                                       const float* __restrict__ g_vh,
                                                       const float* __restrict__ xn, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ wvec, int norm_type,
                                                       int accumulate, float* __restrict__ g_x,
                                                       float* __restrict__ g_vec) {
  const int H = D.H;
  const float invH = 1.0f / (float)H;
  VSN_NODE_LOOP(i, D.N, 1) {
    (void)sub;
    float g[V], n[V], ga[V], w[V];
    ldrow<V>(g_xh + (size_t)i * ldg, lane, g);
    ldrow<V>(xn + (size_t)i * H, lane, n);
    ldrow<V>(gamma, lane, ga);
    ldrow<V>(wvec, lane, w);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      g[c] *= ga[c];
      s1 += g[c];
      s2 += g[c] * n[c];
    }
    const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
    const float rs = rstd[i];
    float out[V];
    if (accumulate)
      ldrow<V>(g_x + (size_t)i * H, lane, out);
    else {
#pragma unroll
      for (int c = 0; c < V; ++c) out[c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < V; ++c) out[c] += rs * (g[c] - m1 - n[c] * m2);
    strow<V>(g_x + (size_t)i * H, lane, out);
    if (norm_type == 0)  // "none": g_vec (+)= g_vh * w ; rms / max_min adjoints live in vecnorm.hip
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float gv[V], o[V];
      ldrow<V>(g_vh + ((size_t)i * S + s) * H, lane, gv);
      if (accumulate)
        ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, o);
      else {
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < V; ++c) o[c] += gv[c] * w[c];
      strow<V>(g_vec + ((size_t)i * S + s) * H, lane, o);
    }
  }
}

// ---- adjoint of LayerNorm + VecLayerNorm("none") of layer l  FUSED WITH  the adjoint of the node update
// of layer l-1 (k_bwd_node_norm + k_bwd_node_update in one pass: both are node-local, g_x / g_vec of the
// node stay in registers between the two)
template <int V, int S, int WPN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_norm_update(Dims D, const float* __restrict__ g_xh, int ldg,
                                                         const float* __restrict__ g_vh,
                                                         const float* __restrict__ xn,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ wvec, int accumulate,
                                                         float* __restrict__ g_x, float* __restrict__ g_vec,
                                                         const float* __restrict__ vp, const float* __restrict__ o,
                                                         float* __restrict__ g_o, float* __restrict__ g_vp) {
  const int H = D.H;
  const float invH = 1.0f / (float)H;
  if constexpr (WPN > 1 && S <= WPN) {
    // Small batches: wave s of the node's workgroup owns vector component s (every wave recomputes the cheap
    // LayerNorm adjoint g_x); the two channel-wise sums over s go through LDS in a fixed order.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    VSN_NODE_LOOP(i, D.N, WPN) {
      float g[V], n[V], ga[V], w[V];
      ldrow<V>(g_xh + (size_t)i * ldg, lane, g);
      ldrow<V>(xn + (size_t)i * H, lane, n);
      ldrow<V>(gamma, lane, ga);
      ldrow<V>(wvec, lane, w);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        g[c] *= ga[c];
        s1 += g[c];
        s2 += g[c] * n[c];
      }
      const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
      const float rs = rstd[i];
      float gx[V];
      if (accumulate)
        ldrow<V>(g_x + (size_t)i * H, lane, gx);
      else {
#pragma unroll
        for (int c = 0; c < V; ++c) gx[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < V; ++c) gx[c] += rs * (g[c] - m1 - n[c] * m2);
      __syncthreads();  // all waves have read the old g_x row; smem of the previous node is free
      if (sub == 0) strow<V>(g_x + (size_t)i * H, lane, gx);
      float vdp[V], go1p[V];
#pragma unroll
      for (int c = 0; c < V; ++c) vdp[c] = go1p[c] = 0.f;
      if (sub < S) {
        const int s = sub;
        float o1[V], o2[V], gv[V], ov[V];
        ldrow<V>(o + (size_t)i * 3 * H, lane, o1);
        ldrow<V>(o + (size_t)i * 3 * H + H, lane, o2);
        ldrow<V>(g_vh + ((size_t)i * S + s) * H, lane, gv);
        if (accumulate)
          ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, ov);
        else {
#pragma unroll
          for (int c = 0; c < V; ++c) ov[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < V; ++c) ov[c] += gv[c] * w[c];
        strow<V>(g_vec + ((size_t)i * S + s) * H, lane, ov);
        const float* row = vp + ((size_t)i * S + s) * 5 * H;
        float* grow = g_vp + ((size_t)i * S + s) * 5 * H;
        float v1[V], v2[V], v3[V], t1[V], t2[V], t3[V];
        ldrow<V>(row, lane, v1);
        ldrow<V>(row + H, lane, v2);
        ldrow<V>(row + 2 * H, lane, v3);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const float gvd = gx[c] * o2[c];
          vdp[c] = v1[c] * v2[c];
          go1p[c] = ov[c] * v3[c];
          t1[c] = gvd * v2[c];
          t2[c] = gvd * v1[c];
          t3[c] = ov[c] * o1[c];
        }
        strow<V>(grow, lane, t1);
        strow<V>(grow + H, lane, t2);
        strow<V>(grow + 2 * H, lane, t3);
        if (sub > 0) {
          float* dst = smem + ((size_t)(sub - 1) * 2 * 64 + lane) * V;
#pragma unroll
          for (int c = 0; c < V; ++c) {
            dst[c] = vdp[c];
            dst[64 * V + c] = go1p[c];
          }
        }
      }
      __syncthreads();
      if (sub == 0) {
        for (int s = 1; s < S; ++s) {
          const float* src_ = smem + ((size_t)(s - 1) * 2 * 64 + lane) * V;
#pragma unroll
          for (int c = 0; c < V; ++c) {
            vdp[c] += src_[c];
            go1p[c] += src_[64 * V + c];
          }
        }
        float go2[V];
#pragma unroll
        for (int c = 0; c < V; ++c) go2[c] = gx[c] * vdp[c];
        strow<V>(g_o + (size_t)i * 3 * H, lane, go1p);
        strow<V>(g_o + (size_t)i * 3 * H + H, lane, go2);
        strow<V>(g_o + (size_t)i * 3 * H + 2 * H, lane, gx);
      }
    }
    return;
  }
  VSN_NODE_LOOP(i, D.N, 1) {
    (void)sub;
    float g[V], n[V], ga[V], w[V];
    ldrow<V>(g_xh + (size_t)i * ldg, lane, g);
    ldrow<V>(xn + (size_t)i * H, lane, n);
    ldrow<V>(gamma, lane, ga);
    ldrow<V>(wvec, lane, w);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      g[c] *= ga[c];
      s1 += g[c];
      s2 += g[c] * n[c];
    }
    const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
    const float rs = rstd[i];
    float gx[V];
    if (accumulate)
      ldrow<V>(g_x + (size_t)i * H, lane, gx);
    else {
#pragma unroll
      for (int c = 0; c < V; ++c) gx[c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < V; ++c) gx[c] += rs * (g[c] - m1 - n[c] * m2);
    strow<V>(g_x + (size_t)i * H, lane, gx);
    // ---- node-update adjoint of the layer below, with this g_x and the g_vec rows produced on the fly
    float o1[V], o2[V], gvd[V], vd[V], go1[V];
    ldrow<V>(o + (size_t)i * 3 * H, lane, o1);
    ldrow<V>(o + (size_t)i * 3 * H + H, lane, o2);
#pragma unroll
    for (int c = 0; c < V; ++c) {
      gvd[c] = gx[c] * o2[c];
      vd[c] = 0.f;
      go1[c] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float gv[V], ov[V];
      ldrow<V>(g_vh + ((size_t)i * S + s) * H, lane, gv);
      if (accumulate)
        ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, ov);
      else {
#pragma unroll
        for (int c = 0; c < V; ++c) ov[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < V; ++c) ov[c] += gv[c] * w[c];
      strow<V>(g_vec + ((size_t)i * S + s) * H, lane, ov);
      const float* row = vp + ((size_t)i * S + s) * 5 * H;
      float* grow = g_vp + ((size_t)i * S + s) * 5 * H;
      float v1[V], v2[V], v3[V], t1[V], t2[V], t3[V];
      ldrow<V>(row, lane, v1);
      ldrow<V>(row + H, lane, v2);
      ldrow<V>(row + 2 * H, lane, v3);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        vd[c] += v1[c] * v2[c];
        go1[c] += ov[c] * v3[c];
        t1[c] = gvd[c] * v2[c];
        t2[c] = gvd[c] * v1[c];
        t3[c] = ov[c] * o1[c];
      }
      strow<V>(grow, lane, t1);
      strow<V>(grow + H, lane, t2);
      strow<V>(grow + 2 * H, lane, t3);
    }
    float go2[V];
#pragma unroll
    for (int c = 0; c < V; ++c) go2[c] = gx[c] * vd[c];
    strow<V>(g_o + (size_t)i * 3 * H, lane, go1);
    strow<V>(g_o + (size_t)i * 3 * H + H, lane, go2);
    strow<V>(g_o + (size_t)i * 3 * H + 2 * H, lane, gx);
  }
}

// ---- adjoint of EdgeEmbedding (utils.py:331-337) -----------------------------------
// g_psi_e = g_f_e (x_i + x_j) ; g_x_i += sum_{in} g_f psi + sum_{out} g_f psi
// g_xh != nullptr: the LayerNorm adjoint of layer 0 (what k_bwd_node_norm adds to g_x) rides in the node epilogue
template <int V, int S, int WPN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_bwd_embed_edge(
    Dims D, const float* __restrict__ x, const float* __restrict__ pp, const float* __restrict__ g_f,
    float* __restrict__ g_pp, float* __restrict__ g_x, const float* __restrict__ g_xh,
    const float* __restrict__ xn, const float* __restrict__ rstd, const float* __restrict__ gamma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = D.H;
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    const int t0 = uni(D.colptr[i]), t1 = uni(D.colptr[i + 1]);
    float xi[V], acc[1][V];
    ldrow<V>(x + (size_t)i * H, lane, xi);
#pragma unroll
    for (int c = 0; c < V; ++c) acc[0][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float xj[V], ps[V], gf[V], gp[V];
      ldrow<V>(x + (size_t)j * H, lane, xj);
      ldrow<V>(pp + (size_t)e * 2 * H + H, lane, ps);
      ldrow<V>(g_f + (size_t)e * H, lane, gf);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        gp[c] = gf[c] * (xi[c] + xj[c]);
        acc[0][c] += gf[c] * ps[c];
      }
      strow<V>(g_pp + (size_t)e * 2 * H + H, lane, gp);
    }
    for (int t = t0 + sub; t < t1; t += WPN) {
      const int e = uni(D.perm[t]);
      float ps[V], gf[V];
      ldrow<V>(pp + (size_t)e * 2 * H + H, lane, ps);
      ldrow<V>(g_f + (size_t)e * H, lane, gf);
#pragma unroll
      for (int c = 0; c < V; ++c) acc[0][c] += gf[c] * ps[c];
    }
    node_reduce<V, 1, WPN>(acc, smem, lane, sub);
    if (sub == 0) {
      float gx[V];
      ldrow<V>(g_x + (size_t)i * H, lane, gx);
      if (g_xh) {  // same arithmetic and order as k_bwd_node_norm (accumulate): g_x += rstd (g - mean(g) - n mean(g n))
        float g[V], n[V], ga[V];
        ldrow<V>(g_xh + (size_t)i * H, lane, g);
        ldrow<V>(xn + (size_t)i * H, lane, n);
        ldrow<V>(gamma, lane, ga);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          g[c] *= ga[c];
          s1 += g[c];
          s2 += g[c] * n[c];
        }
        const float invH = 1.0f / (float)H;
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
        const float rs = rstd[i];
#pragma unroll
        for (int c = 0; c < V; ++c) gx[c] += rs * (g[c] - m1 - n[c] * m2);
      }
#pragma unroll
      for (int c = 0; c < V; ++c) gx[c] += acc[0][c];
      strow<V>(g_x + (size_t)i * H, lane, gx);
    }
  }
}

// ---- adjoint of NeighborEmbedding's aggregation (utils.py:296-317) -------------------
// g_Wn = g_n_i emb2[z_j] (non-loop) ; g_phi = g_Wn C ; g_C += sum_c g_Wn phi
template <int V, int S, int WPN>
__global__ VSN_WALK_BOUNDS(WPN) void k_bwd_embed_node(
    Dims D, const float* __restrict__ emb2, const float* __restrict__ pp, const float* __restrict__ g_n,
    float* __restrict__ g_pp, float* __restrict__ g_geo) {
  const int H = D.H;
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float gn[V];
    ldrow<V>(g_n + (size_t)i * H, lane, gn);
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float gph[V];
      if (j == i) {
#pragma unroll
        for (int c = 0; c < V; ++c) gph[c] = 0.f;
        strow<V>(g_pp + (size_t)e * 2 * H, lane, gph);
        continue;
      }
      const float C = D.geo[(size_t)e * 8 + 1];
      float em[V], ph[V];
      ldrow<V>(emb2 + (size_t)uni(D.zi[j]) * H, lane, em);
      ldrow<V>(pp + (size_t)e * 2 * H, lane, ph);
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const float gW = gn[c] * em[c];
        gph[c] = gW * C;
        p += gW * ph[c];
      }
      strow<V>(g_pp + (size_t)e * 2 * H, lane, gph);
      p = wave_sum(p);
      if (lane == 0) g_geo[(size_t)e * VSN_GEO_W + 8] += p;
    }
  }
}

// ---- launchers -----------------------------------------------------------------------
#define VSN_LAUNCH_ACT(KN, RK, ...)                                                                     \
  do {                                                                                                  \
    const int w__ = pick_wpn(D.N);                                                                      \
    const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;                               \
    VSN_DISPATCH_VSA(D.H, D.S, w__, g__, KN,                                                            \
                     <<<node_grid(D.N, w__), node_block(w__), node_lds(w__, (RK), D.H / 64), st>>>(__VA_ARGS__)); \
  } while (0)
#define VSN_LAUNCH(KN, RK, ...)                                                                         \
  do {                                                                                                  \
    const int w__ = pick_wpn(D.N);                                                                      \
    VSN_DISPATCH_VS(D.H, D.S, w__, KN,                                                                  \
                    <<<node_grid(D.N, w__), node_block(w__), node_lds(w__, (RK), D.H / 64), st>>>(__VA_ARGS__)); \
  } while (0)

int launch_bwd_node_update(hipStream_t st, const Dims& D, const float* g_x, const float* g_vec, const float* vp,
                           const float* o, float* g_o, float* g_vp) {
  if (D.N <= 0) return 0;
  VSN_DISPATCH_VS(D.H, D.S, 1, k_bwd_node_update,
                  <<<node_grid(D.N, 1), 256, 0, st>>>(D, g_x, g_vec, vp, o, g_o, g_vp));
  return 0;
}
int launch_bwd_edge_update(hipStream_t st, const Dims& D, const float* vp, const float* pe, const float* g_f,
                           float* g_pe, float* g_vp, float* g_geo) {
  if (D.N <= 0) return 0;
  const int V = D.H / 64;
  // batches only: with eight waves per node already (single-protein sizes) the per-wave fixed work doubles for nothing
  // (Chignolin: 46.8 vs 40.4 us per launch); VSN_SPLIT_CH=2 forces it there too
  if (g_split_channels && (V & 1) == 0 && D.S == 8 && (pick_wpn(D.N) == 1 || g_split_channels > 1)) {
    // two waves per node, half the channels each (see the kernel)
    const int w = pick_wpn(D.N);
    const bool gen = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
    const dim3 grid(node_grid(D.N, w), 2);
    const size_t lds = node_lds(w, D.S, V / 2);
#define VSN_EUT(VH)                                                                                                   \
  do {                                                                                                                \
    if (w == 1) {                                                                                                     \
      if (gen) k_bwd_edge_update_T<VH, 8, 1, true, 2><<<grid, node_block(w), lds, st>>>(D, vp, pe, g_f, g_pe, g_vp, g_geo); \
      else k_bwd_edge_update_T<VH, 8, 1, false, 2><<<grid, node_block(w), lds, st>>>(D, vp, pe, g_f, g_pe, g_vp, g_geo);    \
    } else {                                                                                                          \
      if (gen)                                                                                                        \
        k_bwd_edge_update_T<VH, 8, VSN_WPN_SMALL, true, 2><<<grid, node_block(w), lds, st>>>(D, vp, pe, g_f, g_pe, g_vp, g_geo); \
      else                                                                                                            \
        k_bwd_edge_update_T<VH, 8, VSN_WPN_SMALL, false, 2><<<grid, node_block(w), lds, st>>>(D, vp, pe, g_f, g_pe, g_vp, g_geo); \
    }                                                                                                                 \
  } while (0)
    switch (V / 2) {
      case 1: VSN_EUT(1); break;
      case 2: VSN_EUT(2); break;
      case 3: VSN_EUT(3); break;
      default: VSN_EUT(4); break;
    }
#undef VSN_EUT
  } else {
    VSN_LAUNCH_ACT(k_bwd_edge_update_T, D.S, D, vp, pe, g_f, g_pe, g_vp, g_geo);
  }
  VSN_LAUNCH_ACT(k_bwd_edge_update_S, D.S, D, vp, pe, g_f, g_vp);
  return 0;
}
// the streamless reverse chain of a layer at single-protein sizes (see k_bwd_hf1 / k_bwd_hf2); false = not applicable
bool bwd_streamless_ok(const Dims& D) { return g_fuse_side >= 2 && pick_wpn(D.N) != 1 && D.N > 0; }
bool bwd_batch_path(const Dims& D) { return pick_wpn(D.N) == 1 && D.N > 0; }
// (launch_maybe_timed: in profile mode these launches carry an event pair on their dispatch packet, kernels.h)
#define VSN_KL4(NAME, KIND)                                                                \
  template <int V, int S, int W, bool G>                                                   \
  struct KL_##NAME {                                                                       \
    template <typename... A>                                                               \
    static void go(dim3 g, dim3 b, unsigned lds, hipStream_t st, A... a) {                 \
      launch_maybe_timed(KIND, NAME<V, S, W, G>, g, b, lds, st, a...);                     \
    }                                                                                      \
  }
VSN_KL4(k_bwd_hf1, WK_BWD_HF1);
VSN_KL4(k_bwd_hf2, WK_BWD_HF2);
VSN_KL4(k_bwd_attn_S, WK_BWD_ATTN_S);
#undef VSN_KL4
template <int V, int S, int W>
struct KL_k_bwd_norm_update {
  template <typename... A>
  static void go(dim3 g, dim3 b, unsigned lds, hipStream_t st, A... a) {
    launch_maybe_timed(WK_BWD_NORM_UPDATE, k_bwd_norm_update<V, S, W>, g, b, lds, st, a...);
  }
};
// blocks per part of the multi-part launches: one per node, rounded up to whole XCD rounds unless layout 0
static inline int part_grid(int N, int w) {
  const int g = node_grid(N, w);
  return g_part_layout ? (g + 7) & ~7 : g;
}
int launch_bwd_hf1(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre, float* g_t,
                   float* g_geo, const float* vp, const float* pe, const float* g_f, float* g_pe, float* g_vp,
                   float* g_vh, bool with_edge_update) {
  const int w = pick_wpn(D.N);
  const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
  const int with_eu = with_edge_update ? 1 : 0;
  const int il = g_part_layout == 2 ? 1 : 0;
  VSN_DISPATCH_VSA(D.H, D.S, w, g__, KL_k_bwd_hf1,
                   ::go((with_eu ? 4 : 2) * part_grid(D.N, w), node_block(w), node_lds(w, D.S, D.H / 64), st,
                        D, g_vec, vh, tpre, g_t, g_geo, vp, pe, g_f, g_pe, g_vp, g_vh, with_eu, il));
  return 0;
}
int launch_bwd_hf2(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A, float* g_m,
                   float* g_pe, float* g_qkv, float* sat_tmp, float* g_geo, Parts g_m_parts, Parts g_A_parts,
                   const float* vp, const float* g_f, float* g_vp) {
  const int w = pick_wpn(D.N);
  const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
  const int il = g_part_layout == 2 ? 1 : 0;
  VSN_DISPATCH_VSA(D.H, D.S, w, g__, KL_k_bwd_hf2,
                   ::go(2 * part_grid(D.N, w), node_block(w), node_lds(w, D.S, D.H / 64), st,
                        D, qkv, pe, g_A, g_m, g_pe, g_qkv, sat_tmp, g_geo, g_m_parts, g_A_parts, vp, g_f, g_vp, il));
  VSN_DISPATCH_VSA(D.H, D.S, w, g__, KL_k_bwd_attn_S,
                   ::go(node_grid(D.N, w), node_block(w), node_lds(w, 2, D.H / 64), st, D, qkv, pe, g_m, sat_tmp, g_qkv));
  return 0;
}
// edge-update adjoint (both sides) + source side of the vector messages; one launch at single-protein sizes
int launch_bwd_side(hipStream_t st, const Dims& D, const float* vp, const float* pe, const float* g_f, float* g_pe,
                    float* g_vp, float* g_geo, const float* g_vec, const float* tpre, float* g_vh) {
  if (D.N <= 0) return 0;
  const int w = pick_wpn(D.N);
  if (w == 1 || !g_fuse_side) {
    int rc = launch_bwd_edge_update(st, D, vp, pe, g_f, g_pe, g_vp, g_geo);
    if (rc) return rc;
    VSN_LAUNCH_ACT(k_bwd_vecmsg_S, D.S, D, g_vec, tpre, g_vh);
    return 0;
  }
  const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
  VSN_DISPATCH_VSA(D.H, D.S, w, g__, k_bwd_side,
                   <<<3 * node_grid(D.N, w), node_block(w), node_lds(w, D.S, D.H / 64), st>>>(
                       D, vp, pe, g_f, g_pe, g_vp, g_geo, g_vec, tpre, g_vh));
  return 0;
}
int launch_bwd_vecmsg(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre,
                      float* g_t, float* g_vh, float* g_geo) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT(k_bwd_vecmsg_T, 0, D, g_vec, vh, tpre, g_t, g_geo);
  if (g_vh) VSN_LAUNCH_ACT(k_bwd_vecmsg_S, D.S, D, g_vec, tpre, g_vh);
  return 0;
}
int launch_bwd_vecmsg_S(hipStream_t st, const Dims& D, const float* g_vec, const float* tpre, float* g_vh) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT(k_bwd_vecmsg_S, D.S, D, g_vec, tpre, g_vh);
  return 0;
}
int launch_bwd_attn(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A, float* g_m,
                    float* g_pe, float* g_qkv, float* sat_tmp, float* g_geo, Parts g_m_parts, Parts g_A_parts) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT(k_bwd_attn_T, 1, D, qkv, pe, g_A, g_m, g_pe, g_qkv, sat_tmp, g_geo, g_m_parts, g_A_parts);
  {
    const int w = pick_wpn(D.N);
    const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
    VSN_DISPATCH_VSA(D.H, D.S, w, g__, KL_k_bwd_attn_S,
                     ::go(node_grid(D.N, w), node_block(w), node_lds(w, 2, D.H / 64), st, D, qkv, pe, g_m, sat_tmp, g_qkv));
  }
  return 0;
}
// source side (dE/dk, dE/dv) + target side dE/dq, after fused.hip::k_bwd_gf_fused
int launch_bwd_attn_QS(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_m,
                       const float* sat_tmp, float* g_qkv) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT(k_bwd_attn_Q, 1, D, qkv, pe, sat_tmp, g_qkv);
  VSN_LAUNCH_ACT(k_bwd_attn_S, 2, D, qkv, pe, g_m, sat_tmp, g_qkv);
  return 0;
}
int launch_bwd_node_norm(hipStream_t st, const Dims& D, const float* g_xh, int ldg, const float* g_vh,
                         const float* xn, const float* rstd, const float* gamma, const float* wvec, int norm_type,
                         int accumulate, float* g_x, float* g_vec) {
  if (D.N <= 0) return 0;
  VSN_DISPATCH_VS(D.H, D.S, 1, k_bwd_node_norm,
                  <<<node_grid(D.N, 1), 256, 0, st>>>(D, g_xh, ldg, g_vh, xn, rstd, gamma, wvec, norm_type,
                                                      accumulate, g_x, g_vec));
  return 0;
}
int launch_bwd_norm_update(hipStream_t st, const Dims& D, const float* g_xh, int ldg, const float* g_vh,
                           const float* xn, const float* rstd, const float* gamma, const float* wvec,
                           int accumulate, float* g_x, float* g_vec, const float* vp, const float* o, float* g_o,
                           float* g_vp) {
  if (D.N <= 0) return 0;
  const int w__ = pick_wpn(D.N);
  VSN_DISPATCH_VS(D.H, D.S, w__, KL_k_bwd_norm_update,
                  ::go(node_grid(D.N, w__), node_block(w__), w__ == 1 ? 0 : (unsigned)((w__ - 1) * 2 * D.H * 4), st,
                       D, g_xh, ldg, g_vh, xn, rstd, gamma, wvec, accumulate, g_x, g_vec, vp, o, g_o, g_vp));
  return 0;
}
int launch_bwd_embed_edge(hipStream_t st, const Dims& D, const float* x, const float* pp, const float* g_f,
                          float* g_pp, float* g_x, const float* g_xh, const float* xn, const float* rstd,
                          const float* gamma) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH(k_bwd_embed_edge, 1, D, x, pp, g_f, g_pp, g_x, g_xh, xn, rstd, gamma);
  return 0;
}
int launch_bwd_embed_node(hipStream_t st, const Dims& D, const float* emb2, const float* pp, const float* g_n,
                          float* g_pp, float* g_geo) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH(k_bwd_embed_node, 0, D, emb2, pp, g_n, g_pp, g_geo);
  return 0;
}

}  // namespace vsn
