// Forward gather / scatter / vector-feature kernels of the ViS-MP layers.
//
// Pattern: one wave64 per TARGET node walks the node's in-edges (CSR by target);
// every lane owns V = H/64 contiguous channels, so a row of H floats is one
// fully coalesced 64-lane access (16 B/lane at H = 256).  Per-target state
// (q_i, w_trg v_i, accumulators) lives in registers for the whole edge loop,
// edge scalars (source index, d_ij, cutoff) come through the scalar path,
// segment sums are register accumulations in a fixed order: no atomics, results
// are bit-reproducible run to run.  The dense products between these kernels
// run on the MFMA GEMM (gemm.hip).
//
// Reference: ViSNet/model/utils.py:296-317 (NeighborEmbedding), :331-341
// (EdgeEmbedding); visnet_block.py:237-312 (ViS_MP.forward/message/aggregate/
// edge_update), :206-209 (vector_rejection); utils.py:165-249 (VecLayerNorm).
#include <cstdlib>

#include <hip/hip_ext.h>

#include "common.h"
#include "kernels.h"

namespace vsn {

// template dispatch on V = H/64 (1,2,4), S (3,8) and WPN (1 or VSN_WPN_SMALL)
#ifndef VSN_WPN_SMALL
#define VSN_WPN_SMALL 8
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

// Lab builds (-DVSN_LAB_STAMPS): lane 0 of every wave of k_node_update stamps the 100 MHz real-time counter at its
// phase boundaries into a buffer set with vsn_lab_set_stamps() (tools/lab/stamps_node_update.py prints the phases).
#ifdef VSN_LAB_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define VSN_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (g_stamps && (threadIdx.x & 63) == 0)                                                                  \
      g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define VSN_STAMP(k) do {} while (0)
#endif
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

// fused LayerNorm of the NEXT layer / read-out on one node row held by a wave (same arithmetic as k_node_norm)
template <int V>
__device__ __forceinline__ void node_layernorm_store(const NextNorm& nn, int i, int H, int lane, float (&xv)[V]) {
  const float invH = 1.0f / (float)H;
  float g[V], bta[V], sm_ = 0.f;
  ldrow<V>(nn.gamma, lane, g);
  ldrow<V>(nn.beta, lane, bta);
#pragma unroll
  for (int c = 0; c < V; ++c) sm_ += xv[c];
  const float mean = wave_sum(sm_) * invH;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < V; ++c) {
    xv[c] -= mean;
    q += xv[c] * xv[c];
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) * invH + 1e-5f);
  float n[V], hh[V];
#pragma unroll
  for (int c = 0; c < V; ++c) {
    n[c] = xv[c] * rs;
    hh[c] = n[c] * g[c] + bta[c];
  }
  strow<V>(nn.xn + (size_t)i * H, lane, n);
  strow<V>(nn.xh + (size_t)i * nn.ldxh, lane, hh);
  if (lane == 0) nn.rstd[i] = rs;
}

// ---- embeddings -------------------------------------------------------------
// cat[i] = [ emb1[z_i] | sum_{j->i, j!=i} emb2[z_j] * phi_e * C_e ]   (utils.py:296-317)
template <int V, int S, int WPN>
__global__ VSN_WALK_BOUNDS(WPN) void k_embed_node(Dims D, const float* __restrict__ emb1,
                                                                          const float* __restrict__ emb2,
                                                                          const float* __restrict__ pp,
                                                                          float* __restrict__ cat) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = D.H;
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float acc[1][V];
#pragma unroll
    for (int c = 0; c < V; ++c) acc[0][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      if (j == i) continue;
      const float C = D.geo[(size_t)e * 8 + 1];
      float em[V], ph[V];
      ldrow<V>(emb2 + (size_t)uni(D.zi[j]) * H, lane, em);
      ldrow<V>(pp + (size_t)e * 2 * H, lane, ph);
#pragma unroll
      for (int c = 0; c < V; ++c) acc[0][c] += em[c] * (ph[c] * C);
    }
    node_reduce<V, 1, WPN>(acc, smem, lane, sub);
    if (sub == 0) {
      float x0[V];
      ldrow<V>(emb1 + (size_t)uni(D.zi[i]) * H, lane, x0);
      strow<V>(cat + (size_t)i * 2 * H, lane, x0);
      strow<V>(cat + (size_t)i * 2 * H + H, lane, acc[0]);
    }
  }
}

// f_e = (x_i + x_j) * psi_e for all edges incl. loops (utils.py:331-337); vec = 0
// nn.xn != nullptr: also layer 0's LayerNorm of x and its VecLayerNorm("none") of vec == 0 (vh = 0), i.e. what
// k_node_norm would do in a launch of its own
template <int V, int S, int WPN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_embed_edge(Dims D, const float* __restrict__ x,
                                                                          const float* __restrict__ pp,
                                                                          float* __restrict__ f,
                                                                          float* __restrict__ vec,
                                                                          float* __restrict__ xcopy, NextNorm nn) {
  const int H = D.H;
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float xi[V];
    ldrow<V>(x + (size_t)i * H, lane, xi);
    if (sub == 0) strow<V>(xcopy + (size_t)i * H, lane, xi);  // running x of the layers starts as a copy
    if (nn.xn && sub == (WPN == 1 ? 0 : WPN - 1)) {  // (the last wave: it has the fewest edges of the node)
      float xv[V];
#pragma unroll
      for (int c = 0; c < V; ++c) xv[c] = xi[c];
      node_layernorm_store<V>(nn, i, H, lane, xv);
    }
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float xj[V], ps[V], o[V];
      ldrow<V>(x + (size_t)j * H, lane, xj);
      ldrow<V>(pp + (size_t)e * 2 * H + H, lane, ps);
#pragma unroll
      for (int c = 0; c < V; ++c) o[c] = (xi[c] + xj[c]) * ps[c];
      strow<V>(f + (size_t)e * H, lane, o);
    }
    float zr[V];
#pragma unroll
    for (int c = 0; c < V; ++c) zr[c] = 0.f;
    for (int s = sub; s < S; s += WPN) {
      strow<V>(vec + ((size_t)i * S + s) * H, lane, zr);
      if (nn.xn) strow<V>(nn.vh + ((size_t)i * S + s) * H, lane, zr);
    }
  }
}

// ---- LayerNorm + VecLayerNorm (visnet_block.py:238-239, utils.py:186-249) -----
template <int V, int S, int WPN>
__global__ __launch_bounds__(256) void k_node_norm(Dims D, const float* __restrict__ x, const float* __restrict__ vec,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ wvec, int norm_type,
                                                   float* __restrict__ xn, float* __restrict__ rstd,
                                                   float* __restrict__ xh, int ldxh, float* __restrict__ vh) {
  const int H = D.H;
  const float invH = 1.0f / (float)H;
  VSN_NODE_LOOP(i, D.N, 1) {
    (void)sub;
    float xv[V], g[V], b[V], w[V];
    ldrow<V>(x + (size_t)i * H, lane, xv);
    ldrow<V>(gamma, lane, g);
    ldrow<V>(beta, lane, b);
    ldrow<V>(wvec, lane, w);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c) s += xv[c];
    const float mean = wave_sum(s) * invH;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      xv[c] -= mean;
      q += xv[c] * xv[c];
    }
    const float var = wave_sum(q) * invH;
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    float n[V], h[V];
#pragma unroll
    for (int c = 0; c < V; ++c) {
      n[c] = xv[c] * rs;
      h[c] = n[c] * g[c] + b[c];
    }
    strow<V>(xn + (size_t)i * H, lane, n);
    strow<V>(xh + (size_t)i * ldxh, lane, h);
    if (lane == 0) rstd[i] = rs;
    // VecLayerNorm, norm_type 0 ("none"): vh = vec * weight  (rms / max_min: vecnorm.hip)
    if (norm_type == 0)
#pragma unroll
    for (int sidx = 0; sidx < S; ++sidx) {
      float v[V];
      ldrow<V>(vec + ((size_t)i * S + sidx) * H, lane, v);
#pragma unroll
      for (int c = 0; c < V; ++c) v[c] *= w[c];
      strow<V>(vh + ((size_t)i * S + sidx) * H, lane, v);
    }
  }
}

// ---- attention + scalar message + its aggregation (visnet_block.py:276-283,305) --
// a_h = silu(sum_c q_i k_j dk) * C ; m_e = v_j * dv * a ; A_i = sum_e m_e
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void edge_attn_body(const Dims& D, const float* __restrict__ qkv,
                                               const float* __restrict__ pe, float* __restrict__ m,
                                               float* __restrict__ A, float* __restrict__ smem, int bid, int nblk) {
  const int H = D.H;
  const int lph = 64 / D.nh;  // lanes per head
  VSN_NODE_LOOP_B(i, D.N, WPN, bid, nblk) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float q[V], acc[1][V];
    ldrow<V>(qkv + (size_t)i * 3 * H, lane, q);
#pragma unroll
    for (int c = 0; c < V; ++c) acc[0][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      const float C = D.geo[(size_t)e * 8 + 1];
      float k[V], v[V], pk[V], pv[V], mv[V];
      ldrow<V>(qkv + (size_t)j * 3 * H + H, lane, k);
      ldrow<V>(qkv + (size_t)j * 3 * H + 2 * H, lane, v);
      ldrow<V>(pe + (size_t)e * 3 * H, lane, pk);
      ldrow<V>(pe + (size_t)e * 3 * H + H, lane, pv);
      float av[V];  // the attention weight of the head of each of the lane's channels
      bool any_heads = false;
      if constexpr (GEN) any_heads = D.hgen != 0;
      if (any_heads) {  // head count that does not divide 64: a lane's channels may straddle heads
        float pc[V], sc[V];
#pragma unroll
        for (int c = 0; c < V; ++c) pc[c] = q[c] * k[c] * act_f(D.act, pk[c]);
        head_sums_any<V>(pc, lane, D.hd, D.nh, sc);
#pragma unroll
        for (int c = 0; c < V; ++c) av[c] = act_f(D.attn_act, sc[c]) * C;
      } else {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) part += q[c] * k[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pk[c]);
        const float sat = group_sum(part, lph);
        const float a = act_f((GEN ? D.attn_act : VSN_ACT_SILU), sat) * C;
#pragma unroll
        for (int c = 0; c < V; ++c) av[c] = a;
      }
#pragma unroll
      for (int c = 0; c < V; ++c) {
        mv[c] = v[c] * act_f((GEN ? D.act : VSN_ACT_SILU), pv[c]) * av[c];
        acc[0][c] += mv[c];
      }
      strow<V>(m + (size_t)e * H, lane, mv);
    }
    node_reduce<V, 1, WPN>(acc, smem, lane, sub);
    if (sub == 0) strow<V>(A + (size_t)i * H, lane, acc[0]);
  }
}
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_edge_attn(Dims D, const float* __restrict__ qkv,
                                                                         const float* __restrict__ pe,
                                                                         float* __restrict__ m,
                                                                         float* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  edge_attn_body<V, S, WPN, GEN>(D, qkv, pe, m, A, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ---- vector messages, their aggregation and the node update ---------------------
// V_i[s] = sum_e vh_j[s]*s1_e + d_e[s]*s2_e ; dx = (sum_s vec1 vec2) o2 + o3 ;
// dvec = vec3 o1 + V ; x += dx ; vec += dvec    (visnet_block.py:284-288,271-274,129-137)
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_node_update(Dims D, const float* __restrict__ tpre,
                                                                           const float* __restrict__ vh,
                                                                           const float* __restrict__ vp,
                                                                           const float* __restrict__ o,
                                                                           float* __restrict__ x,
                                                                           float* __restrict__ vec, NextNorm nn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = D.H;
  VSN_STAMP(0);
  VSN_NODE_LOOP(i, D.N, WPN) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    VSN_STAMP(1);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float Va[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < V; ++c) Va[s][c] = 0.f;
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float s1[V], s2[V];
      ldrow<V>(tpre + (size_t)e * 2 * H, lane, s1);
      ldrow<V>(tpre + (size_t)e * 2 * H + H, lane, s2);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        s1[c] = act_f((GEN ? D.act : VSN_ACT_SILU), s1[c]);
        s2[c] = act_f((GEN ? D.act : VSN_ACT_SILU), s2[c]);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float vj[V];
        ldrow<V>(vh + ((size_t)j * S + s) * H, lane, vj);
        const float ds = D.d[(size_t)e * 8 + s];
#pragma unroll
        for (int c = 0; c < V; ++c) Va[s][c] += vj[c] * s1[c] + ds * s2[c];
      }
      VSN_STAMP(e < e0 + WPN ? 2 : (e < e0 + 2 * WPN ? 3 : 4));  // after the wave's 1st / 2nd / 3rd edge
    }
    VSN_STAMP(5);
    if constexpr (WPN > 1 && S <= WPN && V <= 4) {
      // Small batches: instead of summing everything into wave 0 and letting it walk the S components alone,
      // reduce-SCATTER the partial sums (wave s receives the node total of component s) and let the S waves
      // update their component in parallel; only the channel-wise sum over s and the LayerNorm stay on wave 0.
      float tot[V], vdp[V];
      node_reduce_scatter<V, S, WPN>(Va, tot, smem, lane, sub);
      VSN_STAMP(6);
#pragma unroll
      for (int c = 0; c < V; ++c) vdp[c] = 0.f;
      if (sub < S) {
        float o1[V], v1[V], v2[V], v3[V], vv[V];
        ldrow<V>(o + (size_t)i * 3 * H, lane, o1);
        const float* row = vp + ((size_t)i * S + sub) * 5 * H;
        ldrow<V>(row, lane, v1);
        ldrow<V>(row + H, lane, v2);
        ldrow<V>(row + 2 * H, lane, v3);
        ldrow<V>(vec + ((size_t)i * S + sub) * H, lane, vv);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          vdp[c] = v1[c] * v2[c];
          vv[c] += v3[c] * o1[c] + tot[c];
        }
        strow<V>(vec + ((size_t)i * S + sub) * H, lane, vv);
        if (nn.xn) {  // fused VecLayerNorm("none") of the NEXT layer / read-out: vh = vec * weight
          float w[V];
          ldrow<V>(nn.wvec, lane, w);
#pragma unroll
          for (int c = 0; c < V; ++c) vv[c] *= w[c];
          strow<V>(nn.vh + ((size_t)i * S + sub) * H, lane, vv);
        }
      }
      VSN_STAMP(7);
      __syncthreads();  // everybody is done reading the partial sums
      VSN_STAMP(8);
      if (sub > 0 && sub < S) {
        float* dst = smem + ((size_t)(sub - 1) * 64 + lane) * V;
#pragma unroll
        for (int c = 0; c < V; ++c) dst[c] = vdp[c];
      }
      __syncthreads();
      VSN_STAMP(9);
      if (sub == 0) {
        float vd[V], o2[V], o3[V], xv[V];
#pragma unroll
        for (int c = 0; c < V; ++c) vd[c] = vdp[c];
        for (int s = 1; s < S; ++s) {
          const float* src_ = smem + ((size_t)(s - 1) * 64 + lane) * V;
#pragma unroll
          for (int c = 0; c < V; ++c) vd[c] += src_[c];
        }
        ldrow<V>(o + (size_t)i * 3 * H + H, lane, o2);
        ldrow<V>(o + (size_t)i * 3 * H + 2 * H, lane, o3);
        ldrow<V>(x + (size_t)i * H, lane, xv);
#pragma unroll
        for (int c = 0; c < V; ++c) xv[c] += vd[c] * o2[c] + o3[c];
        strow<V>(x + (size_t)i * H, lane, xv);
        VSN_STAMP(10);
        if (nn.xn) node_layernorm_store<V>(nn, i, H, lane, xv);
        VSN_STAMP(11);
      }
    } else {
      node_reduce<V, S, WPN>(Va, smem, lane, sub);
    if (sub == 0) {
      float o1[V], o2[V], o3[V], vd[V];
      ldrow<V>(o + (size_t)i * 3 * H, lane, o1);
      ldrow<V>(o + (size_t)i * 3 * H + H, lane, o2);
      ldrow<V>(o + (size_t)i * 3 * H + 2 * H, lane, o3);
#pragma unroll
      for (int c = 0; c < V; ++c) vd[c] = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float* row = vp + ((size_t)i * S + s) * 5 * H;
        float v1[V], v2[V], v3[V], vv[V];
        ldrow<V>(row, lane, v1);
        ldrow<V>(row + H, lane, v2);
        ldrow<V>(row + 2 * H, lane, v3);
        ldrow<V>(vec + ((size_t)i * S + s) * H, lane, vv);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          vd[c] += v1[c] * v2[c];
          vv[c] += v3[c] * o1[c] + Va[s][c];
        }
        strow<V>(vec + ((size_t)i * S + s) * H, lane, vv);
        if (nn.xn) {  // fused VecLayerNorm("none") of the NEXT layer / read-out: vh = vec * weight
          float w[V];
          ldrow<V>(nn.wvec, lane, w);
#pragma unroll
          for (int c = 0; c < V; ++c) vv[c] *= w[c];
          strow<V>(nn.vh + ((size_t)i * S + s) * H, lane, vv);
        }
      }
      float xv[V];
      ldrow<V>(x + (size_t)i * H, lane, xv);
#pragma unroll
      for (int c = 0; c < V; ++c) xv[c] += vd[c] * o2[c] + o3[c];
      strow<V>(x + (size_t)i * H, lane, xv);
      if (nn.xn) node_layernorm_store<V>(nn, i, H, lane, xv);
    }
    }
  }
}

// (Round 4: an S-SPLIT form of this kernel at single-protein sizes - waves own the spherical components instead of
//  splitting the edges, the activations of an edge staged through LDS once, four vh rows in flight, no cross-wave
//  reduction - measured 14.4-14.5 us against 13.8-14.0 us for the form above, Chignolin 447.8-448.5 vs 448.8-449.0
//  steps/s, and was removed: LAB_NOTES.md section 11.)

// ---- edge update (visnet_block.py:290-295): f_e += silu(pf_e) * <rej(wt_i,d), rej(ws_j,d)> ---
// <w1,w2> = u1.u2 + (u1.d)(u2.d)(|d|^2 - 2)   (expanded double rejection)
template <int V, int S, int WPN, bool GEN>
__device__ __forceinline__ void edge_update_body(const Dims& D, const float* __restrict__ vp,
                                                 const float* __restrict__ pe, float* __restrict__ f, int bid,
                                                 int nblk) {
  const int H = D.H;
  VSN_NODE_LOOP_B(i, D.N, WPN, bid, nblk) {
    const int e0 = uni(D.rowptr[i]), e1 = uni(D.rowptr[i + 1]);
    const int srcc = edge_cache_load(D.src, e0, e1, lane);
    float wt[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s) ldrow<V>(vp + ((size_t)i * S + s) * 5 * H + 3 * H, lane, wt[s]);
    for (int e = e0 + sub; e < e1; e += WPN) {
      const int j = edge_cache_get(srcc, D.src, e, e0);
      float dot[V], a1[V], a2[V];
#pragma unroll
      for (int c = 0; c < V; ++c) dot[c] = a1[c] = a2[c] = 0.f;
      float cc = -2.0f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float u2[V];
        ldrow<V>(vp + ((size_t)j * S + s) * 5 * H + 4 * H, lane, u2);
        const float ds = D.d[(size_t)e * 8 + s];
        cc += ds * ds;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          dot[c] += wt[s][c] * u2[c];
          a1[c] += wt[s][c] * ds;
          a2[c] += u2[c] * ds;
        }
      }
      float pf[V], fv[V];
      ldrow<V>(pe + (size_t)e * 3 * H + 2 * H, lane, pf);
      ldrow<V>(f + (size_t)e * H, lane, fv);
#pragma unroll
      for (int c = 0; c < V; ++c) fv[c] += act_f((GEN ? D.act : VSN_ACT_SILU), pf[c]) * (dot[c] + a1[c] * a2[c] * cc);
      strow<V>(f + (size_t)e * H, lane, fv);
    }
  }
}
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS_H(WPN, (V <= 4 ? 4 : 0), 0) void k_edge_update(Dims D, const float* __restrict__ vp,
                                                                           const float* __restrict__ pe,
                                                                           float* __restrict__ f) {
  edge_update_body<V, S, WPN, GEN>(D, vp, pe, f, (int)blockIdx.x, (int)gridDim.x);
}
// Horizontal fusion of the two independent edge walks of a layer (both consume the edge linears `pe`):
// blocks [0, G) do the attention, blocks [G, 2G) the edge update.  On a single protein both are latency-bound,
// so one launch runs them side by side without the ~15 us event latency a second stream would cost.
template <int V, int S, int WPN, bool GEN>
__global__ VSN_WALK_BOUNDS(WPN) void k_edge_attn_update(Dims D, const float* __restrict__ qkv,
                                                                                const float* __restrict__ pe,
                                                                                float* __restrict__ m,
                                                                                float* __restrict__ A,
                                                                                const float* __restrict__ vp,
                                                                                float* __restrict__ f) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = (int)gridDim.x >> 1;
  if ((int)blockIdx.x < G) edge_attn_body<V, S, WPN, GEN>(D, qkv, pe, m, A, smem, (int)blockIdx.x, G);
  else edge_update_body<V, S, WPN, GEN>(D, vp, pe, f, (int)blockIdx.x - G, G);
}

// ---- launchers -------------------------------------------------------------------
// Profile mode (engine.hip, ScatterBracket): the scatter-path launches carry a pair of events ON THEIR DISPATCH
// PACKET (hipExtLaunchKernelGGL).  The elapsed time between the two is the kernel's own begin / end timestamp - the
// pair rocprofv3's kernel trace records - not a stream bracket whose marker packets add a size-dependent gap.
static thread_local LaunchEvents tl_ev_q[8];
static thread_local int tl_ev_n = 0, tl_ev_head = 0;
void set_launch_events(const LaunchEvents* ev) {
  tl_ev_n = tl_ev_head = 0;
  if (ev) tl_ev_q[tl_ev_n++] = *ev;
}
void push_launch_events(const LaunchEvents& ev) {
  if (tl_ev_head == tl_ev_n) tl_ev_n = tl_ev_head = 0;
  if (tl_ev_n < 8) tl_ev_q[tl_ev_n++] = ev;
}
bool take_launch_events(int kind, LaunchEvents* out) {
  if (tl_ev_head >= tl_ev_n || tl_ev_q[tl_ev_head].kind != kind) return false;
  *out = tl_ev_q[tl_ev_head++];
  return true;
}
#define VSN_KL(NAME, KIND)                                                                 \
  template <int V, int S, int W, bool G>                                                   \
  struct KL_##NAME {                                                                       \
    template <typename... A>                                                               \
    static void go(dim3 g, dim3 b, unsigned lds, hipStream_t st, A... a) {                 \
      launch_maybe_timed(KIND, NAME<V, S, W, G>, g, b, lds, st, a...);                     \
    }                                                                                      \
  }
VSN_KL(k_edge_attn, WK_EDGE_ATTN);
VSN_KL(k_edge_attn_update, WK_EDGE_ATTN);
VSN_KL(k_node_update, WK_NODE_UPDATE);
#undef VSN_KL
#define VSN_LAUNCH_ACT_TIMED(KN, RK, ...)                                                               \
  do {                                                                                                  \
    const int w__ = pick_wpn(D.N);                                                                      \
    const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;                     \
    VSN_DISPATCH_VSA(D.H, D.S, w__, g__, KL_##KN,                                                       \
                     ::go(node_grid(D.N, w__), node_block(w__), node_lds(w__, (RK), D.H / 64), st, __VA_ARGS__)); \
  } while (0)
#define VSN_LAUNCH_ACT(KN, RK, ...)                                                                     \
  do {                                                                                                  \
    const int w__ = pick_wpn(D.N);                                                                      \
    const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;                     \
    VSN_DISPATCH_VSA(D.H, D.S, w__, g__, KN,                                                            \
                     <<<node_grid(D.N, w__), node_block(w__), node_lds(w__, (RK), D.H / 64), st>>>(__VA_ARGS__)); \
  } while (0)
#define VSN_LAUNCH(KN, RK, ...)                                                                         \
  do {                                                                                                  \
    const int w__ = pick_wpn(D.N);                                                                      \
    VSN_DISPATCH_VS(D.H, D.S, w__, KN,                                                                  \
                    <<<node_grid(D.N, w__), node_block(w__), node_lds(w__, (RK), D.H / 64), st>>>(__VA_ARGS__)); \
  } while (0)

int launch_embed_node(hipStream_t st, const Dims& D, const float* emb1, const float* emb2, const float* pp,
                      float* cat) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH(k_embed_node, 1, D, emb1, emb2, pp, cat);
  return 0;
}
int launch_embed_edge(hipStream_t st, const Dims& D, const float* x, const float* pp, float* f, float* vec,
                      float* xcopy, const NextNorm& nn) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH(k_embed_edge, 0, D, x, pp, f, vec, xcopy, nn);
  return 0;
}
int launch_node_norm(hipStream_t st, const Dims& D, const float* x, const float* vec, const float* gamma,
                     const float* beta, const float* wvec, int norm_type, float* xn, float* rstd, float* xh,
                     int ldxh, float* vh) {
  if (D.N <= 0) return 0;
  VSN_DISPATCH_VS(D.H, D.S, 1, k_node_norm,
                  <<<node_grid(D.N, 1), 256, 0, st>>>(D, x, vec, gamma, beta, wvec, norm_type, xn, rstd, xh, ldxh,
                                                      vh));
  return 0;
}
int launch_edge_attn(hipStream_t st, const Dims& D, const float* qkv, const float* pe, float* m, float* A) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT_TIMED(k_edge_attn, 1, D, qkv, pe, m, A);
  return 0;
}
int launch_edge_attn_update(hipStream_t st, const Dims& D, const float* qkv, const float* pe, float* m, float* A,
                            const float* vp, float* f) {
  if (D.N <= 0) return 0;
  const int w__ = pick_wpn(D.N);
  const bool g__ = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU || D.hgen;
  VSN_DISPATCH_VSA(D.H, D.S, w__, g__, KL_k_edge_attn_update,
                   ::go(2 * node_grid(D.N, w__), node_block(w__), node_lds(w__, 1, D.H / 64), st, D, qkv, pe, m, A,
                        vp, f));
  return 0;
}
int launch_node_update(hipStream_t st, const Dims& D, const float* tpre, const float* vh, const float* vp,
                       const float* o, float* x, float* vec, const NextNorm& nn) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT_TIMED(k_node_update, D.S, D, tpre, vh, vp, o, x, vec, nn);
  return 0;
}
int launch_edge_update(hipStream_t st, const Dims& D, const float* vp, const float* pe, float* f) {
  if (D.N <= 0) return 0;
  VSN_LAUNCH_ACT(k_edge_update, 0, D, vp, pe, f);
  return 0;
}

__global__ void k_fill(float* __restrict__ p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
int launch_fill(hipStream_t st, float* p, size_t n, float v) {
  if (n == 0) return 0;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
  return 0;
}

}  // namespace vsn

#ifdef VSN_LAB_STAMPS
extern "C" int vsn_lab_set_stamps(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vsn::g_stamps), &q, sizeof(q));
}
#endif
