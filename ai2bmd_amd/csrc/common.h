// Shared device helpers for the gfx950 ViSNet kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VSN_WAVE 64
#ifndef VSN_XCD_REMAP
#define VSN_XCD_REMAP 1
#endif

// Launch bounds of a node-walk kernel: WPN waves per node (single-protein sizes) or one wave per node in workgroups of
// four (batches), plus an occupancy HINT per size class (amdgpu_waves_per_eu): BW / MW = the waves per SIMD the
// one-wave-per-node / the WPN-waves-per-node instantiation is compiled for, 0 = the compiler's default.  Why: left alone
// the scheduler aims at 8 waves per SIMD, i.e. <= 64 VGPRs, and gets there by issuing the row loads of an edge ONE AFTER
// THE OTHER into the same registers (k_bwd_vecmsg_S: 8 x {global_load_dwordx4; s_waitcnt vmcnt(0); 4 fmac}); told that
// 4 waves per SIMD are enough it keeps the loads of an edge in flight together (81 VGPRs, 514 -> 425 us on the
// 4096-fragment batch).  A request below the minimum the workgroup size implies is ignored by the compiler, which is how
// "0" is spelled here: (1, 8).  The hinted kernels pass BW = (V <= 4 ? 4 : 0): hidden sizes above 256 need more than the
// 128 registers of that budget per wave and keep the default.  Lab builds: -DVSN_LAB_BATCH_WPE=n / -DVSN_LAB_MD_WPE=n override every kernel's hint.
#if defined(VSN_LAB_BATCH_WPE) || defined(VSN_LAB_MD_WPE)
#ifndef VSN_LAB_BATCH_WPE
#define VSN_LAB_BATCH_WPE 0
#endif
#ifndef VSN_LAB_MD_WPE
#define VSN_LAB_MD_WPE 0
#endif
#define VSN_WPE_B(BW) (VSN_LAB_BATCH_WPE ? VSN_LAB_BATCH_WPE : (BW))
#define VSN_WPE_M(MW) (VSN_LAB_MD_WPE ? VSN_LAB_MD_WPE : (MW))
#else
#define VSN_WPE_B(BW) (BW)
#define VSN_WPE_M(MW) (MW)
#endif
#define VSN_WALK_BOUNDS_H(WPN, BW, MW)                                                                   \
  __launch_bounds__(64 * ((WPN) == 1 ? 4 : (WPN)))                                                       \
      __attribute__((amdgpu_waves_per_eu((WPN) == 1 ? (VSN_WPE_B(BW) ? VSN_WPE_B(BW) : 1) : (VSN_WPE_M(MW) ? 2 : 1), \
                                         (WPN) == 1 ? (VSN_WPE_B(BW) ? VSN_WPE_B(BW) : 8) : (VSN_WPE_M(MW) ? VSN_WPE_M(MW) : 8))))
#define VSN_WALK_BOUNDS(WPN) VSN_WALK_BOUNDS_H(WPN, 0, 0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations (silu == swish; reference utils.py:110-116) ---------------
// sigmoid = v_rcp_f32(1 + v_exp_f32(-x log2 e)) (about 1 ulp each) instead of expf() + an IEEE division.  The
// accurate form is ~28 VALU instructions; with up to 24 sigmoids per edge and layer in the reverse gather kernels the
// activations were the bulk of their instruction stream (Chignolin: 421 -> 434 steps/s, max|dF| against the fp64
// reference 1.5e-6 -> 1.6e-6).  -DVSN_FAST_SIGMOID=0 restores the libm form.
#ifndef VSN_FAST_SIGMOID
#define VSN_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float sigmoid_f(float x) {
#if VSN_FAST_SIGMOID
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
#else
  return 1.0f / (1.0f + expf(-x));
#endif
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ void silu_both(float x, float& y, float& dy) {
  float s = sigmoid_f(x);
  y = x * s;
  dy = s * (1.0f + x * (1.0f - s));
}

// ---- the reference's activation table (ViSNet/model/utils.py:93-116 act_class_mapping) -----------------
// kind is a kernel argument (wave-uniform): silu / swish keep their branch-free fast path.
#define VSN_ACT_SILU 0     // "silu", "swish": x sigmoid(x)
#define VSN_ACT_SSP 1      // "ssp": softplus(x) - ln 2   (F.softplus: beta 1, linear above 20)
#define VSN_ACT_TANH 2     // "tanh"
#define VSN_ACT_SIGMOID 3  // "sigmoid"
__device__ __forceinline__ void act_both(int kind, float x, float& y, float& dy) {
  if (kind == VSN_ACT_SILU) {
    silu_both(x, y, dy);
  } else if (kind == VSN_ACT_SSP) {
    const float s = sigmoid_f(x);
    y = (x > 20.0f ? x : log1pf(expf(x))) - 0.69314718055994531f;
    dy = x > 20.0f ? 1.0f : s;
  } else if (kind == VSN_ACT_TANH) {
    const float t = tanhf(x);
    y = t;
    dy = 1.0f - t * t;
  } else {
    const float s = sigmoid_f(x);
    y = s;
    dy = s * (1.0f - s);
  }
}
__device__ __forceinline__ float act_f(int kind, float x) {
  if (kind == VSN_ACT_SILU) return silu_f(x);
  float y, dy;
  act_both(kind, x, y, dy);
  return y;
}
__device__ __forceinline__ float dact_f(int kind, float x) {
  if (kind == VSN_ACT_SILU) return dsilu_f(x);
  float y, dy;
  act_both(kind, x, y, dy);
  return dy;
}

// ---- wave64 reductions -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Multi-value wave reduction: every lane holds P partial sums p[0..P) (P = 4 or 8); returns, in EVERY lane, the
// total over the 64 lanes of component (lane & (P-1)).  A butterfly that halves the number of live values at each of
// the first log2(P) steps (a lane keeps the components whose index bit matches its lane bit and ships the others to
// its partner): P-1 + (6 - log2 P) cross-lane moves instead of 6 P for P separate wave_sum()s; fixed order.
template <int P>
__device__ __forceinline__ float wave_multi_sum(const float (&p)[P], int lane) {
  static_assert(P == 4 || P == 8, "P must be 4 or 8");
  float v;
  if constexpr (P == 8) {
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float q[4], r[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = b2 ? p[k + 4] : p[k], send = b2 ? p[k] : p[k + 4];
      q[k] = keep + __shfl_xor(send, 4, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float keep = b1 ? q[k + 2] : q[k], send = b1 ? q[k] : q[k + 2];
      r[k] = keep + __shfl_xor(send, 2, 64);
    }
    const float keep = b0 ? r[1] : r[0], send = b0 ? r[0] : r[1];
    v = keep + __shfl_xor(send, 1, 64);
    v += __shfl_xor(v, 8, 64);
  } else {
    const bool b1 = lane & 2, b0 = lane & 1;
    float r[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float keep = b1 ? p[k + 2] : p[k], send = b1 ? p[k] : p[k + 2];
      r[k] = keep + __shfl_xor(send, 2, 64);
    }
    const float keep = b0 ? r[1] : r[0], send = b0 ? r[0] : r[1];
    v = keep + __shfl_xor(send, 1, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
  }
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// Per-head sums for ANY head count: lane l owns channels l V .. l V + V - 1, head(c) = c / hd.  out[t] = the sum of val
// over all channels of the head of the lane's channel t.  One wave reduction per head (the power-of-two fast path is
// group_sum over 64 / nh lanes); fixed order.
template <int V>
__device__ __forceinline__ void head_sums_any(const float (&val)[V], int lane, int hd, int nh, float (&out)[V]) {
  int hidx[V];
#pragma unroll
  for (int t = 0; t < V; ++t) {
    hidx[t] = (lane * V + t) / hd;
    out[t] = 0.f;
  }
  for (int h = 0; h < nh; ++h) {
    float p = 0.f;
#pragma unroll
    for (int t = 0; t < V; ++t) p += hidx[t] == h ? val[t] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
#pragma unroll
    for (int t = 0; t < V; ++t) out[t] = hidx[t] == h ? p : out[t];
  }
}

// sum over aligned groups of `width` consecutive lanes (width power of two)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- per-lane row fragments: a row of H = 64*V floats, lane owns V contiguous --
template <int V>
struct RowVec;
template <>
struct RowVec<1> {
  typedef float T;
};
template <>
struct RowVec<2> {
  typedef float2 T;
};
template <>
struct RowVec<4> {
  typedef float4 T;
};

// V in {1, 2, 4}: one 4/8/16-byte access per lane.  Other widths (hidden = 192, 320, 384, 448, 512) split the
// lane's V contiguous floats into the widest aligned pieces (V = 8: two 16-byte accesses; odd V: dwords).
template <int V>
__device__ __forceinline__ void ldrow(const float* __restrict__ row, int lane, float (&r)[V]) {
  if constexpr (V == 1 || V == 2 || V == 4) {
    typedef typename RowVec<V>::T T;
    T t = *reinterpret_cast<const T*>(row + lane * V);
    const float* p = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = p[i];
  } else if constexpr (V % 4 == 0) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(row + lane * V + 4 * q);
      r[4 * q] = t.x, r[4 * q + 1] = t.y, r[4 * q + 2] = t.z, r[4 * q + 3] = t.w;
    }
  } else if constexpr (V % 2 == 0) {
#pragma unroll
    for (int q = 0; q < V / 2; ++q) {
      const float2 t = *reinterpret_cast<const float2*>(row + lane * V + 2 * q);
      r[2 * q] = t.x, r[2 * q + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = row[lane * V + i];
  }
}
template <int V>
__device__ __forceinline__ void strow(float* __restrict__ row, int lane, const float (&r)[V]) {
  if constexpr (V == 1 || V == 2 || V == 4) {
    typedef typename RowVec<V>::T T;
    T t;
    float* p = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = r[i];
    *reinterpret_cast<T*>(row + lane * V) = t;
  } else if constexpr (V % 4 == 0) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q)
      *reinterpret_cast<float4*>(row + lane * V + 4 * q) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
  } else if constexpr (V % 2 == 0) {
#pragma unroll
    for (int q = 0; q < V / 2; ++q) *reinterpret_cast<float2*>(row + lane * V + 2 * q) = make_float2(r[2 * q], r[2 * q + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) row[lane * V + i] = r[i];
  }
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XCD-aware block remap: the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2.
// Renumbering so that CONSECUTIVE logical blocks share an XCD keeps a fragment's node rows (which the
// ~17 edges per node gather again and again) in ONE L2 instead of all eight.  Bijective for any grid.
__device__ __forceinline__ int xcd_block(int b, int G) {
  const int xcd = b & 7, idx = b >> 3;
  const int q = G >> 3, r = G & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Node loop.  WPN = waves cooperating on ONE node:
//   WPN == 1 : one wave per node, (blockDim/64) nodes per workgroup (large batches);
//   WPN  > 1 : the workgroup's WPN waves split the node's edge list (edge e0+sub, e0+sub+WPN, ...)
//              and combine their partial sums through LDS in a fixed order (small batches,
//              e.g. one protein per MD step, where N waves cannot fill 256 CUs).
#define VSN_NODE_LOOP_B(node, N, WPN, BID, NBLK)                                  \
  const int lane = threadIdx.x & 63;                                             \
  const int wv__ = uni((int)(threadIdx.x >> 6)); /* wave id: make it an SGPR */  \
  const int sub = (WPN) == 1 ? 0 : wv__;                                         \
  const int npb__ = (WPN) == 1 ? (int)(blockDim.x >> 6) : 1;                     \
  const int nblk__ = (NBLK);                                                     \
  const int blk__ = VSN_XCD_REMAP ? xcd_block((BID), nblk__) : (BID);            \
  for (int node = blk__ * npb__ + ((WPN) == 1 ? wv__ : 0); node < (N); node += nblk__ * npb__)
#define VSN_NODE_LOOP(node, N, WPN) VSN_NODE_LOOP_B(node, N, WPN, (int)blockIdx.x, (int)gridDim.x)

// Edge-id cache: one coalesced load brings the ids of (up to) 64 consecutive edges of a node into the
// wave's lanes; the edge loop then picks them with v_readlane instead of a dependent scalar load per edge
// (halves the dependent-load chain of every edge iteration: id -> row instead of ptr -> id -> row).
__device__ __forceinline__ int edge_cache_load(const int* __restrict__ ids, int e0, int e1, int lane) {
  return (e0 + lane < e1) ? ids[e0 + lane] : 0;
}
__device__ __forceinline__ int edge_cache_get(int cache, const int* __restrict__ ids, int e, int e0) {
  const int k = uni(e - e0);
  return k < 64 ? __builtin_amdgcn_readlane(cache, k) : uni(ids[e]);
}

// sums acc[K][V] over the WPN waves of the workgroup into wave 0 (fixed order -> deterministic).
// Rows go through LDS in chunks of at most VSN_REDUCE_ROWS so the scratch stays <= 57 KB.
#define VSN_REDUCE_ROWS 8
template <int V, int K, int WPN>
__device__ __forceinline__ void node_reduce(float (&acc)[K][V], float* __restrict__ smem, int lane, int sub) {
  if (WPN == 1) return;
  constexpr int RMAX = (32 / V) < VSN_REDUCE_ROWS ? (32 / V) : VSN_REDUCE_ROWS;  // <= 57 KB of scratch for any V
  constexpr int KC = K < RMAX ? K : RMAX;
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();  // smem may still be read from the previous use
    if (sub > 0) {
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k0 + k < K) {
#pragma unroll
          for (int c = 0; c < V; ++c) smem[(((sub - 1) * KC + k) * 64 + lane) * V + c] = acc[k0 + k][c];
        }
    }
    __syncthreads();
    if (sub == 0) {
      for (int w = 1; w < WPN; ++w)
#pragma unroll
        for (int k = 0; k < KC; ++k)
          if (k0 + k < K) {
#pragma unroll
            for (int c = 0; c < V; ++c) acc[k0 + k][c] += smem[(((w - 1) * KC + k) * 64 + lane) * V + c];
          }
    }
  }
}

// Reduce-SCATTER over the WPN waves of the workgroup: wave s (s < K <= WPN) receives, in tot[], the sum over all waves
// of acc[s][] (fixed order -> deterministic); every wave then finishes / stores its own row in parallel instead of
// wave 0 walking all K rows.  LDS: K*(WPN-1) rows of 64*V floats (same budget as node_reduce with K rows).
template <int V, int K, int WPN>
__device__ __forceinline__ void node_reduce_scatter(const float (&acc)[K][V], float (&tot)[V],
                                                    float* __restrict__ smem, int lane, int sub) {
  static_assert(K <= WPN, "one wave per row");
  __syncthreads();  // smem may still be read from the previous use
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (k != sub) {
      float* dst = smem + ((size_t)(k * (WPN - 1) + (sub < k ? sub : sub - 1)) * 64 + lane) * V;
#pragma unroll
      for (int c = 0; c < V; ++c) dst[c] = acc[k][c];
    }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < V; ++c) tot[c] = 0.f;
  if (sub < K) {
    float own[V];
#pragma unroll
    for (int c = 0; c < V; ++c) own[c] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k == sub) {
#pragma unroll
        for (int c = 0; c < V; ++c) own[c] = acc[k][c];
      }
    for (int w = 0; w < WPN; ++w) {
      if (w == sub) {
#pragma unroll
        for (int c = 0; c < V; ++c) tot[c] += own[c];
      } else {
        const float* src_ = smem + ((size_t)(sub * (WPN - 1) + (w < sub ? w : w - 1)) * 64 + lane) * V;
#pragma unroll
        for (int c = 0; c < V; ++c) tot[c] += src_[c];
      }
    }
  }
}
