// Shared device helpers for the gfx950 ViSNet kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VSN_WAVE 64
#ifndef VSN_XCD_REMAP
#define VSN_XCD_REMAP 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations (silu == swish; reference utils.py:110-116) ---------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
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

// ---- wave64 reductions -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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

template <int V>
__device__ __forceinline__ void ldrow(const float* __restrict__ row, int lane, float (&r)[V]) {
  typedef typename RowVec<V>::T T;
  T t = *reinterpret_cast<const T*>(row + lane * V);
  const float* p = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) r[i] = p[i];
}
template <int V>
__device__ __forceinline__ void strow(float* __restrict__ row, int lane, const float (&r)[V]) {
  typedef typename RowVec<V>::T T;
  T t;
  float* p = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) p[i] = r[i];
  *reinterpret_cast<T*>(row + lane * V) = t;
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
  constexpr int KC = K < VSN_REDUCE_ROWS ? K : VSN_REDUCE_ROWS;
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
