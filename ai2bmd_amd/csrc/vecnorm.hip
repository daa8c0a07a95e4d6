// VecLayerNorm "rms" and "max_min" (reference: ViSNet/model/utils.py:186-249) and their
// hand-derived adjoints.  ("none" is folded into k_node_norm / k_bwd_node_norm.)
//
// vec [N,S,H]; S = 3 -> one block of components, S = 8 -> blocks [0,3) and [3,8) normalised
// separately (utils.py:230-247).  Per block b and channel c:
//     dist_c = max(||v[b,:,c]||_2, 1e-12)
//   rms     : r = sqrt(mean_c dist_c^2)            out = v / r * w_c
//   max_min : m = min_c dist_c, M = max_c dist_c, D = M - m (1 if 0)
//             out = relu((dist_c - m)/D) * v / dist_c * w_c
// One wave per node; lane owns V = H/64 channels; min/max/sum over channels are wave reductions.
// The reference's `(dist == 0).all()` early-outs return the same values as these formulas.
#include "common.h"
#include "kernels.h"

namespace vsn {

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

template <int V, int S>
__global__ __launch_bounds__(256) void k_vecnorm_fwd(int N, int H, int norm_type, const float* __restrict__ vec,
                                                     const float* __restrict__ w, float* __restrict__ vin,
                                                     float* __restrict__ vh) {
  const float eps = 1e-12f;
  VSN_NODE_LOOP(i, N, 1) {
    (void)sub;
    float wv[V];
    ldrow<V>(w, lane, wv);
    float v[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      ldrow<V>(vec + ((size_t)i * S + s) * H, lane, v[s]);
      strow<V>(vin + ((size_t)i * S + s) * H, lane, v[s]);
    }
#pragma unroll
    for (int b = 0; b < (S == 8 ? 2 : 1); ++b) {
      const int s0 = b == 0 ? 0 : 3, s1 = b == 0 ? 3 : S;
      float dist[V];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        float d2 = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (s >= s0 && s < s1) d2 += v[s][c] * v[s][c];
        dist[c] = fmaxf(sqrtf(d2), eps);
      }
      if (norm_type == 1) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) q += dist[c] * dist[c];
        const float r = sqrtf(wave_sum(q) / (float)H);
        const float rinv = 1.0f / r;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (s >= s0 && s < s1) {
#pragma unroll
            for (int c = 0; c < V; ++c) v[s][c] = v[s][c] * rinv * wv[c];
          }
      } else {
        float lm = dist[0], lM = dist[0];
#pragma unroll
        for (int c = 1; c < V; ++c) {
          lm = fminf(lm, dist[c]);
          lM = fmaxf(lM, dist[c]);
        }
        const float m = wave_min(lm), M = wave_max(lM);
        float D = M - m;
        if (D == 0.f) D = 1.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const float q = fmaxf((dist[c] - m) / D, 0.f) / dist[c] * wv[c];
#pragma unroll
          for (int s = 0; s < S; ++s)
            if (s >= s0 && s < s1) v[s][c] *= q;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) strow<V>(vh + ((size_t)i * S + s) * H, lane, v[s]);
  }
}

// g_vec (+)= J^T g_vh
template <int V, int S>
__global__ __launch_bounds__(256) void k_vecnorm_bwd(int N, int H, int norm_type, const float* __restrict__ vin,
                                                     const float* __restrict__ w, const float* __restrict__ g_vh,
                                                     int accumulate, float* __restrict__ g_vec) {
  const float eps = 1e-12f;
  VSN_NODE_LOOP(i, N, 1) {
    (void)sub;
    float wv[V];
    ldrow<V>(w, lane, wv);
    float v[S][V], g[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      ldrow<V>(vin + ((size_t)i * S + s) * H, lane, v[s]);
      ldrow<V>(g_vh + ((size_t)i * S + s) * H, lane, g[s]);
#pragma unroll
      for (int c = 0; c < V; ++c) g[s][c] *= wv[c];  // gw = g * w
    }
#pragma unroll
    for (int b = 0; b < (S == 8 ? 2 : 1); ++b) {
      const int s0 = b == 0 ? 0 : 3, s1 = b == 0 ? 3 : S;
      float dist[V], G[V];
      bool free_[V];  // clamp inactive
#pragma unroll
      for (int c = 0; c < V; ++c) {
        float d2 = 0.f, gg = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (s >= s0 && s < s1) {
            d2 += v[s][c] * v[s][c];
            gg += g[s][c] * v[s][c];
          }
        const float d = sqrtf(d2);
        free_[c] = d > eps;
        dist[c] = fmaxf(d, eps);
        G[c] = gg;
      }
      if (norm_type == 1) {
        float q = 0.f, t = 0.f;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          q += dist[c] * dist[c];
          t += G[c];
        }
        const float r = sqrtf(wave_sum(q) / (float)H);
        const float T = wave_sum(t);
        const float rinv = 1.0f / r;
        const float k = T * rinv * rinv * rinv / (float)H;
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (s >= s0 && s < s1) {
#pragma unroll
            for (int c = 0; c < V; ++c) g[s][c] = g[s][c] * rinv - (free_[c] ? k * v[s][c] : 0.f);
          }
      } else {
        float lm = dist[0], lM = dist[0];
#pragma unroll
        for (int c = 1; c < V; ++c) {
          lm = fminf(lm, dist[c]);
          lM = fmaxf(lM, dist[c]);
        }
        const float m = wave_min(lm), M = wave_max(lM);
        const bool flat = (M - m) == 0.f;
        const float D = flat ? 1.f : (M - m);
        // first channel attaining the min / max (torch.min/max route the gradient to one index)
        int im = 1 << 30, iM = 1 << 30;
#pragma unroll
        for (int c = 0; c < V; ++c) {
          if (dist[c] == m && lane * V + c < im) im = lane * V + c;
          if (dist[c] == M && lane * V + c < iM) iM = lane * V + c;
        }
        im = wave_min_i(im);
        iM = wave_min_i(iM);
        float gm_part = 0.f, gM_part = 0.f;
        float q[V], gdist[V];
#pragma unroll
        for (int c = 0; c < V; ++c) {
          const float dt = (dist[c] - m) / D;
          const bool act = dt > 0.f;  // relu'(0) = 0
          q[c] = act ? dt / dist[c] : 0.f;
          gdist[c] = 0.f;
          if (act) {
            // q = (dist - m) / (D dist)
            gdist[c] = G[c] * (m / D) / (dist[c] * dist[c]);
            if (!flat) {
              gm_part += G[c] * (-1.0f / D + (dist[c] - m) / (D * D)) / dist[c];
              gM_part += G[c] * (-(dist[c] - m) / (D * D)) / dist[c];
            } else {
              gm_part += G[c] * (-1.0f / D) / dist[c];
            }
          }
        }
        const float gm = wave_sum(gm_part), gM = wave_sum(gM_part);
#pragma unroll
        for (int c = 0; c < V; ++c) {
          if (lane * V + c == im) gdist[c] += gm;
          if (lane * V + c == iM) gdist[c] += gM;
          const float kd = free_[c] ? gdist[c] / dist[c] : 0.f;
#pragma unroll
          for (int s = 0; s < S; ++s)
            if (s >= s0 && s < s1) g[s][c] = g[s][c] * q[c] + kd * v[s][c];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float o[V];
      if (accumulate)
        ldrow<V>(g_vec + ((size_t)i * S + s) * H, lane, o);
      else {
#pragma unroll
        for (int c = 0; c < V; ++c) o[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < V; ++c) o[c] += g[s][c];
      strow<V>(g_vec + ((size_t)i * S + s) * H, lane, o);
    }
  }
}

#define VSN_DISPATCH2V(V_, S_, FN, ...)               \
  do {                                                \
    if ((S_) == 8) FN<V_, 8> __VA_ARGS__;             \
    else if ((S_) == 3) FN<V_, 3> __VA_ARGS__;        \
    else return -22;                                  \
  } while (0)
#define VSN_DISPATCH2(H_, S_, FN, ...)                              \
  do {                                                              \
    switch ((H_) / 64) {                                            \
      case 1: VSN_DISPATCH2V(1, S_, FN, __VA_ARGS__); break;        \
      case 2: VSN_DISPATCH2V(2, S_, FN, __VA_ARGS__); break;        \
      case 3: VSN_DISPATCH2V(3, S_, FN, __VA_ARGS__); break;        \
      case 4: VSN_DISPATCH2V(4, S_, FN, __VA_ARGS__); break;        \
      case 5: VSN_DISPATCH2V(5, S_, FN, __VA_ARGS__); break;        \
      case 6: VSN_DISPATCH2V(6, S_, FN, __VA_ARGS__); break;        \
      case 7: VSN_DISPATCH2V(7, S_, FN, __VA_ARGS__); break;        \
      case 8: VSN_DISPATCH2V(8, S_, FN, __VA_ARGS__); break;        \
      default: return -22;                                          \
    }                                                               \
  } while (0)

int launch_vecnorm_fwd(hipStream_t st, int N, int H, int S, int norm_type, const float* vec, const float* w,
                       float* vin, float* vh) {
  if (N <= 0) return 0;
  int grid = (N + 3) / 4;
  if (grid > 16384) grid = 16384;
  VSN_DISPATCH2(H, S, k_vecnorm_fwd, <<<grid, 256, 0, st>>>(N, H, norm_type, vec, w, vin, vh));
  return 0;
}
int launch_vecnorm_bwd(hipStream_t st, int N, int H, int S, int norm_type, const float* vin, const float* w,
                       const float* g_vh, int accumulate, float* g_vec) {
  if (N <= 0) return 0;
  int grid = (N + 3) / 4;
  if (grid > 16384) grid = 16384;
  VSN_DISPATCH2(H, S, k_vecnorm_bwd, <<<grid, 256, 0, st>>>(N, H, norm_type, vin, w, g_vh, accumulate, g_vec));
  return 0;
}

}  // namespace vsn
