// Gather kernels fused INTO the MFMA products they feed (fragment batches, hidden = 256).
//
// The reverse pass of a ViS-MP layer (visnet_block.py:276-288 message, :290-295 edge update; adjoints in
// layer_bwd.hip) alternates HBM-bound per-edge kernels with MFMA-bound dense products whose A operand those kernels
// have just written: g_t[E,2H] (vector-message adjoint) -> g_m = g_t.Ws, and g_pe[E,3H] (attention / edge-update
// adjoints) -> g_f += g_pe.We3.  On a fragment batch E is ~17 N, so each such operand is gigabytes per layer that
// cross HBM twice, and while the gather kernel runs the matrix pipes idle (and vice versa).
//
// Here the per-edge kernel becomes the PROLOGUE of the panel GEMM (pgemm.h): a workgroup owns 64 consecutive edges,
// computes their operand rows straight into the swizzled LDS panel, and multiplies them by the packed weights - the
// operand never exists in HBM, and with two workgroups per CU the gathers of one run under the MFMAs of the other.
// K is walked in slices of 256 panel columns; the slices are cut by CHANNEL HALF (slice h = the columns of every
// operand part that belong to channels [128 h, 128 h + 128)), so that one pass over the edges produces exactly one
// slice: a half-wave (32 lanes x 4 channels) per edge, 16-byte loads, per-head sums inside 32 / num_heads x ... lanes.
// The packed weight matrices carry the matching K permutation (engine.hip: pack_panel).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "pgemm.h"

namespace vsn {

// LAB BUILDS ONLY (-DVSN_LAB_ABL=1 through tools/build_variant.py, then env VSN_LAB_FUSED_ABL; wrong numbers on purpose):
// bit 0 = the fused products skip their gather prologue, bit 1 = they skip the MFMA slices - what each phase costs
// alone (tools/lab/fused_phases.sh).  The product build has no such switch: the constant below folds it away.
#ifndef VSN_LAB_ABL
#define VSN_LAB_ABL 0
#endif
#if VSN_LAB_ABL
__device__ int d_fused_abl = 0;
#define VSN_FUSED_ABL() d_fused_abl
#else
#define VSN_FUSED_ABL() 0
#endif

// like wave_multi_sum<8> (common.h) but over each 32-lane half of the wave: every lane ends with the total, over its
// half, of component (lane & 7)
__device__ __forceinline__ float half_multi_sum8(const float (&p)[8], int lane) {
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
  float v = keep + __shfl_xor(send, 1, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the (target, source) node ids of the 16 panel rows a wave gathers, fetched ONCE per panel with one coalesced load each
// (lane & 15 = row within the wave's 16) instead of per row inside the gather loop, where every row's loads waited for
// them: one dependent round trip per row instead of two.  Rows past the live edge count read the last live edge, as the
// loop did.
struct PanelIds {
  int tgt, src;
};
__device__ __forceinline__ PanelIds panel_ids(const Dims& D, const int e0, const int Meff, const int tw, const int lane) {
  int e = e0 + tw * 16 + (lane & 15);
  e = e < Meff ? e : Meff - 1;
  return PanelIds{D.tgt[e], D.src[e]};
}
// row 2 t + hw of those 16 (t wave-uniform, hw = the lane's half-wave)
__device__ __forceinline__ int panel_id_at(const int ids, const int t, const int hw) {
  const int a = __builtin_amdgcn_readlane(ids, 2 * t), b = __builtin_amdgcn_readlane(ids, 2 * t + 1);
  return hw ? b : a;
}

// ---------------------------------------------------------------------------------------------------------------
// g_m[E,H] = g_t[E,2H] . Ws      with g_t produced on the fly (adjoint of the vector messages, target side;
// same arithmetic as k_bwd_vecmsg_T):
//   g_s1 = sum_s g_vec_i[s] vh_j[s] ; g_s2 = sum_s g_vec_i[s] d_e[s] ; g_t = [g_s1 act'(t1) | g_s2 act'(t2)]
//   dE/dd_e[s] += sum_c g_vec_i[s][c] act(t2[c])                                  -> g_geo[e][0..7]
// Panel slice h, columns: [0,128) = g_t1 channels 128 h .., [128,256) = g_t2 channels 128 h ..
// ---------------------------------------------------------------------------------------------------------------
// the prologue of one panel slice: the 64 edges [e0, e0 + 64) x channel half h, written into the swizzled LDS panel
// by four waves (`tw` = wave within the team of four, a half-wave per edge)
template <bool GEN, int U = 2>
__device__ __forceinline__ void gm_gather(const Dims& D, const float* __restrict__ g_vec, const float* __restrict__ vh,
                                          const float* __restrict__ tpre, float* __restrict__ g_geo,
                                          float* __restrict__ smem, const int e0, const int Meff, const int h,
                                          const int tw, const int lane, const int abl, const PanelIds ids) {
  const int l5 = lane & 31, hw = lane >> 5;
  const int act = GEN ? D.act : VSN_ACT_SILU;
  const int c0 = 128 * h + 4 * l5;
  // the rows of a half-wave (every second edge of 16 consecutive ones) mostly share their target node: its eight
  // g_vec rows stay in registers and are re-fetched only when the target changes (they were 45 % of the row loads)
  f32x4 gv[8];
  int i_prev = -1;
#pragma unroll U
  for (int t = (abl & 1) ? 8 : 0; t < 8; ++t) {
    const int r = tw * 16 + 2 * t + hw;  // panel row of this half-wave
    const bool valid = e0 + r < Meff;
    const int e = valid ? e0 + r : Meff - 1;
    const int i = panel_id_at(ids.tgt, t, hw), j = panel_id_at(ids.src, t, hw);
    const float* __restrict__ tp = tpre + (size_t)e * 512 + c0;
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(tp);
    const f32x4 t2 = *reinterpret_cast<const f32x4*>(tp + 256);
    const f32x4 dA = *reinterpret_cast<const f32x4*>(D.d + (size_t)e * 8);
    const f32x4 dB = *reinterpret_cast<const f32x4*>(D.d + (size_t)e * 8 + 4);
    const float geo_old = l5 < 8 ? g_geo[(size_t)e * VSN_GEO_W + l5] : 0.f;  // fetched with the other operands
    const float ds[8] = {dA.x, dA.y, dA.z, dA.w, dB.x, dB.y, dB.z, dB.w};
    const float* __restrict__ gvp = g_vec + (size_t)i * 8 * 256 + c0;
    const float* __restrict__ vjp = vh + (size_t)j * 8 * 256 + c0;
    f32x4 vj[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) vj[s] = *reinterpret_cast<const f32x4*>(vjp + s * 256);
    if (i != i_prev) {  // (half-wave uniform)
#pragma unroll
      for (int s = 0; s < 8; ++s) gv[s] = *reinterpret_cast<const f32x4*>(gvp + s * 256);
      i_prev = i;
    }
    float d1[4], s2[4], d2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      d1[c] = dact_f(act, t1[c]);
      act_both(act, t2[c], s2[c], d2[c]);
    }
    f32x4 gs1 = {0.f, 0.f, 0.f, 0.f}, gs2 = {0.f, 0.f, 0.f, 0.f};
    float pp[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float pq = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        gs1[c] += gv[s][c] * vj[s][c];
        gs2[c] += gv[s][c] * ds[s];
        pq += gv[s][c] * s2[c];
      }
      pp[s] = pq;
    }
    const float mine = half_multi_sum8(pp, lane);  // lane l5 < 8: sum over the half-wave of component l5
    if (valid && l5 < 8) g_geo[(size_t)e * VSN_GEO_W + l5] = geo_old + mine;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gs1[c] *= d1[c];
      gs2[c] *= d2[c];
    }
    *reinterpret_cast<f32x4*>(smem + panel_at(r, l5)) = gs1;
    *reinterpret_cast<f32x4*>(smem + panel_at(r, 32 + l5)) = gs2;
  }
}

template <bool GEN>
__global__ __launch_bounds__(256, 2) void k_bwd_gm_fused(Dims D, const float* __restrict__ g_vec,
                                                         const float* __restrict__ vh,
                                                         const float* __restrict__ tpre,
                                                         const float* __restrict__ Bp, float* __restrict__ g_m,
                                                         float* __restrict__ g_geo) {
  typedef PgemmBwd<4> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int Meff = *D.ecount;
  Meff = Meff < D.Emax ? Meff : D.Emax;
  const int live = (Meff + 63) >> 6;
  if ((int)blockIdx.x >= live) return;
  const int p = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int e0 = p * 64;
  typename G::Acc acc;
  G::zero(acc);
  typename G::Ring ring;
  G::prefetch(ring, Bp, 512, 0, wave, lane);
  const int abl = VSN_FUSED_ABL();
  const PanelIds ids = panel_ids(D, e0, Meff, wave, lane);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();  // every wave is done reading slice 0
    gm_gather<GEN>(D, g_vec, vh, tpre, g_geo, smem, e0, Meff, h, wave, lane, abl, ids);
    __syncthreads();
    G::pin(ring);
    if (!(abl & 2)) G::slice(acc, ring, smem, Bp, 512, h, wave, lane);
  }
  G::template store<0>(acc, g_m, 256, e0, Meff, wave, lane);
}

// ---- the same product as a PERSISTENT, TEAM-PHASED kernel (round 5; LAB BUILDS ONLY: -DVSN_LAB_ABL=1) -----------------
#if VSN_LAB_ABL
// Measured on the 4096-fragment batch (tools/lab/fused_phases.sh, products alone): k_bwd_gm_fused 1.97 ms = 1.34 ms
// of MFMA slices + 0.77 ms of gather prologue; k_bwd_gf_fused 2.79 = 2.02 + 1.12.  The two resident workgroups of a
// CU were meant to hide each other's prologue, but free-running they fall into step (both in the MFMA phase share
// the pipe at half speed each, then both gather with the pipe idle): 6 % / 11 % of the prologue was hidden.
// Here ONE workgroup of eight waves owns a CU: two teams of four waves, each with its own 64-KB panel, walk their
// panels through the phases  gather(h=0) | mfma(0) | gather(1) | mfma(1) [| dma(2) | mfma(2)]  ONE TICK APART, with a
// workgroup barrier per tick - by construction one team is in an MFMA phase while the other gathers, every tick.
// The workgroup is persistent: it takes a contiguous range of panels (team 0 the even, team 1 the odd ones of the
// range), so the alternation runs on across panels.  Same arithmetic in the same order as the kernels above: bitwise
// the same results.
// RESULT (MI355X, same lab script with VSN_PANEL_TP=1): gm 2.32 ms = 1.33 ms MFMA-only + 1.11 ms gather-only, gf 3.32 =
// 2.13 + 1.71 - the phases STILL add up.  One MFMA wave per SIMD does saturate the matrix pipe (MFMA-only time
// unchanged), but a SIMD whose matrix pipe is saturated issues next to nothing from its other wave: the gather of the
// co-resident team only advances in the gaps.  Strict alternation therefore buys nothing over the free-running pair,
// and the single gather wave per SIMD (unroll 1 to fit 256 registers) is slower than two: 13.4 k -> 12.7 k
// fragments/s.  Kept as a lab variant (off by default); the fused products stay at MFMA + gather.
template <bool GEN>
__global__ __launch_bounds__(512, 1) void k_bwd_gm_fused_tp(Dims D, const float* __restrict__ g_vec,
                                                            const float* __restrict__ vh,
                                                            const float* __restrict__ tpre,
                                                            const float* __restrict__ Bp, float* __restrict__ g_m,
                                                            float* __restrict__ g_geo) {
  typedef PgemmBwd<4> G;
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  int Meff = *D.ecount;
  Meff = Meff < D.Emax ? Meff : D.Emax;
  const int live = (Meff + 63) >> 6;
  const int nwg = (int)gridDim.x;
  const int w = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, nwg) : (int)blockIdx.x;
  const int p_lo = (int)(((long long)w * live) / nwg), p_hi = (int)(((long long)(w + 1) * live) / nwg);
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int team = wave >> 2, tw = wave & 3;
  float* __restrict__ smem = smem_all + team * (64 * 256);
  const int np = p_hi - p_lo;                       // panels of this workgroup
  const int mine = (np + 1 - team) >> 1;            // ... of this team (team 0: even offsets, team 1: odd ones)
  const int mine0 = (np + 1) >> 1;
  const int abl = VSN_FUSED_ABL();
  typename G::Acc acc;
  typename G::Ring ring;
  // Every wave of the workgroup arrives at the same NUMBER of barriers (mine0 * 4 + 1), the two teams from different
  // places in the code: team 1 starts one tick late, team 0 waits one tick at the end, and a team that has a panel
  // less than the other idles through that panel's four ticks.
  if (team) __syncthreads();
#pragma unroll 1
  for (int k = 0; k < mine; ++k) {
    const int e0 = (p_lo + 2 * k + team) * 64;
    G::zero(acc);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      gm_gather<GEN, 1>(D, g_vec, vh, tpre, g_geo, smem, e0, Meff, h, tw, lane, abl, panel_ids(D, e0, Meff, tw, lane));
      // the B ring is (re)filled per slice, AFTER the gather: 32 registers the gather then has for its rows (the
      // read-ahead at the tail of the previous slice is dropped; the barrier wait covers the latency of this one)
      G::prefetch(ring, Bp, 512, h, tw, lane);
      __syncthreads();
      G::pin(ring);
      if (!(abl & 2)) G::slice(acc, ring, smem, Bp, 512, h, tw, lane);
      if (h) G::template store<0>(acc, g_m, 256, e0, Meff, tw, lane);  // (before the barrier: the other team gathers meanwhile)
      __syncthreads();
    }
  }
  for (int k = mine; k < mine0; ++k) {
    __syncthreads();
    __syncthreads();
    __syncthreads();
    __syncthreads();
  }
  if (!team) __syncthreads();
}
#endif  // VSN_LAB_ABL (team-phased lab variant)

// ---------------------------------------------------------------------------------------------------------------
// g_f[E,H] (+)= g_pe[E,3H] . We3    with the attention part of g_pe produced on the fly (adjoint of attention /
// scalar message, target side; same arithmetic as k_bwd_attn_T except that dE/dq is summed by k_bwd_attn_QS):
//   gm = g_m_e + g_A_i (written back to g_m for the source-side sums) ; sat, a recomputed
//   g_a[h] = sum_{c in h} gm v_j dv ; g_sat = g_a act'(sat) C ; dE/dC += sum_h g_a act(sat)
//   g_pk = g_sat q_i k_j act'(pk) ; g_pv = gm v_j a act'(pv) ; sat_tmp[e] = [g_sat | a]
// Panel slice h (h = 0, 1), columns: [0,128) = g_pk channels 128 h .., [128,256) = g_pv channels 128 h ..;
// slice 2 (layers with an edge update) = g_pf, copied by LDS-DMA from g_pe[:, 2H:3H] (written by the edge-update
// adjoint).  K = 512 or 768.
// ---------------------------------------------------------------------------------------------------------------
template <bool GEN, int U = 2>
__device__ __forceinline__ void gf_gather(const Dims& D, const float* __restrict__ qkv, const float* __restrict__ pe,
                                          const float* __restrict__ g_A, float* __restrict__ g_m,
                                          float* __restrict__ sat_tmp, float* __restrict__ g_geo,
                                          float* __restrict__ smem, const int e0, const int Meff, const int h,
                                          const int tw, const int lane, const int abl, const PanelIds ids) {
  const int l5 = lane & 31, hw = lane >> 5;
  const int act = GEN ? D.act : VSN_ACT_SILU, aact = GEN ? D.attn_act : VSN_ACT_SILU;
  const int nh = D.nh;
  const int lph = 64 / nh;  // lanes per head (4 channels per lane): 256 / nh / 4
  const int c0 = 128 * h + 4 * l5;
  const int head = (c0 * nh) >> 8;  // head of this lane's channels
  f32x4 q = {0.f, 0.f, 0.f, 0.f}, gA = {0.f, 0.f, 0.f, 0.f};  // rows of the target node: re-fetched when it changes
  int i_prev = -1;
#pragma unroll U
  for (int t = (abl & 1) ? 8 : 0; t < 8; ++t) {
    const int r = tw * 16 + 2 * t + hw;
    const bool valid = e0 + r < Meff;
    const int e = valid ? e0 + r : Meff - 1;
    const int i = panel_id_at(ids.tgt, t, hw), j = panel_id_at(ids.src, t, hw);
    const float C = D.geo[(size_t)e * 8 + 1];
    const float gC_old = l5 == 0 ? g_geo[(size_t)e * VSN_GEO_W + 8] : 0.f;
    if (i != i_prev) {
      q = *reinterpret_cast<const f32x4*>(qkv + (size_t)i * 768 + c0);
      gA = *reinterpret_cast<const f32x4*>(g_A + (size_t)i * 256 + c0);
      i_prev = i;
    }
    const f32x4 k = *reinterpret_cast<const f32x4*>(qkv + (size_t)j * 768 + 256 + c0);
    const f32x4 v = *reinterpret_cast<const f32x4*>(qkv + (size_t)j * 768 + 512 + c0);
    const f32x4 pk = *reinterpret_cast<const f32x4*>(pe + (size_t)e * 768 + c0);
    const f32x4 pv = *reinterpret_cast<const f32x4*>(pe + (size_t)e * 768 + 256 + c0);
    f32x4 gm = *reinterpret_cast<const f32x4*>(g_m + (size_t)e * 256 + c0);
    float dk[4], ddk[4], dv[4], ddv[4];
    float part = 0.f, gpart = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gm[c] += gA[c];
      act_both(act, pk[c], dk[c], ddk[c]);
      act_both(act, pv[c], dv[c], ddv[c]);
      part += q[c] * k[c] * dk[c];
      gpart += gm[c] * v[c] * dv[c];
    }
    if (valid) *reinterpret_cast<f32x4*>(g_m + (size_t)e * 256 + c0) = gm;
    const float sat = group_sum(part, lph);
    const float ga = group_sum(gpart, lph);
    float ssat, dssat;
    act_both(aact, sat, ssat, dssat);
    const float a = ssat * C;
    const float gsat = ga * dssat * C;
    const bool head_lead = (l5 & (lph - 1)) == 0;
    const float gC = half_sum(head_lead ? ga * ssat : 0.f);
    if (valid && l5 == 0) g_geo[(size_t)e * VSN_GEO_W + 8] = gC_old + gC;
    if (valid && head_lead) {
      sat_tmp[(size_t)e * 2 * nh + head] = gsat;
      sat_tmp[(size_t)e * 2 * nh + nh + head] = a;
    }
    f32x4 gpk, gpv;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gpk[c] = gsat * q[c] * k[c] * ddk[c];
      gpv[c] = gm[c] * v[c] * a * ddv[c];
    }
    *reinterpret_cast<f32x4*>(smem + panel_at(r, l5)) = gpk;
    *reinterpret_cast<f32x4*>(smem + panel_at(r, 32 + l5)) = gpv;
  }
}

template <bool GEN, int EPI>
__global__ __launch_bounds__(256, 2) void k_bwd_gf_fused(Dims D, const float* __restrict__ qkv,
                                                         const float* __restrict__ pe,
                                                         const float* __restrict__ g_A, float* __restrict__ g_m,
                                                         const float* __restrict__ g_pe,
                                                         float* __restrict__ sat_tmp, float* __restrict__ g_geo,
                                                         const float* __restrict__ Bp, float* __restrict__ g_f,
                                                         int K) {
  typedef PgemmBwd<4> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int Meff = *D.ecount;
  Meff = Meff < D.Emax ? Meff : D.Emax;
  const int live = (Meff + 63) >> 6;
  if ((int)blockIdx.x >= live) return;
  const int p = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int e0 = p * 64;
  typename G::Acc acc;
  G::zero(acc);
  typename G::Ring ring;
  G::prefetch(ring, Bp, K, 0, wave, lane);
  const int abl = VSN_FUSED_ABL();
  const PanelIds ids = panel_ids(D, e0, Meff, wave, lane);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    gf_gather<GEN>(D, qkv, pe, g_A, g_m, sat_tmp, g_geo, smem, e0, Meff, h, wave, lane, abl, ids);
    __syncthreads();
    G::pin(ring);
    if (!(abl & 2)) G::slice(acc, ring, smem, Bp, K, h, wave, lane);
  }
  if (K > 512) {
    __syncthreads();
    panel_load_dma<64, 4>(smem, g_pe, 768, e0, Meff, 512, wave, lane);
    __syncthreads();
    G::pin(ring);
    if (!(abl & 2)) G::slice(acc, ring, smem, Bp, K, 2, wave, lane);
  }
  G::template store<EPI>(acc, g_f, 256, e0, Meff, wave, lane);
}
#if VSN_LAB_ABL

// persistent, team-phased form (see k_bwd_gm_fused_tp): phases per panel  gather(0) | mfma(0) | gather(1) | mfma(1)
// and, with K = 768, | dma(2) | mfma(2) - gather-like and MFMA phases alternate, the teams run one tick apart
template <bool GEN, int EPI>
__global__ __launch_bounds__(512, 1) void k_bwd_gf_fused_tp(Dims D, const float* __restrict__ qkv,
                                                            const float* __restrict__ pe,
                                                            const float* __restrict__ g_A, float* __restrict__ g_m,
                                                            const float* __restrict__ g_pe,
                                                            float* __restrict__ sat_tmp, float* __restrict__ g_geo,
                                                            const float* __restrict__ Bp, float* __restrict__ g_f,
                                                            int K) {
  typedef PgemmBwd<4> G;
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  int Meff = *D.ecount;
  Meff = Meff < D.Emax ? Meff : D.Emax;
  const int live = (Meff + 63) >> 6;
  const int nwg = (int)gridDim.x;
  const int w = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, nwg) : (int)blockIdx.x;
  const int p_lo = (int)(((long long)w * live) / nwg), p_hi = (int)(((long long)(w + 1) * live) / nwg);
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int team = wave >> 2, tw = wave & 3;
  float* __restrict__ smem = smem_all + team * (64 * 256);
  const int C = K > 512 ? 6 : 4;  // phases (= barriers) per panel
  const int np = p_hi - p_lo;
  const int mine = (np + 1 - team) >> 1;
  const int mine0 = (np + 1) >> 1;
  const int abl = VSN_FUSED_ABL();
  typename G::Acc acc;
  typename G::Ring ring;
  if (team) __syncthreads();  // (barrier bookkeeping: see k_bwd_gm_fused_tp)
#pragma unroll 1
  for (int k = 0; k < mine; ++k) {
    const int e0 = (p_lo + 2 * k + team) * 64;
    G::zero(acc);
    const int nsl = K > 512 ? 3 : 2;
#pragma unroll 1
    for (int h = 0; h < nsl; ++h) {
      if (h < 2) gf_gather<GEN, 1>(D, qkv, pe, g_A, g_m, sat_tmp, g_geo, smem, e0, Meff, h, tw, lane, abl, panel_ids(D, e0, Meff, tw, lane));
      else panel_load_dma<64, 4>(smem, g_pe, 768, e0, Meff, 512, tw, lane);  // (drained by the barrier)
      G::prefetch(ring, Bp, K, h, tw, lane);
      __syncthreads();
      G::pin(ring);
      if (!(abl & 2)) G::slice(acc, ring, smem, Bp, K, h, tw, lane);
      if (h == nsl - 1) G::template store<EPI>(acc, g_f, 256, e0, Meff, tw, lane);
      __syncthreads();
    }
  }
  for (int k = mine; k < mine0; ++k)
    for (int c = 0; c < C; ++c) __syncthreads();
  if (!team) __syncthreads();
}
#endif  // VSN_LAB_ABL (team-phased lab variant)

// ---- hosts ------------------------------------------------------------------------------------------------
static int g_fuse_panel = 1;  // env VSN_FUSE_PANEL=0: never take the fused / panel kernels (A/B aid)
static int g_lab_abl = 0;
static const bool g_fused_env = [] {
  if (const char* e = getenv("VSN_FUSE_PANEL")) g_fuse_panel = atoi(e);
  if (const char* e = getenv("VSN_LAB_FUSED_ABL")) g_lab_abl = atoi(e);
  return true;
}();
static void lab_push_abl() {  // (per device: once per process is enough for the lab runs, which use one GPU)
#if VSN_LAB_ABL
  static bool done = false;
  if (done || !g_lab_abl) return;
  hipMemcpyToSymbol(HIP_SYMBOL(d_fused_abl), &g_lab_abl, sizeof(int));
  done = true;
#else
  (void)g_lab_abl;
#endif
}

bool panel_ok(const Dims& D) {  // (head counts that divide 64: the fused prologues sum heads over lane groups)
  return g_fuse_panel && D.H == 256 && D.S == 8 && D.nh >= 2 && D.nh <= 64 && !D.hgen;
}

// the attribute is per DEVICE: remembered per (current device, kernel), so a thread that drives engines on several
// GPUs sets it on each of them.  Returns false when the device refuses that much LDS per workgroup: the caller then
// fails the evaluation instead of launching a kernel that cannot run (stale g_m / g_f would give wrong forces).
template <typename K>
static inline bool panel_lds(K kern, int bytes = 65536) {
  struct Key {
    int dev;
    const void* k;
  };
  static thread_local Key done[64];
  static thread_local int ndone = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  for (int i = 0; i < ndone; ++i)
    if (done[i].dev == dev && done[i].k == (const void*)kern) return true;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (ndone < 64) done[ndone++] = Key{dev, (const void*)kern};
  return true;
}
static inline int launched() { return hipGetLastError() == hipSuccess ? 0 : -1; }

#if VSN_LAB_ABL
// env VSN_PANEL_TP / vsn_set_option "panel_tp" (LAB BUILDS ONLY, process-wide): 1 = the persistent team-phased kernels,
// 0 (default: measured faster, see k_bwd_gm_fused_tp) = one workgroup per panel
static int g_panel_tp = [] {
  const char* e = getenv("VSN_PANEL_TP");
  return e ? atoi(e) : 0;
}();
int set_panel_tp(int v) {
  g_panel_tp = v;
  return 0;
}
static int tp_grid() {  // one persistent workgroup per CU of the current device
  static thread_local int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& n = cus[dev & 63];
  if (!n) {
    hipDeviceProp_t pr;
    n = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    n = (n + 7) & ~7;
  }
  return n;
}
#else
int set_panel_tp(int v) { return v ? -1 : 0; }  // the team-phased variant does not exist in product builds
#endif

int launch_bwd_gm_fused(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre,
                        const float* WsTp, float* g_m, float* g_geo) {
  if (D.Emax <= 0) return 0;
  lab_push_abl();
  const bool gen = D.act != VSN_ACT_SILU;
  const int grid = (D.Emax + 63) / 64;
#if VSN_LAB_ABL
  if (g_panel_tp) {
    const int g = std::min(tp_grid(), (grid + 1) / 2);
    if (gen) {
      if (!panel_lds(k_bwd_gm_fused_tp<true>, 131072)) return -1;
      k_bwd_gm_fused_tp<true><<<g, 512, 131072, st>>>(D, g_vec, vh, tpre, WsTp, g_m, g_geo);
    } else {
      if (!panel_lds(k_bwd_gm_fused_tp<false>, 131072)) return -1;
      k_bwd_gm_fused_tp<false><<<g, 512, 131072, st>>>(D, g_vec, vh, tpre, WsTp, g_m, g_geo);
    }
    return launched();
  }
#endif
  if (gen) {
    if (!panel_lds(k_bwd_gm_fused<true>)) return -1;
    k_bwd_gm_fused<true><<<grid, 256, 65536, st>>>(D, g_vec, vh, tpre, WsTp, g_m, g_geo);
  } else {
    if (!panel_lds(k_bwd_gm_fused<false>)) return -1;
    k_bwd_gm_fused<false><<<grid, 256, 65536, st>>>(D, g_vec, vh, tpre, WsTp, g_m, g_geo);
  }
  return launched();
}

int launch_bwd_gf_fused(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A,
                        float* g_m, const float* g_pe, float* sat_tmp, float* g_geo, const float* We3Tp, float* g_f,
                        int K, int accumulate) {
  if (D.Emax <= 0) return 0;
  lab_push_abl();
  const bool gen = D.act != VSN_ACT_SILU || D.attn_act != VSN_ACT_SILU;
  const int grid = (D.Emax + 63) / 64;
#if VSN_LAB_ABL
#define VSN_GF_TP(G_, E_)                                                                                           \
  if (g_panel_tp) {                                                                                                 \
    if (!panel_lds(k_bwd_gf_fused_tp<G_, E_>, 131072)) return -1;                                                   \
    k_bwd_gf_fused_tp<G_, E_><<<std::min(tp_grid(), (grid + 1) / 2), 512, 131072, st>>>(                            \
        D, qkv, pe, g_A, g_m, g_pe, sat_tmp, g_geo, We3Tp, g_f, K);                                                 \
    return launched();                                                                                              \
  }
#else
#define VSN_GF_TP(G_, E_)
#endif
#define VSN_GF(G_, E_)                                                                                              \
  do {                                                                                                              \
    VSN_GF_TP(G_, E_)                                                                                               \
    if (!panel_lds(k_bwd_gf_fused<G_, E_>)) return -1;                                                              \
    k_bwd_gf_fused<G_, E_><<<grid, 256, 65536, st>>>(D, qkv, pe, g_A, g_m, g_pe, sat_tmp, g_geo, We3Tp, g_f, K);    \
  } while (0)
  if (gen) {
    if (accumulate) VSN_GF(true, 2);
    else VSN_GF(true, 0);
  } else {
    if (accumulate) VSN_GF(false, 2);
    else VSN_GF(false, 0);
  }
#undef VSN_GF
#undef VSN_GF_TP
  return launched();
}

}  // namespace vsn
