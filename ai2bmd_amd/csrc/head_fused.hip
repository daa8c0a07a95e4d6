// Fused read-out head for single-protein sizes: everything between the first projection of the head
// (pv0 = vec_out_norm(vec) . [W1 | W2]^T, a batch-sized GEMM) and the last product of its reverse pass
// (g_vo = g_pv0 . [W1 | W2]) is node-local - two GatedEquivariantBlocks' MLPs, norms over the spherical
// components, gates, and their adjoints (output_modules.py:52-62,136-140; visnet.py:139-149).  On one protein the
// unfused chain is 17 launches of 4-12 us each for a few hundred rows; here ONE workgroup carries a tile of T = 8
// nodes (T*S vector rows) through all 15 stages with every intermediate in LDS, streaming each weight matrix from L2
// exactly once per workgroup.  The small products run on v_mfma_f32_16x16x4_f32 (exact fp32): a tile has 8 scalar
// rows, so the 16-row MFMA wastes half of its rows where the 32-row one would waste three quarters.
//
// LDS: ~146 KB at H = 256 (one workgroup per CU); wider networks (H > 256) keep the unfused path (head.hip).
#include "common.h"
#include "kernels.h"

namespace vsn {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef HF_T
#define HF_T 8          // nodes per workgroup (A/B builds: -DHF_T=4)
#endif
#ifndef HF_PD
#define HF_PD 8         // weight fragments (1 KiB each) in flight per wave (A/B builds: -DHF_PD=16)
#endif
static_assert(HF_T <= 8, "lds_gemm8 carries two 4-row groups");
#define HF_WAVES 16     // 1024 threads: 4 waves per SIMD of the one workgroup a CU holds (LDS-bound occupancy)

// Out[r][n] = sum_k A[r][k] * W[n][k] (+ bias[n]) for r < M, n < Nc.  A, Out in LDS (row strides lda, ldo), W global,
// PACKED in fragment order at load time (engine.hip::pack_head): block (16 columns cb, 16 k kg) = 1 KiB = one coalesced
// 16-byte load per lane (the nn.Linear layout cost a load instruction 16 half-used cache lines: the workgroup's
// weight stream, not its MFMAs, is what a tile waits for).  Wave w owns the 16-column blocks w, w + 16, ...; for each
// it keeps one accumulator per 16-row block (RB <= 4) so a weight fragment is fetched once and used for every row block.
// MFMA 16x16x4: lane (i = lane & 15, q = lane >> 4) supplies A[i][k0 + 4 q + t] and W[n0 + i][k0 + 4 q + t] for
// t = 0..3 (one 16-byte access each), i.e. 16 k-values per group of four MFMAs; result lane holds
// C[4 q + r][n0 + i], r = 0..3.
template <int RB, int K, int NC>
__device__ __forceinline__ void lds_gemm(const float* __restrict__ As, int lda, int M, const float* __restrict__ W,
                                         const float* __restrict__ bias, float* __restrict__ Out, int ldo, int wave,
                                         int lane) {
  constexpr int KG = K / 16, CB = NC / 16;
  // weight fragments in flight per wave (an L2 round trip spans ~8 k-groups): the largest divisor of KG up to 8, so
  // that the ring slot of fragment kg is kg % PD in EVERY column block (the ring runs on across block boundaries)
  constexpr int PD = KG % HF_PD == 0 ? HF_PD : KG % 8 == 0 ? 8 : KG % 7 == 0 ? 7 : KG % 6 == 0 ? 6 : KG % 5 == 0 ? 5
                   : KG % 4 == 0 ? 4 : KG % 3 == 0 ? 3 : KG % 2 == 0 ? 2 : 1;
  static_assert(KG % PD == 0, "ring depth must divide the k-groups of a column block");
  constexpr int NCB = (CB + HF_WAVES - 1) / HF_WAVES;
  const int i = lane & 15, q = lane >> 4;
  if (wave >= CB) return;
  const float* ap[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int r = rb * 16 + i;
    r = r < M ? r : M - 1;  // rows past the tile repeat its last row; their results are never stored
    ap[rb] = As + r * lda + q * 4;
  }
  const float* wp = W + (size_t)wave * KG * 256 + lane * 4;  // block (cb = wave, kg = 0), this lane's 16 bytes
  f32x4v wq[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p) wq[p] = *reinterpret_cast<const f32x4v*>(wp + p * 256);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int j = 0; j < NCB; ++j) {
    const int cb = wave + j * HF_WAVES;
    if (cb >= CB) break;
    // the ring keeps running into the next column block of this wave (clamped to the current one at the end)
    const float* wn = cb + HF_WAVES < CB ? wp + (size_t)HF_WAVES * KG * 256 : wp;
    f32x4v acc[RB], an[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      acc[rb] = f32x4v{0.f, 0.f, 0.f, 0.f};
      an[rb] = *reinterpret_cast<const f32x4v*>(ap[rb]);
    }
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {  // fully unrolled: every register index below is static
      const f32x4v w = wq[kg % PD];
      wq[kg % PD] = kg + PD < KG ? *reinterpret_cast<const f32x4v*>(wp + (kg + PD) * 256)
                                 : *reinterpret_cast<const f32x4v*>(wn + (kg + PD - KG) * 256);
      f32x4v ac[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {  // the A fragments one k-group ahead as well (LDS latency off the MFMA chain)
        ac[rb] = an[rb];
        an[rb] = *reinterpret_cast<const f32x4v*>(ap[rb] + (kg + 1 < KG ? kg + 1 : 0) * 16);
      }
      // pin the refills HERE, ahead of their use: hipcc otherwise sinks them next to the MFMAs that consume them
      // (seen in the ISA: two loads in flight instead of eight), which makes every group wait for L2
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const f32x4v a = ac[rb];
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[rb], 0, 0, 0);
      }
    }
    const int col = cb * 16 + i;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb * 16 + 4 * q + r;
        if (row < M) Out[row * ldo + col] = acc[rb][r] + bv;
      }
    wp = wn;
  }
}

// The same product for the SCALAR stages (M <= 8 rows: one row per node of the tile).  A 16-row MFMA would spend half
// of its passes on padding rows, and the matrix pipe of the ONE CU a tile lives on is what these stages wait for
// (15 stages = 18.9 MFLOP padded, 30 us of a CU's fp32 pipe); v_mfma_f32_4x4x1_16B_f32 has sixteen independent 4 x 4
// blocks instead: block b = 4 kk + cg takes rows (4 g .. 4 g + 3) x columns (4 cg .. 4 cg + 3) of the wave's 16-column
// block and the k-subset (4 kk .. 4 kk + 3) of every 16-k group, two row groups g = all 8 rows, nothing padded; the
// four k-subsets are summed across lanes l, l + 16, l + 32, l + 48 at the end of a column block.  The lane's weight
// operand is W[n0 + (lane & 15)][k0 + 4 (lane >> 4) + t] - exactly the packed fragment the 16x16x4 form reads.
template <int K, int NC>
__device__ __forceinline__ void lds_gemm8(const float* __restrict__ As, int lda, int M, const float* __restrict__ W,
                                          const float* __restrict__ bias, float* __restrict__ Out, int ldo, int wave,
                                          int lane) {
  constexpr int KG = K / 16, CB = NC / 16;
  constexpr int PD = KG % HF_PD == 0 ? HF_PD : KG % 8 == 0 ? 8 : KG % 7 == 0 ? 7 : KG % 6 == 0 ? 6 : KG % 5 == 0 ? 5
                   : KG % 4 == 0 ? 4 : KG % 3 == 0 ? 3 : KG % 2 == 0 ? 2 : 1;
  static_assert(KG % PD == 0, "ring depth must divide the k-groups of a column block");
  constexpr int NCB = (CB + HF_WAVES - 1) / HF_WAVES;
  const int c = lane & 15, kk = lane >> 4, i = lane & 3;
  if (wave >= CB) return;
  const float* ap0 = As + (i < M ? i : M - 1) * lda + kk * 4;          // rows past the tile repeat its last row
  const float* ap1 = As + (4 + i < M ? 4 + i : M - 1) * lda + kk * 4;  // (never stored)
  const float* wp = W + (size_t)wave * KG * 256 + lane * 4;
  f32x4v wq[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p) wq[p] = *reinterpret_cast<const f32x4v*>(wp + p * 256);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int j = 0; j < NCB; ++j) {
    const int cb = wave + j * HF_WAVES;
    if (cb >= CB) break;
    const float* wn = cb + HF_WAVES < CB ? wp + (size_t)HF_WAVES * KG * 256 : wp;
    f32x4v acc0 = f32x4v{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    f32x4v an0 = *reinterpret_cast<const f32x4v*>(ap0), an1 = *reinterpret_cast<const f32x4v*>(ap1);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const f32x4v w = wq[kg % PD];
      wq[kg % PD] = kg + PD < KG ? *reinterpret_cast<const f32x4v*>(wp + (kg + PD) * 256)
                                 : *reinterpret_cast<const f32x4v*>(wn + (kg + PD - KG) * 256);
      const f32x4v a0 = an0, a1 = an1;
      an0 = *reinterpret_cast<const f32x4v*>(ap0 + (kg + 1 < KG ? kg + 1 : 0) * 16);
      an1 = *reinterpret_cast<const f32x4v*>(ap1 + (kg + 1 < KG ? kg + 1 : 0) * 16);
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, w.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, w.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, w.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, w.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, w.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, w.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, w.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, w.w, acc1, 0, 0, 0);
    }
    const int col = cb * 16 + c;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = acc0[r], v1 = acc1[r];
      v0 += __shfl_xor(v0, 16);
      v1 += __shfl_xor(v1, 16);
      v0 += __shfl_xor(v0, 32);
      v1 += __shfl_xor(v1, 32);
      if (kk == 0) {
        if (r < M) Out[r * ldo + col] = v0 + bv;
        if (4 + r < M) Out[(4 + r) * ldo + col] = v1 + bv;
      }
    }
    wp = wn;
  }
}

// Lab builds (-DHF_LAB_STAMPS): thread 0 of every workgroup stamps the 100 MHz real-time counter after each stage into a
// buffer set with vsn_lab_set_hf_stamps() (tools/lab/stamps_head.py prints the stage times).
#ifdef HF_LAB_STAMPS
__device__ unsigned long long* g_hf_stamps = nullptr;
#define HF_STAMP(k)                                                                                 \
  do {                                                                                              \
    if (g_hf_stamps && threadIdx.x == 0) g_hf_stamps[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define HF_STAMP(k) do {} while (0)
#endif

struct HeadFusedArgs {
  int N, S, H, act;
  const float* cat0g;   // [N][2H], first H columns = out_norm(x)
  const float* pv0;     // [N*S][H + h2]
  HeadW W;
  const int* zi;
  float* y;             // [N]
  float* g_cat0g;       // [N][2H]: first H columns <- dE/d out_norm(x)
  float* g_pv0;         // [N*S][H + h2]
};

template <int H, int S, bool GEN>
__global__ __launch_bounds__(64 * HF_WAVES) void hk_fused(HeadFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int h2 = H / 2, ldp = H + h2;
  constexpr int l2H = 2 * H + 4, lH = H + 4, lh = h2 + 4;
  constexpr int T = HF_T, R = HF_T * S;
  constexpr int RBV = (R + 15) / 16;
  float* cat0 = lds;                    // [T][2H+4]
  float* a0 = cat0 + T * l2H;           // [T][H+4]
  float* ta = a0 + T * lH;              // [T][H+4]  act(a0), later g_u0
  float* u0 = ta + T * lH;              // [T][H+4]
  float* cat1 = u0 + T * lH;            // [T][H+4]  later g_h0
  float* gcat1 = cat1 + T * lH;         // [T][H+4]
  float* vec1o = gcat1 + T * lH;        // [R][h2+4] later g_vec1o
  float* p1 = vec1o + R * lh;           // [R][h2+4] later g_p1
  float* a1b = p1 + R * lh;             // [T][h2+4] later g_a1
  float* gcat0 = a1b + T * lh;          // [T][2H+4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int akind = GEN ? a.act : VSN_ACT_SILU;
  const int n0 = blockIdx.x * T;
  const int M = a.N - n0 < T ? a.N - n0 : T;  // live nodes of this tile
  const int MR = M * S;
  HF_STAMP(0);

  // S1: cat0 = [out_norm(x) | || pv0[:, :, :H] ||_s]
  for (int idx = tid; idx < M * H; idx += nthr) {
    const int i = idx / H, c = idx - i * H;
    cat0[i * l2H + c] = a.cat0g[(size_t)(n0 + i) * 2 * H + c];
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float v = a.pv0[((size_t)(n0 + i) * S + s) * ldp + c];
      s2 += v * v;
    }
    cat0[i * l2H + H + c] = sqrtf(s2);
  }
  __syncthreads();
  HF_STAMP(1);
  // S2: a0 = cat0 . Wa0^T + ba0 ; ta = act(a0)
  lds_gemm8<2 * H, H>(cat0, l2H, M, a.W.Wa0p, a.W.ba0, a0, lH, wave, lane);
  __syncthreads();
  HF_STAMP(2);
  for (int idx = tid; idx < M * H; idx += nthr) {
    const int i = idx / H, c = idx - i * H;
    ta[i * lH + c] = act_f(akind, a0[i * lH + c]);
  }
  __syncthreads();
  HF_STAMP(3);
  // S3: u0 = ta . Wb0^T + bb0 = [xs | gate]
  lds_gemm8<H, H>(ta, lH, M, a.W.Wb0p, a.W.bb0, u0, lH, wave, lane);
  __syncthreads();
  HF_STAMP(4);
  // S4: cat1[:, :h2] = act(xs) ; vec1o[s] = gate * pv0[s, H:]
  for (int idx = tid; idx < M * h2; idx += nthr) {
    const int i = idx / h2, c = idx - i * h2;
    cat1[i * lH + c] = act_f(akind, u0[i * lH + c]);
    const float gate = u0[i * lH + h2 + c];
#pragma unroll
    for (int s = 0; s < S; ++s)
      vec1o[(i * S + s) * lh + c] = gate * a.pv0[((size_t)(n0 + i) * S + s) * ldp + H + c];
  }
  __syncthreads();
  HF_STAMP(5);
  // S5: p1 = vec1o . W11^T
  lds_gemm<RBV, h2, h2>(vec1o, lh, MR, a.W.W11p, nullptr, p1, lh, wave, lane);
  __syncthreads();
  HF_STAMP(6);
  // S6: cat1[:, h2:] = || p1 ||_s
  for (int idx = tid; idx < M * h2; idx += nthr) {
    const int i = idx / h2, c = idx - i * h2;
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float v = p1[(i * S + s) * lh + c];
      s2 += v * v;
    }
    cat1[i * lH + h2 + c] = sqrtf(s2);
  }
  __syncthreads();
  HF_STAMP(7);
  // S7: a1b = cat1 . Wa1^T + ba1
  lds_gemm8<H, h2>(cat1, lH, M, a.W.Wa1p, a.W.ba1, a1b, lh, wave, lane);
  __syncthreads();
  HF_STAMP(8);
  // S8: y = std (wb1 . act(a1b) + bb1) + atomref[z] ; g_a1 = std wb1 act'(a1b)   (dE/dy = 1)
  if (wave < M) {
    const int i = wave;
    float acc = 0.f;
    for (int c = lane; c < h2; c += 64) {
      float av, dav;
      act_both(akind, a1b[i * lh + c], av, dav);
      acc += av * a.W.wb1[c];
      a1b[i * lh + c] = a.W.stdv * a.W.wb1[c] * dav;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = (acc + a.W.bb1) * a.W.stdv;
      if (a.W.atomref) v += a.W.atomref[a.zi[n0 + i]];
      a.y[n0 + i] = v;
    }
  }
  __syncthreads();
  HF_STAMP(9);
  // S10: g_cat1 = g_a1 . Wa1
  lds_gemm8<h2, H>(a1b, lh, M, a.W.Wa1Tp, nullptr, gcat1, lH, wave, lane);
  __syncthreads();
  HF_STAMP(10);
  // S11: g_p1[s] = g_v1b / v1b * p1[s]   (0 where v1b == 0, like torch.norm)
  for (int idx = tid; idx < M * h2; idx += nthr) {
    const int i = idx / h2, c = idx - i * h2;
    const float v = cat1[i * lH + h2 + c];
    const float sc = v > 0.f ? gcat1[i * lH + h2 + c] / v : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) p1[(i * S + s) * lh + c] *= sc;
  }
  __syncthreads();
  HF_STAMP(11);
  // S12: g_vec1o = g_p1 . W11
  lds_gemm<RBV, h2, h2>(p1, lh, MR, a.W.W11Tp, nullptr, vec1o, lh, wave, lane);
  __syncthreads();
  HF_STAMP(12);
  // S13: g_gate = sum_s g_vec1o[s] v2[s] ; g_v2[s] = g_vec1o[s] gate ; g_xs = g_x1 act'(xs)
  for (int idx = tid; idx < M * h2; idx += nthr) {
    const int i = idx / h2, c = idx - i * h2;
    const float gate = u0[i * lH + h2 + c];
    float gg = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float gv = vec1o[(i * S + s) * lh + c];
      const size_t pr = ((size_t)(n0 + i) * S + s) * ldp + H + c;
      gg += gv * a.pv0[pr];
      a.g_pv0[pr] = gv * gate;
    }
    ta[i * lH + c] = gcat1[i * lH + c] * dact_f(akind, u0[i * lH + c]);
    ta[i * lH + h2 + c] = gg;
  }
  __syncthreads();
  HF_STAMP(13);
  // S14: g_h0 = (g_u0 . Wb0) * act'(a0)
  lds_gemm8<H, H>(ta, lH, M, a.W.Wb0Tp, nullptr, cat1, lH, wave, lane);
  __syncthreads();
  HF_STAMP(14);
  for (int idx = tid; idx < M * H; idx += nthr) {
    const int i = idx / H, c = idx - i * H;
    cat1[i * lH + c] *= dact_f(akind, a0[i * lH + c]);
  }
  __syncthreads();
  HF_STAMP(15);
  // S15: g_cat0 = g_h0 . Wa0
  lds_gemm8<H, 2 * H>(cat1, lH, M, a.W.Wa0Tp, nullptr, gcat0, l2H, wave, lane);
  __syncthreads();
  HF_STAMP(16);
  // S16: dE/d out_norm(x) -> global ; g_pv0[s, :H] = g_v1 / v1 * pv0[s, :H]
  for (int idx = tid; idx < M * H; idx += nthr) {
    const int i = idx / H, c = idx - i * H;
    a.g_cat0g[(size_t)(n0 + i) * 2 * H + c] = gcat0[i * l2H + c];
    const float v = cat0[i * l2H + H + c];
    const float sc = v > 0.f ? gcat0[i * l2H + H + c] / v : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const size_t pr = ((size_t)(n0 + i) * S + s) * ldp + c;
      a.g_pv0[pr] = sc * a.pv0[pr];
    }
  }
  HF_STAMP(17);
}

static size_t head_fused_lds(int H, int S) {
  const size_t h2 = H / 2, T = HF_T, R = (size_t)HF_T * S;
  return 4 * (2 * T * (2 * H + 4) + 5 * T * (H + 4) + 2 * R * (h2 + 4) + T * (h2 + 4));
}

// (the packed weight copies exist for H % 64 == 0, H <= 256: engine.hip)
bool head_fused_supported(const Dims& D) {
  return (D.S == 3 || D.S == 8) && D.N < 4096 && (D.H == 64 || D.H == 128 || D.H == 192 || D.H == 256) &&
         head_fused_lds(D.H, D.S) <= 160 * 1024;
}

// forward + reverse of the head between pv0 and g_pv0 in one launch (see the file header).  Needs Bf.pv0 (the
// projection GEMM) and out_norm(x) in Bf.cat0[:, :H]; leaves Bf.y, Bf.g_cat0[:, :H] and Bf.g_pv0.
int launch_head_fused(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf) {
  if (D.N <= 0) return 0;
  HeadFusedArgs a;
  a.N = D.N;
  a.S = D.S;
  a.H = D.H;
  a.act = D.act;
  a.cat0g = Bf.cat0;
  a.pv0 = Bf.pv0;
  a.W = W;
  a.zi = D.zi;
  a.y = Bf.y;
  a.g_cat0g = Bf.g_cat0;
  a.g_pv0 = Bf.g_pv0;
  const size_t lds = head_fused_lds(D.H, D.S);
  const dim3 grid((D.N + HF_T - 1) / HF_T), blk(64 * HF_WAVES);
  const bool gen = D.act != VSN_ACT_SILU;
#define HF_GO(H_, S_, G_)                                                                                        \
  do {                                                                                                           \
    static unsigned long long attr_set = 0; /* per device (one bit each): > 64 KB of dynamic LDS is opt-in */    \
    int dev_ = 0;                                                                                                \
    (void)hipGetDevice(&dev_);                                                                                   \
    if (!((attr_set >> (dev_ & 63)) & 1ull)) {                                                                   \
      if (hipFuncSetAttribute((const void*)hk_fused<H_, S_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              160 * 1024) != hipSuccess)                                                         \
        return -5;                                                                                               \
      attr_set |= 1ull << (dev_ & 63);                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL((hk_fused<H_, S_, G_>), grid, blk, lds, st, a);                                           \
  } while (0)
#define HF_H(H_)                    \
  do {                              \
    if (D.S == 8) {                 \
      if (gen) HF_GO(H_, 8, true);  \
      else HF_GO(H_, 8, false);     \
    } else {                        \
      if (gen) HF_GO(H_, 3, true);  \
      else HF_GO(H_, 3, false);     \
    }                               \
  } while (0)
  switch (D.H) {
    case 64: HF_H(64); break;
    case 128: HF_H(128); break;
    case 192: HF_H(192); break;
    case 256: HF_H(256); break;
    default: return -22;
  }
#undef HF_H
#undef HF_GO
  return 0;
}

}  // namespace vsn

#ifdef HF_LAB_STAMPS
extern "C" int vsn_lab_set_hf_stamps(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vsn::g_hf_stamps), &q, sizeof(q));
}
#endif
