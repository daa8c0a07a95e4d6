// fp32 MFMA GEMM for the dense linear blocks of ViSNet (reference: every
// nn.Linear in ViSNet/model/visnet_block.py:183-203, utils.py:282-283,323,
// output_modules.py:30-39 and their input-gradient products).
//
//   C[M,Nc] (+)= f(A)[M,K] * Bt[Nc,K]^T (+ bias[Nc])
//
// A is row-major with K contiguous (activations), Bt is the nn.Linear weight
// layout [out,in] (K contiguous) so forward products need no transpose; the
// reverse pass uses pre-transposed weight copies made once at load time.
// Arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 (an fmaf chain), 157 TF peak.
//
// Tile: BM x BN per 256-thread workgroup (4 waves as WM x WN), BK = 32.
// LDS rows are padded to 36 floats so the ds_read_b128 fragment reads of a
// 16-lane service group land in 16 distinct 16-byte slots (conflict free).
// Each lane reads 4 consecutive k of its row: lanes 0-31 take k0..k0+3, lanes
// 32-63 take k0+4..k0+7; MFMA #t pairs (k0+t, k0+4+t) for A and B alike, so
// the k-sum is just reordered.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

#ifndef VSN_LAB_PRIO
#define VSN_LAB_PRIO 0  // tools/lab/gemm_direct.hip: s_setprio placement experiments (0 = none, the product build)
#endif

namespace vsn {

// SILU: apply the activation to A on its way into LDS (one product of the read-out head); a template parameter so
// that the ~250 VALU instructions of the exact sigmoid are not sitting in the k-loop of every other product
// (1 = silu, 2 = any kind of the reference's table, carried in flags bits 8..).
template <int BM, int BN, int WM, int WN, bool DB, int SILU = 0, int BK = 32, int PF = 1>
__device__ __forceinline__ void gemm_body(const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                          int ldb, float* __restrict__ C, int ldc,
                                          const float* __restrict__ bias, int M, const int* __restrict__ Mptr,
                                          int Nc, int K, int flags, int ksplit, float* __restrict__ part,
                                          int block_id, float* __restrict__ smem) {
  constexpr int LS = BK + 4;  // padded LDS row stride (floats)
  constexpr int C4 = BK / 4;  // 16-byte chunks per tile row
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int LA = BM * C4 / 256, LB = BN * C4 / 256;  // 16-byte loads per thread per k-tile
  constexpr int STAGE = (BM + BN) * LS;

  int Meff = M;
  if (Mptr) {
    int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int tiles_n = Nc / BN;
  // live blocks = those covering rows < Meff (the grid is sized by the host-side bound M).  The XCD-aware
  // renumbering is done over the LIVE blocks only, so every XCD gets the same share of real work and the
  // column tiles of one row tile (which share the A tile) meet in one L2.
  const int live = ((Meff + BM - 1) / BM) * tiles_n * ksplit;
  if (block_id >= live) return;
  const int bid = VSN_XCD_REMAP ? xcd_block(block_id, live) : block_id;
  const int tile = bid / ksplit, ks = bid % ksplit;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr bool silu_a = SILU != 0;  // 1: silu (folds to the branch-free form), 2: kind from flags bits 8..
  const int akind = SILU == 2 ? (flags >> 8) : VSN_ACT_SILU;
  // this workgroup's K range (split-K: partial sums go to `part`, reduced by k_gemm_reduce)
  const int nkt_all = K / BK;
  const int kt0 = (int)((long long)nkt_all * ks / ksplit), kt1 = (int)((long long)nkt_all * (ks + 1) / ksplit);
  const int nkt = kt1 - kt0;
  const int kbase = kt0 * BK;

  // PF = global-load prefetch depth of the double-buffered pipeline, in k-tiles held in registers (1 or 2)
  f32x4 rra[PF][LA], rrb[PF][LB];
#define VSN_GLOAD(k0) VSN_GLOAD_S(0, k0)
#define VSN_SSTORE(buf) VSN_SSTORE_S(0, buf)
#define VSN_GLOAD_S(SET, k0)                                                                   \
  {                                                                                          \
    _Pragma("unroll") for (int it = 0; it < LA; ++it) {                                      \
      const int f_ = tid + it * 256;                                                         \
      const int r_ = f_ / C4, c4_ = f_ % C4;                                                  \
      int gr_ = row0 + r_;                                                                   \
      gr_ = gr_ < Meff ? gr_ : Meff - 1; /* clamp: rows >= Meff are never stored */          \
      rra[SET][it] = *reinterpret_cast<const f32x4*>(A + (size_t)gr_ * lda + (k0) + c4_ * 4); \
    }                                                                                        \
    _Pragma("unroll") for (int it = 0; it < LB; ++it) {                                      \
      const int f_ = tid + it * 256;                                                         \
      const int r_ = f_ / C4, c4_ = f_ % C4;                                                  \
      rrb[SET][it] = *reinterpret_cast<const f32x4*>(Bt + (size_t)(col0 + r_) * ldb + (k0) + c4_ * 4); \
    }                                                                                        \
  }
#define VSN_SSTORE_S(SET, buf)                                                \
  {                                                                           \
    float* As_ = smem + (buf) * STAGE;                                        \
    float* Bs_ = As_ + BM * LS;                                               \
    _Pragma("unroll") for (int it = 0; it < LA; ++it) {                       \
      const int f_ = tid + it * 256;                                          \
      const int r_ = f_ / C4, c4_ = f_ % C4;                                   \
      f32x4 v_ = rra[SET][it];                                                \
      if (silu_a) { /* activation kind (VSN_ACT_*) rides in flags bits 8.. */  \
        v_.x = act_f(akind, v_.x);                                            \
        v_.y = act_f(akind, v_.y);                                            \
        v_.z = act_f(akind, v_.z);                                            \
        v_.w = act_f(akind, v_.w);                                            \
      }                                                                       \
      *reinterpret_cast<f32x4*>(As_ + r_ * LS + c4_ * 4) = v_;                \
    }                                                                         \
    _Pragma("unroll") for (int it = 0; it < LB; ++it) {                       \
      const int f_ = tid + it * 256;                                          \
      const int r_ = f_ / C4, c4_ = f_ % C4;                                   \
      *reinterpret_cast<f32x4*>(Bs_ + r_ * LS + c4_ * 4) = rrb[SET][it];      \
    }                                                                         \
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // accumulate mode on the one-accumulator tile (64x64, the grouped reverse-pass products): fetch the old C values
  // now, so that the epilogue does not wait for a memory round trip per tile
  constexpr bool PREC = (MI * NI == 1);
  const bool accum = (flags & 1) != 0;
  float cold[PREC ? 16 : 1];
  float bvs[NI];  // the bias too: one load per lane, but at the end of the tile it is a full round trip on the tail
#pragma unroll
  for (int j = 0; j < NI; ++j) bvs[j] = (bias && ksplit == 1) ? bias[col0 + wn * TN + j * 32 + l31] : 0.f;
  if (PREC && accum && ksplit == 1) {
    const int col = col0 + wn * TN + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = row0 + wm * TM + (r & 3) + 8 * (r >> 2) + 4 * hi;
      row = row < Meff ? row : Meff - 1;
      cold[r] = C[(size_t)row * ldc + col];
    }
  }
  if (DB) {
    // software pipeline: LDS holds tile kt (buffer kt&1), registers hold tile kt+1, one barrier per tile
    VSN_GLOAD(kbase);
    VSN_SSTORE(0);
    {
      const int kn = (1 < nkt ? 1 : 0) * BK + kbase;
      VSN_GLOAD(kn);  // set 0 <- tile 1
    }
    if constexpr (PF == 2) {
      const int kn = (2 < nkt ? 2 : nkt - 1) * BK + kbase;
      VSN_GLOAD_S(PF - 1, kn);  // set 1 <- tile 2
    }
    __syncthreads();
  } else {
    VSN_GLOAD(kbase);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    if (!DB) {
      VSN_SSTORE(0);
      __syncthreads();
      const int kn = (kt + 1 < nkt ? kt + 1 : kt) * BK + kbase;
      VSN_GLOAD(kn);
      // keep the prefetch HERE: without this hipcc sinks the loads below the MFMA block (to shorten their
      // live ranges), which exposes the full global-load latency once per k-tile
      __builtin_amdgcn_sched_barrier(0);
    }
    const float* As = smem + (DB ? (kt & 1) : 0) * STAGE;
    const float* Bs = As + BM * LS;
#if VSN_LAB_PRIO == 2
    __builtin_amdgcn_s_setprio(2);  // lab: favour waves inside their MFMA block
#endif
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(As + (wm * TM + i * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * TN + j * 32 + l31) * LS + kk * 8 + hi * 4);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
#if VSN_LAB_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#elif VSN_LAB_PRIO == 1
    __builtin_amdgcn_s_setprio(3);  // lab: favour waves in their LDS-store / prefetch / barrier phase
#endif
    if (DB && PF == 1 && kt + 1 < nkt) {
      // tile kt+1 (in registers) -> the other LDS buffer (last read in iteration kt-1, a barrier ago)
      VSN_SSTORE((kt + 1) & 1);
      const int kn = (kt + 2 < nkt ? kt + 2 : kt + 1) * BK + kbase;
      VSN_GLOAD(kn);
      __builtin_amdgcn_sched_barrier(0);  // issue the prefetch before the barrier / next MFMA block
    }
    if constexpr (DB && PF == 2) {
      // two tiles in flight: set (kt & 1) holds tile kt+1 (loaded two iterations ago), the other set tile kt+2;
      // after the store the freed set fetches tile kt+3 - a load now has two MFMA blocks to land, which a workgroup
      // that is alone on its CU (grid tails, launches of ~3 tiles per CU) needs
      if (kt + 1 < nkt) {
        const int kn = (kt + 3 < nkt ? kt + 3 : nkt - 1) * BK + kbase;
        if ((kt & 1) == 0) {
          VSN_SSTORE_S(0, (kt + 1) & 1);
          VSN_GLOAD_S(0, kn);
        } else {
          VSN_SSTORE_S(PF - 1, (kt + 1) & 1);
          VSN_GLOAD_S(PF - 1, kn);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
#if VSN_LAB_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
  }

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + wn * TN + j * 32 + l31;
      const float bv = bvs[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < Meff) {
          if (ksplit == 1) {
            float* cp = C + (size_t)row * ldc + col;
            float v = acc[i][j][r] + bv;
            if (accum) v += PREC ? cold[r] : *cp;
            *cp = v;
          } else {
            part[((size_t)ks * M + row) * Nc + col] = acc[i][j][r];
          }
        }
      }
    }
#undef VSN_GLOAD
#undef VSN_SSTORE
#undef VSN_GLOAD_S
#undef VSN_SSTORE_S
}

template <int BM, int BN, int WM, int WN, bool DB, int SILU = 0, int BK = 32, int PF = 1>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda,
                                              const float* __restrict__ Bt, int ldb,
                                              float* __restrict__ C, int ldc,
                                              const float* __restrict__ bias, int M,
                                              const int* __restrict__ Mptr, int Nc, int K, int flags,
                                              int ksplit, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * (BM + BN) * (BK + 4)];
  gemm_body<BM, BN, WM, WN, DB, SILU, BK, PF>(A, lda, Bt, ldb, C, ldc, bias, M, Mptr, Nc, K, flags, ksplit, part,
                                      (int)blockIdx.x, smem);
}

// Several independent products in ONE launch (64x64 tiles): block -> (problem, tile).  Used for the
// per-layer groups {qkv, vector projections, edge linears}, {s_proj, o_proj}, {dX products}: on a
// single-protein MD step the small members (N = a few hundred rows) cannot fill 256 CUs alone.
template <int PF>
__global__ __launch_bounds__(256) void k_gemm_group(GemmGroup g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 64) * 36];
  int b = (int)blockIdx.x, p = 0;
#pragma unroll
  for (int q = 0; q < GemmGroup::MAXP - 1; ++q)
    if (p == q && q + 1 < g.n && b >= g.p[q].blocks) {
      b -= g.p[q].blocks;
      p = q + 1;
    }
  const GemmDesc& d = g.p[p];
  gemm_body<64, 64, 2, 2, true, 0, 32, PF>(d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K,
                                           d.flags, d.ksplit, d.part, b, smem);
}

// split-K epilogue: C (+)= sum_s part[s] (+ bias), fixed summation order
__global__ void k_gemm_reduce(const float* __restrict__ part, int ksplit, float* __restrict__ C, int ldc,
                              const float* __restrict__ bias, int M, const int* __restrict__ Mptr, int Nc,
                              int flags) {
  int Meff = M;
  if (Mptr) {
    int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int nc4 = Nc >> 2;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = gid / nc4;
  const int c4 = (int)(gid % nc4);
  if (row >= Meff) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + ((size_t)row) * Nc + c4 * 4);
  for (int k = 1; k < ksplit; ++k)
    s += *reinterpret_cast<const f32x4*>(part + ((size_t)k * M + row) * Nc + c4 * 4);
  if (bias) s += *reinterpret_cast<const f32x4*>(bias + c4 * 4);
  float* cp = C + (size_t)row * ldc + c4 * 4;
  if (flags & 1) s += *reinterpret_cast<const f32x4*>(cp);
  *reinterpret_cast<f32x4*>(cp) = s;
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream ----
int g_splitk_tiles = 400;   // split-K only below this many output tiles (env VSN_SPLITK_TILES; swept on Chignolin: 0:297, 200:306, 400:306, 768:301, 1200:289 steps/s)
int g_gemm_force = 0;     // micro-benchmark aid (env VSN_GEMM_FORCE): force an experimental tile variant
int g_gemm_pf = 1;  // global-load prefetch depth of the grouped 64x64 kernel (env VSN_GEMM_PF: 1 | 2)
int g_gemm_db128 = 0;  // A/B switch (env VSN_GEMM_DB128): measured 3-6 % slower than single-buffered at 128x128
static thread_local GemmProfiler* tl_prof = nullptr;
void set_gemm_profiler(GemmProfiler* p) { tl_prof = p; }
// split-K scratch (set by the engine per chunk; nullptr disables split-K)
static thread_local float* tl_splitk_ws = nullptr;
static thread_local size_t tl_splitk_elems = 0;
void set_gemm_splitk_workspace(float* p, size_t elems) {
  tl_splitk_ws = p;
  tl_splitk_elems = elems;
}

// 128x128 tiles only when they still give >= 4 workgroups per CU; otherwise the 4x finer
// 64x64 tiling fills the 256 CUs better (one protein per MD step: M of a few thousand rows)
static bool gemm_env_init() {
  const char* e = getenv("VSN_GEMM_DB128");
  if (e) g_gemm_db128 = atoi(e);
  e = getenv("VSN_SPLITK_TILES");
  if (e) g_splitk_tiles = atoi(e);
  e = getenv("VSN_GEMM_PF");
  if (e) g_gemm_pf = atoi(e);
  e = getenv("VSN_GEMM_FORCE");
  if (e) g_gemm_force = atoi(e);
  return true;
}

int gemm_variant(int M, int Nc) {
  if ((Nc % 128) == 0 && (long long)((M + 127) / 128) * (Nc / 128) >= 1024) return 0;
  if ((Nc % 64) == 0) return 1;
  return 2;
}

int launch_gemm(hipStream_t st, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc,
                const float* bias, int M, const int* Mptr, int Nc, int K, int flags) {
  if (M <= 0) return 0;
  if ((K & 31) || (Nc & 31) || (lda & 3) || (ldb & 3)) return -22;
  GemmProfiler::Rec* rec = nullptr;
  if (tl_prof) {
    tl_prof->recs.emplace_back();
    rec = &tl_prof->recs.back();
    rec->variant = gemm_variant(M, Nc);
    rec->M = M;
    rec->dev_m = Mptr != nullptr;
    rec->group_n = 0;
    rec->flops_per_row = 2.0 * (double)Nc * (double)K;
    rec->bytes_per_row = 4.0 * ((double)K + (double)Nc * ((flags & 1) ? 2.0 : 1.0));
    hipEventCreate(&rec->a);
    hipEventCreate(&rec->b);
    hipEventRecord(rec->a, st);
  }
  struct Fin {
    GemmProfiler::Rec* r;
    hipStream_t s;
    ~Fin() {
      if (r) hipEventRecord(r->b, s);
    }
  } fin{rec, st};
  static const bool env_init = gemm_env_init();
  (void)env_init;
  if (g_gemm_force >= 10 && !tl_prof) {
    // experimental variants, single launch, no split-K
#define VSN_TRY(ID, BM_, BN_, WM_, WN_, DB_)                                                                   \
    if (g_gemm_force == ID && (Nc % BN_) == 0) {                                                                \
      const int grid_ = ((M + BM_ - 1) / BM_) * (Nc / BN_);                                                     \
      hipLaunchKernelGGL((k_gemm<BM_, BN_, WM_, WN_, DB_>), dim3(grid_), dim3(256), 0, st, A, lda, Bt, ldb, C,  \
                         ldc, bias, M, Mptr, Nc, K, flags, 1, nullptr);                                         \
      return 0;                                                                                                 \
    }
    VSN_TRY(10, 64, 64, 2, 2, false)
    VSN_TRY(11, 64, 128, 2, 2, true)
    VSN_TRY(12, 128, 64, 2, 2, true)
    VSN_TRY(13, 64, 128, 2, 2, false)
    VSN_TRY(14, 128, 64, 2, 2, false)
    VSN_TRY(15, 64, 64, 2, 2, true)
    VSN_TRY(16, 128, 128, 2, 2, false)
#undef VSN_TRY
  }
  if (flags & 2) {  // silu(A): its own instantiations, no split-K
    const bool gen = (flags >> 8) != VSN_ACT_SILU;
    const dim3 g64(((M + 63) / 64) * (Nc / 64)), g32(((M + 127) / 128) * (Nc / 32));
    if ((Nc % 64) == 0 && !gen)
      hipLaunchKernelGGL((k_gemm<64, 64, 2, 2, true, 1>), g64, dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M, Mptr,
                         Nc, K, flags, 1, nullptr);
    else if ((Nc % 64) == 0)
      hipLaunchKernelGGL((k_gemm<64, 64, 2, 2, true, 2>), g64, dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M, Mptr,
                         Nc, K, flags, 1, nullptr);
    else if (!gen)
      hipLaunchKernelGGL((k_gemm<128, 32, 4, 1, true, 1>), g32, dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M, Mptr,
                         Nc, K, flags, 1, nullptr);
    else
      hipLaunchKernelGGL((k_gemm<128, 32, 4, 1, true, 2>), g32, dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M, Mptr,
                         Nc, K, flags, 1, nullptr);
    return 0;
  }
  const int variant = gemm_variant(M, Nc);
  const int bm = variant == 1 ? 64 : 128, bn = variant == 0 ? 128 : (variant == 1 ? 64 : 32);
  const int tiles = ((M + bm - 1) / bm) * (Nc / bn);
  // split-K: few output tiles and a long K (the dX = dY.W products of the reverse pass on small
  // batches) would leave most CUs idle; cut K over `ks` workgroups and reduce deterministically.
  int ks = 1;
  if (tiles < g_splitk_tiles && K >= 512 && tl_splitk_ws && (ldc & 3) == 0) {
    ks = (1024 + tiles - 1) / tiles;
    const int kmax = K / 128;  // keep >= 4 k-tiles per split
    if (ks > kmax) ks = kmax;
    if (ks > 8) ks = 8;
    while (ks > 1 && (size_t)ks * M * Nc > tl_splitk_elems) --ks;
    if (ks < 1) ks = 1;
  }
  float* part = ks > 1 ? tl_splitk_ws : nullptr;
  const int grid = tiles * ks;
  if (variant == 0) {
    if (g_gemm_db128) hipLaunchKernelGGL((k_gemm<128, 128, 2, 2, true>), dim3(grid), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M,
                       Mptr, Nc, K, flags, ks, part);
    else hipLaunchKernelGGL((k_gemm<128, 128, 2, 2, false>), dim3(grid), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M,
                       Mptr, Nc, K, flags, ks, part);
  } else if (variant == 1) {
    hipLaunchKernelGGL((k_gemm<64, 64, 2, 2, true>), dim3(grid), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M,
                       Mptr, Nc, K, flags, ks, part);
  } else {
    hipLaunchKernelGGL((k_gemm<128, 32, 4, 1, true>), dim3(grid), dim3(256), 0, st, A, lda, Bt, ldb, C, ldc, bias, M,
                       Mptr, Nc, K, flags, ks, part);
  }
  if (ks > 1) {
    long long n4 = (long long)M * (Nc / 4);
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, ks, C, ldc, bias,
                       M, Mptr, Nc, flags);
  }
  return 0;
}

int launch_gemm_group(hipStream_t st, const GemmDesc* descs, int n) {
  static const bool env_init = gemm_env_init();
  (void)env_init;
  // fall back to individual launches when the group would not use the 64x64 tiling anyway
  bool groupable = n > 1 && n <= GemmGroup::MAXP;
  long long tiles = 0;
  for (int i = 0; i < n && groupable; ++i) {
    const GemmDesc& d = descs[i];
    if (d.M <= 0) continue;
    if ((d.K & 31) || (d.Nc & 63) || (d.lda & 3) || (d.ldb & 3)) groupable = false;
    if (d.keep_parts > 1 && (!d.part || (d.K / 32) < d.keep_parts || (d.flags & 1) || d.bias)) return -22;
    if (gemm_variant(d.M, d.Nc) == 0) groupable = false;  // big enough to fill the chip alone
    if (d.flags & 2) groupable = false;                   // silu(A) has its own kernels
    tiles += (long long)((d.M + 63) / 64) * (d.Nc / 64);
  }
  if (!groupable) {
    for (int i = 0; i < n; ++i)
      if (descs[i].keep_parts > 1) return -22;  // partial slices are a grouped-launch feature (caller's guard)
    for (int i = 0; i < n; ++i) {
      const GemmDesc& d = descs[i];
      int rc = launch_gemm(st, d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K, d.flags);
      if (rc) return rc;
    }
    return 0;
  }
  GemmGroup g;
  g.n = 0;
  int grid = 0;
  GemmProfiler::Rec* rec = nullptr;
  if (tl_prof) {
    tl_prof->recs.emplace_back();
    rec = &tl_prof->recs.back();
    rec->variant = 3;  // k_gemm_group
    rec->M = 1;
    rec->dev_m = false;
    rec->flops_per_row = 0;
    rec->bytes_per_row = 0;
    rec->group_n = 0;
    hipEventCreate(&rec->a);
    hipEventCreate(&rec->b);
    hipEventRecord(rec->a, st);
  }
  // split-K members share the scratch buffer: carve it
  size_t ws_off = 0;
  GemmDesc red[GemmGroup::MAXP];
  int nred = 0;
  for (int i = 0; i < n; ++i) {
    GemmDesc d = descs[i];
    if (d.M <= 0) continue;
    const int t = ((d.M + 63) / 64) * (d.Nc / 64);
    int ks = 1;
    if (d.keep_parts > 1) {
      ks = d.keep_parts;  // slices stay in d.part (caller's buffer); the consumer sums them
      d.ksplit = ks;
    } else {
      if (tiles < g_splitk_tiles && d.K >= 512 && tl_splitk_ws && (d.ldc & 3) == 0) {
        ks = (int)((1024 + tiles - 1) / tiles);
        const int kmax = d.K / 128;
        if (ks > kmax) ks = kmax;
        if (ks > 8) ks = 8;
        while (ks > 1 && ws_off + (size_t)ks * d.M * d.Nc > tl_splitk_elems) --ks;
        if (ks < 1) ks = 1;
      }
      d.ksplit = ks;
      d.part = ks > 1 ? tl_splitk_ws + ws_off : nullptr;
      if (ks > 1) {
        ws_off += (size_t)ks * d.M * d.Nc;
        red[nred++] = d;
      }
    }
    d.blocks = (t * ks + 7) & ~7;  // multiple of 8: local block id % 8 stays the XCD id
    grid += d.blocks;
    if (rec) {
      rec->gM[rec->group_n] = d.M;
      rec->gdev[rec->group_n] = d.Mptr != nullptr;
      rec->gflops[rec->group_n] = 2.0 * d.Nc * d.K;
      rec->gbytes[rec->group_n] = 4.0 * (d.K + d.Nc * ((d.flags & 1) ? 2.0 : 1.0));
      rec->group_n++;
    }
    g.p[g.n++] = d;
  }
  if (g.n > 0) {
    if (g_gemm_pf == 2) hipLaunchKernelGGL(k_gemm_group<2>, dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(k_gemm_group<1>, dim3(grid), dim3(256), 0, st, g);
  }
  for (int i = 0; i < nred; ++i) {
    const GemmDesc& d = red[i];
    long long n4 = (long long)d.M * (d.Nc / 4);
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, d.part, d.ksplit, d.C,
                       d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.flags);
  }
  if (rec) hipEventRecord(rec->b, st);
  return 0;
}

}  // namespace vsn
