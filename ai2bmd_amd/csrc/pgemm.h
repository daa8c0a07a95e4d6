// Panel GEMM: fp32 MFMA products with the A operand STATIONARY in LDS and the B operand (a weight matrix,
// constant for the life of the model) streamed straight from L2 into registers in MFMA-fragment order.
//
//   C[M,Nc] (+)= A[M,K] * Bt[Nc,K]^T (+ bias)          v_mfma_f32_32x32x2_f32, exact fp32
//
// Why: the LDS-tiled kernel of gemm.hip stages A AND B through LDS with one barrier per 32-deep
// k-tile; its workgroup owns one output tile, so nothing row-local can be fused around it without recomputing it per
// column tile.  Here a workgroup owns a PANEL of BM rows for ALL its output columns:
//   * the panel's K-slice of 256 floats per row (BM KiB) is written into LDS once, by LDS-DMA
//     (global_load_lds_dwordx4: one instruction = one 1-KiB row, no VGPRs, no ds_write pass), or PRODUCED there by a
//     fused per-row prologue (the gather kernels that used to write the operand to HBM);
//   * the weights are packed once at load time so that the 64 lanes of a wave read their B fragment of a
//     (32 columns x 8 k) block as ONE coalesced 1-KiB global_load_dwordx4 - no LDS, no barrier, a register ring of
//     D blocks in flight per wave;
//   * the k-loop therefore has NO barrier at all: waves run free, two workgroups (8 waves) per CU keep each SIMD's
//     matrix pipe fed while a neighbour is in its prologue or epilogue.
// Shapes: "fwd" K == 256 (the panel is the whole K), any Nc % (32 WN) == 0: loop over column tiles, fresh
// accumulators per tile; "bwd" Nc == 256 (the accumulators hold the whole row panel), K % 256 == 0: loop over
// K-slices, the panel is re-filled per slice.  Other hidden sizes keep the LDS-tiled kernels.
//
// Operand roles are SWAPPED in the MFMA (weights as its "A", activations as its "B"): D[i][j] = sum_k W[i][k] X[j][k],
// so a lane ends up with ONE output row (j = lane & 31) and 16 output columns in four runs of four consecutive ones
// ((r & 3) + 8 (r >> 2) + 4 (lane >> 5)): the epilogue is 4 x 16-byte stores per accumulator instead of 16 x 4-byte
// ones (the store burst of the row-per-register layout cost 9 % of the kernel).  Same k order, bitwise the same sums.
//
// LDS image of the panel: row r = 256 floats = 64 chunks of 16 B; chunk c lives at chunk position c ^ (r & 15), so
// the ds_read_b128 of a 16-lane service group (16 consecutive rows mod 16, one logical chunk) hits 16 distinct
// 16-byte slots of the 256-byte bank row.  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane
// SOURCE address (lane L of row r fetches chunk L ^ (r & 15)).
#pragma once
#include "common.h"

namespace vsn {

#define VSN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
template <typename T>
__device__ __forceinline__ T* uni_ptr(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// element Bt[n][k] of a [Nc, K] weight matrix in packed order: blocks of (32 columns x 8 k) = 64 lanes x 4 floats,
// block (nb, ks) at ((nb * K/8) + ks) * 256 floats; lane = (n % 32) + 32 * ((k % 8) / 4), component k % 4
#define VSN_PGEMM_PAD_FLOATS (16 * 256)  // read-ahead past the end of a packed matrix (ring depth <= 16 blocks)
static inline size_t pgemm_pack_index(int n, int k, int K) {
  return ((size_t)(n >> 5) * (size_t)(K >> 3) + (size_t)(k >> 3)) * 256 + (size_t)(((n & 31) + 32 * ((k & 7) >> 2)) * 4 + (k & 3));
}

// per-lane LDS float offsets of the activation fragment of k-step ks (8 k: lanes 0-31 take k0..k0+3, lanes 32-63
// k0+4..k0+7) for panel row (l31 + 32 i): fo[ks & 7] + (ks >> 3) * 64 + i * 32 * 256
__device__ __forceinline__ void panel_frag_offsets(int lane, int (&fo)[8]) {
  const int l31 = lane & 31, hi = lane >> 5, x = l31 & 15;
#pragma unroll
  for (int q = 0; q < 8; ++q) fo[q] = l31 * 256 + ((((q << 1) | hi) ^ x) << 2);
}

// LDS float offset of logical 16-byte chunk c (0..63) of panel row r (for prologues that WRITE the panel)
__device__ __forceinline__ int panel_at(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 2); }

// fill panel rows [0, BM) with A[row0 + r][k0 .. k0 + 256) by LDS-DMA; NW waves, wave w takes rows w*BM/NW ...
// (with BM / NW a multiple of 16, r & 15 is the compile-time loop index)
template <int BM, int NW>
__device__ __forceinline__ void panel_load_dma(float* __restrict__ smem, const float* __restrict__ A, int lda,
                                               int row0, int Meff, int k0, int wave, int lane) {
  constexpr int RPW = BM / NW;
#pragma unroll
  for (int it = 0; it < RPW; ++it) {
    const int r = wave * RPW + it;
    int gr = row0 + r;
    gr = gr < Meff ? gr : Meff - 1;  // rows past the end are clamped (computed, never stored)
    const float* g = A + (size_t)gr * lda + k0 + ((lane ^ (r & 15)) << 2);
    __builtin_amdgcn_global_load_lds(g, VSN_LDS_PTR(smem + r * 256), 16, 0, 0);
  }
}

// the 16 values a lane holds of one 32 x 32 accumulator = row j of the panel block, columns c0 + 8 q + 4 hi + (0..3):
// four 16-byte pieces.  Branch-free epilogue through a buffer resource that ends at the last valid row: rows past it
// are dropped by the hardware, and the compiler sees ONE straight-line path, so its s_waitcnt counts for the B ring
// stay exact across the stores (with a branchy epilogue it waited for every store of a tile before the next tile's
// first MFMA).  EPI: 0 = store, 1 = store + bias, 2 = accumulate into C.
template <int EPI>
__device__ __forceinline__ void panel_store_acc(const f32x16& a, const __amdgpu_buffer_rsrc_t rs, int byte_off,
                                                const f32x4 (&bv)[4]) {
  f32x4 old[4];
  if (EPI == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      old[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + q * 32, 0, 0));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
    if (EPI == 1) v += bv[q];
    if (EPI == 2) v += old[q];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off + q * 32, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fwd: K == 256.  Workgroup = WN waves over a panel of BM = 32 MI rows; every wave owns ALL rows (MI accumulators of
// 32 x 32) and the 32 columns `wave` of every 32 WN-wide column tile.
// ---------------------------------------------------------------------------------------------------------------
// ABL (lab only): 1 = no B refills, 2 = no stores, 4 = no A fragment reads
template <int MI, int WN, int D, int ABL = 0>
struct PgemmFwd {
  static constexpr int BM = 32 * MI;
  static constexpr int BN = 32 * WN;
  static constexpr int NT = 64 * WN;
  static constexpr int LDS_FLOATS = BM * 256;

  // the wave's B stream: blocks (nb = ct * WN + wave, ks = 0..31), 1 KiB each, 32 KiB per column tile; the first D
  // blocks are requested BEFORE the panel barrier (they do not depend on the panel)
  struct Ring {
    f32x4 rb[D];
  };
  static __device__ __forceinline__ void prefetch(Ring& R, const float* __restrict__ Bp, int ct0, int wave, int lane) {
    const char* __restrict__ bcur = reinterpret_cast<const char*>(Bp + ((size_t)(ct0 * WN + wave) * 32) * 256);
    const unsigned lo = (unsigned)lane * 16u;
#pragma unroll
    for (int d = 0; d < D; ++d) R.rb[d] = *reinterpret_cast<const f32x4*>(bcur + lo + d * 1024);
  }
  // an empty asm "using" the ring right behind the barrier: LLVM otherwise sinks the prefetch loads into the tile
  // loop's preheader, BEHIND the barrier's vmcnt(0); with the ring complete on entry, the s_waitcnt counts hipcc
  // derives for the tile loop are the back-edge's exact ones
  static __device__ __forceinline__ void pin(Ring& R) {
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(R.rb[d]));
  }
  // the k-loop + epilogue over column tiles [ct0, ct1) once the panel is in LDS (caller has synchronised)
  template <int EPI>
  static __device__ __forceinline__ void compute(Ring& R, const float* __restrict__ smem,
                                                 const float* __restrict__ Bp, float* __restrict__ C, int ldc,
                                                 const float* __restrict__ bias, int row0, int Meff, int ct0, int ct1,
                                                 int wave, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    int fo[8];
    panel_frag_offsets(lane, fo);
    // B stream: wave-uniform base (SGPRs) + 32-bit lane offset
    const char* __restrict__ bcur = reinterpret_cast<const char*>(Bp + ((size_t)(ct0 * WN + wave) * 32) * 256);
    const unsigned lo = (unsigned)lane * 16u;
    f32x4(&rb)[D] = R.rb;
    // activation fragments are read one k-step ahead of their MFMAs (the panel is stationary: step 0 of the next
    // column tile reads the same fragments as step 0 of this one, so the pipeline simply wraps around)
    f32x4 fa[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(smem + fo[0] + i * 32 * 256);
    int rows = Meff - row0;
    rows = rows < 0 ? 0 : (rows > BM ? BM : rows);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(uni_ptr(C + (size_t)row0 * ldc), 0, uni(rows * ldc * 4), 0x00020000);
#pragma unroll 1
    for (int ct = ct0; ct < ct1; ++ct) {
      const int col0 = ct * BN + wave * 32;  // first column of this wave's 32
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (EPI == 1) bv[q] = *reinterpret_cast<const f32x4*>(bias + col0 + 8 * q + 4 * hi);
        else bv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const char* __restrict__ bnext =
          ct + 1 < ct1 ? reinterpret_cast<const char*>(Bp + ((size_t)((ct + 1) * WN + wave) * 32) * 256) : bcur;
      f32x16 acc[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        // one k-step = {MI ds_read_b128 (the NEXT step's activation fragments), 1 global_load_dwordx4 (ring refill, D
        // steps ahead; past this tile's 32 blocks it is the next tile's stream), 4 MI MFMA}, issued in that order
        const int kn = (ks + 1) & 31;
        f32x4 na[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (ABL & 4) na[i] = fa[i];
          else na[i] = *reinterpret_cast<const f32x4*>(smem + fo[kn & 7] + (kn >> 3) * 64 + i * 32 * 256);
        }
        const f32x4 fb = rb[ks % D];
        if (!(ABL & 1))
          rb[ks % D] = *reinterpret_cast<const f32x4*>(ks + D < 32 ? bcur + lo + (ks + D) * 1024
                                                                    : bnext + lo + (ks + D - 32) * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.x, fa[i].x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.y, fa[i].y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.z, fa[i].z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.w, fa[i].w, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = na[i];
        __builtin_amdgcn_sched_group_barrier(0x100, MI, 0);      // DS read
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MI, 0);  // MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
      bcur = bnext;
      if (ABL & 2) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][r];
        if (sum == 12345.678f) C[0] = sum;
      } else {
        // (row offsets recomputed per tile from an opaque copy of ldc: hoisted out of the tile loop they would sit in
        //  VGPRs across the k-loop)
        int ldo = ldc * 4;
        asm volatile("" : "+s"(ldo));
#pragma unroll
        for (int i = 0; i < MI; ++i)
          panel_store_acc<EPI>(acc[i], rs, (i * 32 + l31) * ldo + (col0 + 4 * hi) * 4, bv);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// bwd: Nc == 256.  Workgroup = 4 waves over a panel of 64 rows; wave wn owns all 64 rows and columns wn*64 .. +64
// (2 x 2 accumulators of 32 x 32) for the whole K; the panel is re-filled per K-slice of 256.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
struct PgemmBwd {
  static constexpr int BM = 64;
  static constexpr int NT = 256;
  static constexpr int LDS_FLOATS = BM * 256;

  struct Acc {
    f32x16 a00, a01, a10, a11;  // a<row block><column block>
  };
  static __device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a.a00[r] = a.a01[r] = a.a10[r] = a.a11[r] = 0.f;
  }
  struct Ring {
    f32x4 rb0[D], rb1[D];
  };
  // the first D blocks of the wave's two B streams (slice_idx = first slice): requested BEFORE the first panel
  // barrier; the ring then runs on across the slices (the stream of a 32-column block is contiguous over ALL of K)
  static __device__ __forceinline__ void prefetch(Ring& R, const float* __restrict__ Bp, int K, int slice_idx,
                                                  int wave, int lane) {
    const size_t nbstride = (size_t)(K >> 3) * 256;
    const char* __restrict__ b0 =
        reinterpret_cast<const char*>(Bp + (size_t)(2 * wave) * nbstride + (size_t)slice_idx * 32 * 256);
    const char* __restrict__ b1 = b0 + nbstride * 4;
    const unsigned lo = (unsigned)lane * 16u;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      R.rb0[d] = *reinterpret_cast<const f32x4*>(b0 + lo + d * 1024);
      R.rb1[d] = *reinterpret_cast<const f32x4*>(b1 + lo + d * 1024);
    }
  }
  static __device__ __forceinline__ void pin(Ring& R) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      asm volatile("" : "+v"(R.rb0[d]));
      asm volatile("" : "+v"(R.rb1[d]));
    }
  }
  // one K-slice (32 k-steps) from the panel in LDS; Bp = packed weights [256, K], slice_idx = the 256-wide slice of K
  static __device__ __forceinline__ void slice(Acc& a, Ring& R, const float* __restrict__ smem,
                                               const float* __restrict__ Bp, int K, int slice_idx, int wave, int lane) {
    int fo[8];
    panel_frag_offsets(lane, fo);
    const size_t nbstride = (size_t)(K >> 3) * 256;  // floats between consecutive 32-column blocks
    const char* __restrict__ b0 =
        reinterpret_cast<const char*>(Bp + (size_t)(2 * wave) * nbstride + (size_t)slice_idx * 32 * 256);
    const char* __restrict__ b1 = b0 + nbstride * 4;
    const unsigned lo = (unsigned)lane * 16u;
    f32x4(&rb0)[D] = R.rb0;
    f32x4(&rb1)[D] = R.rb1;
    f32x4 fa0 = *reinterpret_cast<const f32x4*>(smem + fo[0]);
    f32x4 fa1 = *reinterpret_cast<const f32x4*>(smem + fo[0] + 32 * 256);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const int kn = (ks + 1) & 31;  // (the wrap-around read after the last step is unused)
      const f32x4 na0 = *reinterpret_cast<const f32x4*>(smem + fo[kn & 7] + (kn >> 3) * 64);
      const f32x4 na1 = *reinterpret_cast<const f32x4*>(smem + fo[kn & 7] + (kn >> 3) * 64 + 32 * 256);
      const f32x4 fb0 = rb0[ks % D], fb1 = rb1[ks % D];
      // the tail of a slice already requests the head of the next one (the last slice runs D blocks past its
      // columns' range: the packed buffer is padded)
      rb0[ks % D] = *reinterpret_cast<const f32x4*>(b0 + lo + (ks + D) * 1024);
      rb1[ks % D] = *reinterpret_cast<const f32x4*>(b1 + lo + (ks + D) * 1024);
#define VSN_STEP16(T)                                                              \
  a.a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0.T, fa0.T, a.a00, 0, 0, 0);      \
  a.a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb1.T, fa0.T, a.a01, 0, 0, 0);      \
  a.a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0.T, fa1.T, a.a10, 0, 0, 0);      \
  a.a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb1.T, fa1.T, a.a11, 0, 0, 0);
      VSN_STEP16(x)
      VSN_STEP16(y)
      VSN_STEP16(z)
      VSN_STEP16(w)
#undef VSN_STEP16
      fa0 = na0;
      fa1 = na1;
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read x2
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read x2
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // MFMA x16
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // EPI: 0 = store, 2 = accumulate into C
  template <int EPI>
  static __device__ __forceinline__ void store(const Acc& a, float* __restrict__ C, int ldc, int row0, int Meff,
                                               int wave, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    int rows = Meff - row0;
    rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(uni_ptr(C + (size_t)row0 * ldc), 0, uni(rows * ldc * 4), 0x00020000);
    const int ldo = ldc * 4, cb = (wave * 64 + 4 * hi) * 4;
    const f32x4 nob[4] = {};
    panel_store_acc<EPI>(a.a00, rs, l31 * ldo + cb, nob);
    panel_store_acc<EPI>(a.a01, rs, l31 * ldo + cb + 128, nob);
    panel_store_acc<EPI>(a.a10, rs, (32 + l31) * ldo + cb, nob);
    panel_store_acc<EPI>(a.a11, rs, (32 + l31) * ldo + cb + 128, nob);
  }
};

// ---- plain kernels: the panel is a copy of A rows (LDS-DMA) -------------------------------------------------
template <int MI, int WN, int D, int EPI = 0, int MINW = 2, int ABL = 0>
__global__ __launch_bounds__(64 * WN, MINW) void k_pgemm_fwd(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ Bp, float* __restrict__ C,
                                                             int ldc, const float* __restrict__ bias, int M,
                                                             const int* __restrict__ Mptr, int Nc, int nsplit) {
  typedef PgemmFwd<MI, WN, D, ABL> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int Meff = M;
  if (Mptr) {
    const int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int live = ((Meff + G::BM - 1) / G::BM) * nsplit;
  if ((int)blockIdx.x >= live) return;
  const int bid = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int p = bid / nsplit, sp = bid % nsplit;
  const int nct = Nc / G::BN;
  const int ct0 = nct * sp / nsplit, ct1 = nct * (sp + 1) / nsplit;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  panel_load_dma<G::BM, WN>(smem, A, lda, p * G::BM, Meff, 0, wave, lane);
  typename G::Ring ring;
  G::prefetch(ring, Bp, ct0, wave, lane);
  __syncthreads();  // (drains the LDS-DMA queue: vmcnt(0) + barrier)
  G::pin(ring);
  G::template compute<EPI>(ring, smem, Bp, C, ldc, bias, p * G::BM, Meff, ct0, ct1, wave, lane);
}

template <int D, int EPI = 0>
__global__ __launch_bounds__(256, 2) void k_pgemm_bwd(const float* __restrict__ A, int lda,
                                                      const float* __restrict__ Bp, float* __restrict__ C, int ldc,
                                                      int M, const int* __restrict__ Mptr, int K) {
  typedef PgemmBwd<D> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int Meff = M;
  if (Mptr) {
    const int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int live = (Meff + G::BM - 1) / G::BM;
  if ((int)blockIdx.x >= live) return;
  const int p = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  typename G::Acc acc;
  G::zero(acc);
  typename G::Ring ring;
  G::prefetch(ring, Bp, K, 0, wave, lane);
  const int nsl = K >> 8;
  for (int s = 0; s < nsl; ++s) {
    if (s) __syncthreads();  // every wave is done reading the previous slice
    panel_load_dma<G::BM, 4>(smem, A, lda, p * G::BM, Meff, s * 256, wave, lane);
    __syncthreads();
    G::pin(ring);
    G::slice(acc, ring, smem, Bp, K, s, wave, lane);
  }
  G::template store<EPI>(acc, C, ldc, p * G::BM, Meff, wave, lane);
}

}  // namespace vsn
