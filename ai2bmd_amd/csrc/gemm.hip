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
// LDS rows are 32 floats (128 B, no padding) with the eight 16-byte chunks of a
// row XOR-swizzled by (row >> 1) & 7, so the ds_read_b128 fragment reads of a
// 16-lane service group (16 consecutive rows, one logical chunk) land in 16
// distinct 16-byte slots.  Unpadded, the double-buffered 64x64 tile takes
// exactly 32 KiB (padded rows: 36 KiB).  That is still FOUR workgroups per CU,
// not five: LDS is allocated in 1 280-byte steps (HW_REG_LDS_ALLOC reads 130
// 256-byte granules for 32 768 bytes), so five would need <= 32 000 bytes each.
// Each lane reads 4 consecutive k of its row: lanes 0-31 take k0..k0+3, lanes
// 32-63 take k0+4..k0+7; MFMA #t pairs (k0+t, k0+4+t) for A and B alike, so
// the k-sum is just reordered.
#include <cstdlib>
#include <map>
#include <type_traits>

#include "common.h"
#include "kernels.h"

#ifndef VSN_LAB_TRACE
#define VSN_LAB_TRACE 0  // tools/lab/gemm_direct.hip: per-wave phase timestamps of the k-loop (1 = on; lab builds only)
#endif
#ifndef VSN_LAB_PRIO
#define VSN_LAB_PRIO 0  // tools/lab/gemm_direct.hip: s_setprio placement experiments (0 = none, the product build)
#endif

namespace vsn {

// SILU: apply the activation to A on its way into LDS (one product of the read-out head); a template parameter so
// that the ~250 VALU instructions of the exact sigmoid are not sitting in the k-loop of every other product
// (1 = silu, 2 = any kind of the reference's table, carried in flags bits 8..).
//
// What the per-wave phase trace of the 64x64 kernel (tools/lab/gemm_direct.hip, -DVSN_LAB_TRACE=1) showed and this
// body is built around: with K = 256 a tile is only eight k-iterations long, so everything OUTSIDE the k-loop
// (prologue 6.6 k cycles, epilogue 7.7 k, workgroup turnover 2.5 k, against 30 k in the loop) decides how many of a
// SIMD's wave slots are inside an MFMA block at any time - the pipe saturates with three, and it had 2.3 of 4.
//  * swizzled, unpadded LDS rows (32 KiB per workgroup; the fifth resident workgroup needs <= 32 000 B, see above);
//  * the epilogue is ~100 instructions (uniform tile base + 32-bit lane offsets, no per-row predicates on whole
//    tiles, accumulate-mode products start their accumulator FROM C instead of adding C at the end);
//  * fragment reads are double-buffered over the four k-steps of a tile and the LDS stores of the next tile sit in
//    the middle of the MFMA block, so a lone wave (small launches, tails) does not expose their latency either;
//  * global operands are addressed as uniform base + 32-bit lane offset: no 64-bit VALU arithmetic in the loop.
template <int BM, int BN, int WM, int WN, bool DB, int SILU = 0, int BK = 32>
__device__ __forceinline__ void gemm_body(const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                          int ldb, float* __restrict__ C, int ldc,
                                          const float* __restrict__ bias, int M, const int* __restrict__ Mptr,
                                          int Nc, int K, int flags, int ksplit, float* __restrict__ part,
                                          int block_id, float* __restrict__ smem) {
  constexpr bool SWZ = BK == 32;         // XOR-swizzled rows (BK = 32, every product instantiation) or padded rows (lab)
  constexpr int LS = SWZ ? BK : BK + 4;  // LDS row stride (floats)
  constexpr int C4 = BK / 4;             // 16-byte chunks per tile row
  constexpr int KK = BK / 8;             // MFMA k-steps per tile (8 k each: lanes 0-31 take k..k+3, lanes 32-63 k+4..k+7)
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int NT = WM * WN * 64;                       // threads per workgroup (256 in every product instantiation)
  constexpr int LA = BM * C4 / NT, LB = BN * C4 / NT;    // 16-byte loads per thread per k-tile
  static_assert(LA * NT == BM * C4 && LB * NT == BN * C4, "tile rows must divide over the workgroup");
  constexpr int STAGE = (BM + BN) * LS;
  // LDS float offset of logical chunk c4 of tile row r
#define VSN_LDS_AT(r, c4) ((r) * LS + (SWZ ? (((c4) ^ (((r) >> 1) & 7)) * 4) : (c4) * 4))

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
#if VSN_LAB_TRACE
  // lab: lane 0 of every wave stamps its phases straight into `part` ([block][wave][64] dwords; ksplit must be 1)
  unsigned* const trace = reinterpret_cast<unsigned*>(part) + ((size_t)block_id * 4 + wave) * 64;
  int trn = 2;
#define VSN_STAMP_AT(slot)                                                                  \
  do {                                                                                      \
    if (part && lane == 0) trace[slot] = (unsigned)__builtin_amdgcn_s_memtime();            \
  } while (0)
#define VSN_STAMP()                    \
  do {                                 \
    if (trn < 62) VSN_STAMP_AT(trn);   \
    ++trn;                             \
  } while (0)
  if (part && lane == 0) trace[0] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // HW_ID[15:0]
  if (part && lane == 0) trace[61] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);  // LDS_ALLOC
  VSN_STAMP_AT(62);  // kernel entry
#else
#define VSN_STAMP_AT(slot)
#define VSN_STAMP()
#endif
  constexpr bool silu_a = SILU != 0;  // 1: silu (folds to the branch-free form), 2: kind from flags bits 8..
  const int akind = SILU == 2 ? (flags >> 8) : VSN_ACT_SILU;
  // this workgroup's K range (split-K: partial sums go to `part`, reduced by k_gemm_reduce or by the consumer)
  const int nkt_all = K / BK;
  const int kt0 = (int)((long long)nkt_all * ks / ksplit), kt1 = (int)((long long)nkt_all * (ks + 1) / ksplit);
  const int nkt = kt1 - kt0;

  // Global operands: one uniform base per tile, 32-bit lane offsets (rows >= Meff are clamped: never stored).
  const float* __restrict__ Ab = A + (size_t)row0 * lda + (size_t)kt0 * BK;
  const float* __restrict__ Bb = Bt + (size_t)col0 * ldb + (size_t)kt0 * BK;
  unsigned aoff[LA], boff[LB];
  int lsa[LA], lsb[LB];  // where the lane's chunks go in an LDS stage
#pragma unroll
  for (int it = 0; it < LA; ++it) {
    const int f = tid + it * NT, r = f / C4, c4 = f % C4;
    const int rr = row0 + r < Meff ? r : Meff - 1 - row0;
    aoff[it] = (unsigned)rr * (unsigned)lda + (unsigned)(c4 * 4);
    lsa[it] = VSN_LDS_AT(r, c4);
  }
#pragma unroll
  for (int it = 0; it < LB; ++it) {
    const int f = tid + it * NT, r = f / C4, c4 = f % C4;
    boff[it] = (unsigned)r * (unsigned)ldb + (unsigned)(c4 * 4);
    lsb[it] = BM * LS + VSN_LDS_AT(r, c4);
  }
  f32x4 rra[LA], rrb[LB];  // the k-tile in flight from global memory
#define VSN_GLOAD(ktile)                                                                          \
  {                                                                                               \
    const float* __restrict__ Ak_ = Ab + (ktile) * BK;                                            \
    const float* __restrict__ Bk_ = Bb + (ktile) * BK;                                            \
    _Pragma("unroll") for (int it = 0; it < LA; ++it) rra[it] = *reinterpret_cast<const f32x4*>(Ak_ + aoff[it]); \
    _Pragma("unroll") for (int it = 0; it < LB; ++it) rrb[it] = *reinterpret_cast<const f32x4*>(Bk_ + boff[it]); \
  }
#define VSN_SSTORE(stage)                                                      \
  {                                                                            \
    float* St_ = smem + (stage) * STAGE;                                       \
    _Pragma("unroll") for (int it = 0; it < LA; ++it) {                        \
      f32x4 v_ = rra[it];                                                      \
      if (silu_a) { /* activation kind (VSN_ACT_*) rides in flags bits 8.. */   \
        v_.x = act_f(akind, v_.x);                                             \
        v_.y = act_f(akind, v_.y);                                             \
        v_.z = act_f(akind, v_.z);                                             \
        v_.w = act_f(akind, v_.w);                                             \
      }                                                                        \
      *reinterpret_cast<f32x4*>(St_ + lsa[it]) = v_;                           \
    }                                                                          \
    _Pragma("unroll") for (int it = 0; it < LB; ++it) *reinterpret_cast<f32x4*>(St_ + lsb[it]) = rrb[it]; \
  }

  // Fragment addresses: row (wm*TM + i*32 + l31) of A, (wn*TN + j*32 + l31) of B; both have the same (row >> 1) & 7,
  // so the swizzled in-row offset of k-step kk is shared.
  int fo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) fo[kk] = VSN_LDS_AT(l31, kk * 2 + hi) - l31 * LS;
  const int fra = (wm * TM + l31) * LS, frb = BM * LS + (wn * TN + l31) * LS;

  constexpr bool PREC = (MI * NI == 1);  // one-accumulator tile (64x64, the grouped reverse-pass products)
  const bool accum = (flags & 1) != 0;
  const bool acc_out = accum && ksplit == 1;
  // MFMA operand roles are SWAPPED (weights as its "A", activations as its "B": D[n][m] = sum_k W[n][k] X[m][k]), so a
  // lane ends up with ONE output row (m = l31) and 16 output columns in four runs of four consecutive ones
  // (col = 8 q + 4 hi + 0..3): the epilogue is 4 x 16-byte stores per accumulator instead of 16 x 4-byte ones (and the
  // accumulate-from-C prologue 4 x 16-byte loads).  Same products in the same k order: bitwise the same sums.
  // bias: the one-accumulator tile (K = 256: eight k-iterations, the tail matters) fetches its sixteen values now -
  // at the end of the tile they would be a full round trip on the tail; the 128-wide tiles would pay 32 registers
  // across the k-loop for that (a wave per SIMD less) and fetch them in the epilogue instead
  constexpr bool BIAS_EARLY = (MI * NI == 1);
  const bool has_bias = bias && ksplit == 1;
  f32x4 bvs[BIAS_EARLY ? 4 : 1];
  if (BIAS_EARLY) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bvs[q] = has_bias ? *reinterpret_cast<const f32x4*>(bias + col0 + wn * TN + 8 * q + 4 * hi)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 acc[MI][NI];
  if (PREC && acc_out) {
    // accumulate mode: the accumulator STARTS from the old C values (no second pass over C in the epilogue, and
    // no sixteen registers holding them across the k-loop)
    const float* cp = C + (size_t)row0 * ldc + col0;
    const bool row_ok = row0 + wm * TM + l31 < Meff;
    const unsigned off = (unsigned)(wm * TM + l31) * (unsigned)ldc + (unsigned)(wn * TN + 4 * hi);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 o_ = row_ok ? *reinterpret_cast<const f32x4*>(cp + off + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[0][0][4 * q] = o_.x, acc[0][0][4 * q + 1] = o_.y, acc[0][0][4 * q + 2] = o_.z, acc[0][0][4 * q + 3] = o_.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  VSN_GLOAD(0);
  if (DB) {
    // software pipeline: LDS stage (kt & 1) holds tile kt, registers hold tile kt+1, one barrier per tile
    VSN_SSTORE(0);
    if (1 < nkt) VSN_GLOAD(1);
    __syncthreads();
  }
  VSN_STAMP_AT(1);  // k-loop starts

#define VSN_FRAG(set, stage_, kk_)                                                                            \
  {                                                                                                           \
    const float* Sr_ = smem + (stage_) * STAGE + fo[kk_];                                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(Sr_ + fra + i * 32 * LS); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(Sr_ + frb + j * 32 * LS); \
  }
  // one k-tile; ST (the LDS stage that holds it) is a compile-time constant so that every LDS address is a register
  // plus an immediate
  auto ktile = [&](auto stc, const int kt) __attribute__((always_inline)) {
    constexpr int st = decltype(stc)::value;
    VSN_STAMP();  // top of the iteration
    if (!DB) {
      VSN_SSTORE(0);
      __syncthreads();
      if (kt + 1 < nkt) VSN_GLOAD(kt + 1);
      // keep the prefetch HERE: without this hipcc sinks the loads below the MFMA block (to shorten their
      // live ranges), which exposes the full global-load latency once per k-tile
      __builtin_amdgcn_sched_barrier(0);
    }
#if VSN_LAB_PRIO == 2
    __builtin_amdgcn_s_setprio(2);  // lab: favour waves inside their MFMA block
#endif
    f32x4 fa[2][MI], fb[2][NI];
    VSN_FRAG(0, st, 0);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) VSN_FRAG((kk + 1) & 1, st, kk + 1);
      const int cur = kk & 1;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j].x, fa[cur][i].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j].y, fa[cur][i].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j].z, fa[cur][i].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j].w, fa[cur][i].w, acc[i][j], 0, 0, 0);
        }
      if (DB && kk == KK / 2 - 1) {
        // Half-way through the MFMA block: tile kt+1 (in registers since the last iteration) goes to the other
        // stage - last read in iteration kt-1, a barrier ago - and the loads of tile kt+2 start.  The stores
        // complete under the second half of the block instead of in front of the barrier.
        __builtin_amdgcn_sched_barrier(0);
        VSN_STAMP();
        if (kt + 1 < nkt) {
          VSN_SSTORE(st ^ 1);
          if (kt + 2 < nkt) VSN_GLOAD(kt + 2);
        }
        __builtin_amdgcn_sched_barrier(0);  // and the prefetch is issued here, not sunk below the MFMAs
      }
    }
    VSN_STAMP();  // MFMA block issued
#if VSN_LAB_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#elif VSN_LAB_PRIO == 1
    __builtin_amdgcn_s_setprio(3);  // lab: favour waves in their barrier phase
#endif
    __syncthreads();
    VSN_STAMP();  // past the barrier
#if VSN_LAB_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  {
    int kt = 0;
    if (DB) {
      for (; kt + 1 < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, kt);
        ktile(std::integral_constant<int, 1>{}, kt + 1);
      }
      if (kt < nkt) ktile(std::integral_constant<int, 0>{}, kt);
    } else {
      for (; kt < nkt; ++kt) ktile(std::integral_constant<int, 0>{}, kt);
    }
  }

  // Epilogue.  It runs while the other workgroups of the CU are inside their MFMA blocks and every instruction of it
  // competes with them for issue slots: one uniform base pointer per tile, 32-bit lane offsets, and no per-row
  // predicates on a tile that lies wholly below Meff.
  float* __restrict__ Ct = ksplit == 1 ? C + (size_t)row0 * ldc + col0 : part + ((size_t)ks * M + row0) * Nc + col0;
  const unsigned ldo = (unsigned)(ksplit == 1 ? ldc : Nc);
  const bool rmw = acc_out && !PREC;  // multi-accumulator tiles add the old C here
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const bool row_ok = row0 + wm * TM + i * 32 + l31 < Meff;  // this lane's output row
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float* cp = Ct + ((unsigned)(wm * TM + i * 32 + l31) * ldo + (unsigned)(wn * TN + j * 32 + 4 * hi));
      if (row_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          if (BIAS_EARLY) v += bvs[q];
          else if (has_bias) v += *reinterpret_cast<const f32x4*>(bias + col0 + wn * TN + j * 32 + 8 * q + 4 * hi);
          if (rmw) v += *reinterpret_cast<const f32x4*>(cp + 8 * q);
          *reinterpret_cast<f32x4*>(cp + 8 * q) = v;
        }
      }
    }
  }
  VSN_STAMP_AT(63);  // epilogue stores issued
#undef VSN_GLOAD
#undef VSN_SSTORE
#undef VSN_FRAG
#undef VSN_STAMP
#undef VSN_STAMP_AT
#undef VSN_LDS_AT
}

#include "gemm_s3.h"  // opt-in 3 x bf16 split form of the grouped 64 x 64 tile (gemm_body3, Split3Table)

template <int BM, int BN, int WM, int WN, bool DB, int SILU = 0, int BK = 32>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm(const float* __restrict__ A, int lda,
                                              const float* __restrict__ Bt, int ldb,
                                              float* __restrict__ C, int ldc,
                                              const float* __restrict__ bias, int M,
                                              const int* __restrict__ Mptr, int Nc, int K, int flags,
                                              int ksplit, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * (BM + BN) * (BK == 32 ? BK : BK + 4)];
  gemm_body<BM, BN, WM, WN, DB, SILU, BK>(A, lda, Bt, ldb, C, ldc, bias, M, Mptr, Nc, K, flags, ksplit, part,
                                      (int)blockIdx.x, smem);
}

// Several independent products in ONE launch (64x64 tiles): block -> (problem, tile).  Used for the
// per-layer groups {qkv, vector projections, edge linears}, {s_proj, o_proj}, {dX products}: on a
// single-protein MD step the small members (N = a few hundred rows) cannot fill 256 CUs alone.
__global__ __launch_bounds__(256) void k_gemm_group(GemmGroup g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 64) * 32];  // 32 KiB
  int b = (int)blockIdx.x, p = 0;
#pragma unroll
  for (int q = 0; q < GemmGroup::MAXP - 1; ++q)
    if (p == q && q + 1 < g.n && b >= g.p[q].blocks) {
      b -= g.p[q].blocks;
      p = q + 1;
    }
  const GemmDesc& d = g.p[p];
  gemm_body<64, 64, 2, 2, true, 0, 32>(d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K,
                                           d.flags, d.ksplit, d.part, b, smem);
}
// the grouped launch of the opt-in mode gemm_split3: members marked VSN_S3_FLAG carry packed bf16 planes as their
// weight operand and run the split tile, the others the fp32 tile.  A kernel of its own, so that the
// default fp32 launch above is the same binary whether or not the mode exists.
#ifndef VSN_S3_MINWAVES
#define VSN_S3_MINWAVES 4  // waves per SIMD asked of the compiler (A/B builds: tools/build_variant.py -DVSN_S3_MINWAVES=5)
#endif
__global__ __launch_bounds__(256, VSN_S3_MINWAVES) void k_gemm_group_s3(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 24 KiB (split tiles only) or 32 KiB (an fp32 member)
  int b = (int)blockIdx.x, p = 0;
#pragma unroll
  for (int q = 0; q < GemmGroup::MAXP - 1; ++q)
    if (p == q && q + 1 < g.n && b >= g.p[q].blocks) {
      b -= g.p[q].blocks;
      p = q + 1;
    }
  const GemmDesc& d = g.p[p];
  if (d.flags & VSN_S3_FLAG)
    gemm_body3(d.A, d.lda, reinterpret_cast<const unsigned short*>(d.Bt), d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr,
               d.Nc, d.K, d.flags, d.ksplit, d.part, b, smem);
  else
    gemm_body<64, 64, 2, 2, true, 0, 32>(d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K,
                                         d.flags, d.ksplit, d.part, b, smem);
}

// split-K epilogue: C (+)= sum_s part[s] (+ bias), fixed summation order
__device__ __forceinline__ void gemm_reduce_body(const float* __restrict__ part, int ksplit, float* __restrict__ C,
                                                 int ldc, const float* __restrict__ bias, int M,
                                                 const int* __restrict__ Mptr, int Nc, int flags, long long gid) {
  int Meff = M;
  if (Mptr) {
    int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int nc4 = Nc >> 2;
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
__global__ void k_gemm_reduce(const float* __restrict__ part, int ksplit, float* __restrict__ C, int ldc,
                              const float* __restrict__ bias, int M, const int* __restrict__ Mptr, int Nc,
                              int flags) {
  gemm_reduce_body(part, ksplit, C, ldc, bias, M, Mptr, Nc, flags, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
// the split members of one grouped launch, summed in ONE launch (`blocks` = reduction workgroups of the member)
__global__ void k_gemm_reduce_group(GemmGroup g) {
  int b = (int)blockIdx.x, p = 0;
#pragma unroll
  for (int q = 0; q < GemmGroup::MAXP - 1; ++q)
    if (p == q && q + 1 < g.n && b >= g.p[q].blocks) {
      b -= g.p[q].blocks;
      p = q + 1;
    }
  const GemmDesc& d = g.p[p];
  gemm_reduce_body(d.part, d.ksplit, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.flags,
                   (long long)b * blockDim.x + threadIdx.x);
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream ----
int g_splitk_tiles = 400;   // split-K only below this many output tiles (env VSN_SPLITK_TILES; swept on Chignolin: 0:297, 200:306, 400:306, 768:301, 1200:289 steps/s)
int g_splitk_mink = 512;  // plain launches: split-K from this K on (env VSN_SPLITK_MINK, A/B aid)
int g_gemm_force = 0;     // micro-benchmark aid (env VSN_GEMM_FORCE): force an experimental tile variant
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
  e = getenv("VSN_SPLITK_MINK");
  if (e) g_splitk_mink = atoi(e);
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
  if ((K & 31) || (Nc & 31) || (lda & 3) || (ldb & 3) || (ldc & 3)) return -22;  // (16-byte row pieces everywhere)
  // the epilogue / accumulate prologue move C and bias as f32x4: 16-byte aligned bases
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bt) | reinterpret_cast<uintptr_t>(C) |
       reinterpret_cast<uintptr_t>(bias)) & 15)
    return -22;
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
    VSN_TRY(17, 256, 256, 2, 2, false)
    VSN_TRY(18, 256, 128, 2, 2, false)
    VSN_TRY(19, 128, 256, 2, 2, false)
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
  if (tiles < g_splitk_tiles && K >= g_splitk_mink && tl_splitk_ws && (ldc & 3) == 0) {
    ks = (1024 + tiles - 1) / tiles;
    const int kmax = K / 128;  // keep >= 4 k-tiles per split
    if (ks > kmax) ks = kmax;
    if (ks > 8) ks = 8;
    while (ks > 1 && (size_t)ks * M * Nc > tl_splitk_elems) --ks;
    if (ks < 1) ks = 1;
  }
  float* part = ks > 1 ? tl_splitk_ws : nullptr;
  const int grid = tiles * ks;
  if (variant == 0 && ks == 1 && tl_s3) {  // opt-in mode gemm_split3: the batch tile as 3 x bf16 split products
    if (const unsigned short* pl = s3_planes(Bt, ldb, Nc, K, flags, st)) {
      hipLaunchKernelGGL(k_gemm3_128, dim3(grid), dim3(256), 0, st, A, lda, pl, ldb, C, ldc, bias, M, Mptr, Nc, K,
                         flags);
      return 0;
    }
  }
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

// would launch_gemm_group run these products as ONE grouped launch (the only form that can leave K-slices to a consumer)?
bool gemm_group_ok(const GemmDesc* descs, int n) {
  static const bool env_init = gemm_env_init();
  (void)env_init;
  if (!(n > 1 && n <= GemmGroup::MAXP)) return false;
  for (int i = 0; i < n; ++i) {
    const GemmDesc& d = descs[i];
    if (d.M <= 0) continue;
    if ((d.K & 31) || (d.Nc & 63) || (d.lda & 3) || (d.ldb & 3) || (d.ldc & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.Bt) | reinterpret_cast<uintptr_t>(d.C) |
         reinterpret_cast<uintptr_t>(d.bias)) & 15)
      return false;  // (f32x4 epilogue: falls back to launch_gemm, which reports -22)
    if (gemm_variant(d.M, d.Nc) == 0) return false;  // big enough to fill the chip alone
    if (d.flags & 2) return false;                   // silu(A) has its own kernels
  }
  return true;
}

int launch_gemm_group(hipStream_t st, const GemmDesc* descs, int n) {
  static const bool env_init = gemm_env_init();
  (void)env_init;
  // fall back to individual launches when the group would not use the 64x64 tiling anyway
  const bool groupable = gemm_group_ok(descs, n);
  long long tiles = 0;
  for (int i = 0; i < n && groupable; ++i) {
    const GemmDesc& d = descs[i];
    if (d.M <= 0) continue;
    if (d.keep_parts > 1 && (!d.part || (d.K / 32) < d.keep_parts || (d.flags & 1) || d.bias)) return -22;
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
  bool any_s3 = false;
  unsigned lds_s3 = 24576;
  for (int i = 0; i < n; ++i) {
    GemmDesc d = descs[i];
    if (d.M <= 0) continue;
    s3_patch(d, st);  // (opt-in mode gemm_split3)
    any_s3 |= (d.flags & VSN_S3_FLAG) != 0;
    if (!(d.flags & VSN_S3_FLAG)) lds_s3 = 32768;  // an fp32 member in the split launch: its tile needs 32 KiB
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
    if (any_s3) hipLaunchKernelGGL(k_gemm_group_s3, dim3(grid), dim3(256), lds_s3, st, g);
    else hipLaunchKernelGGL(k_gemm_group, dim3(grid), dim3(256), 0, st, g);
  }
  if (nred == 1) {
    const GemmDesc& d = red[0];
    long long n4 = (long long)d.M * (d.Nc / 4);
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, d.part, d.ksplit, d.C,
                       d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.flags);
  } else if (nred > 1) {
    GemmGroup r;
    r.n = nred;
    unsigned rgrid = 0;
    for (int i = 0; i < nred; ++i) {
      r.p[i] = red[i];
      r.p[i].blocks = (int)(((long long)red[i].M * (red[i].Nc / 4) + 255) / 256);
      rgrid += (unsigned)r.p[i].blocks;
    }
    hipLaunchKernelGGL(k_gemm_reduce_group, dim3(rgrid), dim3(256), 0, st, r);
  }
  if (rec) hipEventRecord(rec->b, st);
  return 0;
}

}  // namespace vsn
