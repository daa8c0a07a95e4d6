// Opt-in product mode `gemm_split3` (vsn_set_option; default OFF - the default arithmetic of every product is the
// fp32 MFMA chain of gemm.hip): the 64 x 64 tile of the grouped launches with both fp32 operands split into three
// bf16 terms, x = hi + mid + lo (8 + 8 + 8 mantissa bits), and six v_mfma_f32_32x32x16_bf16 products per 16-k block,
//     hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi        (smallest terms first, fp32 accumulate),
// i.e. every product term above 2^-24 relative: measured rms error 2.4e-7 of rms(C) against 2.9e-7 for the fp32 MFMA
// chain on the same operands (DESIGN.md "3 x bf16 split"; the bf16 products are exact in the fp32 accumulator).
// The matrix-pipe time of a product falls to 6/16 of the fp32 form.
//
// Same tile numbering, split-K slices, accumulate-from-C and epilogue as gemm_body<64, 64, 2, 2, true>, so it drops
// into k_gemm_group behind a flag bit of the member.  The WEIGHT planes are made once per weight matrix
// (k_s3_pack, cached per engine in a Split3Table), packed in MFMA-fragment order: block (32 columns, 16 k, plane)
// = 1 KiB = 64 lanes x 8 bf16, so a wave's B fragment is ONE coalesced 1-KiB load from L2 straight into registers,
// one k-tile ahead.  The ACTIVATIONS are split on their way into LDS, which holds the three A planes only (two
// stages of 12 KiB, one barrier per k-tile).
//
// Measured and left out (round 4, LAB_NOTES section 11): 128-row tiles in the grouped launch (a wave owning two MFMA
// tiles that share its B fragment: a third fewer bytes through the L1 per product, but 146 registers = three waves per
// SIMD and half the tiles - Chignolin 488 -> 469 steps/s with every member tall, 490 with the E-row members only).
//
// (spliced into gemm.hip INSIDE namespace vsn, after gemm_body)
#pragma once

typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
#define VSN_S3_FLAG (1 << 30)

__device__ __forceinline__ void s3_split8(const f32x4 x0, const f32x4 x1, s3_bf16x8& hi, s3_bf16x8& mid, s3_bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = t < 4 ? x0[t] : x1[t - 4];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    hi[t] = h;
    mid[t] = m;
    lo[t] = (__bf16)(r1 - (float)m);
  }
}

// byte offset of 16-byte chunk c (8 bf16 k-values) of row r in a 64-row x 64-byte plane (XOR swizzle: the
// ds_read_b128 of 16 consecutive rows land in distinct 16-byte slots)
__device__ __forceinline__ int s3_at(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 3)) << 4); }

// element index of (column n, k, plane p) in the packed planes of a [Nc][ldb] weight matrix
__host__ __device__ __forceinline__ size_t s3_pack(size_t n, size_t k, int p, size_t ldb) {
  return ((((n >> 5) * (ldb >> 4) + (k >> 4)) * 3 + p) * 64 + (n & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7);
}

__device__ __forceinline__ void gemm_body3(const float* __restrict__ A, int lda, const unsigned short* __restrict__ B3,
                                           int ldb, float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                           int M, const int* __restrict__ Mptr, int Nc, int K, int flags, int ksplit,
                                           float* __restrict__ part, int block_id, float* __restrict__ smem_f) {
  constexpr int BM = 64, BN = 64, BK = 32;
  constexpr int PLANE = 64 * 64, STAGE = 3 * PLANE;  // bytes
  unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_f);
  int Meff = M;
  if (Mptr) {
    int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int tiles_n = Nc / BN;
  const int live = ((Meff + BM - 1) / BM) * tiles_n * ksplit;
  if (block_id >= live) return;
  const int bid = VSN_XCD_REMAP ? xcd_block(block_id, live) : block_id;
  const int tile = bid / ksplit, ks = bid % ksplit;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nkt_all = K / BK;
  const int kt0 = (int)((long long)nkt_all * ks / ksplit), kt1 = (int)((long long)nkt_all * (ks + 1) / ksplit);
  const int nkt = kt1 - kt0;
  // staging role: tile row sr, 8-k chunk sj
  const int sr = tid >> 2, sj = tid & 3;
  const int ar = row0 + sr < Meff ? sr : Meff - 1 - row0;  // rows >= Meff are clamped: never stored
  const float* __restrict__ ag = A + (size_t)(row0 + ar) * lda + (size_t)kt0 * BK + sj * 8;
  const int soff = s3_at(sr, sj);
  // this wave's 32-column block, its first 16-k block: fragments follow at 3 KiB per 16-k block
  const unsigned short* __restrict__ bw =
      B3 + ((((size_t)((col0 >> 5) + wn) * (size_t)(ldb >> 4) + (size_t)kt0 * 2) * 3) * 64 + lane) * 8;
  f32x4 ra0, ra1;
  auto gloadA = [&](int kt) {
    ra0 = *reinterpret_cast<const f32x4*>(ag + kt * BK);
    ra1 = *reinterpret_cast<const f32x4*>(ag + kt * BK + 4);
  };
  auto sstoreA = [&](int stage) {
    unsigned char* base = smem + stage * STAGE + soff;
    s3_bf16x8 h, m, l;
    s3_split8(ra0, ra1, h, m, l);
    *reinterpret_cast<s3_bf16x8*>(base) = h;
    *reinterpret_cast<s3_bf16x8*>(base + PLANE) = m;
    *reinterpret_cast<s3_bf16x8*>(base + 2 * PLANE) = l;
  };
  // B fragments: a register ring, the loop below unrolled over its positions (no copies).  VSN_S3_BRING = 1 (default):
  // two slots, k-tile kt + 1 requested when kt starts.  = 2: three slots, kt + 2 requested when kt starts - the review's
  // "B ring two k-tiles ahead"; measured (round 5): 24 more registers push the tile to 128 VGPRs + 16 spills at four
  // workgroups per CU, k_gemm_group_s3 28.8 -> 39.8 us, Chignolin 471 -> 396 steps/s in the mode.  Rejected.
#ifndef VSN_S3_BRING
#define VSN_S3_BRING 1
#endif
  typedef s3_bf16x8 BFrag[2][3];
  BFrag b0, b1, b2;
  auto gloadB = [&](int kt, BFrag& dst) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dst[kc][p] = *reinterpret_cast<const s3_bf16x8*>(bw + ((size_t)(kt * 2 + kc) * 3 + p) * 512);
  };
  const bool accum = (flags & 1) != 0;
  const bool acc_out = accum && ksplit == 1;
  const float bv = (bias && ksplit == 1) ? bias[col0 + wn * 32 + l31] : 0.f;
  f32x16 acc;
  if (acc_out) {  // accumulate mode: the accumulator starts from the old C values
    const float* cp = C + (size_t)row0 * ldc + col0;
    const int rlim = Meff - row0 - (wm * 32 + 4 * hi);
    const unsigned off = (unsigned)(wm * 32 + 4 * hi) * (unsigned)ldc + (unsigned)(wn * 32 + l31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      acc[r] = dr < rlim ? cp[off + (unsigned)dr * (unsigned)ldc] : 0.f;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  const int fra = wm * 32 + l31;
  auto mfma6 = [&](const unsigned char* st, int kc, const BFrag& bc) {
    const int c = kc * 2 + hi;
    s3_bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const s3_bf16x8*>(st + p * PLANE + s3_at(fra, c));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bc[kc][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bc[kc][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bc[kc][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bc[kc][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bc[kc][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bc[kc][0], acc, 0, 0, 0);
  };
  // one k-tile: multiply with `bc`, request k-tile kt + VSN_S3_BRING into `bl` (the ring slot that has just been freed)
  auto step = [&](int kt, const BFrag& bc, BFrag& bl) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
    if (kt + VSN_S3_BRING < nkt) gloadB(kt + VSN_S3_BRING, bl);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(st, 0, bc);
    if (kt + 1 < nkt) {  // tile kt+1 (in registers) -> the other stage; the loads of tile kt+2 start
      __builtin_amdgcn_sched_barrier(0);
      sstoreA((kt & 1) ^ 1);
      if (kt + 2 < nkt) gloadA(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma6(st, 1, bc);
    __syncthreads();
  };
  gloadA(0);
  gloadB(0, b0);
#if VSN_S3_BRING == 2
  if (1 < nkt) gloadB(1, b1);
#endif
  sstoreA(0);
  if (1 < nkt) gloadA(1);
  __syncthreads();
#if VSN_S3_BRING == 2
  for (int kt = 0; kt < nkt; kt += 3) {
    step(kt, b0, b2);
    if (kt + 1 < nkt) step(kt + 1, b1, b0);
    if (kt + 2 < nkt) step(kt + 2, b2, b1);
  }
#else
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, b0, b1);
    if (kt + 1 < nkt) step(kt + 1, b1, b0);
  }
#endif
  float* __restrict__ Ct = ksplit == 1 ? C + (size_t)row0 * ldc + col0 : part + ((size_t)ks * M + row0) * Nc + col0;
  const unsigned ldo = (unsigned)(ksplit == 1 ? ldc : Nc);
  const unsigned off = (unsigned)(wm * 32 + 4 * hi) * ldo + (unsigned)(wn * 32 + l31);
  const int rlim = Meff - row0 - (wm * 32 + 4 * hi);
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((r & 3) + 8 * (r >> 2) < rlim) Ct[off + (unsigned)((r & 3) + 8 * (r >> 2)) * ldo] = acc[r] + bv;
}

// The batch tile in the same mode: 128 x 128 per workgroup, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA tiles), BK = 32,
// one LDS stage of six planes (48 KiB) + register prefetch of the next k-tile.  The weight planes come from the SAME
// fragment-ordered packing as above (thread (row lr, k-half lh) picks its two 16-byte pieces of a 16-k block).
// byte offset of 16-byte chunk c of row r in a 128-row x 64-byte plane
__device__ __forceinline__ int s3_at128(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 3)) << 4); }
__global__ __launch_bounds__(256) void k_gemm3_128(const float* __restrict__ A, int lda,
                                                   const unsigned short* __restrict__ B3, int ldb,
                                                   float* __restrict__ C, int ldc, const float* __restrict__ bias, int M,
                                                   const int* __restrict__ Mptr, int Nc, int K, int flags) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 8192];
  unsigned char* const lA = lds;
  unsigned char* const lB = lds + 3 * 8192;
  int Meff = M;
  if (Mptr) {
    int md = *Mptr;
    Meff = md < M ? md : M;
  }
  const int tiles_n = Nc / 128;
  const int live = ((Meff + 127) / 128) * tiles_n;
  if ((int)blockIdx.x >= live) return;
  const int bid = VSN_XCD_REMAP ? xcd_block((int)blockIdx.x, live) : (int)blockIdx.x;
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int row0 = tm * 128, col0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi5 = lane >> 5;
  // staging role: row lr of the tile, k-half lh (16 k-values = chunks 2 lh, 2 lh + 1)
  const int lr = tid >> 1, lh = tid & 1;
  const int ar = row0 + lr < Meff ? row0 + lr : Meff - 1;  // rows past Meff repeat the last row; never stored
  const float* __restrict__ ag = A + (size_t)ar * lda + lh * 16;
  const int n = col0 + lr;
  // 16-k block kb of column n, plane p, k-half c: (((n >> 5) * (ldb >> 4) + kb) * 3 + p) * 64 + (n & 31) + 32 c  (x 8 bf16)
  const unsigned short* __restrict__ bg = B3 + ((((size_t)(n >> 5) * (size_t)(ldb >> 4) + lh) * 3) * 64 + (n & 31)) * 8;
  f32x4 ra[4];
  s3_bf16x8 rb[3][2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const f32x4*>(ag + kt * 32 + q * 4);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        rb[p][c] = *reinterpret_cast<const s3_bf16x8*>(bg + (((size_t)kt * 2 * 3 + p) * 64 + 32 * c) * 8);
  };
  auto sstore = [&]() {
    s3_bf16x8 rap[3][2];
    s3_split8(ra[0], ra[1], rap[0][0], rap[1][0], rap[2][0]);
    s3_split8(ra[2], ra[3], rap[0][1], rap[1][1], rap[2][1]);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        *reinterpret_cast<s3_bf16x8*>(lA + p * 8192 + s3_at128(lr, 2 * lh + c)) = rap[p][c];
        *reinterpret_cast<s3_bf16x8*>(lB + p * 8192 + s3_at128(lr, 2 * lh + c)) = rb[p][c];
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nkt = K / 32;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();  // the previous stage's fragments are consumed
    sstore();
    __syncthreads();
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int c = kc * 2 + hi5;
      s3_bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const s3_bf16x8*>(lA + p * 8192 + s3_at128(r, c));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const s3_bf16x8*>(lB + p * 8192 + s3_at128(r, c));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }
  const bool rmw = (flags & 1) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + l31;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + i * 32 + 4 * hi5 + (r & 3) + 8 * (r >> 2);
        if (row < Meff) {
          float* cp = C + (size_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (rmw) v += *cp;
          *cp = v;
        }
      }
    }
}

// W [Nc][ldb] fp32 -> packed planes (hi, mid, lo) in fragment order
__global__ void k_s3_pack(const float* __restrict__ W, size_t n, int ldb, unsigned short* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / ldb, k = i % ldb;
  const float x = W[i];
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  out[s3_pack(r, k, 0, (size_t)ldb)] = __builtin_bit_cast(unsigned short, h);
  out[s3_pack(r, k, 1, (size_t)ldb)] = __builtin_bit_cast(unsigned short, m);
  out[s3_pack(r, k, 2, (size_t)ldb)] = __builtin_bit_cast(unsigned short, l);
}

// per-engine cache: fp32 weight operand of a product (a pointer into the engine's packed-weight arena, fixed after
// vsn_finalize) -> its packed bf16 planes, made on first sight on the launch stream
struct Split3Table {
  struct Entry {
    unsigned short* planes;
    size_t elems;
  };
  std::map<const float*, Entry> cache;
};
Split3Table* split3_table_create() { return new Split3Table(); }
void split3_table_destroy(Split3Table* t) {
  if (!t) return;
  for (auto& kv : t->cache) hipFree(kv.second.planes);
  delete t;
}
static thread_local Split3Table* tl_s3 = nullptr;
void set_gemm_split3(Split3Table* t) { tl_s3 = t; }

// the packed planes of a weight operand [Nc][ldb] of which a product reads K columns (made on first sight, on the
// launch stream); nullptr where the split tiles do not apply
static const unsigned short* s3_planes(const float* Bt, int ldb, int Nc, int K, int flags, hipStream_t st) {
  if (!tl_s3 || (flags & 2) || (ldb & 15) || (K & 31) || (Nc & 63)) return nullptr;
  // the operand may be a K-window of a wider matrix (ldb > K): only what the product reads is read here
  const size_t elems = (size_t)(Nc - 1) * ldb + K;
  auto it = tl_s3->cache.find(Bt);
  if (it == tl_s3->cache.end() || it->second.elems < elems) {
    unsigned short* p = nullptr;
    if (hipMalloc((void**)&p, 3 * (size_t)Nc * ldb * sizeof(unsigned short)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(k_s3_pack, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, Bt, elems, ldb, p);
    // once per weight matrix: the planes are complete before anybody - a later call on ANOTHER stream included - can
    // read them (and a smaller cached view of the same operand has no users left when it is freed)
    hipStreamSynchronize(st);
    if (it != tl_s3->cache.end()) hipFree(it->second.planes);
    tl_s3->cache[Bt] = Split3Table::Entry{p, elems};
    it = tl_s3->cache.find(Bt);
  }
  return it->second.planes;
}
// swap a grouped member's weight operand for its planes and mark it (members the split tile does not cover stay fp32)
static void s3_patch(GemmDesc& d, hipStream_t st) {
  const unsigned short* pl = s3_planes(d.Bt, d.ldb, d.Nc, d.K, d.flags, st);
  if (!pl) return;
  d.Bt = reinterpret_cast<const float*>(pl);
  d.flags |= VSN_S3_FLAG;
}
