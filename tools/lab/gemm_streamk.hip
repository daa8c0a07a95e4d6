// Lab only (not part of the product): the persistent stream-K form of the grouped 64x64 GEMM that round 1's review
// asked for, kept with its measurements.  Included by gemm_direct.hip after ai2bmd_amd/csrc/gemm.hip.
//
// Result on MI355X (tools/lab/README.md): correct for every shape (deterministic: the last arriver of a tile adds
// the partial slabs in k order), but SLOWER than the workgroup-per-tile kernel - 60 vs 84 TFLOP/s on the Chignolin
// forward product (nearly every tile straddles a range boundary at ~10 units per workgroup, and each straddle costs a
// 16 KiB write-through slab, a counter round trip and a slab read-back), and still 7-13 % slower where no tile is
// split at all (the tile-switching pipeline needs 124 VGPRs and ~30 KiB of code).
#pragma once
namespace vsn {
// ---------------------------------------------------------------------------------------------------------------
// Stream-K, persistent form of the grouped 64x64 kernel (single-protein sizes).
//
// Why: a K = 256 tile is eight k-iterations; a workgroup per tile spends ~30 % of its life outside the k-loop
// (prologue, epilogue, workgroup turnover) and 2 324 tiles on 1 024 resident slots are 2.27 "rounds", i.e. three.
// Here the launch is G = 1 024 workgroups (four per CU, all resident) and the work is the flat list of k-tile UNITS
// of every live tile of every member, units of a tile adjacent.  Workgroup v takes units [U v / G, U (v+1) / G):
// every SIMD gets the same number of MFMA blocks, the global->LDS pipeline runs straight through tile boundaries
// (the loads of the next tile's first k-tiles are issued under the MFMAs of the current one), and nothing is
// re-dispatched.  A tile that straddles a range boundary is finished by whoever ARRIVES last (an arrival counter
// per tile): every contributor writes its partial accumulator to a workspace slot, the last one adds the slots up
// in k order - a fixed order, so the result does not depend on who that was - and runs the epilogue.  No waiting on
// other workgroups anywhere, so no residency assumption.
// Hand-off protocol: write-through (sc1) slab stores -> vmcnt(0) in every wave -> barrier -> relaxed agent-scope
// fetch_add by one lane; the reducer does one agent-scope acquire fence, then plain loads.
struct SkArgs {
  GemmDesc p[GemmGroup::MAXP];
  int n;
  int* cnt;   // one arrival counter per live tile; zero between launches (the reducer re-zeroes its own)
  float* ws;  // partial accumulators [G][2][16][256]
};

__global__ __launch_bounds__(256) void k_gemm_sk(SkArgs g) {
  constexpr int BM = 64, BN = 64, BK = 32, LS = 32, C4 = 8, KK = 4, STAGE = (BM + BN) * LS, MAXP = GemmGroup::MAXP;
  // ONE __shared__ object (a second one, however small, makes hipcc serialise the k-loop's LDS traffic behind
  // vmcnt(0)): two operand stages + the reducer flag
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 64];
  int* const sflag = reinterpret_cast<int*>(smem + 2 * STAGE);
#define VSN_LDS_AT(r, c4) ((r) * LS + (((c4) ^ (((r) >> 1) & 7)) * 4))

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int G = (int)gridDim.x;
  // XCD x (= blockIdx % 8) gets a contiguous eighth of the virtual ids, so the column tiles that share an A row
  // tile meet in one L2
  const int v = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);

  // ---- the unit list (uniform): live tiles and k-tiles of every member ----
  // (scalars, not arrays: hipcc turns a select chain over a small private array into an indexed scratch / LDS access)
  int un[MAXP], tl[MAXP];
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    un[q] = 0;
    tl[q] = 0;
    if (q < g.n) {
      int Meff = g.p[q].M;
      if (g.p[q].Mptr) {
        const int md = *g.p[q].Mptr;
        Meff = md < Meff ? md : Meff;
      }
      tl[q] = ((Meff + BM - 1) / BM) * (g.p[q].Nc / BN);
      un[q] = tl[q] * (g.p[q].K / BK);
    }
  }
  const int ub1 = un[0], ub2 = ub1 + un[1], ub3 = ub2 + un[2], ub4 = ub3 + un[3];
  const int tb1 = tl[0], tb2 = tb1 + tl[1], tb3 = tb2 + tl[2];
  const long long U = ub4;
  const int u0 = (int)(U * v / G), u1 = (int)(U * (v + 1) / G);
  if (u0 >= u1) return;

  // a segment = the part of ONE tile inside [u0, u1)
  struct Seg {
    int p, tile, kt, end;  // member, tile within the member, first k-tile, unit index one past the segment
    int nkt, ustart;       // k-tiles of the tile, unit index of its k-tile 0
    int tidx;              // arrival counter of the tile
  };
  auto decode = [&](const int u) __attribute__((always_inline)) {
    Seg sg;
    sg.p = u >= ub3 ? 3 : u >= ub2 ? 2 : u >= ub1 ? 1 : 0;
    const int ubase = u >= ub3 ? ub3 : u >= ub2 ? ub2 : u >= ub1 ? ub1 : 0;
    const int tbase = u >= ub3 ? tb3 : u >= ub2 ? tb2 : u >= ub1 ? tb1 : 0;
    const int kq = u >= ub3 ? g.p[3].K : u >= ub2 ? g.p[2].K : u >= ub1 ? g.p[1].K : g.p[0].K;
    sg.nkt = kq / BK;
    const int local = u - ubase;
    sg.tile = local / sg.nkt;
    sg.kt = local - sg.tile * sg.nkt;
    sg.ustart = u - sg.kt;
    sg.tidx = tbase + sg.tile;
    const int tend = sg.ustart + sg.nkt;
    sg.end = tend < u1 ? tend : u1;
    return sg;
  };
  // member fields by (uniform) index
#define VSN_SEL(field, pp) \
  ((pp) == 0 ? g.p[0].field : (pp) == 1 ? g.p[1].field : (pp) == 2 ? g.p[2].field : g.p[3].field)

  // ---- load cursor: runs two units ahead of the MFMAs, straight through tile boundaries ----
  const float* __restrict__ Ak = nullptr;  // A + row0*lda + kt*BK of the next unit to fetch
  const float* __restrict__ Bk = nullptr;
  unsigned aoff[2], boff[2];
  int lu = u0, lend = u0;
  int lsa[2], lsb[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int f = tid + it * 256, r = f / C4, c4 = f % C4;
    lsa[it] = VSN_LDS_AT(r, c4);
    lsb[it] = BM * LS + VSN_LDS_AT(r, c4);
  }
  f32x4 rra[2], rrb[2];
  auto load_unit = [&]() __attribute__((always_inline)) {
    if (lu == lend) {  // first unit of a segment: where are its operands
      const Seg sg = decode(lu);
      lend = sg.end;
      const int pp = sg.p;
      const int tiles_n = VSN_SEL(Nc, pp) / BN;
      const int tm = sg.tile / tiles_n, tn = sg.tile - tm * tiles_n;
      const int lda = VSN_SEL(lda, pp), ldb = VSN_SEL(ldb, pp);
      int Meff = VSN_SEL(M, pp);
      const int* mp = VSN_SEL(Mptr, pp);
      if (mp) {
        const int md = *mp;
        Meff = md < Meff ? md : Meff;
      }
      const int row0 = tm * BM;
      Ak = VSN_SEL(A, pp) + (size_t)row0 * lda + (size_t)sg.kt * BK;
      Bk = VSN_SEL(Bt, pp) + (size_t)(tn * BN) * ldb + (size_t)sg.kt * BK;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int f = tid + it * 256, r = f / C4, c4 = f % C4;
        const int rr = row0 + r < Meff ? r : Meff - 1 - row0;  // rows >= Meff: clamped, never stored
        aoff[it] = (unsigned)rr * (unsigned)lda + (unsigned)(c4 * 4);
        boff[it] = (unsigned)r * (unsigned)ldb + (unsigned)(c4 * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) rra[it] = *reinterpret_cast<const f32x4*>(Ak + aoff[it]);
#pragma unroll
    for (int it = 0; it < 2; ++it) rrb[it] = *reinterpret_cast<const f32x4*>(Bk + boff[it]);
    Ak += BK;
    Bk += BK;
    ++lu;
  };
  auto store_unit = [&](const int stage) __attribute__((always_inline)) {
    float* St = smem + stage * STAGE;
#pragma unroll
    for (int it = 0; it < 2; ++it) *reinterpret_cast<f32x4*>(St + lsa[it]) = rra[it];
#pragma unroll
    for (int it = 0; it < 2; ++it) *reinterpret_cast<f32x4*>(St + lsb[it]) = rrb[it];
  };

  // ---- compute cursor ----
  int fo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) fo[kk] = VSN_LDS_AT(l31, kk * 2 + hi) - l31 * LS;
  const int fra = (wm * 32 + l31) * LS, frb = BM * LS + (wn * 32 + l31) * LS;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  Seg cs = decode(u0);

  // C (+)= tile + bias for the compute segment's tile (uniform control flow)
  auto write_tile = [&](const f32x16& t) __attribute__((always_inline)) {
    const int pp = cs.p;
    const int tiles_n = VSN_SEL(Nc, pp) / BN;
    const int tm = cs.tile / tiles_n, tn = cs.tile - tm * tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    int Meff = VSN_SEL(M, pp);
    const int* mp = VSN_SEL(Mptr, pp);
    if (mp) {
      const int md = *mp;
      Meff = md < Meff ? md : Meff;
    }
    const unsigned ldc = (unsigned)VSN_SEL(ldc, pp);
    float* __restrict__ Ct = VSN_SEL(C, pp) + (size_t)row0 * ldc + col0;
    const float* bp = VSN_SEL(bias, pp);
    const float bv = bp ? bp[col0 + wn * 32 + l31] : 0.f;
    const bool rmw = (VSN_SEL(flags, pp) & 1) != 0;
    const bool full = row0 + BM <= Meff;
    const unsigned off = (unsigned)(wm * 32 + 4 * hi) * ldc + (unsigned)(wn * 32 + l31);
    if (full && !rmw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Ct[off + (unsigned)((r & 3) + 8 * (r >> 2)) * ldc] = t[r] + bv;
    } else {
      const int rlim = full ? BM : Meff - row0 - (wm * 32 + 4 * hi);
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        old[r] = (rmw && dr < rlim) ? Ct[off + (unsigned)dr * ldc] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < rlim) Ct[off + (unsigned)dr * ldc] = t[r] + bv + old[r];
      }
    }
  };
  // range start of virtual workgroup w, and the workgroup that owns unit u
  auto rstart = [&](const int w) __attribute__((always_inline)) { return (int)(U * w / G); };
  auto owner = [&](const int u) __attribute__((always_inline)) {
    int w = (int)(((long long)u * G) / U);
    while (w + 1 < G && rstart(w + 1) <= u) ++w;
    while (w > 0 && rstart(w) > u) --w;
    return w;
  };
  // end of a segment: whole tile -> epilogue; part of a tile -> slab + arrival, the last arriver reduces
  auto finish_segment = [&]() __attribute__((always_inline)) {
    const bool whole = cs.kt == 0 && cs.end == cs.ustart + cs.nkt;
    if (whole) {
      write_tile(acc);
    } else {
      const int slot = u0 >= cs.ustart ? 0 : 1;  // the range starts inside this tile, or before it
      float* const slab = g.ws + ((size_t)v * 2 + slot) * (16 * 256);
#pragma unroll
      for (int r = 0; r < 16; ++r)  // write-through (sc1) stores: no L2 write-back fence needed to publish them
        __hip_atomic_store(slab + r * 256 + tid, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int tend = cs.ustart + cs.nkt;
      const int wfirst = owner(cs.ustart), wlast = owner(tend - 1);
      if (tid == 0) {
        int* const cp = g.cnt + cs.tidx;
        int nc = 0;  // contributors = the workgroups of [wfirst, wlast] whose range is not empty
        for (int w = wfirst; w <= wlast; ++w) nc += rstart(w + 1) > rstart(w) ? 1 : 0;
        const int old = __hip_atomic_fetch_add(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == nc - 1;
        if (last) __hip_atomic_store(cp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        sflag[0] = last;
      }
      __syncthreads();
      if (sflag[0]) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.f;
        for (int w = wfirst; w <= wlast; ++w) {  // k order: a fixed summation order whoever reduces
          if (rstart(w + 1) == rstart(w)) continue;
          const int sl = rstart(w) >= cs.ustart ? 0 : 1;
          const float* sp = g.ws + ((size_t)w * 2 + sl) * (16 * 256);
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] += sp[r * 256 + tid];
        }
        write_tile(t);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };

  // ---- prologue: unit u0 into stage 0, unit u0+1 into registers ----
  load_unit();
  store_unit(0);
  if (u0 + 1 < u1) load_unit();
  __syncthreads();

  auto unit = [&](auto stc, const int u) __attribute__((always_inline)) {
    constexpr int st = decltype(stc)::value;
    const float* Sr = smem + st * STAGE;
    f32x4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const f32x4*>(Sr + fra + fo[0]);
    fb[0] = *reinterpret_cast<const f32x4*>(Sr + frb + fo[0]);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) {
        fa[(kk + 1) & 1] = *reinterpret_cast<const f32x4*>(Sr + fra + fo[kk + 1]);
        fb[(kk + 1) & 1] = *reinterpret_cast<const f32x4*>(Sr + frb + fo[kk + 1]);
      }
      const int cur = kk & 1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc, 0, 0, 0);
      if (kk == KK / 2 - 1) {
        // half-way through the MFMA block: unit u+1 (in registers) -> the other stage, unit u+2 -> registers
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < u1) {
          store_unit(st ^ 1);
          if (u + 2 < u1) load_unit();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  // segments in order; the LDS stage of unit u is (u - u0) & 1 whatever segment it belongs to
  int u = u0;
  while (u < u1) {
    const int send = cs.end;
    if ((u - u0) & 1) {
      unit(std::integral_constant<int, 1>{}, u);
      ++u;
    }
    for (; u + 1 < send; u += 2) {
      unit(std::integral_constant<int, 0>{}, u);
      unit(std::integral_constant<int, 1>{}, u + 1);
    }
    if (u < send) {
      unit(std::integral_constant<int, 0>{}, u);
      ++u;
    }
    finish_segment();
    if (u < u1) cs = decode(u);
  }
#undef VSN_SEL
#undef VSN_LDS_AT
}

// stream-K workspace (set by the engine per handle; nullptr disables the stream-K path)
static thread_local int* tl_sk_cnt = nullptr;
static thread_local int tl_sk_cnt_elems = 0;
static thread_local float* tl_sk_ws = nullptr;
static thread_local int tl_sk_G = 0;
void set_gemm_sk_workspace(int* cnt, int cnt_elems, float* ws, int G) {
  tl_sk_cnt = cnt;
  tl_sk_cnt_elems = cnt_elems;
  tl_sk_ws = ws;
  tl_sk_G = G;
}
size_t gemm_sk_ws_floats(int G) { return (size_t)G * 2 * 16 * 256; }

// 1 = launched, 0 = not applicable (caller falls back), < 0 = error
int launch_gemm_sk(hipStream_t st, const GemmDesc* descs, int n) {
  if (!tl_sk_ws || !tl_sk_cnt || tl_sk_G < 8 || (tl_sk_G & 7) || n < 1 || n > GemmGroup::MAXP) return 0;
  SkArgs a;
  a.n = 0;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    const GemmDesc& d = descs[i];
    if (d.M <= 0) continue;
    if ((d.K & 31) || (d.Nc & 63) || (d.lda & 3) || (d.ldb & 3) || (d.flags & ~1) || d.keep_parts > 1) return 0;
    tiles += (long long)((d.M + 63) / 64) * (d.Nc / 64);
    a.p[a.n++] = d;
  }
  if (a.n == 0) return 1;
  if (tiles > tl_sk_cnt_elems) return 0;
  for (int i = a.n; i < GemmGroup::MAXP; ++i) {
    a.p[i] = a.p[0];
    a.p[i].M = 0;
  }
  a.cnt = tl_sk_cnt;
  a.ws = tl_sk_ws;
  GemmProfiler::Rec* rec = nullptr;
  if (false) {
    tl_prof->recs.emplace_back();
    rec = &tl_prof->recs.back();
    rec->variant = 3;  // grouped launch
    rec->M = 1;
    rec->dev_m = false;
    rec->flops_per_row = 0;
    rec->bytes_per_row = 0;
    rec->group_n = 0;
    for (int i = 0; i < a.n; ++i) {
      const GemmDesc& d = a.p[i];
      rec->gM[rec->group_n] = d.M;
      rec->gdev[rec->group_n] = d.Mptr != nullptr;
      rec->gflops[rec->group_n] = 2.0 * d.Nc * d.K;
      rec->gbytes[rec->group_n] = 4.0 * (d.K + d.Nc * ((d.flags & 1) ? 2.0 : 1.0));
      rec->group_n++;
    }
    hipEventCreate(&rec->a);
    hipEventCreate(&rec->b);
    hipEventRecord(rec->a, st);
  }
  hipLaunchKernelGGL(k_gemm_sk, dim3(tl_sk_G), dim3(256), 0, st, a);
  if (rec) hipEventRecord(rec->b, st);
  return 1;
}

}  // namespace vsn
