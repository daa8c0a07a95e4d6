// Neighbour list + pair geometry for a batch of fragments.
//
// Reference: ViSNet/model/utils.py:259-276 (Distance.forward ->
// torch_cluster.radius_graph(loop=True, max_num_neighbors)), visnet_block.py:111-117
// (unit vectors, Sphere), utils.py:16-19,53-57 (CosineCutoff, ExpNormalSmearing),
// utils.py:131-160 (real spherical harmonics).
//
// Output: edges (j -> i) in CSR order by target i, sources ascending (the order
// torch_cluster's CUDA kernel produces), at most max_nb sources per target
// keeping the lowest indices, self loops kept; plus the same edge set grouped
// by source (perm/colptr) for the reverse pass' deterministic segmented sums.
// Fragments of AI2BMD are tiny (<= 44 atoms): one workgroup per fragment, everything in LDS.  Batches holding a
// larger fragment (whole-molecule mode, B = 1, up to VSN_MAX_FRAG_ATOMS) take the node-parallel passes below.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace vsn {

__device__ __forceinline__ float dist2(const float* __restrict__ pos, int a, int b) {
  float dx = pos[3 * a + 0] - pos[3 * b + 0];
  float dy = pos[3 * a + 1] - pos[3 * b + 1];
  float dz = pos[3 * a + 2] - pos[3 * b + 2];
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// pass 1: truncated in-degree per target; also narrows z to int32.  An atomic number outside the embedding /
// atomref tables (nn.Embedding raises IndexError in the reference) is clamped to row 0 and raises the chunk's
// status flag; the final kernels then poison energies and forces with NaN and vsn_last_status() reports it.
__global__ void k_graph_count(const float* __restrict__ pos, const long long* __restrict__ z64,
                              const int* __restrict__ fstart, const int* __restrict__ fend, int* __restrict__ deg,
                              int* __restrict__ zi, float rc2, int max_nb, int z_limit, int* __restrict__ status,
                              int epoch) {
  const int b = blockIdx.x;
  const int s = fstart[b], n = fend[b] - s;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int cnt = 0;
    for (int j = 0; j < n && cnt < max_nb; ++j)
      if (dist2(pos, s + j, s + i) < rc2) ++cnt;
    deg[s + i] = cnt;
    const long long zv = z64[s + i];
    const bool bad = zv < 0 || zv >= (long long)z_limit;
    if (bad) atomicMax(status, epoch);
    zi[s + i] = bad ? 0 : (int)zv;
  }
}

// exclusive scan of deg[0..N) -> rowptr[0..N], total -> *ecount ; one block
__global__ __launch_bounds__(1024) void k_scan(const int* __restrict__ deg, int* __restrict__ rowptr,
                                               int* __restrict__ colptr, int N, int* __restrict__ ecount) {
  __shared__ int part[1024];
  __shared__ int carry_s;
  const int tid = threadIdx.x;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    int i = base + tid;
    int v = i < N ? deg[i] : 0;
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    int incl = part[tid];
    int carry = carry_s;
    if (i < N) rowptr[i] = carry + incl - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + incl;
    __syncthreads();
  }
  if (tid == 0) {
    rowptr[N] = carry_s;
    colptr[N] = carry_s;
    *ecount = carry_s;
  }
}

// Small-fragment variant of pass 2 (n <= 64, every AI2BMD fragment): one wave per fragment,
// positions and the dense n x n edge-id matrix live in LDS, so the by-source view needs no
// searches: thread j just walks column j in ascending target order.
// SCAN (single-protein sizes, N < 4096): the wave also derives its rows of rowptr itself - base = sum of deg over all
// earlier atoms (a few hundred to a few thousand ints), then the prefix inside the fragment - so the one-block k_scan
// launch between the degree count and this kernel disappears (every launch of the MD step sits at a ~5 us floor).
template <bool SCAN>
__global__ __launch_bounds__(64) void k_graph_fill_small(const float* __restrict__ pos,
                                                         const int* __restrict__ fstart,
                                                         const int* __restrict__ fend, int* __restrict__ rowptr,
                                                         int* __restrict__ src, int* __restrict__ tgt,
                                                         int* __restrict__ colptr, int* __restrict__ perm, float rc2,
                                                         int max_nb, const int* __restrict__ deg, int B, int N,
                                                         int* __restrict__ ecount) {
  __shared__ float ps[64 * 3];
  __shared__ int eid[64 * 65];  // eid[i*65 + j] = edge (j -> i) or -1 ; padded against bank conflicts
  __shared__ int outdeg[64];
  const int b = blockIdx.x;
  const int s = fstart[b], n = fend[b] - s;
  const int t = threadIdx.x;
  int my_row = 0, base = 0;
  if (SCAN) {
    int part = 0;
    for (int a = t; a < s; a += 64) part += deg[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    base = part;
    const int dv = t < n ? deg[s + t] : 0;
    int incl = dv;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (t >= o) incl += v;
    }
    my_row = base + incl - dv;
    if (t < n) rowptr[s + t] = my_row;
    if (b == B - 1 && t == 63) {  // the last fragment (possibly empty) closes the arrays: incl of lane 63 = its edge total
      rowptr[N] = base + incl;
      colptr[N] = base + incl;
      *ecount = base + incl;
    }
  }
  if (n <= 0) return;
  for (int k = t; k < 3 * n; k += 64) ps[k] = pos[3 * (size_t)s + k];
  __syncthreads();
  if (t < n) {
    const int i = t;
    int e = SCAN ? my_row : rowptr[s + i];
    int cnt = 0;
    const float xi = ps[3 * i], yi = ps[3 * i + 1], zi_ = ps[3 * i + 2];
    for (int j = 0; j < n; ++j) {
      float dx = ps[3 * j] - xi, dy = ps[3 * j + 1] - yi, dz = ps[3 * j + 2] - zi_;
      float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      int id = -1;
      if (d2 < rc2 && cnt < max_nb) {
        src[e] = s + j;
        tgt[e] = s + i;
        id = e;
        ++e;
        ++cnt;
      }
      eid[i * 65 + j] = id;
    }
  }
  __syncthreads();
  int cnt = 0;
  if (t < n)
    for (int i = 0; i < n; ++i) cnt += eid[i * 65 + t] >= 0 ? 1 : 0;
  outdeg[t] = t < n ? cnt : 0;
  __syncthreads();
  // exclusive prefix over the wave (n <= 64)
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(incl, o, 64);
    if (t >= o) incl += v;
  }
  if (t < n) {
    int out = (SCAN ? base : rowptr[s]) + incl - cnt;
    colptr[s + t] = out;
    for (int i = 0; i < n; ++i) {
      const int e = eid[i * 65 + t];
      if (e >= 0) perm[out++] = e;
    }
  }
}

// ---- fragments larger than a wavefront (whole-molecule mode, `--mode visnet`: ONE fragment of 10^3..10^4 atoms,
// Calculators/visnet_calculator.py:139-155).  Same edges, same order, same truncation rule as above, but one THREAD
// per node over as many workgroups as the batch needs instead of one 64-thread workgroup per fragment: every pass
// is n threads x n ascending candidates (all lanes of a wave read the same candidate position: a broadcast), the
// out-degree prefix is a global scan (edges never leave their fragment, so the global exclusive scan of the
// out-degrees IS colptr).
__device__ __forceinline__ int frag_of(const int* __restrict__ fend, int B, int i) {
  int lo = 0, hi = B - 1;  // first fragment whose end is > i
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (fend[mid] > i) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

__global__ void k_graph_count_big(const float* __restrict__ pos, const long long* __restrict__ z64,
                                  const int* __restrict__ fstart, const int* __restrict__ fend, int B, int N,
                                  int* __restrict__ deg, int* __restrict__ zi, float rc2, int max_nb, int z_limit,
                                  int* __restrict__ status, int epoch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int b = frag_of(fend, B, i);
  const int s = fstart[b], e = fend[b];
  int cnt = 0;
  for (int j = s; j < e && cnt < max_nb; ++j)
    if (dist2(pos, j, i) < rc2) ++cnt;
  deg[i] = cnt;
  const long long zv = z64[i];
  const bool bad = zv < 0 || zv >= (long long)z_limit;
  if (bad) atomicMax(status, epoch);
  zi[i] = bad ? 0 : (int)zv;
}

__global__ void k_graph_fill_big(const float* __restrict__ pos, const int* __restrict__ fstart,
                                 const int* __restrict__ fend, int B, int N, const int* __restrict__ rowptr,
                                 int* __restrict__ src, int* __restrict__ tgt, float rc2, int max_nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int b = frag_of(fend, B, i);
  const int s = fstart[b], e = fend[b];
  int out = rowptr[i], cnt = 0;
  for (int j = s; j < e && cnt < max_nb; ++j)
    if (dist2(pos, j, i) < rc2) {
      src[out] = j;
      tgt[out] = i;
      ++out;
      ++cnt;
    }
}

// position of source j in target i's (ascending) row, or -1 when the truncation dropped it
__device__ __forceinline__ int row_find(const int* __restrict__ rowptr, const int* __restrict__ src, int i, int j) {
  int lo = rowptr[i], hi = rowptr[i + 1];
  const int end = hi;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (src[mid] < j) lo = mid + 1;
    else hi = mid;
  }
  return (lo < end && src[lo] == j) ? lo : -1;
}

// FILL == false: outdeg[j] = number of targets that kept j ; FILL == true: perm[colptr[j]..] = those edges, targets ascending
template <bool FILL>
__global__ void k_graph_bysrc_big(const float* __restrict__ pos, const int* __restrict__ fstart,
                                  const int* __restrict__ fend, int B, int N, const int* __restrict__ rowptr,
                                  const int* __restrict__ src, int* __restrict__ outdeg,
                                  const int* __restrict__ colptr, int* __restrict__ perm, float rc2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int b = frag_of(fend, B, j);
  const int s = fstart[b], e = fend[b];
  int cnt = 0, out = FILL ? colptr[j] : 0;
  for (int i = s; i < e; ++i) {
    if (!(dist2(pos, j, i) < rc2)) continue;
    const int at = row_find(rowptr, src, i, j);
    if (at < 0) continue;
    if (FILL) perm[out++] = at;
    else ++cnt;
  }
  if (!FILL) outdeg[j] = cnt;
}

// ---- single-protein sizes: the WHOLE graph + geometry in one launch (round 6) -----------------------------------------
// An MD step of one protein is a chain of dependent launches, each with a ~5 us life of its own; degree count, CSR fill
// and per-edge geometry were three of them (5.7 + 10.5 + 5.8 us on Chignolin) for a few microseconds of work.  What
// forced the kernel boundaries was ONE number per fragment: the edge count of all earlier fragments (the base of its CSR
// rows).  Here every workgroup derives it itself - the truncated in-degrees of the atoms in front of its fragment, a few
// thousand distance tests spread over 256 threads (the same `<` on the same fp32 expression, so the same counts) - and
// then builds rows, by-source view and geometry of its own edges with everything staged in LDS.
// Same edges in the same order as the passes above (CSR by target, sources ascending, lowest-index max_nb kept).
// Geometry: r, C, dC, exp(-alpha r) once per edge (k_edge_geom recomputes them for each of the Rp basis functions).
#define VSN_GRAPH_MAXE (64 * 32)  // edges of one fragment: <= 64 targets x <= max_nb (<= 32 on this path) sources
#define VSN_GRAPH_WAVES 16
// Every neighbour test is ONE wave-wide comparison: lane j tests candidate j of the fragment, the ballot is the row of
// the adjacency matrix, its population count the degree, the count of set bits below a lane the position of that source
// in the (ascending) row - no serial scan over candidates anywhere.
#define VSN_GRAPH_POOL (64 * 65 + 3 * VSN_GRAPH_MAXE)  // floats: eid + eC + edC + et ... or the staged positions
#define VSN_GRAPH_MAXB 1024                            // fragment offsets staged in LDS
template <int DUMMY>
__global__ __launch_bounds__(64 * VSN_GRAPH_WAVES) void k_graph_small_all(GraphArgs a) {
  constexpr int NW = VSN_GRAPH_WAVES, NT = 64 * NW;
  __shared__ float ps[64 * 3];
  // one pool, two lives: first the positions of every atom in FRONT of this fragment (the base count below walks
  // earlier fragments one after the other: from global memory that was one dependent ~1 us round trip per fragment,
  // 20 us for the last fragment of Chignolin), then the edge-id matrix and the per-edge geometry terms
  __shared__ __attribute__((aligned(16))) float pool[VSN_GRAPH_POOL];
  __shared__ int sfs[VSN_GRAPH_MAXB + 1];
  int* eid = reinterpret_cast<int*>(pool);  // eid[i*65 + j] = edge (j -> i) or -1 ; padded against bank conflicts
  float* eC = pool + 64 * 65;               // C, dC, exp(-alpha r) (gauss: r) per local edge
  float* edC = eC + VSN_GRAPH_MAXE;
  float* et = edC + VSN_GRAPH_MAXE;
  __shared__ unsigned short epair[VSN_GRAPH_MAXE];  // local edge -> (target << 8) | source
  __shared__ int red[NW];
  __shared__ int sdeg[64], srow[64], sout[64], scol[64];
  __shared__ int s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = a.fstart[b], n = a.fend[b] - s;
  for (int k = tid; k < 3 * s; k += NT) pool[k] = a.pos[k];
  for (int k = tid; k < b; k += NT) sfs[k] = a.fstart[k];
  if (tid == 0) sfs[b] = s;  // (fragments are contiguous: end of fb = start of fb + 1)
  __syncthreads();
  // ---- base = sum of the truncated in-degrees of all atoms in front of this fragment (a wave per atom)
  int part = 0;
  for (int fb = 0; fb < b; ++fb) {
    const int fs = sfs[fb], fn = sfs[fb + 1] - fs;
    float xj = 0.f, yj = 0.f, zj = 0.f;
    if (lane < fn) {
      xj = pool[3 * (fs + lane)];
      yj = pool[3 * (fs + lane) + 1];
      zj = pool[3 * (fs + lane) + 2];
    }
    for (int i = wave; i < fn; i += NW) {
      // (candidate - target, like dist2(pos, j, i): the squares do not see the sign)
      const float dx = xj - __shfl(xj, i, 64), dy = yj - __shfl(yj, i, 64), dz = zj - __shfl(zj, i, 64);
      const bool in = lane < fn && __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)) < a.rc2;
      const int cnt = __popcll(__ballot(in));
      part += cnt < a.max_nb ? cnt : a.max_nb;  // (wave-uniform)
    }
  }
  __syncthreads();  // (the pool is about to change hands)
  if (lane == 0) red[wave] = part;
  for (int k = tid; k < 3 * n; k += NT) ps[k] = a.pos[3 * (size_t)s + k];
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) base += red[w];
  // ---- rows of this fragment: a wave per target, lane = candidate source
  unsigned long long keepm[(64 + NW - 1) / NW];
#pragma unroll
  for (int q = 0; q < (64 + NW - 1) / NW; ++q) {
    const int i = wave + q * NW;
    keepm[q] = 0ull;
    if (i < n) {  // (wave-uniform)
      bool in = false;
      if (lane < n) {
        const float dx = ps[3 * lane] - ps[3 * i], dy = ps[3 * lane + 1] - ps[3 * i + 1], dz = ps[3 * lane + 2] - ps[3 * i + 2];
        in = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)) < a.rc2;
      }
      const unsigned long long m = __ballot(in);
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      keepm[q] = __ballot(in && rank < a.max_nb);  // the lowest-index max_nb sources
      if (lane == 0) {
        const int c = __popcll(m);
        sdeg[i] = c < a.max_nb ? c : a.max_nb;
      }
    }
  }
  if (tid < n) {
    const long long zv = a.z64[s + tid];
    const bool bad = zv < 0 || zv >= (long long)a.z_limit;
    if (bad) atomicMax(a.status, a.epoch);
    a.zi[s + tid] = bad ? 0 : (int)zv;
  }
  __syncthreads();
  if (wave == 0) {
    const int dv = lane < n ? sdeg[lane] : 0;
    int incl = dv;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    srow[lane] = base + incl - dv;
    if (lane < n) a.rowptr[s + lane] = base + incl - dv;
    if (lane == 63) {
      s_cnt = incl;
      if (b == a.B - 1) {  // the last fragment (possibly empty) closes the arrays
        a.rowptr[a.N] = base + incl;
        a.colptr[a.N] = base + incl;
        *a.ecount = base + incl;
      }
    }
  }
  __syncthreads();
  if (n <= 0) return;  // (uniform per workgroup)
#pragma unroll
  for (int q = 0; q < (64 + NW - 1) / NW; ++q) {
    const int i = wave + q * NW;
    if (i < n && lane < n) {
      const unsigned long long m = keepm[q];
      int id = -1;
      if ((m >> lane) & 1ull) {
        id = srow[i] + __popcll(m & ((1ull << lane) - 1ull));
        a.src[id] = s + lane;
        a.tgt[id] = s + i;
        epair[id - base] = (unsigned short)((i << 8) | lane);
      }
      eid[i * 65 + lane] = id;
    }
  }
  __syncthreads();
  // ---- the same edges by source: a wave per source j, lane = target i (ascending)
#pragma unroll
  for (int q = 0; q < (64 + NW - 1) / NW; ++q) {
    const int j = wave + q * NW;
    keepm[q] = 0ull;
    if (j < n) {
      const bool has = lane < n && eid[lane * 65 + j] >= 0;
      keepm[q] = __ballot(has);
      if (lane == 0) sout[j] = __popcll(keepm[q]);
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int dv = lane < n ? sout[lane] : 0;
    int incl = dv;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    scol[lane] = base + incl - dv;
    if (lane < n) a.colptr[s + lane] = base + incl - dv;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (64 + NW - 1) / NW; ++q) {
    const int j = wave + q * NW;
    if (j < n && lane < n) {
      const unsigned long long m = keepm[q];
      if ((m >> lane) & 1ull) a.perm[scol[j] + __popcll(m & ((1ull << lane) - 1ull))] = eid[lane * 65 + j];
    }
  }
  // ---- geometry of this fragment's edges
  const int ne = s_cnt;
  const float pi_rc = 3.14159265358979323846f / a.rc;
  for (int le = tid; le < ne; le += NT) {
    const int pr = epair[le], i = pr >> 8, j = pr & 255;
    const size_t e = (size_t)base + le;
    const float ex = ps[3 * j] - ps[3 * i], ey = ps[3 * j + 1] - ps[3 * i + 1], ez = ps[3 * j + 2] - ps[3 * i + 2];
    const bool loop = (i == j);
    const float r = loop ? 0.f : sqrtf(ex * ex + ey * ey + ez * ez);
    const float rinv = loop ? 0.f : 1.0f / r;
    const float inside = (r < a.rc) ? 1.f : 0.f;
    const float C = 0.5f * (cosf(r * pi_rc) + 1.0f) * inside;
    const float dC = -0.5f * pi_rc * sinf(r * pi_rc) * inside;
    eC[le] = C;
    edC[le] = dC;
    et[le] = a.rbf_type == 1 ? r : expf(-a.alpha * r);
    const float ux = ex * rinv, uy = ey * rinv, uz = ez * rinv;
    float4* g = reinterpret_cast<float4*>(a.geo + e * 8);
    g[0] = make_float4(r, C, dC, ux);
    g[1] = make_float4(uy, uz, rinv, 0.f);
    const float s3 = 1.7320508075688772f;
    float4* dd = reinterpret_cast<float4*>(a.d + e * 8);
    if (a.S == 8) {
      dd[0] = make_float4(ux, uy, uz, s3 * ux * uz);
      dd[1] = make_float4(s3 * ux * uy, uy * uy - 0.5f * (ux * ux + uz * uz), s3 * uy * uz, 0.5f * s3 * (uz * uz - ux * ux));
    } else {
      dd[0] = make_float4(ux, uy, uz, 0.f);
      dd[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // rbf / drbf rows (and the cleared g_geo row): a wave walks whole edges, lane = basis function (Rp = 32: two edges)
  const int Rp = a.Rp, R = a.R;
  const float coeff = a.rbf_type == 1 ? a.betas[0] : 0.f;
  const int per = Rp <= 32 ? 2 : 1;              // edges per wave pass
  const int k = Rp <= 32 ? (lane & 31) : lane;   // (Rp is 32 or 64 on this path: num_rbf <= 64)
  const float mu = k < R ? a.means[k] : 0.f, be = (k < R && a.rbf_type != 1) ? a.betas[k] : 0.f;
  for (int le = wave * per + (Rp <= 32 ? (lane >> 5) : 0); le < ne; le += NW * per) {
    const size_t e = (size_t)base + le;
    if (k < VSN_GEO_W && k < Rp) a.g_geo[e * VSN_GEO_W + k] = 0.f;  // the reverse pass accumulates dE/dd, dE/dC per edge here
    float v = 0.f, dv = 0.f;
    if (k < R) {
      const float C = eC[le], dC = edC[le], t = et[le];
      if (a.rbf_type == 1) {  // GaussianSmearing: exp(coeff (r - offset_k)^2), no cutoff factor
        const float dr = t - mu;
        const float ek = expf(coeff * dr * dr);
        v = ek;
        dv = 2.0f * coeff * dr * ek;
      } else {
        const float ek = expf(-be * (t - mu) * (t - mu));
        const float dek = 2.0f * a.alpha * be * t * (t - mu) * ek;
        v = C * ek;
        dv = dC * ek + C * dek;
      }
    }
    if (k < Rp) {
      a.rbf[e * Rp + k] = v;
      a.drbf[e * Rp + k] = dv;
    }
  }
}

// per-edge geometry: one thread per (edge, rbf index); Rp = padded rbf count (multiple of 32)
__global__ void k_edge_geom(const float* __restrict__ pos, const int* __restrict__ src, const int* __restrict__ tgt,
                            const int* __restrict__ ecount, const float* __restrict__ means,
                            const float* __restrict__ betas, int rbf_type, int R, int Rp, float rc, float alpha, int S,
                            float* __restrict__ geo /*[E,8]: r,C,dC,ux,uy,uz,rinv,pad*/, float* __restrict__ d,
                            float* __restrict__ rbf, float* __restrict__ drbf, float* __restrict__ g_geo) {
  const int E = *ecount;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e = gid / Rp;
  const int k = (int)(gid % Rp);
  if (e >= E) return;
  if (k < VSN_GEO_W) g_geo[e * VSN_GEO_W + k] = 0.f;  // the reverse pass accumulates dE/dd, dE/dC per edge here (Rp >= 32)
  const int j = src[e], i = tgt[e];
  float ex = pos[3 * j + 0] - pos[3 * i + 0];
  float ey = pos[3 * j + 1] - pos[3 * i + 1];
  float ez = pos[3 * j + 2] - pos[3 * i + 2];
  const bool loop = (i == j);
  float r = loop ? 0.f : sqrtf(ex * ex + ey * ey + ez * ez);
  float rinv = loop ? 0.f : 1.0f / r;
  const float pi_rc = 3.14159265358979323846f / rc;
  float inside = (r < rc) ? 1.f : 0.f;
  float C = 0.5f * (cosf(r * pi_rc) + 1.0f) * inside;
  float dC = -0.5f * pi_rc * sinf(r * pi_rc) * inside;
  if (k < R && rbf_type == 1) {
    // GaussianSmearing (utils.py:60-87): exp(coeff (r - offset_k)^2), NO cutoff factor; means = offset, betas[0] = coeff
    const float coeff = betas[0], dr = r - means[k];
    const float ek = expf(coeff * dr * dr);
    rbf[e * Rp + k] = ek;
    drbf[e * Rp + k] = 2.0f * coeff * dr * ek;
  } else if (k < R) {
    float t = expf(-alpha * r);
    float mu = means[k], be = betas[k];
    float ek = expf(-be * (t - mu) * (t - mu));
    float dek = 2.0f * alpha * be * t * (t - mu) * ek;
    rbf[e * Rp + k] = C * ek;
    drbf[e * Rp + k] = dC * ek + C * dek;
  } else {
    rbf[e * Rp + k] = 0.f;
    drbf[e * Rp + k] = 0.f;
  }
  if (k == 0) {
    float ux = ex * rinv, uy = ey * rinv, uz = ez * rinv;
    float* g = geo + e * 8;
    g[0] = r;
    g[1] = C;
    g[2] = dC;
    g[3] = ux;
    g[4] = uy;
    g[5] = uz;
    g[6] = rinv;
    g[7] = 0.f;
    float* dd = d + e * 8;
    const float s3 = 1.7320508075688772f;
    dd[0] = ux;
    dd[1] = uy;
    dd[2] = uz;
    if (S == 8) {
      dd[3] = s3 * ux * uz;
      dd[4] = s3 * ux * uy;
      dd[5] = uy * uy - 0.5f * (ux * ux + uz * uz);
      dd[6] = s3 * uy * uz;
      dd[7] = 0.5f * s3 * (uz * uz - ux * ux);
    } else {
      dd[3] = dd[4] = dd[5] = dd[6] = dd[7] = 0.f;
    }
  }
}

// reverse of the geometry: dE/d(edge vector) per edge from the accumulated
// dE/dr-ish terms (g_rbf . drbf + g_C * dC) and dE/dd (through the SH Jacobian
// and the unit-vector normalisation).
__device__ __forceinline__ void edge_force(int e, const float* __restrict__ geo, const float* __restrict__ g_rbf,
                                           const float* __restrict__ drbf, int Rp,
                                           const float* __restrict__ g_geo /*[E,32]: g_d 0..7, 16..23 and 24..31, g_C at 8*/,
                                           int S, float& ox, float& oy, float& oz) {
  const float* g = geo + (size_t)e * 8;
  const float dC = g[2], ux = g[3], uy = g[4], uz = g[5], rinv = g[6];
  float gr = g_geo[(size_t)e * VSN_GEO_W + 8] * dC;
  // 16-byte loads (Rp is a multiple of 32: rows are 128-byte aligned); same summation order as a scalar loop
  const float4* gb = reinterpret_cast<const float4*>(g_rbf + (size_t)e * Rp);
  const float4* db = reinterpret_cast<const float4*>(drbf + (size_t)e * Rp);
#pragma unroll 8
  for (int k = 0; k < Rp / 4; ++k) {
    const float4 a = gb[k], b = db[k];
    gr += a.x * b.x;
    gr += a.y * b.y;
    gr += a.z * b.z;
    gr += a.w * b.w;
  }
  const float* gd0 = g_geo + (size_t)e * VSN_GEO_W;
  float gd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) gd[k] = gd0[k] + (gd0[16 + k] + gd0[24 + k]);  // vector-message + edge-update shares (two channel halves)
  const float s3 = 1.7320508075688772f;
  float gx = gd[0], gy = gd[1], gz = gd[2];
  if (S == 8) {
    gx += s3 * uz * gd[3] + s3 * uy * gd[4] - ux * gd[5] - s3 * ux * gd[7];
    gy += s3 * ux * gd[4] + 2.0f * uy * gd[5] + s3 * uz * gd[6];
    gz += s3 * ux * gd[3] - uz * gd[5] + s3 * uy * gd[6] + s3 * uz * gd[7];
  }
  float dot = gx * ux + gy * uy + gz * uz;
  ox = gr * ux + (gx - dot * ux) * rinv;
  oy = gr * uy + (gy - dot * uy) * rinv;
  oz = gr * uz + (gz - dot * uz) * rinv;
  if (rinv == 0.f) ox = oy = oz = 0.f;  // self loop: no position dependence
}

// one thread per edge -> g_ev [E,4]
__global__ void k_bwd_geom(const int* __restrict__ ecount, const float* __restrict__ geo,
                           const float* __restrict__ g_rbf, const float* __restrict__ drbf, int Rp,
                           const float* __restrict__ g_geo, int S, float* __restrict__ g_ev /*[E,4]*/) {
  const int E = *ecount;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float ox, oy, oz;
  edge_force(e, geo, g_rbf, drbf, Rp, g_geo, S, ox, oy, oz);
  float* o = g_ev + (size_t)e * 4;
  o[0] = ox;
  o[1] = oy;
  o[2] = oz;
  o[3] = 0.f;
}

// F_i = -dE/dpos_i = sum_{e: tgt=i} g_ev_e - sum_{e: src=i} g_ev_e  (ev = pos_src - pos_tgt), fixed order.
// 16 lanes per atom stride over its ~17 in- and ~17 out-edges (one thread per atom walked 34 dependent loads).
__global__ void k_force_gather(int N, const int* __restrict__ rowptr, const int* __restrict__ colptr,
                               const int* __restrict__ perm, const float* __restrict__ g_ev,
                               float* __restrict__ f_out, const int* __restrict__ status, int epoch) {
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), l = threadIdx.x & 15;
  const bool live = i < N;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  if (live) {
    for (int e = rowptr[i] + l; e < rowptr[i + 1]; e += 16) {
      fx += g_ev[4 * (size_t)e + 0];
      fy += g_ev[4 * (size_t)e + 1];
      fz += g_ev[4 * (size_t)e + 2];
    }
    for (int t = colptr[i] + l; t < colptr[i + 1]; t += 16) {
      const int e = perm[t];
      fx -= g_ev[4 * (size_t)e + 0];
      fy -= g_ev[4 * (size_t)e + 1];
      fz -= g_ev[4 * (size_t)e + 2];
    }
  }
  fx = group_sum(fx, 16);
  fy = group_sum(fy, 16);
  fz = group_sum(fz, 16);
  if (*status == epoch) fx = fy = fz = __builtin_nanf("");  // invalid input (atomic number out of range): fail loudly
  if (live && l == 0) {
    f_out[3 * (size_t)i + 0] = fx;
    f_out[3 * (size_t)i + 1] = fy;
    f_out[3 * (size_t)i + 2] = fz;
  }
}

// Single-protein sizes: the two passes above in ONE launch - a WAVE per atom, one lane per incident edge (in-edges,
// then out-edges through perm), every lane evaluates the adjoint of its edge vector itself (each edge twice: once at
// its target, once at its source; a few hundred bytes of L2-resident rows) and the wave adds the contributions up;
// g_ev never exists.  (The dependent chain rowptr -> edge -> rows is walked ONCE per lane: 16 lanes per atom walking
// two or three edges each took 14 us, longer than the two launches it replaced.)
__global__ __launch_bounds__(256) void k_force_gather_geom(int N, const int* __restrict__ rowptr,
                                                           const int* __restrict__ colptr, const int* __restrict__ perm,
                                                           const float* __restrict__ geo, const float* __restrict__ g_rbf,
                                                           const float* __restrict__ drbf, int Rp,
                                                           const float* __restrict__ g_geo, int S,
                                                           float* __restrict__ f_out, const int* __restrict__ status,
                                                           int epoch, EnergyFold ef) {
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), l = threadIdx.x & 63;
  if (i < ef.B) {  // the per-fragment energy sums of the read-out (head.hip::hk_energy): wave i sums fragment i
    float acc = 0.f;
    for (int k = ef.fstart[i] + l; k < ef.fend[i]; k += 64) acc += ef.y[k];
    acc = wave_sum(acc);
    if (l == 0) ef.e_out[i] = (*status == epoch) ? __builtin_nanf("") : acc + ef.mean;
  }
  if (i >= N) return;  // (wave-uniform)
  const int r0 = rowptr[i], din = rowptr[i + 1] - r0, c0 = colptr[i], tot = din + colptr[i + 1] - c0;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  for (int t = l; t < tot; t += 64) {
    const bool in = t < din;
    const int e = in ? r0 + t : perm[c0 + t - din];
    float ox, oy, oz;
    edge_force(e, geo, g_rbf, drbf, Rp, g_geo, S, ox, oy, oz);
    fx += in ? ox : -ox;
    fy += in ? oy : -oy;
    fz += in ? oz : -oz;
  }
  fx = wave_sum(fx);
  fy = wave_sum(fy);
  fz = wave_sum(fz);
  if (*status == epoch) fx = fy = fz = __builtin_nanf("");  // invalid input (atomic number out of range): fail loudly
  if (l == 0) {
    f_out[3 * (size_t)i + 0] = fx;
    f_out[3 * (size_t)i + 1] = fy;
    f_out[3 * (size_t)i + 2] = fz;
  }
}

// env VSN_GRAPH_FUSED=0: the three separate launches at single-protein sizes too (A/B aid)
static int g_graph_fused = [] {
  const char* e = getenv("VSN_GRAPH_FUSED");
  return e ? atoi(e) : 1;
}();

int launch_graph(hipStream_t st, const GraphArgs& a) {
  if (a.B <= 0 || a.N <= 0) return 0;
  if (g_graph_fused && a.max_frag <= 64 && 3 * a.N <= VSN_GRAPH_POOL && a.B <= VSN_GRAPH_MAXB && a.max_nb <= 32 &&
      a.Rp <= 64 && VSN_GEO_W <= 32) {
    hipLaunchKernelGGL(k_graph_small_all<0>, dim3(a.B), dim3(64 * VSN_GRAPH_WAVES), 0, st, a);
    return 0;
  }
  if (a.max_frag <= 64) {
    hipLaunchKernelGGL(k_graph_count, dim3(a.B), dim3(64), 0, st, a.pos, a.z64, a.fstart, a.fend, a.deg, a.zi, a.rc2,
                       a.max_nb, a.z_limit, a.status, a.epoch);
    if (a.N < 4096) {  // single-protein sizes: the fill kernel scans the degrees itself (one launch less)
      hipLaunchKernelGGL(k_graph_fill_small<true>, dim3(a.B), dim3(64), 0, st, a.pos, a.fstart, a.fend, a.rowptr,
                         a.src, a.tgt, a.colptr, a.perm, a.rc2, a.max_nb, a.deg, a.B, a.N, a.ecount);
    } else {
      hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, a.deg, a.rowptr, a.colptr, a.N, a.ecount);
      hipLaunchKernelGGL(k_graph_fill_small<false>, dim3(a.B), dim3(64), 0, st, a.pos, a.fstart, a.fend, a.rowptr,
                         a.src, a.tgt, a.colptr, a.perm, a.rc2, a.max_nb, a.deg, a.B, a.N, a.ecount);
    }
  } else {
    // a fragment larger than a wavefront somewhere in the batch: node-parallel passes (any mix of sizes)
    const dim3 grid((a.N + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_graph_count_big, grid, blk, 0, st, a.pos, a.z64, a.fstart, a.fend, a.B, a.N, a.deg, a.zi,
                       a.rc2, a.max_nb, a.z_limit, a.status, a.epoch);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, a.deg, a.rowptr, a.colptr, a.N, a.ecount);
    hipLaunchKernelGGL(k_graph_fill_big, grid, blk, 0, st, a.pos, a.fstart, a.fend, a.B, a.N, a.rowptr, a.src, a.tgt,
                       a.rc2, a.max_nb);
    // deg is free again: reuse it for the out-degrees, whose exclusive scan is colptr
    hipLaunchKernelGGL(k_graph_bysrc_big<false>, grid, blk, 0, st, a.pos, a.fstart, a.fend, a.B, a.N, a.rowptr, a.src,
                       a.deg, nullptr, nullptr, a.rc2);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, a.deg, a.colptr, a.colptr, a.N, a.ecount);
    hipLaunchKernelGGL(k_graph_bysrc_big<true>, grid, blk, 0, st, a.pos, a.fstart, a.fend, a.B, a.N, a.rowptr, a.src,
                       nullptr, a.colptr, a.perm, a.rc2);
  }
  long long tot = (long long)a.Emax * a.Rp;
  int blocks = (int)((tot + 255) / 256);
  if (blocks > 0)
    hipLaunchKernelGGL(k_edge_geom, dim3(blocks), dim3(256), 0, st, a.pos, a.src, a.tgt, a.ecount, a.means, a.betas,
                       a.rbf_type, a.R, a.Rp, a.rc, a.alpha, a.S, a.geo, a.d, a.rbf, a.drbf, a.g_geo);
  return 0;
}

// reduce_op = "mean" (ViSNet/model/visnet.py:146: scatter(x, batch, reduce="mean")): the per-fragment energy is the
// MEAN of the atomic terms, E_b = (sum_i y_i) / n_b + mean.  Fragments are independent, so against the "add" evaluation
// this is a per-fragment scale of the energy sum and of the forces of the fragment's atoms - one wave per fragment.
__global__ __launch_bounds__(256) void k_reduce_mean(int B, const int* __restrict__ fstart,
                                                     const int* __restrict__ fend, float mean,
                                                     float* __restrict__ e_out, float* __restrict__ f_out) {
  const int b = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), l = threadIdx.x & 63;
  if (b >= B) return;
  const int a0 = fstart[b], a1 = fend[b];
  if (a1 <= a0) return;  // (an empty fragment keeps `mean`, as in the "add" evaluation)
  const float inv = 1.0f / (float)(a1 - a0);
  if (l == 0) e_out[b] = (e_out[b] - mean) * inv + mean;
  for (int k = 3 * a0 + l; k < 3 * a1; k += 64) f_out[k] *= inv;
}
int launch_reduce_mean(hipStream_t st, int B, const int* fstart, const int* fend, float mean, float* e_out,
                       float* f_out) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(k_reduce_mean, dim3((B + 3) / 4), dim3(256), 0, st, B, fstart, fend, mean, e_out, f_out);
  return 0;
}

int launch_bwd_geom(hipStream_t st, const GraphArgs& a, const float* g_rbf, const float* g_geo, float* g_ev,
                    float* f_out, bool keep_g_ev, const EnergyFold& ef) {
  if (a.N <= 0) return ef.B > 0 ? -22 : 0;
  if (a.N < 4096 && !keep_g_ev) {  // single-protein sizes: one launch, g_ev is not materialised
    const int blocks = std::max((a.N + 3) / 4, (ef.B + 3) / 4);
    hipLaunchKernelGGL(k_force_gather_geom, dim3(blocks), dim3(256), 0, st, a.N, a.rowptr, a.colptr, a.perm, a.geo,
                       g_rbf, a.drbf, a.Rp, g_geo, a.S, f_out, a.status, a.epoch, ef);
    return 0;
  }
  if (ef.B > 0) return -22;  // (the head only defers its energy sums when this launch takes the fused form)
  int blocks = (a.Emax + 255) / 256;
  if (blocks > 0)
    hipLaunchKernelGGL(k_bwd_geom, dim3(blocks), dim3(256), 0, st, a.ecount, a.geo, g_rbf, a.drbf, a.Rp, g_geo, a.S,
                       g_ev);
  hipLaunchKernelGGL(k_force_gather, dim3((a.N + 15) / 16), dim3(256), 0, st, a.N, a.rowptr, a.colptr, a.perm,
                     g_ev, f_out, a.status, a.epoch);
  return 0;
}

}  // namespace vsn
