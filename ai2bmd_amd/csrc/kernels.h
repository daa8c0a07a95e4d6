// Host-side launch interface between the engine (engine.cpp) and the kernel
// translation units.  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <vector>

namespace vsn {

struct GraphArgs {
  const float* pos;
  const long long* z64;
  const int* fstart;
  const int* fend;
  int B, N, Emax, max_frag;
  float rc, rc2, alpha;
  int max_nb, R, Rp, S;
  int rbf_type;        // 0 expnorm (means, betas), 1 gauss (means = offset, betas[0] = coeff)
  const float* means;
  const float* betas;
  int* deg;
  int* zi;
  int* rowptr;
  int* colptr;
  int* src;
  int* tgt;
  int* perm;
  int* ecount;
  int z_limit;   // valid atomic numbers: 0 <= z < z_limit (embedding rows, atomref rows)
  int* status;   // device word: epoch of the last chunk that saw an atomic number out of range (never cleared)
  int epoch;     // this chunk's epoch: *status == epoch <=> this chunk is invalid
  float* g_geo;  // [E,VSN_GEO_W] reverse-pass accumulators, cleared per live edge by k_edge_geom
  float* geo;   // [E,8]  r, C, dC, ux, uy, uz, 1/r, 0
  float* d;     // [E,8]  spherical harmonics (first S used)
  float* rbf;   // [E,Rp]
  float* drbf;  // [E,Rp]
};

// g_geo row: dE/dd (vector messages) 0..7, dE/dC 8, dE/dd (edge update) 16..23 and 24..31 (one group per channel half)
#define VSN_GEO_W 32

// Everything the per-layer kernels need (one chunk).
struct Dims {
  int N, Emax, H, S, nh, R, Rp;
  int act, attn_act;  // VSN_ACT_* of hparams "activation" / "attn_activation" (common.h)
  // heads: hd = H / nh channels per head.  hgen = 0: nh divides 64 (a head = 64 / nh whole lanes: group_sum fast path);
  // hgen = 1: any other nh with H % nh == 0 (visnet_block.py:158-166 asks no more) - a lane's channels may straddle
  // heads; the kernels that form per-head sums take their generic instantiation (head_sums_any, common.h)
  int hd, hgen;
  const int* ecount;
  const int* rowptr;
  const int* colptr;
  const int* src;
  const int* tgt;
  const int* perm;
  const int* zi;
  const float* geo;
  const float* d;
};

struct GemmDesc {
  const float* A;
  const float* Bt;
  float* C;
  const float* bias;
  const int* Mptr;
  float* part;
  int lda, ldb, ldc, M, Nc, K, flags, ksplit, blocks;
  // keep_parts > 1: cut K into exactly that many slices and LEAVE the partial products in `part`
  // ([keep_parts][M][Nc], M = the host-side row bound) - no reduction launch; the consumer kernel adds the
  // slices up in a fixed order while it reads them (grouped launches only)
  int keep_parts;
};
struct GemmGroup {
  static constexpr int MAXP = 4;
  GemmDesc p[MAXP];
  int n;
};
inline GemmDesc gemm_desc(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, const float* bias,
                          int M, const int* Mptr, int Nc, int K, int flags) {
  GemmDesc d;
  d.A = A;
  d.Bt = Bt;
  d.C = C;
  d.bias = bias;
  d.Mptr = Mptr;
  d.part = nullptr;
  d.lda = lda;
  d.ldb = ldb;
  d.ldc = ldc;
  d.M = M;
  d.Nc = Nc;
  d.K = K;
  d.flags = flags;
  d.ksplit = 1;
  d.blocks = 0;
  d.keep_parts = 0;
  return d;
}

struct GemmProfiler {
  struct Rec {
    hipEvent_t a, b;
    int variant, M;
    bool dev_m;
    double flops_per_row, bytes_per_row;
    int group_n;  // > 0: grouped launch, per-member rows/flops below
    int gM[4];
    bool gdev[4];
    double gflops[4], gbytes[4];
  };
  std::vector<Rec> recs;
};
void set_gemm_profiler(GemmProfiler* p);  // thread-local; nullptr disables
// profile mode, scatter path: the NEXT launch of k_edge_attn / k_edge_attn_update / k_node_update on this thread
// carries these two events on its dispatch packet (their elapsed time = the kernel's begin..end timestamps)
// Every queued pair is TAGGED with the walk it is meant for: a launch only takes a pair of its own kind, so routing
// another launch through launch_maybe_timed (or skipping one) cannot charge time and bytes to the wrong kernel.
enum WalkKind { WK_EDGE_ATTN = 0, WK_NODE_UPDATE = 1, WK_BWD_HF1 = 2, WK_BWD_HF2 = 3, WK_BWD_ATTN_S = 4, WK_BWD_NORM_UPDATE = 5 };
struct LaunchEvents {
  hipEvent_t a, b;
  int kind;  // WalkKind
};
// thread-local FIFO (layer_fwd.hip): a launch that goes through launch_maybe_timed(kind, ...) takes the front pair IF it
// carries the same kind.  Timed-capable launches: k_edge_attn[_update], k_node_update (forward); k_bwd_hf1, k_bwd_hf2,
// k_bwd_attn_S, k_bwd_norm_update (reverse walks).  set_launch_events(nullptr) empties the queue.
void set_launch_events(const LaunchEvents* ev);  // nullptr: clear; else: clear + push one
void push_launch_events(const LaunchEvents& ev);
bool take_launch_events(int kind, LaunchEvents* out);
template <typename... KA, typename... A>
static inline void launch_maybe_timed(int kind, void (*kern)(KA...), dim3 g, dim3 b, unsigned lds, hipStream_t st,
                                      A... a) {
  LaunchEvents ev;
  if (take_launch_events(kind, &ev)) {
    hipExtLaunchKernelGGL<KA...>(kern, g, b, lds, st, ev.a, ev.b, 0, static_cast<KA>(a)...);
  } else {
    hipLaunchKernelGGL(kern, g, b, lds, st, static_cast<KA>(a)...);
  }
}
void set_gemm_splitk_workspace(float* p, size_t elems);  // thread-local scratch for split-K partials
// opt-in product mode gemm_split3 (gemm_s3.h): the grouped launches of this thread run their members as 3 x bf16 split
// products while a table is set; the table caches the packed weight planes of one engine
struct Split3Table;
Split3Table* split3_table_create();
void split3_table_destroy(Split3Table* t);
void set_gemm_split3(Split3Table* t);  // thread-local; nullptr (default) = fp32 MFMA products

int set_panel_tp(int v);  // fused.hip: team-phased fused panel products - LAB BUILDS ONLY (-1 when asked for in a product build)
int launch_gemm(hipStream_t st, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc,
                const float* bias, int M, const int* Mptr, int Nc, int K, int flags);
// independent products in one launch (falls back to separate launches when not worthwhile)
int launch_gemm_group(hipStream_t st, const GemmDesc* descs, int n);
// the predicate launch_gemm_group uses: true = one grouped launch (callers that set keep_parts must check it)
bool gemm_group_ok(const GemmDesc* descs, int n);

int launch_graph(hipStream_t st, const GraphArgs& a);
// per-fragment energies e_out[b] = sum_{i in fragment b} y[i] + mean, folded into the force gather (B = 0: nothing)
struct EnergyFold {
  int B;
  const int *fstart, *fend;
  const float* y;
  float mean;
  float* e_out;
};
// keep_g_ev: the debug tap wants the per-edge adjoint in HBM; otherwise single-protein sizes fold it into the gather
int launch_bwd_geom(hipStream_t st, const GraphArgs& a, const float* g_rbf, const float* g_geo, float* g_ev,
                    float* f_out, bool keep_g_ev, const EnergyFold& ef);

// reduce_op = "mean": per-fragment scale of the finished "add" evaluation (energies and forces), graph.hip
int launch_reduce_mean(hipStream_t st, int B, const int* fstart, const int* fend, float mean, float* e_out,
                       float* f_out);

// ---- forward ----
int launch_embed_node(hipStream_t st, const Dims& D, const float* emb1, const float* emb2, const float* pp,
                      float* cat);
// LayerNorm + VecLayerNorm("none") of the NEXT layer (or the read-out) fused into the node update that
// produces their input; xn == nullptr disables the fusion
struct NextNorm {
  const float *gamma, *beta, *wvec;
  float *xn, *rstd, *xh, *vh;
  int ldxh;
};
// nn.xn != nullptr: layer 0's norms (LayerNorm of x; vh = 0) ride in the same launch
int launch_embed_edge(hipStream_t st, const Dims& D, const float* x, const float* pp, float* f, float* vec,
                      float* xcopy, const NextNorm& nn);
int launch_node_norm(hipStream_t st, const Dims& D, const float* x, const float* vec, const float* gamma,
                     const float* beta, const float* wvec, int norm_type, float* xn, float* rstd, float* xh,
                     int ldxh, float* vh);
int launch_edge_attn(hipStream_t st, const Dims& D, const float* qkv, const float* pe, float* m, float* A);
int launch_node_update(hipStream_t st, const Dims& D, const float* tpre, const float* vh, const float* vp,
                       const float* o, float* x, float* vec, const NextNorm& nn);
// adjoint of LayerNorm (+ VecLayerNorm "none") of layer l fused with the adjoint of the node update of
// layer l-1 (both are node-local); see k_bwd_norm_update
int launch_bwd_norm_update(hipStream_t st, const Dims& D, const float* g_xh, int ldg, const float* g_vh,
                           const float* xn, const float* rstd, const float* gamma, const float* wvec,
                           int accumulate, float* g_x, float* g_vec, const float* vp, const float* o, float* g_o,
                           float* g_vp);
int launch_edge_attn_update(hipStream_t st, const Dims& D, const float* qkv, const float* pe, float* m, float* A,
                            const float* vp, float* f);
int launch_edge_update(hipStream_t st, const Dims& D, const float* vp, const float* pe, float* f);

// ---- reverse ----
int launch_bwd_node_update(hipStream_t st, const Dims& D, const float* g_x, const float* g_vec, const float* vp,
                           const float* o, float* g_o, float* g_vp);
int launch_bwd_side(hipStream_t st, const Dims& D, const float* vp, const float* pe, const float* g_f, float* g_pe,
                    float* g_vp, float* g_geo, const float* g_vec, const float* tpre, float* g_vh);
int launch_bwd_edge_update(hipStream_t st, const Dims& D, const float* vp, const float* pe, const float* g_f,
                           float* g_pe, float* g_vp, float* g_geo);
int launch_bwd_vecmsg(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre,
                      float* g_t, float* g_vh, float* g_geo);
int launch_bwd_vecmsg_S(hipStream_t st, const Dims& D, const float* g_vec, const float* tpre, float* g_vh);
// K-slices of a product left un-summed by the GEMM (GemmDesc::keep_parts): slice k of row r is p + k * stride + r * ld
struct Parts {
  const float* p;
  size_t stride;
  int n;  // 0: not split, read the plain array
};
bool bwd_streamless_ok(const Dims& D);
bool bwd_batch_path(const Dims& D);  // one wave per node (fragment batches), not the several-waves-per-node kernels
int launch_bwd_hf1(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre, float* g_t,
                   float* g_geo, const float* vp, const float* pe, const float* g_f, float* g_pe, float* g_vp,
                   float* g_vh, bool with_edge_update);
int launch_bwd_hf2(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A, float* g_m,
                   float* g_pe, float* g_qkv, float* sat_tmp, float* g_geo, Parts g_m_parts, Parts g_A_parts,
                   const float* vp, const float* g_f, float* g_vp);
// g_m_parts / g_A_parts: when n > 0 the incoming dE/dm rows / dE/dA rows are read as sums of those slices
int launch_bwd_attn(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A, float* g_m,
                    float* g_pe, float* g_qkv, float* sat_tmp, float* g_geo, Parts g_m_parts = Parts{nullptr, 0, 0},
                    Parts g_A_parts = Parts{nullptr, 0, 0});
int launch_bwd_attn_QS(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_m,
                       const float* sat_tmp, float* g_qkv);

// ---- fused.hip: gather kernels as prologues of panel GEMMs (fragment batches, hidden = 256) ----
bool panel_ok(const Dims& D);
// g_m = g_t . Ws with g_t (adjoint of the vector messages, target side) produced on the fly; WsTp = packed + K-permuted
int launch_bwd_gm_fused(hipStream_t st, const Dims& D, const float* g_vec, const float* vh, const float* tpre,
                        const float* WsTp, float* g_m, float* g_geo);
// g_f (+)= g_pe . We3 with the attention part of g_pe produced on the fly (g_pf read from g_pe[:, 2H:3H] when K = 3H);
// leaves g_m += g_A and sat_tmp = [g_sat | a] for launch_bwd_attn_QS
int launch_bwd_gf_fused(hipStream_t st, const Dims& D, const float* qkv, const float* pe, const float* g_A,
                        float* g_m, const float* g_pe, float* sat_tmp, float* g_geo, const float* We3Tp, float* g_f,
                        int K, int accumulate);
int launch_bwd_node_norm(hipStream_t st, const Dims& D, const float* g_xh, int ldg, const float* g_vh, const float* xn,
                         const float* rstd, const float* gamma, const float* wvec, int norm_type, int accumulate,
                         float* g_x, float* g_vec);
// g_xh != nullptr: layer 0's LayerNorm adjoint (g_x += ...) rides in the same launch
int launch_bwd_embed_edge(hipStream_t st, const Dims& D, const float* x, const float* pp, const float* g_f,
                          float* g_pp, float* g_x, const float* g_xh, const float* xn, const float* rstd,
                          const float* gamma);
int launch_bwd_embed_node(hipStream_t st, const Dims& D, const float* emb2, const float* pp, const float* g_n,
                          float* g_pp, float* g_geo);

// ---- read-out head ----
struct HeadW {
  const float *Wpv0, *Wpv0T;  // [H+h2,H], [H,H+h2]
  const float *Wa0, *ba0, *Wa0T;
  const float *Wb0, *bb0, *Wb0T;
  const float *W11, *W11T;
  const float *Wa1, *ba1, *Wa1T;
  // the same matrices in the fused head's MFMA 16x16x4 fragment order (head_fused.hip::lds_gemm: 1-KiB blocks of 16
  // columns x 16 k, one coalesced 16-byte load per lane); null when the fused head is not applicable
  const float *Wa0p, *Wb0p, *W11p, *Wa1p, *Wa1Tp, *W11Tp, *Wb0Tp, *Wa0Tp;
  const float* wb1;  // [h2] row 0 of update_net.2 of block 1
  float bb1, mean, stdv;
  const float* atomref;  // [Z] or null
  const int* status;     // chunk status word and epoch (see GraphArgs): *status == epoch -> energies are NaN
  int epoch;
  int fuse;              // 1: single-protein sizes take the fused head kernel (head_fused.hip) when it fits in LDS
  int defer_energy;      // 1 (fused head only): the per-fragment energy sums ride in the evaluation's LAST launch
                         // (launch_bwd_geom's EnergyFold) instead of a launch of their own
};
struct HeadBuf {
  float *cat0, *pv0, *a0, *u0, *vec1o, *cat1, *p1, *a1b, *y;        // forward
  float *g_a1, *g_cat1, *g_p1, *g_vec1o, *g_u0, *g_h0, *g_cat0, *g_pv0;  // reverse
};
// true when launch_head_forward will leave the energy sums to launch_bwd_geom (both sides ask the same predicate)
bool head_defers_energy(const Dims& D, const HeadW& W);
// expects out_norm(x) already in Bf.cat0[:, :H] (row stride 2H)
int launch_head_forward(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf, const float* vo,
                        const int* fstart, const int* fend, int B, float* e_out);
// leaves dE/d out_norm(x) in Bf.g_cat0[:, :H] (row stride 2H), dE/d vec_out_norm(vec) in g_vo
int launch_head_backward(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf, float* g_vo);

// the node-local middle of the head (forward AND reverse) in one launch; see head_fused.hip
bool head_fused_supported(const Dims& D);
int launch_head_fused(hipStream_t st, const Dims& D, const HeadW& W, const HeadBuf& Bf);

int launch_fill(hipStream_t st, float* p, size_t n, float v);
// VecLayerNorm rms (1) / max_min (2): vh = norm(vec), also saves vec into vin for the adjoint
int launch_vecnorm_fwd(hipStream_t st, int N, int H, int S, int norm_type, const float* vec, const float* w,
                       float* vin, float* vh);
int launch_vecnorm_bwd(hipStream_t st, int N, int H, int S, int norm_type, const float* vin, const float* w,
                       const float* g_vh, int accumulate, float* g_vec);

// ---- combine (Calculators/combiner.py:24-41) ----
int launch_combine(hipStream_t st, int n_prot, const int* off, const int* rows, const float* sign,
                   const float* f_frag, float* f_prot, int n_e = 0, const int* e_idx = nullptr,
                   const float* e_sign = nullptr, float* e_out = nullptr);

int launch_build_fragments(hipStream_t st, int n, const int* src, const int* acc, const int* tow, const float* len,
                           const float* prot, float* out);

}  // namespace vsn
