// Host-side engine + C ABI (include/vsn.h) of the MI355X ViSNet calculator.
//
// Owns: packed weights (fused / padded / pre-transposed for the reverse pass),
// the per-chunk workspace arena, and the launch sequence of one energy+force
// evaluation (graph -> embeddings -> L x ViS-MP -> read-out -> hand-written
// reverse pass -> forces).  Replaces the TorchScript module + autograd the
// reference runs in ViSNetModel.dl_potential_loader
// (/root/reference/src/Calculators/visnet_calculator.py:54-63,
//  /root/reference/src/ViSNet/model/visnet.py:135-166).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/vsn.h"
#include "kernels.h"
#include "pgemm.h"
#include "tail.h"

using namespace vsn;

#define VSN_MAX_FRAG_ATOMS 262144  // int32 edge ids: n * max_num_neighbors must stay below 2^31

namespace {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

struct LayerW {
  float *Wqkv, *bqkv, *WqkvT;
  float *Wv5, *Wv5T;
  float *We3, *be3, *We3T;
  float *Ws, *bs, *WsT;
  float *Wo, *bo, *WoT;
  float *ln_g, *ln_b, *vln_w;
  // panel-GEMM copies (pgemm.h; hidden = 256 only, else null): MFMA-fragment order, K permuted by channel half for
  // the fused products of fused.hip
  float *WsTp, *We3Tp;
};

struct LayerBuf {
  float *xn, *rstd, *vh, *qkv, *vp, *pe, *tpre, *o;
  float* vin;  // pre-norm vec, saved only for vecnorm rms / max_min
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
  bool dry = true;
  template <typename T>
  T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += bytes;
    return p;
  }
};

}  // namespace

struct vsn_ctx {
  vsn_hparams hp;
  int device = 0;
  std::string err;
  std::map<std::string, HostTensor> raw;
  bool finalized = false;
  int H = 0, L = 0, R = 0, Rp = 0, S = 0, nh = 0, Z = 0;
  // packed weights
  float* warena = nullptr;
  float *emb1, *emb2, *means, *betas, *Wrbf, *brbf, *WrbfT, *Wc, *bc, *WcnT, *on_g, *on_b, *vo_w;
  std::vector<LayerW> lw;
  HeadW hw;
  // workspace
  Arena ws;
  int capN = 0, capE = 0, capB = 0;
  bool debug = false;
  bool profile = false;
  double prof[4][4] = {{0}};  // per GEMM kernel (128x128, 64x64, 128x32, grouped 64x64): launches, ms, flops, bytes
  // profile mode, scatter path: brackets around the forward edge-attention (0) and node-update (1) launches:
  // {launches, ms, algorithmic bytes, 0}
  // (+ round 5) the reverse node walks of single-protein sizes: k_bwd_hf1 (2), k_bwd_hf2 (3), k_bwd_attn_S (4),
  // k_bwd_norm_update (5)
  static constexpr int NWALK = 6;
  double sprof[NWALK][4] = {{0}};
  struct SRec {
    hipEvent_t a, b;
    int kind, N;
    int f0, f1;  // kind 0: with_update; 1: fused_norm; 2: with_eu; 3: K-slices of g_m, of g_A; 5: accumulate
  };
  std::vector<SRec> srecs;
  double prof_empty_ms = 0;   // profile mode: total time of EMPTY event brackets (the cost an event pair adds) ...
  double prof_empty_n = 0;    // ... and how many were measured
  // edge SLOTS per chunk (host bound sum n*min(n, max_nb)); ~105 GB of workspace at H=256, L=9.  Swept on the
  // 4096-fragment batch (1.9 M slots, 1.37 M edges), fragments/s: 262144 12.6k, 524288 13.0k, 786432 13.2k, 1048576 13.4k,
  // 1310720 13.5k, 1572864 13.3k, 2097152 (ONE chunk, 170 GB) 12.0k - past ~1.3 M slots the per-layer arrays outgrow
  // what the Infinity Cache / TLB reach keeps warm between producer and consumer kernels
  int64_t max_chunk_edges = 1310720;
  // buffers
  int *fstart, *fend, *deg, *zi, *rowptr, *colptr, *src, *tgt, *perm, *ecount;
  float *geo, *d, *rbf, *drbf;
  float *pp, *cat, *x_emb, *x, *vec, *f;
  std::vector<LayerBuf> lb;
  float *xh, *m, *A;
  float *xn_o, *rstd_o, *vo, *vin_o;
  HeadBuf hb;
  float* g_vo;
  float *g_x, *g_vec, *g_f;
  float *g_o, *g_vp, *g_A, *g_t, *g_m, *g_pe, *g_qkv, *g_vh, *g_xh, *sat_tmp;
  float *g_pp, *g_n, *g_rbf, *g_geo, *g_ev;
  float* splitk;
  size_t splitk_elems = 0;
  // second stream for work that is independent of the main per-layer chain (edge update and its adjoints)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // bit 0: forward edge update on the side stream, bit 1: reverse-pass side work.  Measured on Chignolin
  // (steps/s): none 339, forward only 327, reverse only 349, both 350 - a fork/join costs ~15 us of event latency,
  // more than the 11-16 us forward edge update it hides, so only the reverse pass (50-60 us of side work per layer) forks.
  // bit 1 at single-protein sizes = the streamless form (side kernels ride in k_bwd_hf1 / k_bwd_hf2 of the main chain);
  // bit 2 (round 5, default OFF) = fragment batches put the edge-update adjoints + source side of the vector messages on
  // the side stream: 13.39 k fragments/s with, 13.39 k without (tools/lab/batch_ab.sh) - the fused panel products next
  // to them just run twice as long (3.4 vs 2.0 ms), so the default keeps one stream.
  int overlap = 2;
  bool fuse_fwd = true, fuse_bwd_opt = true;
  bool fuse_head = true;  // fused node-local head kernel (head_fused.hip) on single-protein sizes
  bool split_rev = true;  // K-slices of the g_m / g_A products summed by their consumer (single-protein sizes)
  bool fuse_panel = true;  // fragment batches, hidden 256: gather kernels as prologues of panel GEMMs (fused.hip)
  int64_t panel_min_edges = (int64_t)1 << 40;  // ... and below the batch regime from this many edge slots on (A/B aid: off)
  bool reduce_mean = false;   // hparams reduce_op == "mean" (visnet.py:146): per-fragment mean instead of sum
  bool gemm_split3 = false;   // opt-in: grouped products as 3 x bf16 split MFMA products (gemm_s3.h); default fp32 MFMA
  Split3Table* s3 = nullptr;  // ... and the packed bf16 planes of this engine's weights (made on first use)
  // debug snapshots: name -> per-layer device copies
  std::map<std::string, std::vector<float*>> snap;
  std::map<std::string, size_t> snap_elems;
  // last chunk dims
  int lastN = 0, lastB = 0, lastEmax = 0;
  // fragment offsets of the last chunk as uploaded (MD: static fragmentation -> the upload is skipped).  The
  // cache is valid only for the workspace it was uploaded into and the stream it was ordered on.
  std::vector<int> h_fs, h_fe;
  hipStream_t h_fs_stream = nullptr;
  int* pin_fs = nullptr;       // pinned staging for the upload (2 x pin_cap ints)
  size_t pin_cap = 0;
  hipEvent_t ev_upload = nullptr;  // recorded after the last upload: the staging buffer is free once it has fired
  int n_atomref = 0;           // rows of the Atomref table (prior_args.max_z; independent of hparams.max_z)
  int* status = nullptr;       // device status word: epoch of the last chunk that saw an invalid atomic number
  int epoch = 0;               // chunk counter (the status word is compared with it instead of being cleared)
};

static int fail(vsn_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define HIPCHK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return fail(c, -5, std::string(#call) + ": " + hipGetErrorString(e__));                    \
  } while (0)

extern "C" int vsn_create(vsn_handle* out, const vsn_hparams* hp, int device_id) {
  if (!out || !hp) return -22;
  vsn_ctx* c = new vsn_ctx();
  c->hp = *hp;
  c->device = device_id;
  c->H = hp->hidden;
  c->L = hp->num_layers;
  c->R = hp->num_rbf;
  c->Rp = (hp->num_rbf + 31) / 32 * 32;
  c->S = (hp->lmax + 1) * (hp->lmax + 1) - 1;
  c->nh = hp->num_heads;
  c->Z = hp->max_z;
  *out = c;
  if (c->H < 64 || c->H > 512 || (c->H % 64)) return fail(c, -22, "hidden must be a multiple of 64 in [64, 512]");
  if (hp->rbf_type < 0 || hp->rbf_type > 1) return fail(c, -22, "unknown rbf_type");
  if (hp->activation < 0 || hp->activation > 3 || hp->attn_activation < 0 || hp->attn_activation > 3)
    return fail(c, -22, "unknown activation");
  if (hp->num_rbf < 1) return fail(c, -22, "num_rbf must be >= 1");
  if (!(hp->lmax == 1 || hp->lmax == 2)) return fail(c, -22, "lmax must be 1 or 2");
  // the reference asks only hidden % num_heads == 0 (visnet_block.py:158-166); head counts that divide 64 take the
  // lane-group fast path, any other the generic per-head sums (Dims::hgen)
  if (c->nh <= 0 || c->nh > 64 || (c->H % c->nh))
    return fail(c, -22, "num_heads must divide hidden (and be <= 64)");
  if (c->L < 1) return fail(c, -22, "num_layers must be >= 1");
  if (hp->vecnorm_type < 0 || hp->vecnorm_type > 2) return fail(c, -22, "unknown vecnorm_type");
  if (hipSetDevice(device_id) != hipSuccess) return fail(c, -19, "hipSetDevice failed (no MI355X visible?)");
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming) != hipSuccess)
    return fail(c, -5, "stream/event creation failed");
  return 0;
}

extern "C" void vsn_destroy(vsn_handle c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->warena) hipFree(c->warena);
  if (c->ws.base) hipFree(c->ws.base);
  split3_table_destroy(c->s3);
  if (c->side) hipStreamDestroy(c->side);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_upload) hipEventDestroy(c->ev_upload);
  if (c->pin_fs) hipHostFree(c->pin_fs);
  for (auto& kv : c->snap)
    for (float* p : kv.second)
      if (p) hipFree(p);
  delete c;
}

extern "C" const char* vsn_last_error(vsn_handle c) { return c ? c->err.c_str() : "null handle"; }

extern "C" int vsn_load_weight(vsn_handle c, const char* name, const void* ptr, const int64_t* shape, int ndim) {
  if (!c || !name || !ptr) return -22;
  HIPCHK(c, hipSetDevice(c->device));
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.data.resize(n);
  HIPCHK(c, hipMemcpy(t.data.data(), ptr, n * sizeof(float), hipMemcpyDefault));
  c->raw[name] = std::move(t);
  c->finalized = false;
  return 0;
}

extern "C" int vsn_set_option(vsn_handle c, const char* key, int64_t value) {
  if (!c || !key) return -22;
  std::string k(key);
  if (k == "max_chunk_edges") {
    if (value < 1024) return fail(c, -22, "max_chunk_edges too small");
    c->max_chunk_edges = value;
  } else if (k == "debug") {
    c->debug = value != 0;
  } else if (k == "fuse_fwd") {
    c->fuse_fwd = value != 0;
  } else if (k == "fuse_bwd") {
    c->fuse_bwd_opt = value != 0;
  } else if (k == "fuse_head") {
    c->fuse_head = value != 0;
  } else if (k == "split_rev") {
    c->split_rev = value != 0;
  } else if (k == "fuse_panel") {
    c->fuse_panel = value != 0;
  } else if (k == "panel_min_edges") {
    c->panel_min_edges = value;
  } else if (k == "reduce_mean") {
    c->reduce_mean = value != 0;
  } else if (k == "gemm_split3") {
    // opt-in arithmetic mode (default 0 = fp32 MFMA everywhere): the grouped products of single-protein sizes as
    // 3 x bf16 split products with fp32 accumulation (gemm_s3.h).  So do the plain products that take the 128x128 batch
    // tile (k_gemm3_128: rbf / embedding / read-out / per-layer linears of fragment batches); the fused panel products
    // (fused.hip) and the 64x64 / split-K plain launches keep their fp32 MFMA kernels.
    c->gemm_split3 = value != 0;
    if (c->gemm_split3 && !c->s3) c->s3 = split3_table_create();
  } else if (k == "panel_tp") {
    // team-phased fused panel products (fused.hip): a LAB-BUILD variant (-DVSN_LAB_ABL=1), absent from the product
    if (set_panel_tp((int)value)) return fail(c, -22, "panel_tp exists in lab builds only (-DVSN_LAB_ABL=1)");
  } else if (k == "overlap") {
    c->overlap = (int)value;
  } else if (k == "profile") {
    c->profile = value != 0;
    memset(c->prof, 0, sizeof(c->prof));
    memset(c->sprof, 0, sizeof(c->sprof));
    c->prof_empty_ms = c->prof_empty_n = 0;
  } else {
    return fail(c, -22, "unknown option " + k);
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------
namespace {
struct Packer {
  std::vector<float> host;
  size_t add(const std::vector<float>& v) {
    size_t off = host.size();
    host.insert(host.end(), v.begin(), v.end());
    while (host.size() % 64) host.push_back(0.f);  // 256-byte alignment
    return off;
  }
};
std::vector<float> transposed(const std::vector<float>& w, int rows, int cols) {
  std::vector<float> t((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int cidx = 0; cidx < cols; ++cidx) t[(size_t)cidx * rows + r] = w[(size_t)r * cols + cidx];
  return t;
}
// [Nc, K] weight matrix (row stride ld >= K) -> panel-GEMM fragment order (pgemm.h), logical column k placed at
// kperm(k); padded for the kernels' read-ahead
template <typename F>
std::vector<float> pack_panel(const std::vector<float>& w, int Nc, int K, int ld, F kperm) {
  std::vector<float> o((size_t)Nc * K + VSN_PGEMM_PAD_FLOATS, 0.f);
  for (int n = 0; n < Nc; ++n)
    for (int k = 0; k < K; ++k) o[pgemm_pack_index(n, kperm(k), K)] = w[(size_t)n * ld + k];
  return o;
}
// [Nc, K] weight matrix -> fragment order of the fused head's 16x16x4 MFMA products (head_fused.hip::lds_gemm):
// block (cb = n / 16, kg = k / 16) = 64 lanes x 4 floats, lane = (n % 16) + 16 * ((k % 16) / 4), component k % 4
std::vector<float> pack_head(const std::vector<float>& w, int Nc, int K) {
  std::vector<float> o((size_t)Nc * K + 16 * 256, 0.f);  // + read-ahead of the weight ring
  const int KG = K / 16;
  for (int n = 0; n < Nc; ++n)
    for (int k = 0; k < K; ++k)
      o[(((size_t)(n >> 4) * KG + (k >> 4)) * 64 + (n & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)] = w[(size_t)n * K + k];
  return o;
}
// K layout of the fused products: slice h (256 columns) = channels [128 h, 128 h + 128) of operand part 0, then of
// part 1; a third part (columns 512..767) stays where it is
inline int kperm_half(int k) {
  if (k >= 512) return k;
  const int part = k >> 8, c = k & 255;
  return (c >> 7) * 256 + part * 128 + (c & 127);
}
std::vector<float> vcat(std::initializer_list<const std::vector<float>*> parts) {
  std::vector<float> o;
  for (auto* p : parts) o.insert(o.end(), p->begin(), p->end());
  return o;
}
}  // namespace

static const std::vector<float>* need(vsn_ctx* c, const std::string& name, size_t elems, std::string& missing) {
  auto it = c->raw.find(name);
  if (it == c->raw.end()) {
    missing = "missing tensor " + name;
    return nullptr;
  }
  if (it->second.data.size() != elems) {
    missing = "tensor " + name + " has " + std::to_string(it->second.data.size()) + " elements, expected " +
              std::to_string(elems);
    return nullptr;
  }
  return &it->second.data;
}

extern "C" int vsn_finalize(vsn_handle c) {
  if (!c) return -22;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->H, L = c->L, R = c->R, Rp = c->Rp, Z = c->Z, h2 = H / 2;
  std::string miss;
  Packer P;
  const std::string rm = "representation_model.";
#define NEED(var, name, elems)                              \
  const std::vector<float>* var = need(c, name, elems, miss); \
  if (!var) return fail(c, -2, miss);

  NEED(emb1, rm + "embedding.weight", (size_t)Z * H);
  // expnorm: means / betas (utils.py:40-46); gauss: offset [R] and the scalar coeff (utils.py:75-79) ride in the
  // same two device arrays (means = offset, betas[0] = coeff)
  const bool gauss = c->hp.rbf_type == 1;
  NEED(means, rm + (gauss ? "distance_expansion.offset" : "distance_expansion.means"), (size_t)R);
  NEED(betas, rm + (gauss ? "distance_expansion.coeff" : "distance_expansion.betas"), (size_t)(gauss ? 1 : R));
  NEED(emb2, rm + "neighbor_embedding.embedding.weight", (size_t)Z * H);
  NEED(Wd, rm + "neighbor_embedding.distance_proj.weight", (size_t)H * R);
  NEED(bd, rm + "neighbor_embedding.distance_proj.bias", (size_t)H);
  NEED(Wc, rm + "neighbor_embedding.combine.weight", (size_t)H * 2 * H);
  NEED(bc, rm + "neighbor_embedding.combine.bias", (size_t)H);
  NEED(We, rm + "edge_embedding.edge_proj.weight", (size_t)H * R);
  NEED(be, rm + "edge_embedding.edge_proj.bias", (size_t)H);
  NEED(on_g, rm + "out_norm.weight", (size_t)H);
  NEED(on_b, rm + "out_norm.bias", (size_t)H);
  NEED(vo_w, rm + "vec_out_norm.weight", (size_t)H);
  NEED(meanv, "mean", 1);
  NEED(stdv, "std", 1);

  std::map<std::string, size_t> off;
  off["emb1"] = P.add(*emb1);
  off["emb2"] = P.add(*emb2);
  {
    std::vector<float> m(Rp, 0.f), b(Rp, 0.f);
    std::copy(means->begin(), means->end(), m.begin());
    std::copy(betas->begin(), betas->end(), b.begin());
    off["means"] = P.add(m);
    off["betas"] = P.add(b);
  }
  {
    // Wrbf [2H, Rp]: rows 0..H-1 distance_proj (phi), rows H..2H-1 edge_proj (psi); K zero-padded
    std::vector<float> W((size_t)2 * H * Rp, 0.f), b(2 * H);
    for (int r = 0; r < H; ++r)
      for (int k = 0; k < R; ++k) {
        W[(size_t)r * Rp + k] = (*Wd)[(size_t)r * R + k];
        W[(size_t)(H + r) * Rp + k] = (*We)[(size_t)r * R + k];
      }
    for (int r = 0; r < H; ++r) {
      b[r] = (*bd)[r];
      b[H + r] = (*be)[r];
    }
    off["Wrbf"] = P.add(W);
    off["brbf"] = P.add(b);
    off["WrbfT"] = P.add(transposed(W, 2 * H, Rp));
  }
  off["Wc"] = P.add(*Wc);
  off["bc"] = P.add(*bc);
  {
    std::vector<float> t((size_t)H * H);  // WcnT[c][k] = Wc[k][H + c]
    for (int k = 0; k < H; ++k)
      for (int cc = 0; cc < H; ++cc) t[(size_t)cc * H + k] = (*Wc)[(size_t)k * 2 * H + H + cc];
    off["WcnT"] = P.add(t);
  }
  off["on_g"] = P.add(*on_g);
  off["on_b"] = P.add(*on_b);
  off["vo_w"] = P.add(*vo_w);

  std::vector<std::map<std::string, size_t>> loff(L);
  for (int l = 0; l < L; ++l) {
    const bool last = (l == L - 1);
    const std::string p = rm + "vis_mp_layers." + std::to_string(l) + ".";
    NEED(lg, p + "layernorm.weight", (size_t)H);
    NEED(lbias, p + "layernorm.bias", (size_t)H);
    NEED(vw, p + "vec_layernorm.weight", (size_t)H);
    NEED(Wvec, p + "vec_proj.weight", (size_t)3 * H * H);
    NEED(Wq, p + "q_proj.weight", (size_t)H * H);
    NEED(bq, p + "q_proj.bias", (size_t)H);
    NEED(Wk, p + "k_proj.weight", (size_t)H * H);
    NEED(bk, p + "k_proj.bias", (size_t)H);
    NEED(Wv, p + "v_proj.weight", (size_t)H * H);
    NEED(bv, p + "v_proj.bias", (size_t)H);
    NEED(Wdk, p + "dk_proj.weight", (size_t)H * H);
    NEED(bdk, p + "dk_proj.bias", (size_t)H);
    NEED(Wdv, p + "dv_proj.weight", (size_t)H * H);
    NEED(bdv, p + "dv_proj.bias", (size_t)H);
    NEED(Wsp, p + "s_proj.weight", (size_t)2 * H * H);
    NEED(bsp, p + "s_proj.bias", (size_t)2 * H);
    NEED(Wop, p + "o_proj.weight", (size_t)3 * H * H);
    NEED(bop, p + "o_proj.bias", (size_t)3 * H);
    std::vector<float> zerosHH((size_t)H * H, 0.f), zerosH(H, 0.f);
    const std::vector<float>*Wf = &zerosHH, *bf = &zerosH, *Wsrc = &zerosHH, *Wtrg = &zerosHH;
    if (!last) {
      NEED(Wf_, p + "f_proj.weight", (size_t)H * H);
      NEED(bf_, p + "f_proj.bias", (size_t)H);
      NEED(Wsrc_, p + "w_src_proj.weight", (size_t)H * H);
      NEED(Wtrg_, p + "w_trg_proj.weight", (size_t)H * H);
      Wf = Wf_;
      bf = bf_;
      Wsrc = Wsrc_;
      Wtrg = Wtrg_;
    }
    auto& o = loff[l];
    std::vector<float> Wqkv = vcat({Wq, Wk, Wv}), bqkv = vcat({bq, bk, bv});
    o["Wqkv"] = P.add(Wqkv);
    o["bqkv"] = P.add(bqkv);
    o["WqkvT"] = P.add(transposed(Wqkv, 3 * H, H));
    std::vector<float> Wv5 = vcat({Wvec, Wtrg, Wsrc});  // vec1|vec2|vec3|w_trg|w_src
    o["Wv5"] = P.add(Wv5);
    o["Wv5T"] = P.add(transposed(Wv5, 5 * H, H));
    std::vector<float> We3 = vcat({Wdk, Wdv, Wf}), be3 = vcat({bdk, bdv, bf});
    o["We3"] = P.add(We3);
    o["be3"] = P.add(be3);
    o["We3T"] = P.add(transposed(We3, 3 * H, H));
    o["Ws"] = P.add(*Wsp);
    o["bs"] = P.add(*bsp);
    const std::vector<float> WsT = transposed(*Wsp, 2 * H, H);
    o["WsT"] = P.add(WsT);
    if (H == 256) {
      // g_m = g_t . Ws: [H, 2H], K = [g_t1 | g_t2] cut by channel half; g_f += g_pe . We3: [H, 3H] ([H, 2H] for the
      // layers without an edge update), K = [g_pk | g_pv | g_pf]
      o["WsTp"] = P.add(pack_panel(WsT, H, 2 * H, 2 * H, kperm_half));
      const int Ke = (last || l == 0) ? 2 * H : 3 * H;
      o["We3Tp"] = P.add(pack_panel(transposed(We3, 3 * H, H), H, Ke, 3 * H, kperm_half));
    }
    o["Wo"] = P.add(*Wop);
    o["bo"] = P.add(*bop);
    o["WoT"] = P.add(transposed(*Wop, 3 * H, H));
    o["ln_g"] = P.add(*lg);
    o["ln_b"] = P.add(*lbias);
    o["vln_w"] = P.add(*vw);
  }
  const std::string on = "output_model.output_network.";
  NEED(W10, on + "0.vec1_proj.weight", (size_t)H * H);
  NEED(W20, on + "0.vec2_proj.weight", (size_t)h2 * H);
  NEED(Wa0, on + "0.update_net.0.weight", (size_t)H * 2 * H);
  NEED(ba0, on + "0.update_net.0.bias", (size_t)H);
  NEED(Wb0, on + "0.update_net.2.weight", (size_t)H * H);
  NEED(bb0, on + "0.update_net.2.bias", (size_t)H);
  NEED(W11, on + "1.vec1_proj.weight", (size_t)h2 * h2);
  NEED(Wa1, on + "1.update_net.0.weight", (size_t)h2 * H);
  NEED(ba1, on + "1.update_net.0.bias", (size_t)h2);
  NEED(Wb1, on + "1.update_net.2.weight", (size_t)2 * h2);
  NEED(bb1, on + "1.update_net.2.bias", (size_t)2);
  std::vector<float> Wpv0 = vcat({W10, W20});
  off["Wpv0"] = P.add(Wpv0);
  off["Wpv0T"] = P.add(transposed(Wpv0, H + h2, H));
  off["Wa0"] = P.add(*Wa0);
  off["ba0"] = P.add(*ba0);
  off["Wa0T"] = P.add(transposed(*Wa0, H, 2 * H));
  off["Wb0"] = P.add(*Wb0);
  off["bb0"] = P.add(*bb0);
  off["Wb0T"] = P.add(transposed(*Wb0, H, H));
  off["W11"] = P.add(*W11);
  off["W11T"] = P.add(transposed(*W11, h2, h2));
  off["Wa1"] = P.add(*Wa1);
  off["ba1"] = P.add(*ba1);
  off["Wa1T"] = P.add(transposed(*Wa1, h2, H));
  {
    std::vector<float> wb1(Wb1->begin(), Wb1->begin() + h2);
    off["wb1"] = P.add(wb1);
  }
  const bool head_packed = (H % 64) == 0 && H <= 256;
  if (head_packed) {
    off["Wa0p"] = P.add(pack_head(*Wa0, H, 2 * H));
    off["Wb0p"] = P.add(pack_head(*Wb0, H, H));
    off["W11p"] = P.add(pack_head(*W11, h2, h2));
    off["Wa1p"] = P.add(pack_head(*Wa1, h2, H));
    off["Wa1Tp"] = P.add(pack_head(transposed(*Wa1, h2, H), H, h2));
    off["W11Tp"] = P.add(pack_head(transposed(*W11, h2, h2), h2, h2));
    off["Wb0Tp"] = P.add(pack_head(transposed(*Wb0, H, H), H, H));
    off["Wa0Tp"] = P.add(pack_head(transposed(*Wa0, H, 2 * H), 2 * H, H));
  }
  bool has_ar = c->hp.has_atomref != 0;
  if (has_ar) {
    // the Atomref table takes its size from prior_args["max_z"] (ViSNet/model/priors.py:62-77), not from the
    // model's max_z: accept any length and bounds-check z against both tables (graph.hip::k_graph_count)
    auto it = c->raw.find("prior_model.atomref.weight");
    if (it == c->raw.end() || it->second.data.empty()) return fail(c, -2, "missing tensor prior_model.atomref.weight");
    c->n_atomref = (int)it->second.data.size();
    off["atomref"] = P.add(it->second.data);
  }
#undef NEED

  if (c->warena) {
    hipFree(c->warena);
    c->warena = nullptr;
  }
  // the split-3 planes are keyed by weight address in the arena just freed: a re-finalize (vsn_load_weight +
  // vsn_finalize with new values) may get the SAME addresses back, so the cache of the old weights must go
  if (c->s3) {
    hipDeviceSynchronize();
    split3_table_destroy(c->s3);
    c->s3 = c->gemm_split3 ? split3_table_create() : nullptr;
  }
  HIPCHK(c, hipMalloc((void**)&c->warena, P.host.size() * sizeof(float)));
  HIPCHK(c, hipMemcpy(c->warena, P.host.data(), P.host.size() * sizeof(float), hipMemcpyHostToDevice));
  float* B = c->warena;
  c->emb1 = B + off["emb1"];
  c->emb2 = B + off["emb2"];
  c->means = B + off["means"];
  c->betas = B + off["betas"];
  c->Wrbf = B + off["Wrbf"];
  c->brbf = B + off["brbf"];
  c->WrbfT = B + off["WrbfT"];
  c->Wc = B + off["Wc"];
  c->bc = B + off["bc"];
  c->WcnT = B + off["WcnT"];
  c->on_g = B + off["on_g"];
  c->on_b = B + off["on_b"];
  c->vo_w = B + off["vo_w"];
  c->lw.resize(L);
  for (int l = 0; l < L; ++l) {
    auto& o = loff[l];
    LayerW& w = c->lw[l];
    w.Wqkv = B + o["Wqkv"];
    w.bqkv = B + o["bqkv"];
    w.WqkvT = B + o["WqkvT"];
    w.Wv5 = B + o["Wv5"];
    w.Wv5T = B + o["Wv5T"];
    w.We3 = B + o["We3"];
    w.be3 = B + o["be3"];
    w.We3T = B + o["We3T"];
    w.Ws = B + o["Ws"];
    w.bs = B + o["bs"];
    w.WsT = B + o["WsT"];
    w.WsTp = o.count("WsTp") ? B + o["WsTp"] : nullptr;
    w.We3Tp = o.count("We3Tp") ? B + o["We3Tp"] : nullptr;
    w.Wo = B + o["Wo"];
    w.bo = B + o["bo"];
    w.WoT = B + o["WoT"];
    w.ln_g = B + o["ln_g"];
    w.ln_b = B + o["ln_b"];
    w.vln_w = B + o["vln_w"];
  }
  HeadW& hw = c->hw;
  hw.Wpv0 = B + off["Wpv0"];
  hw.Wpv0T = B + off["Wpv0T"];
  hw.Wa0 = B + off["Wa0"];
  hw.ba0 = B + off["ba0"];
  hw.Wa0T = B + off["Wa0T"];
  hw.Wb0 = B + off["Wb0"];
  hw.bb0 = B + off["bb0"];
  hw.Wb0T = B + off["Wb0T"];
  hw.W11 = B + off["W11"];
  hw.W11T = B + off["W11T"];
  hw.Wa1 = B + off["Wa1"];
  hw.ba1 = B + off["ba1"];
  hw.Wa1T = B + off["Wa1T"];
  hw.wb1 = B + off["wb1"];
  hw.Wa0p = head_packed ? B + off["Wa0p"] : nullptr;
  hw.Wb0p = head_packed ? B + off["Wb0p"] : nullptr;
  hw.W11p = head_packed ? B + off["W11p"] : nullptr;
  hw.Wa1p = head_packed ? B + off["Wa1p"] : nullptr;
  hw.Wa1Tp = head_packed ? B + off["Wa1Tp"] : nullptr;
  hw.W11Tp = head_packed ? B + off["W11Tp"] : nullptr;
  hw.Wb0Tp = head_packed ? B + off["Wb0Tp"] : nullptr;
  hw.Wa0Tp = head_packed ? B + off["Wa0Tp"] : nullptr;
  hw.bb1 = (*bb1)[0];
  hw.mean = (*meanv)[0];
  hw.stdv = (*stdv)[0];
  hw.atomref = has_ar ? B + off["atomref"] : nullptr;
  c->finalized = true;
  return 0;
}

// ---------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------
static void carve(vsn_ctx* c, int N, int E, int Bn) {
  Arena& a = c->ws;
  a.off = 0;
  const size_t H = c->H, S = c->S, Rp = c->Rp, h2 = c->H / 2, nh = c->nh;
  const size_t n = N, e = E;
  c->fstart = a.take<int>(Bn + 1);
  c->fend = a.take<int>(Bn + 1);
  c->deg = a.take<int>(n + 1);
  c->zi = a.take<int>(n + 1);
  c->rowptr = a.take<int>(n + 2);
  c->colptr = a.take<int>(n + 2);
  c->src = a.take<int>(e + 1);
  c->tgt = a.take<int>(e + 1);
  c->perm = a.take<int>(e + 1);
  c->ecount = a.take<int>(64);
  c->geo = a.take<float>(e * 8);
  c->d = a.take<float>(e * 8);
  c->rbf = a.take<float>(e * Rp);
  c->drbf = a.take<float>(e * Rp);
  c->pp = a.take<float>(e * 2 * H);
  c->cat = a.take<float>(n * 2 * H);
  c->x_emb = a.take<float>(n * H);
  c->x = a.take<float>(n * H);
  c->vec = a.take<float>(n * S * H);
  c->f = a.take<float>(e * H);
  c->lb.resize(c->L);
  for (int l = 0; l < c->L; ++l) {
    LayerBuf& b = c->lb[l];
    b.xn = a.take<float>(n * H);
    b.rstd = a.take<float>(n);
    b.vh = a.take<float>(n * S * H);
    b.qkv = a.take<float>(n * 3 * H);
    b.vp = a.take<float>(n * S * 5 * H);
    b.pe = a.take<float>(e * 3 * H);
    b.tpre = a.take<float>(e * 2 * H);
    b.o = a.take<float>(n * 3 * H);
    b.vin = c->hp.vecnorm_type ? a.take<float>(n * S * H) : nullptr;
  }
  c->xh = a.take<float>(n * H);
  c->m = a.take<float>(e * H);
  c->A = a.take<float>(n * H);
  c->xn_o = a.take<float>(n * H);
  c->rstd_o = a.take<float>(n);
  c->vo = a.take<float>(n * S * H);
  c->vin_o = c->hp.vecnorm_type ? a.take<float>(n * S * H) : nullptr;
  HeadBuf& hb = c->hb;
  hb.cat0 = a.take<float>(n * 2 * H);
  hb.pv0 = a.take<float>(n * S * (H + h2));
  hb.a0 = a.take<float>(n * H);
  hb.u0 = a.take<float>(n * H);
  hb.vec1o = a.take<float>(n * S * h2);
  hb.cat1 = a.take<float>(n * H);
  hb.p1 = a.take<float>(n * S * h2);
  hb.a1b = a.take<float>(n * h2);
  hb.y = a.take<float>(n + 1);
  hb.g_a1 = a.take<float>(n * h2);
  hb.g_cat1 = a.take<float>(n * H);
  hb.g_p1 = a.take<float>(n * S * h2);
  hb.g_vec1o = a.take<float>(n * S * h2);
  hb.g_u0 = a.take<float>(n * H);
  hb.g_h0 = a.take<float>(n * H);
  hb.g_cat0 = a.take<float>(n * 2 * H);
  hb.g_pv0 = a.take<float>(n * S * (H + h2));
  c->g_vo = a.take<float>(n * S * H);
  c->g_x = a.take<float>(n * H);
  c->g_vec = a.take<float>(n * S * H);
  c->g_f = a.take<float>(e * H);
  c->g_o = a.take<float>(n * 3 * H);
  c->g_vp = a.take<float>(n * S * 5 * H);
  c->g_A = a.take<float>(n * H);
  c->g_t = a.take<float>(e * 2 * H);
  c->g_m = a.take<float>(e * H);
  c->g_pe = a.take<float>(e * 3 * H);
  c->g_qkv = a.take<float>(n * 3 * H);
  c->g_vh = a.take<float>(n * S * H);
  c->g_xh = a.take<float>(n * H);
  c->sat_tmp = a.take<float>(e * 2 * nh);
  c->g_pp = a.take<float>(e * 2 * H);
  c->g_n = a.take<float>(n * H);
  c->g_rbf = a.take<float>(e * Rp);
  c->g_geo = a.take<float>(e * VSN_GEO_W);
  c->g_ev = a.take<float>(e * 4);
  // split-K partials: up to 8 slices of the widest narrow output ([E,H] or [S*N,H])
  c->splitk_elems = 8 * std::max(e, n * S) * H;
  c->splitk = a.take<float>(c->splitk_elems);
}

static int ensure_ws(vsn_ctx* c, int N, int E, int Bn) {
  if (N <= c->capN && E <= c->capE && Bn <= c->capB && c->ws.base) return 0;
  int nN = std::max(N, c->capN), nE = std::max(E, c->capE), nB = std::max(Bn, c->capB);
  // grow with head-room so MD loops with slightly varying edge bounds do not realloc
  nN = std::max(nN, 64);
  nE = std::max(nE, 1024);
  nB = std::max(nB, 16);
  c->ws.dry = true;
  carve(c, nN, nE, nB);
  size_t need = c->ws.off;
  // the cached fragment offsets live in the workspace: a re-carve (or a failed one) invalidates them
  c->h_fs.clear();
  c->h_fe.clear();
  c->h_fs_stream = nullptr;
  if (c->ws.base) {
    HIPCHK(c, hipDeviceSynchronize());
    hipFree(c->ws.base);
    c->ws.base = nullptr;
  }
  hipError_t e = hipMalloc((void**)&c->ws.base, need);
  if (e != hipSuccess) {
    c->capN = c->capE = c->capB = 0;
    return fail(c, -12, "workspace hipMalloc of " + std::to_string(need >> 20) + " MiB failed");
  }
  c->ws.cap = need;
  c->ws.dry = false;
  carve(c, nN, nE, nB);
  c->capN = nN;
  c->capE = nE;
  c->capB = nB;
  // layer 0 sees vec == 0 (visnet_block.py:119-121): its vector projections are identically zero and are
  // never computed; the buffer is cleared once per allocation.
  HIPCHK(c, hipMemset(c->lb[0].vp, 0, (size_t)nN * c->S * 5 * c->H * sizeof(float)));
  HIPCHK(c, hipMemset(c->ecount, 0, 64 * sizeof(int)));
  c->epoch = 0;
  return 0;
}

static void snapshot(vsn_ctx* c, hipStream_t st, const char* name, int layer, const float* p, size_t elems) {
  if (!c->debug) return;
  auto& v = c->snap[name];
  if ((int)v.size() < c->L + 1) v.resize(c->L + 1, nullptr);
  size_t& cap = c->snap_elems[std::string(name) + "#" + std::to_string(layer)];
  if (!v[layer] || cap < elems) {
    if (v[layer]) hipFree(v[layer]);
    hipMalloc((void**)&v[layer], std::max<size_t>(elems, 1) * sizeof(float));
    cap = elems;
  }
  hipMemcpyAsync(v[layer], p, elems * sizeof(float), hipMemcpyDeviceToDevice, st);
}

// profile mode: the scatter-path launch made inside this scope carries two events on its dispatch packet
// (layer_fwd.hip, launch_maybe_timed): their elapsed time is the kernel's own begin..end, resolved after the chunk
struct ScatterBracket {
  vsn_ctx* c;
  bool on;
  ScatterBracket(vsn_ctx* c_, hipStream_t, int kind, int N, int f0, int f1) : c(c_), on(c_->profile) {
    if (!on) return;
    vsn_ctx::SRec r;
    r.kind = kind;
    r.N = N;
    r.f0 = f0;
    r.f1 = f1;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    c->srecs.push_back(r);
    push_launch_events(LaunchEvents{r.a, r.b, kind});  // FIFO: taken by the next launch of THIS kind on this thread
  }
  ~ScatterBracket() {
    if (on) set_launch_events(nullptr);  // (a launch that was skipped must not leak its pair to a later one)
  }
};

static void drop_scatter_records(vsn_ctx* c) {
  for (auto& r : c->srecs) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  c->srecs.clear();
}

// ---------------------------------------------------------------------------------
// THE byte model of the node walks (the only statement of it: the live figures of bench.py's roofline.hbm /
// roofline.reverse_walks come from here through vsn_profile_read_walks, tools/walk_table.py calls it through ctypes)
// ---------------------------------------------------------------------------------
// Algorithmic (compulsory) HBM bytes of ONE launch of a node walk over n nodes and e edges: every distinct array the
// launch reads or writes, once; 4-byte indices counted like floats (DESIGN.md sections 3 / 4).  f0 / f1 per kernel:
//   k_node_update        f0 = 1: the next layer's LayerNorm / VecLayerNorm is fused in
//   k_bwd_hf1            f0 = 1: with the edge-update halves (every layer but the last)
//   k_bwd_hf2            f0 / f1 = K-slices of g_m / g_A summed by this consumer (split_rev; 0 counts as 1)
//   k_bwd_norm_update    f0 = 1: accumulates into g_x / g_vec
extern "C" double vsn_walk_alg_bytes(const char* kernel, int H_, int S_, int nh_, double n, double e, int f0, int f1) {
  if (!kernel) return -1.0;
  const double H = H_, S = S_, nh = nh_;
  const std::string k(kernel);
  double fl;
  if (k == "k_edge_attn") {  // pe[dk|dv], C, src | qkv | m, A
    fl = e * (2 * H + 2 + H) + n * (3 * H + H + 1);
  } else if (k == "k_edge_attn_update") {  // + pe[f], f r/w, d, vp[wt|ws]
    fl = e * (2 * H + 2 + H) + n * (3 * H + H + 1) + e * (3 * H + 8) + n * S * 2 * H;
  } else if (k == "k_edge_update") {  // (batches: its own launch) pe[f], f r/w, d, src | vp[wt|ws]
    fl = e * (3 * H + 8 + 1) + n * S * 2 * H;
  } else if (k == "k_node_update") {  // tpre, d, src | vh, vp[vec1..3], o, x r/w, vec r/w (+ xn, xh, rstd, vh of the next layer)
    fl = e * (2 * H + 8 + 1) + n * (S * H * (1 + 3 + 2) + 3 * H + 2 * H + 1) + (f0 ? n * (2 * H + 1 + S * H) : 0.0);
  } else if (k == "k_bwd_hf1") {  // vector messages (both sides) [+ edge update: per-edge half, source side]
    // read tpre[2H], d, src|tgt|perm, g_geo[0..S) r/w | vh, g_vec; write g_t[2H] | g_vh
    fl = e * (2 * H + 8 + 3 + 2 * S + 2 * H) + n * (3 * S * H + 2);
    // + read pe[f], g_f, g_geo[16..16+S) r/w | vp[wt|ws]; write g_pe[f] | g_vp[ws]
    if (f0) fl += e * (2 * H + 2 * S + H) + n * (3 * S * H);
  } else if (k == "k_bwd_hf2") {  // attention target side + edge update per-node half
    // read pe[dk|dv], the K-slices of g_m, C, g_geo[8] r/w, src | qkv, the K-slices of g_A;
    // write g_m, g_pe[dk|dv], sat_tmp | g_q   + read pe[f], g_f, d | vp[ws]; write g_vp[wt]
    fl = e * (2 * H + std::max(f0, 1) * H + 1 + 2 + 1 + H + 2 * H + 2 * nh) + n * (3 * H + std::max(f1, 1) * H + H + 2) +
         e * (2 * H + 8) + n * (2 * S * H);
  } else if (k == "k_bwd_attn_S") {  // read pe[dk|dv], g_m, sat_tmp, perm|tgt | q; write g_k, g_v
    fl = e * (3 * H + 2 * nh + 2) + n * (3 * H + 1);
  } else if (k == "k_bwd_norm_update") {  // read g_xh, xn, rstd, o[2H], g_vh, vp[3H] (+ g_x, g_vec when accumulating);
                                          // write g_x, g_vec, g_vp[3H], g_o[3H]
    fl = n * (2 * H + 1 + 2 * H + S * H + 3 * S * H + H + S * H + 3 * S * H + 3 * H) + (f0 ? n * (H + S * H) : 0.0);
  } else if (k == "k_bwd_gm_fused") {  // (batches) tpre, d, g_geo r/w, src|tgt | vh, g_vec; write g_m
    fl = e * (2 * H + 8 + 2 * S + 2 + H) + n * 2 * S * H;
  } else if (k == "k_bwd_gf_fused") {  // pe[dk|dv], g_m r/w, g_pe[f], C, g_geo r/w, ids, sat_tmp, g_f r/w | qkv, g_A
    fl = e * (2 * H + 2 * H + H + 1 + 2 + 2 + 2 * nh + 2 * H) + n * 4 * H;
  } else if (k == "k_bwd_edge_update_T") {
    fl = e * (3 * H + 8 + 4 * S + 1) + n * 3 * S * H;
  } else if (k == "k_bwd_edge_update_S") {
    fl = e * (2 * H + 10) + n * 2 * S * H;
  } else if (k == "k_bwd_vecmsg_S") {
    fl = e * (H + 2) + n * 2 * S * H;
  } else {
    return -1.0;
  }
  return 4.0 * fl;
}

// ---------------------------------------------------------------------------------
// one chunk: forward + reverse
// ---------------------------------------------------------------------------------
static int run_chunk(vsn_ctx* c, hipStream_t st, const int64_t* z, const float* pos, const std::vector<int>& fs,
                     const std::vector<int>& fe, int N, int Bn, int Emax, int maxfrag, float* e_out, float* f_out) {
  int rc = ensure_ws(c, N, Emax, Bn);
  if (rc) return rc;
  const int H = c->H, L = c->L, S = c->S, Rp = c->Rp;
  c->lastN = N;
  c->lastB = Bn;
  c->lastEmax = Emax;
  // fragment offsets (small) - skip the upload when unchanged (MD: static fragmentation)
  if (fs != c->h_fs || fe != c->h_fe || st != c->h_fs_stream) {
    // pinned staging buffer (an async copy from pageable memory may still read the source after the call
    // returns); it is rewritten only after the previous upload has completed
    if ((size_t)Bn > c->pin_cap) {
      if (c->pin_fs) {
        HIPCHK(c, hipEventSynchronize(c->ev_upload));
        hipHostFree(c->pin_fs);
        c->pin_fs = nullptr;
      }
      c->pin_cap = std::max<size_t>((size_t)Bn * 2, 1024);
      HIPCHK(c, hipHostMalloc((void**)&c->pin_fs, 2 * c->pin_cap * sizeof(int), hipHostMallocDefault));
    } else {
      HIPCHK(c, hipEventSynchronize(c->ev_upload));
    }
    c->h_fs = fs;
    c->h_fe = fe;
    c->h_fs_stream = st;
    memcpy(c->pin_fs, fs.data(), Bn * sizeof(int));
    memcpy(c->pin_fs + c->pin_cap, fe.data(), Bn * sizeof(int));
    HIPCHK(c, hipMemcpyAsync(c->fstart, c->pin_fs, Bn * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->fend, c->pin_fs + c->pin_cap, Bn * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_upload, st));
  }
  GraphArgs g;
  g.pos = pos;
  g.z64 = (const long long*)z;
  g.fstart = c->fstart;
  g.fend = c->fend;
  g.B = Bn;
  g.N = N;
  g.Emax = Emax;
  g.max_frag = maxfrag;
  g.rc = c->hp.cutoff;
  g.rc2 = c->hp.cutoff * c->hp.cutoff;
  g.alpha = 5.0f / c->hp.cutoff;
  g.max_nb = c->hp.max_num_neighbors;
  g.R = c->R;
  g.Rp = Rp;
  g.S = S;
  g.rbf_type = c->hp.rbf_type;
  g.means = c->means;
  g.betas = c->betas;
  g.deg = c->deg;
  g.zi = c->zi;
  g.rowptr = c->rowptr;
  g.colptr = c->colptr;
  g.src = c->src;
  g.tgt = c->tgt;
  g.perm = c->perm;
  g.ecount = c->ecount;
  c->status = c->ecount + 1;  // zeroed when the workspace is carved; holds the epoch of the last chunk with a bad z
  c->epoch = c->epoch >= 0x7ffffff0 ? 1 : c->epoch + 1;
  g.status = c->status;
  g.epoch = c->epoch;
  g.g_geo = c->g_geo;
  c->hw.epoch = c->epoch;
  c->hw.fuse = (c->fuse_head && !c->debug) ? 1 : 0;
  c->hw.defer_energy = c->debug ? 0 : 1;  // (the debug build of the step keeps g_ev and the two-launch force fold)
  g.z_limit = c->hw.atomref ? std::min(c->Z, c->n_atomref) : c->Z;
  c->hw.status = c->status;
  g.geo = c->geo;
  g.d = c->d;
  g.rbf = c->rbf;
  g.drbf = c->drbf;
  Dims D;
  D.N = N;
  D.Emax = Emax;
  D.H = H;
  D.S = S;
  D.nh = c->nh;
  D.hd = H / c->nh;
  D.hgen = (64 % c->nh) != 0 ? 1 : 0;
  D.R = c->R;
  D.Rp = Rp;
  D.act = c->hp.activation;
  D.attn_act = c->hp.attn_activation;
  D.ecount = c->ecount;
  D.rowptr = c->rowptr;
  D.colptr = c->colptr;
  D.src = c->src;
  D.tgt = c->tgt;
  D.perm = c->perm;
  D.zi = c->zi;
  D.geo = c->geo;
  D.d = c->d;
  const int* EP = c->ecount;

#define RC(call)            \
  do {                      \
    int r__ = (call);       \
    if (r__) return fail(c, r__, std::string("launch failed: ") + #call); \
  } while (0)

  // no memsets per evaluation: k_edge_geom clears the g_geo row of every live edge, the first writer of g_f (the
  // last layer's dX group) does not accumulate, and the status word is an epoch stamp
  RC(launch_graph(st, g));
  // ---- embeddings ----
  RC(launch_gemm(st, c->rbf, Rp, c->Wrbf, Rp, c->pp, 2 * H, c->brbf, Emax, EP, 2 * H, Rp, 0));
  RC(launch_embed_node(st, D, c->emb1, c->emb2, c->pp, c->cat));
  RC(launch_gemm(st, c->cat, 2 * H, c->Wc, 2 * H, c->x_emb, H, c->bc, N, nullptr, H, 2 * H, 0));
  // layer 0's LayerNorm (and vh = 0 under VecLayerNorm "none") rides in the edge-embedding launch
  const bool fuse_norm0 = c->fuse_fwd && c->hp.vecnorm_type == 0 && !c->debug;
  {
    NextNorm n0;
    memset(&n0, 0, sizeof(n0));
    if (fuse_norm0) {
      const LayerW& w0 = c->lw[0];
      LayerBuf& b0 = c->lb[0];
      n0 = NextNorm{w0.ln_g, w0.ln_b, w0.vln_w, b0.xn, b0.rstd, c->xh, b0.vh, H};
    }
    RC(launch_embed_edge(st, D, c->x_emb, c->pp, c->f, c->vec, c->x, n0));
  }
  // ---- ViS-MP layers ----
  for (int l = 0; l < L; ++l) {
    const bool last = (l == L - 1);
    const LayerW& w = c->lw[l];
    LayerBuf& b = c->lb[l];
    snapshot(c, st, "x_in", l, c->x, (size_t)N * H);
    snapshot(c, st, "vec_in", l, c->vec, (size_t)N * S * H);
    snapshot(c, st, "f_in", l, c->f, (size_t)Emax * H);
    // the norms of layer l > 0 (and of the read-out) were already produced by the node update of layer l-1
    // when the fusion is active (vecnorm "none", not in debug mode)
    const bool fuse_norm = c->fuse_fwd && c->hp.vecnorm_type == 0 && !c->debug;
    if (!fuse_norm) {
      RC(launch_node_norm(st, D, c->x, c->vec, w.ln_g, w.ln_b, w.vln_w, c->hp.vecnorm_type, b.xn, b.rstd, c->xh, H,
                          b.vh));
      if (c->hp.vecnorm_type) RC(launch_vecnorm_fwd(st, N, H, S, c->hp.vecnorm_type, c->vec, w.vln_w, b.vin, b.vh));
    }
    // layer 0: vec == 0 -> vec1..3, w_trg.v, w_src.v are 0 (buffer pre-cleared) and df == 0 (no f_proj needed)
    const bool l0 = (l == 0);
    {
      GemmDesc gd[3];
      int ng = 0;
      gd[ng++] = gemm_desc(c->xh, H, w.Wqkv, H, b.qkv, 3 * H, w.bqkv, N, nullptr, 3 * H, H, 0);
      if (!l0) gd[ng++] = gemm_desc(b.vh, H, w.Wv5, H, b.vp, 5 * H, nullptr, N * S, nullptr, last ? 3 * H : 5 * H, H, 0);
      gd[ng++] = gemm_desc(c->f, H, w.We3, H, b.pe, 3 * H, w.be3, Emax, EP, (last || l0) ? 2 * H : 3 * H, H, 0);
      RC(launch_gemm_group(st, gd, ng));
    }
    // (moving the vector-projection GEMMs vp = vh.Wv5 / g_vh += g_vp.Wv5 to the side stream as well was
    //  measured: 350 -> 331 steps/s on Chignolin - two MFMA kernels sharing the CUs slow each other down more
    //  than the shorter critical path gains - so only the latency-bound gather kernels are overlapped.)
    // the edge update (f += df) only needs vp / pe / f: it runs on the side stream next to the
    // attention -> s_proj/o_proj -> node update chain and is joined before the next layer reads f
    const bool side_eu = (c->overlap & 1) && !c->debug && !last && !l0;
    if (side_eu) {
      HIPCHK(c, hipEventRecord(c->ev_fork, st));
      HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
      RC(launch_edge_update(c->side, D, b.vp, b.pe, c->f));
      HIPCHK(c, hipEventRecord(c->ev_join, c->side));
    }
    // single-protein sizes: the edge update rides in the attention launch (horizontal fusion)
    const bool fused_eu = c->fuse_fwd && !side_eu && !c->debug && !last && !l0 && N < 4096;
    {
      ScatterBracket sb(c, st, 0, N, fused_eu ? 1 : 0, 0);
      if (fused_eu) RC(launch_edge_attn_update(st, D, b.qkv, b.pe, c->m, c->A, b.vp, c->f));
      else RC(launch_edge_attn(st, D, b.qkv, b.pe, c->m, c->A));
    }
    snapshot(c, st, "m", l, c->m, (size_t)Emax * H);
    snapshot(c, st, "A", l, c->A, (size_t)N * H);
    {
      GemmDesc gd[2];
      gd[0] = gemm_desc(c->m, H, w.Ws, H, b.tpre, 2 * H, w.bs, Emax, EP, 2 * H, H, 0);
      gd[1] = gemm_desc(c->A, H, w.Wo, H, b.o, 3 * H, w.bo, N, nullptr, 3 * H, H, 0);
      RC(launch_gemm_group(st, gd, 2));
    }
    {
      NextNorm nn;
      memset(&nn, 0, sizeof(nn));
      if (fuse_norm) {
        if (!last) {
          const LayerW& wn = c->lw[l + 1];
          LayerBuf& bn = c->lb[l + 1];
          nn = NextNorm{wn.ln_g, wn.ln_b, wn.vln_w, bn.xn, bn.rstd, c->xh, bn.vh, H};
        } else {
          nn = NextNorm{c->on_g, c->on_b, c->vo_w, c->xn_o, c->rstd_o, c->hb.cat0, c->vo, 2 * H};
        }
      }
      ScatterBracket sb(c, st, 1, N, fuse_norm ? 1 : 0, 0);
      RC(launch_node_update(st, D, b.tpre, b.vh, b.vp, b.o, c->x, c->vec, nn));
    }
    if (side_eu) HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
    else if (!last && !l0 && !fused_eu) RC(launch_edge_update(st, D, b.vp, b.pe, c->f));
  }
  snapshot(c, st, "x_in", L, c->x, (size_t)N * H);
  snapshot(c, st, "vec_in", L, c->vec, (size_t)N * S * H);
  // ---- read-out ----
  if (!(c->fuse_fwd && c->hp.vecnorm_type == 0 && !c->debug)) {
    RC(launch_node_norm(st, D, c->x, c->vec, c->on_g, c->on_b, c->vo_w, c->hp.vecnorm_type, c->xn_o, c->rstd_o,
                        c->hb.cat0, 2 * H, c->vo));
    if (c->hp.vecnorm_type)
      RC(launch_vecnorm_fwd(st, N, H, S, c->hp.vecnorm_type, c->vec, c->vo_w, c->vin_o, c->vo));
  }
  RC(launch_head_forward(st, D, c->hw, c->hb, c->vo, c->fstart, c->fend, Bn, e_out));
  // ---- reverse pass ----
  RC(launch_head_backward(st, D, c->hw, c->hb, c->g_vo));
  const bool fuse_bwd = c->fuse_bwd_opt && c->hp.vecnorm_type == 0 && !c->debug;
  if (fuse_bwd) {
    ScatterBracket sb(c, st, 5, N, 0, 0);
    RC(launch_bwd_norm_update(st, D, c->hb.g_cat0, 2 * H, c->g_vo, c->xn_o, c->rstd_o, c->on_g, c->vo_w, 0, c->g_x,
                              c->g_vec, c->lb[L - 1].vp, c->lb[L - 1].o, c->g_o, c->g_vp));
  } else {
    RC(launch_bwd_node_norm(st, D, c->hb.g_cat0, 2 * H, c->g_vo, c->xn_o, c->rstd_o, c->on_g, c->vo_w,
                            c->hp.vecnorm_type, 0, c->g_x, c->g_vec));
    if (c->hp.vecnorm_type)
      RC(launch_vecnorm_bwd(st, N, H, S, c->hp.vecnorm_type, c->vin_o, c->vo_w, c->g_vo, 0, c->g_vec));
  }
  snapshot(c, st, "g_x_in", L, c->g_x, (size_t)N * H);
  snapshot(c, st, "g_vec_in", L, c->g_vec, (size_t)N * S * H);
  for (int l = L - 1; l >= 0; --l) {
    const bool last = (l == L - 1);
    const LayerW& w = c->lw[l];
    LayerBuf& b = c->lb[l];
    if (!fuse_bwd) RC(launch_bwd_node_update(st, D, c->g_x, c->g_vec, b.vp, b.o, c->g_o, c->g_vp));
    // layer 0: the edge update and everything flowing into vec_in (== 0, position independent) vanish
    const bool l0 = (l == 0);
    // (a fused walk over the out-edges doing the three source-side adjoints at once measured SLOWER
    //  than three separate kernels - 609 vs 527 us on a 512-fragment batch, 53 vs 34 us on Chignolin:
    //  these kernels are latency-bound and the fused one loses occupancy - so they stay separate)
    // side stream: the edge-update adjoints (need g_f, vp, pe) and the source side of the vector messages
    // (needs g_vec, tpre) do not depend on this layer's main chain; they are joined before the dX products.
    // (edge_update_T accumulates its dE/dd into its own g_geo slots 16..23, so it cannot race vecmsg_T.)
    // single-protein sizes: no second stream at all - the same kernels ride in two launches of the main chain
    // (k_bwd_hf1 before the g_m / g_A products, k_bwd_hf2 after them); a fork/join costs ~7 us on the main stream at
    // each end, every layer
    // fragment batches at hidden 256: the target-side vector-message and attention adjoints are prologues of the
    // products they feed (fused.hip) - g_t and the attention part of g_pe never reach HBM.  Below `panel_min_edges`
    // (single-protein sizes) the 64-edge panels cannot fill the chip and the several-waves-per-node kernels stay.
    const bool panel = c->fuse_panel && !c->debug && w.WsTp && panel_ok(D) &&
                       (bwd_batch_path(D) || (int64_t)Emax >= c->panel_min_edges);
    const bool streamless = (c->overlap & 2) && !c->debug && !l0 && bwd_streamless_ok(D) && !panel;
    // (a second stream only pays on batches: its fork / join costs ~7 us at each end, every layer)
    const bool side_bw = (c->overlap & 4) && !c->debug && !l0 && !streamless && bwd_batch_path(D);
    if (panel) {
      // edge-update adjoints + source side of the vector messages: on the side stream, or (overlap off) in line
      if (!l0) {
        hipStream_t ss = side_bw ? c->side : st;
        if (side_bw) {
          HIPCHK(c, hipEventRecord(c->ev_fork, st));
          HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        }
        if (!last) RC(launch_bwd_side(ss, D, b.vp, b.pe, c->g_f, c->g_pe, c->g_vp, c->g_geo, c->g_vec, b.tpre, c->g_vh));
        else RC(launch_bwd_vecmsg_S(ss, D, c->g_vec, b.tpre, c->g_vh));
        if (side_bw) HIPCHK(c, hipEventRecord(c->ev_join, c->side));
      }
      RC(launch_bwd_gm_fused(st, D, c->g_vec, b.vh, b.tpre, w.WsTp, c->g_m, c->g_geo));
      RC(launch_gemm(st, c->g_o, 3 * H, w.WoT, 3 * H, c->g_A, H, nullptr, N, nullptr, H, 3 * H, 0));
      // the edge-update adjoint (side stream) writes g_pe[:, 2H:3H], the third K-slice of the fused product
      if (side_bw) HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
      RC(launch_bwd_gf_fused(st, D, b.qkv, b.pe, c->g_A, c->g_m, c->g_pe, c->sat_tmp, c->g_geo, w.We3Tp, c->g_f,
                             (last || l0) ? 2 * H : 3 * H, last ? 0 : 1));
      RC(launch_bwd_attn_QS(st, D, b.qkv, b.pe, c->g_m, c->sat_tmp, c->g_qkv));
      GemmDesc gd[2];
      int ng = 0;
      if (!l0)
        gd[ng++] = gemm_desc(c->g_vp, 5 * H, w.Wv5T, 5 * H, c->g_vh, H, nullptr, N * S, nullptr, H,
                             last ? 3 * H : 5 * H, 1);
      gd[ng++] = gemm_desc(c->g_qkv, 3 * H, w.WqkvT, 3 * H, c->g_xh, H, nullptr, N, nullptr, H, 3 * H, 0);
      RC(launch_gemm_group(st, gd, ng));
    } else {
    if (streamless) {
      ScatterBracket sb(c, st, 2, N, last ? 0 : 1, 0);
      RC(launch_bwd_hf1(st, D, c->g_vec, b.vh, b.tpre, c->g_t, c->g_geo, b.vp, b.pe, c->g_f, c->g_pe, c->g_vp, c->g_vh,
                        !last));
    } else if (side_bw) {
      HIPCHK(c, hipEventRecord(c->ev_fork, st));
      HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
      if (!last) RC(launch_bwd_side(c->side, D, b.vp, b.pe, c->g_f, c->g_pe, c->g_vp, c->g_geo, c->g_vec, b.tpre, c->g_vh));
      else RC(launch_bwd_vecmsg_S(c->side, D, c->g_vec, b.tpre, c->g_vh));
      HIPCHK(c, hipEventRecord(c->ev_join, c->side));
      RC(launch_bwd_vecmsg(st, D, c->g_vec, b.vh, b.tpre, c->g_t, nullptr, c->g_geo));
    } else {
      if (!last && !l0) RC(launch_bwd_edge_update(st, D, b.vp, b.pe, c->g_f, c->g_pe, c->g_vp, c->g_geo));
      RC(launch_bwd_vecmsg(st, D, c->g_vec, b.vh, b.tpre, c->g_t, l0 ? nullptr : c->g_vh, c->g_geo));
    }
    snapshot(c, st, "g_t", l, c->g_t, (size_t)Emax * 2 * H);
    Parts mparts{nullptr, 0, 0}, aparts{nullptr, 0, 0};
    GemmDesc gdr[2];
    // g_A = g_o.Wo (N rows, tiny) rides along with g_m = g_t.Ws (E rows) in one grouped launch
    gdr[0] = gemm_desc(c->g_t, 2 * H, w.WsT, 2 * H, c->g_m, H, nullptr, Emax, EP, H, 2 * H, 0);
    gdr[1] = gemm_desc(c->g_o, 3 * H, w.WoT, 3 * H, c->g_A, H, nullptr, N, nullptr, H, 3 * H, 0);
    // the K-slices below are a grouped-launch feature: decide with the predicate launch_gemm_group itself uses
    const bool split_rev = c->split_rev && !c->debug && N < 4096 && Emax < 32768 && gemm_group_ok(gdr, 2) &&
                           (size_t)(2 * (size_t)Emax + 3 * (size_t)N) * H <= c->splitk_elems;
    {
      if (split_rev) {
        // single-protein sizes: these two long-K products give only ~1.75 64x64 tiles per CU (two waves per SIMD,
        // a latency-bound k-loop).  Cut K into slices of H (uniform 8-k-tile units, ~3.6 per CU) and let the
        // attention adjoint, which reads both results row by row anyway, add the slices up: no reduction launch.
        gdr[0].keep_parts = 2;
        gdr[0].part = c->splitk;
        gdr[1].keep_parts = 3;
        gdr[1].part = c->splitk + (size_t)2 * Emax * H;
        mparts = Parts{gdr[0].part, (size_t)Emax * H, 2};
        aparts = Parts{gdr[1].part, (size_t)N * H, 3};
      }
      RC(launch_gemm_group(st, gdr, 2));
    }
    if (streamless && !last) {
      ScatterBracket sb2(c, st, 3, N, mparts.n, aparts.n);  // (FIFO: the k_bwd_hf2 launch takes this pair ...
      ScatterBracket sb4(c, st, 4, N, 0, 0);                //  ... and the k_bwd_attn_S launch behind it this one)
      RC(launch_bwd_hf2(st, D, b.qkv, b.pe, c->g_A, c->g_m, c->g_pe, c->g_qkv, c->sat_tmp, c->g_geo, mparts, aparts, b.vp,
                        c->g_f, c->g_vp));
    } else {
      ScatterBracket sb4(c, st, 4, N, 0, 0);  // (k_bwd_attn_T is not a timed launch: the pair goes to k_bwd_attn_S)
      RC(launch_bwd_attn(st, D, b.qkv, b.pe, c->g_A, c->g_m, c->g_pe, c->g_qkv, c->sat_tmp, c->g_geo, mparts, aparts));
    }
    snapshot(c, st, "g_m", l, c->g_m, (size_t)Emax * H);
    snapshot(c, st, "g_pe", l, c->g_pe, (size_t)Emax * 3 * H);
    snapshot(c, st, "g_qkv", l, c->g_qkv, (size_t)N * 3 * H);
    snapshot(c, st, "g_vp", l, c->g_vp, (size_t)N * S * 5 * H);
    snapshot(c, st, "g_A", l, c->g_A, (size_t)N * H);
    if (side_bw) HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
    {
      GemmDesc gd[3];
      int ng = 0;
      gd[ng++] = gemm_desc(c->g_pe, 3 * H, w.We3T, 3 * H, c->g_f, H, nullptr, Emax, EP, H,
                           (last || l0) ? 2 * H : 3 * H, last ? 0 : 1);  // the last layer is g_f's first writer
      if (!l0)
        gd[ng++] = gemm_desc(c->g_vp, 5 * H, w.Wv5T, 5 * H, c->g_vh, H, nullptr, N * S, nullptr, H,
                             last ? 3 * H : 5 * H, 1);
      gd[ng++] = gemm_desc(c->g_qkv, 3 * H, w.WqkvT, 3 * H, c->g_xh, H, nullptr, N, nullptr, H, 3 * H, 0);
      RC(launch_gemm_group(st, gd, ng));
    }
    }
    snapshot(c, st, "g_vh", l, c->g_vh, (size_t)N * S * H);
    snapshot(c, st, "g_xh", l, c->g_xh, (size_t)N * H);
    if (fuse_bwd && l > 0) {  // norm adjoint of this layer + node-update adjoint of the layer below, one pass
      ScatterBracket sb(c, st, 5, N, 1, 0);
      RC(launch_bwd_norm_update(st, D, c->g_xh, H, c->g_vh, b.xn, b.rstd, w.ln_g, w.vln_w, 1, c->g_x, c->g_vec,
                                c->lb[l - 1].vp, c->lb[l - 1].o, c->g_o, c->g_vp));
    } else if (!(fuse_bwd && l0))  // (fused reverse pass: layer 0's LayerNorm adjoint rides in the edge-embedding adjoint)
      RC(launch_bwd_node_norm(st, D, c->g_xh, H, c->g_vh, b.xn, b.rstd, w.ln_g, w.vln_w,
                              l0 ? 3 /* skip the vec part */ : c->hp.vecnorm_type, 1, c->g_x, c->g_vec));
    // layer 0 normalises vec == 0, which does not depend on the positions: nothing to propagate
    if (c->hp.vecnorm_type && l > 0)
      RC(launch_vecnorm_bwd(st, N, H, S, c->hp.vecnorm_type, b.vin, w.vln_w, c->g_vh, 1, c->g_vec));
    snapshot(c, st, "g_x_in", l, c->g_x, (size_t)N * H);
    snapshot(c, st, "g_vec_in", l, c->g_vec, (size_t)N * S * H);
    snapshot(c, st, "g_f_in", l, c->g_f, (size_t)Emax * H);
  }
  // ---- embeddings, reverse ----
  if (fuse_bwd)
    RC(launch_bwd_embed_edge(st, D, c->x_emb, c->pp, c->g_f, c->g_pp, c->g_x, c->g_xh, c->lb[0].xn, c->lb[0].rstd,
                             c->lw[0].ln_g));
  else
    RC(launch_bwd_embed_edge(st, D, c->x_emb, c->pp, c->g_f, c->g_pp, c->g_x, nullptr, nullptr, nullptr, nullptr));
  RC(launch_gemm(st, c->g_x, H, c->WcnT, H, c->g_n, H, nullptr, N, nullptr, H, H, 0));
  RC(launch_bwd_embed_node(st, D, c->emb2, c->pp, c->g_n, c->g_pp, c->g_geo));
  RC(launch_gemm(st, c->g_pp, 2 * H, c->WrbfT, 2 * H, c->g_rbf, Rp, nullptr, Emax, EP, Rp, 2 * H, 0));
  {
    EnergyFold ef{0, nullptr, nullptr, nullptr, 0.f, nullptr};
    if (head_defers_energy(D, c->hw)) ef = EnergyFold{Bn, c->fstart, c->fend, c->hb.y, c->hw.mean, e_out};
    RC(launch_bwd_geom(st, g, c->g_rbf, c->g_geo, c->g_ev, f_out, c->debug, ef));
  }
  if (c->reduce_mean) RC(launch_reduce_mean(st, Bn, c->fstart, c->fend, c->hw.mean, e_out, f_out));
#undef RC
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(c, -5, std::string("kernel launch error: ") + hipGetErrorString(le));
  return 0;
}

extern "C" int vsn_forces(vsn_handle c, const int64_t* dev_z, const float* dev_pos, const int64_t* host_start,
                          const int64_t* host_end, int64_t N, int64_t B, float* dev_e_out, float* dev_f_out,
                          void* stream) {
  if (!c) return -22;
  if (!c->finalized) return fail(c, -22, "vsn_finalize() has not been called");
  if (N < 0 || B < 0) return fail(c, -22, "negative sizes");
  if (B == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  // validate layout: contiguous ascending fragments covering [0,N)
  int64_t prev = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (host_start[b] != prev || host_end[b] < host_start[b])
      return fail(c, -22, "fragments must be contiguous and ascending (start[b] == end[b-1])");
    if (host_end[b] - host_start[b] > VSN_MAX_FRAG_ATOMS)
      return fail(c, -22, "fragment larger than VSN_MAX_FRAG_ATOMS");
    prev = host_end[b];
  }
  if (prev != N) return fail(c, -22, "fragment offsets do not cover N atoms");
  const int64_t mnb = c->hp.max_num_neighbors;
  int64_t b0 = 0;
  while (b0 < B) {
    // greedy chunk by the host-side edge bound sum n*min(n,max_nb)
    int64_t eb = 0, b1 = b0, maxfrag = 0;
    while (b1 < B) {
      int64_t n = host_end[b1] - host_start[b1];
      int64_t add = n * std::min<int64_t>(n, mnb);
      if (b1 > b0 && eb + add > c->max_chunk_edges) break;
      eb += add;
      maxfrag = std::max(maxfrag, n);
      ++b1;
    }
    const int64_t a0 = host_start[b0], a1 = host_end[b1 - 1];
    std::vector<int> fs((size_t)(b1 - b0)), fe((size_t)(b1 - b0));
    for (int64_t b = b0; b < b1; ++b) {
      fs[(size_t)(b - b0)] = (int)(host_start[b] - a0);
      fe[(size_t)(b - b0)] = (int)(host_end[b] - a0);
    }
    GemmProfiler gp;
    hipEvent_t empty[16];
    drop_scatter_records(c);  // (records of a call that failed half-way never reach the sums)
    if (c->profile) {
      set_gemm_profiler(&gp);
      // what an event bracket itself costs on this stream: 8 brackets around nothing, mid-stream (the device is
      // busy with the previous chunk / step, as it is for the real brackets)
      for (int k = 0; k < 16; ++k) {
        hipEventCreate(&empty[k]);
        hipEventRecord(empty[k], st);
      }
    }
    int rc0 = ensure_ws(c, (int)(a1 - a0), (int)eb, (int)(b1 - b0));
    if (rc0) {
      set_gemm_profiler(nullptr);
      if (c->profile) {
        hipStreamSynchronize(st);
        for (int k = 0; k < 16; ++k) hipEventDestroy(empty[k]);
        for (auto& r : gp.recs) {
          hipEventDestroy(r.a);
          hipEventDestroy(r.b);
        }
      }
      return rc0;
    }
    set_gemm_splitk_workspace(c->splitk, c->splitk_elems);
    set_gemm_split3(c->gemm_split3 ? c->s3 : nullptr);
    int rc = run_chunk(c, st, dev_z + a0, dev_pos + 3 * a0, fs, fe, (int)(a1 - a0), (int)(b1 - b0), (int)eb,
                       (int)maxfrag, dev_e_out + b0, dev_f_out + 3 * a0);
    set_gemm_profiler(nullptr);
    set_gemm_splitk_workspace(nullptr, 0);
    set_gemm_split3(nullptr);
    if (c->profile) {
      hipStreamSynchronize(st);
      int E = 0;
      hipMemcpy(&E, c->ecount, sizeof(int), hipMemcpyDeviceToHost);
      for (int k = 0; k < 16; k += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, empty[k], empty[k + 1]) == hipSuccess) {
          c->prof_empty_ms += ms;
          c->prof_empty_n += 1;
        }
      }
      for (int k = 0; k < 16; ++k) hipEventDestroy(empty[k]);
      for (auto& r : c->srecs) {
        // algorithmic (compulsory) HBM bytes of the launch: every array it touches once
        float ms = 0.f;
        const bool timed = hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess;
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
        if (!timed) continue;
        static const char* const kname[6] = {"k_edge_attn", "k_node_update", "k_bwd_hf1", "k_bwd_hf2", "k_bwd_attn_S",
                                              "k_bwd_norm_update"};
        const char* nm = (r.kind == WK_EDGE_ATTN && r.f0) ? "k_edge_attn_update" : kname[r.kind];
        const double fl = vsn_walk_alg_bytes(nm, c->H, c->S, c->nh, r.N, E, r.f0, r.f1) / 4.0;
        c->sprof[r.kind][0] += 1;
        c->sprof[r.kind][1] += ms;
        c->sprof[r.kind][2] += 4.0 * fl;
      }
      c->srecs.clear();
      for (auto& r : gp.recs) {
        float ms = 0.f;
        const bool timed = hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess;
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
        if (!timed) continue;
        c->prof[r.variant][0] += 1;
        c->prof[r.variant][1] += ms;
        if (r.group_n > 0) {
          for (int g = 0; g < r.group_n; ++g) {
            double rows = r.gdev[g] ? std::min(r.gM[g], E) : r.gM[g];
            c->prof[r.variant][2] += rows * r.gflops[g];
            c->prof[r.variant][3] += rows * r.gbytes[g];
          }
        } else {
          double rows = r.dev_m ? std::min(r.M, E) : r.M;
          c->prof[r.variant][2] += rows * r.flops_per_row;
          c->prof[r.variant][3] += rows * r.bytes_per_row;
        }
      }
    }
    if (rc) return rc;
    b0 = b1;
  }
  return 0;
}

extern "C" int vsn_profile_read(vsn_handle c, double* out16) {
  if (!c || !out16) return -22;
  for (int v = 0; v < 4; ++v)
    for (int k = 0; k < 4; ++k) out16[v * 4 + k] = c->prof[v][k];
  return 0;
}

extern "C" int vsn_profile_read_scatter(vsn_handle c, double* out8) {
  if (!c || !out8) return -22;
  for (int v = 0; v < 2; ++v)
    for (int k = 0; k < 4; ++k) out8[v * 4 + k] = c->sprof[v][k];
  return 0;
}

extern "C" int vsn_profile_read_walks(vsn_handle c, double* out, int max_kinds) {
  if (!c || !out || max_kinds < 0) return -22;
  const int n = std::min(max_kinds, (int)vsn_ctx::NWALK);
  for (int v = 0; v < n; ++v)
    for (int k = 0; k < 4; ++k) out[v * 4 + k] = c->sprof[v][k];
  return n;
}

extern "C" double vsn_profile_bracket_ms(vsn_handle c) {
  if (!c || c->prof_empty_n <= 0) return 0.0;
  return c->prof_empty_ms / c->prof_empty_n;
}

extern "C" int64_t vsn_last_num_edges(vsn_handle c) {
  if (!c || !c->ws.base) return -22;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  int e = 0;
  if (hipMemcpy(&e, c->ecount, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -5;
  return e;
}

extern "C" int vsn_last_status(vsn_handle c) {
  if (!c || !c->ws.base || !c->status) return -22;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  int v = 0;
  if (hipMemcpy(&v, c->status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -5;
  return v == c->epoch ? 1 : 0;
}

extern "C" int64_t vsn_debug_read(vsn_handle c, const char* name, int layer, void* host_out, int64_t max_elems) {
  if (!c || !name || !c->ws.base) return -22;
  hipSetDevice(c->device);
  if (hipDeviceSynchronize() != hipSuccess) return fail(c, -5, "device sync failed");
  int E = 0;
  hipMemcpy(&E, c->ecount, sizeof(int), hipMemcpyDeviceToHost);
  const size_t N = c->lastN, H = c->H, S = c->S, Rp = c->Rp, h2 = c->H / 2, e = E, Bn = c->lastB;
  const std::string k(name);
  const void* p = nullptr;
  size_t n = 0;
  auto L_ok = layer >= 0 && layer < c->L;
#define TAP(nm, ptr, cnt) \
  if (k == nm) {          \
    p = (ptr);            \
    n = (cnt);            \
  }
  TAP("rowptr", c->rowptr, N + 1)
  TAP("colptr", c->colptr, N + 1)
  TAP("src", c->src, e)
  TAP("tgt", c->tgt, e)
  TAP("perm", c->perm, e)
  TAP("geo", c->geo, e * 8)
  TAP("d", c->d, e * 8)
  TAP("rbf", c->rbf, e * Rp)
  TAP("drbf", c->drbf, e * Rp)
  TAP("pp", c->pp, e * 2 * H)
  TAP("cat", c->cat, N * 2 * H)
  TAP("x_emb", c->x_emb, N * H)
  TAP("x", c->x, N * H)
  TAP("vec", c->vec, N * S * H)
  TAP("f", c->f, e * H)
  TAP("cat0", c->hb.cat0, N * 2 * H)
  TAP("vo", c->vo, N * S * H)
  TAP("pv0", c->hb.pv0, N * S * (H + h2))
  TAP("a0", c->hb.a0, N * H)
  TAP("u0", c->hb.u0, N * H)
  TAP("p1", c->hb.p1, N * S * h2)
  TAP("cat1", c->hb.cat1, N * H)
  TAP("a1b", c->hb.a1b, N * h2)
  TAP("y", c->hb.y, N)
  TAP("g_cat0", c->hb.g_cat0, N * 2 * H)
  TAP("g_vo", c->g_vo, N * S * H)
  TAP("g_x", c->g_x, N * H)
  TAP("g_n", c->g_n, N * H)
  TAP("g_pp", c->g_pp, e * 2 * H)
  TAP("g_rbf", c->g_rbf, e * Rp)
  TAP("g_geo", c->g_geo, e * VSN_GEO_W)
  TAP("g_ev", c->g_ev, e * 4)
  if (L_ok) {
    const LayerBuf& b = c->lb[layer];
    TAP("xn", b.xn, N * H)
    TAP("rstd", b.rstd, N)
    TAP("vh", b.vh, N * S * H)
    TAP("qkv", b.qkv, N * 3 * H)
    TAP("vp", b.vp, N * S * 5 * H)
    TAP("pe", b.pe, e * 3 * H)
    TAP("tpre", b.tpre, e * 2 * H)
    TAP("o", b.o, N * 3 * H)
  }
#undef TAP
  if (!p) {
    auto it = c->snap.find(k);
    if (it != c->snap.end() && layer >= 0 && layer < (int)it->second.size() && it->second[layer]) {
      p = it->second[layer];
      size_t cap = c->snap_elems[k + "#" + std::to_string(layer)];
      // edge-sized snapshots were taken with Emax rows; report the live rows only
      n = cap;
      if (k == "f_in" || k == "m" || k == "g_m" || k == "g_f_in") n = e * H;
      if (k == "g_t") n = e * 2 * H;
      if (k == "g_pe") n = e * 3 * H;
    }
  }
  (void)Bn;
  if (!p) return fail(c, -2, "unknown debug tap " + k);
  if ((int64_t)n > max_elems) n = (size_t)max_elems;
  if (hipMemcpy(host_out, p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, -5, "debug copy failed");
  return (int64_t)n;
}

extern "C" int vsn_gemm(vsn_handle c, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc,
                        const float* bias, int M, int Nc, int K, int flags, void* stream) {
  if (!c) return -22;
  HIPCHK(c, hipSetDevice(c->device));
  // test/bench tap: a persistent scratch buffer so the split-K path is exercised too
  static thread_local float* scratch = nullptr;
  static thread_local size_t scratch_elems = 0;
  const size_t elems = (size_t)8 * (size_t)std::max(M, 1) * (size_t)Nc;
  if (K >= 512 && (size_t)M * Nc <= ((size_t)1 << 26)) {
    if (elems > scratch_elems) {
      if (scratch) {
        hipDeviceSynchronize();
        hipFree(scratch);
      }
      scratch_elems = 0;
      if (hipMalloc((void**)&scratch, elems * sizeof(float)) == hipSuccess) scratch_elems = elems;
      else scratch = nullptr;
    }
    if (scratch) set_gemm_splitk_workspace(scratch, scratch_elems);
  }
  set_gemm_split3(c->gemm_split3 ? c->s3 : nullptr);  // (the tap follows the handle's arithmetic mode like vsn_forces)
  int rc = launch_gemm((hipStream_t)stream, A, lda, Bt, ldb, C, ldc, bias, M, nullptr, Nc, K, flags);
  set_gemm_split3(nullptr);
  set_gemm_splitk_workspace(nullptr, 0);
  if (rc) return fail(c, rc, "gemm: unsupported shape (K, Nc multiples of 32; lda/ldb/ldc multiples of 4; A, Bt, C, bias 16-byte aligned)");
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(c, -5, hipGetErrorString(le));
  return 0;
}

// ---------------------------------------------------------------------------------
// combine plan (Calculators/combiner.py:24-41)
// ---------------------------------------------------------------------------------
struct vsn_combine_plan {
  int device = 0;
  int n_prot = 0;
  int* off = nullptr;
  int* rows = nullptr;
  float* sign = nullptr;
  int n_e = 0;
  int* e_idx = nullptr;
  float* e_sign = nullptr;
};

extern "C" int vsn_combine_plan_create(vsn_combine_handle* out, int device_id, int64_t n_prot, int64_t n_cat,
                                       int64_t n_dip_rows, const int64_t* row_of_cat, const int64_t* select,
                                       const int64_t* origin, int64_t n_select) {
  if (!out || n_prot < 0 || n_select < 0) return -22;
  for (int64_t k = 0; k < n_select; ++k) {
    if (select[k] < 0 || select[k] >= n_cat || origin[k] < 0 || origin[k] >= n_prot) return -22;
  }
  std::vector<int> off((size_t)n_prot + 1, 0), rows((size_t)n_select);
  std::vector<float> sign((size_t)n_select);
  for (int64_t k = 0; k < n_select; ++k) off[(size_t)origin[k] + 1]++;
  for (int64_t a = 0; a < n_prot; ++a) off[(size_t)a + 1] += off[(size_t)a];
  std::vector<int> cur(off.begin(), off.end() - 1);
  for (int64_t k = 0; k < n_select; ++k) {  // stable: ascending k inside each origin
    int slot = cur[(size_t)origin[k]]++;
    rows[(size_t)slot] = (int)row_of_cat[select[k]];
    sign[(size_t)slot] = select[k] < n_dip_rows ? 1.0f : -1.0f;
  }
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  vsn_combine_plan* p = new vsn_combine_plan();
  p->device = device_id;
  p->n_prot = (int)n_prot;
  if (hipMalloc((void**)&p->off, off.size() * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&p->rows, std::max<size_t>(rows.size(), 1) * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&p->sign, std::max<size_t>(sign.size(), 1) * sizeof(float)) != hipSuccess) {
    delete p;
    return -12;
  }
  hipMemcpy(p->off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice);
  if (n_select) {
    hipMemcpy(p->rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(p->sign, sign.data(), sign.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  *out = p;
  return 0;
}

extern "C" void vsn_combine_plan_destroy(vsn_combine_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipFree(p->off);
  hipFree(p->rows);
  hipFree(p->sign);
  hipFree(p->e_idx);
  hipFree(p->e_sign);
  delete p;
}

extern "C" int vsn_combine_plan_set_energy(vsn_combine_handle p, int64_t n, const int64_t* host_index,
                                           const float* host_sign) {
  if (!p || n < 0 || (n > 0 && (!host_index || !host_sign))) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipFree(p->e_idx);
  hipFree(p->e_sign);
  p->e_idx = nullptr;
  p->e_sign = nullptr;
  p->n_e = 0;
  if (n == 0) return 0;
  std::vector<int> idx((size_t)n);
  for (int64_t k = 0; k < n; ++k) {
    if (host_index[k] < 0 || host_index[k] > 0x7fffffff) return -22;
    idx[(size_t)k] = (int)host_index[k];
  }
  if (hipMalloc((void**)&p->e_idx, (size_t)n * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&p->e_sign, (size_t)n * sizeof(float)) != hipSuccess)
    return -12;
  hipMemcpy(p->e_idx, idx.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice);
  hipMemcpy(p->e_sign, host_sign, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  p->n_e = (int)n;
  return 0;
}

extern "C" int vsn_combine_with_energy(vsn_combine_handle p, const float* dev_buf, float* dev_f_prot,
                                       float* dev_e_out, void* stream) {
  if (!p || !dev_e_out || p->n_e <= 0) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  launch_combine((hipStream_t)stream, p->n_prot, p->off, p->rows, p->sign, dev_buf, dev_f_prot, p->n_e, p->e_idx,
                 p->e_sign, dev_e_out);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_combine(vsn_combine_handle p, const float* dev_f_frag, float* dev_f_prot, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  launch_combine((hipStream_t)stream, p->n_prot, p->off, p->rows, p->sign, dev_f_frag, dev_f_prot);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------------
// per-step fragment geometry plan (Fragmentation/distancefrag.py:35-54)
// ---------------------------------------------------------------------------------
struct vsn_fragplan {
  int device = 0;
  int n = 0;
  int *src = nullptr, *acc = nullptr, *tow = nullptr;
  float* len = nullptr;
};

extern "C" int vsn_fragplan_create(vsn_fragplan_handle* out, int device_id, int64_t n, const int64_t* src,
                                   const int64_t* acc, const int64_t* tow, const float* len) {
  if (!out || n < 0 || !src || !acc || !tow || !len) return -22;
  std::vector<int> s((size_t)n), a((size_t)n), t((size_t)n);
  for (int64_t k = 0; k < n; ++k) {
    s[(size_t)k] = (int)src[k];
    a[(size_t)k] = (int)std::max<int64_t>(acc[k], 0);
    t[(size_t)k] = (int)std::max<int64_t>(tow[k], 0);
    if (src[k] < 0 && (acc[k] < 0 || tow[k] < 0 || acc[k] == tow[k])) return -22;
  }
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  vsn_fragplan* p = new vsn_fragplan();
  p->device = device_id;
  p->n = (int)n;
  size_t nb = std::max<size_t>((size_t)n, 1);
  if (hipMalloc((void**)&p->src, nb * 4) != hipSuccess || hipMalloc((void**)&p->acc, nb * 4) != hipSuccess ||
      hipMalloc((void**)&p->tow, nb * 4) != hipSuccess || hipMalloc((void**)&p->len, nb * 4) != hipSuccess) {
    delete p;
    return -12;
  }
  if (n) {
    hipMemcpy(p->src, s.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipMemcpy(p->acc, a.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipMemcpy(p->tow, t.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipMemcpy(p->len, len, (size_t)n * 4, hipMemcpyHostToDevice);
  }
  *out = p;
  return 0;
}

extern "C" void vsn_fragplan_destroy(vsn_fragplan_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipFree(p->src);
  hipFree(p->acc);
  hipFree(p->tow);
  hipFree(p->len);
  delete p;
}

extern "C" int vsn_build_fragments(vsn_fragplan_handle p, const float* prot, float* out, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  launch_build_fragments((hipStream_t)stream, p->n, p->src, p->acc, p->tow, p->len, prot, out);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// device views of the two plans for the launches fused with the integrator halves (md.hip)
int vsn_combine_plan_view(vsn_combine_plan* p, vsn::CombineView* out) {
  if (!p || !out) return -22;
  *out = vsn::CombineView{p->device, p->n_prot, p->n_e, p->off, p->rows, p->e_idx, p->sign, p->e_sign};
  return 0;
}
int vsn_fragplan_view(vsn_fragplan* p, vsn::FragView* out) {
  if (!p || !out) return -22;
  *out = vsn::FragView{p->device, p->n, p->src, p->acc, p->tow, p->len};
  return 0;
}

// ---------------------------------------------------------------------------------
// work partitions (Calculators/device_strategy.py:84-127): contiguous, atom-balanced
// blocks per device, each cut into chunks of at most ~chunk_atoms atoms, a
// straddling fragment going to the nearer side.
// ---------------------------------------------------------------------------------
static int64_t bisect_right(const int64_t* a, int64_t n, int64_t x) {
  return std::upper_bound(a, a + n, x) - a;
}

extern "C" int vsn_partition(const int64_t* start, const int64_t* end, int64_t B, int n_devices, int64_t chunk,
                             int64_t* out, int max_out) {
  if (!start || !end || B <= 0 || n_devices <= 0 || chunk <= 0) return -22;
  int cnt = 0;
  int64_t b_prev = 0;
  const int64_t a_end = B;
  for (int i = 0; i < n_devices; ++i) {
    if (b_prev >= a_end) break;  // (reference would raise IndexError here)
    int64_t block = (end[B - 1] - start[b_prev]) / (n_devices - i);
    int64_t b_end = bisect_right(start, B, block + start[b_prev]);
    int64_t b_idx = b_end - 1;
    int64_t block_end = block + start[b_prev];
    if ((block_end - start[b_idx]) < (end[b_idx] - block_end)) b_end -= 1;
    b_end = std::min(b_end, a_end);
    if (i == n_devices - 1) b_end = a_end;
    int64_t c_prev = b_prev;
    while (c_prev != b_end) {
      int64_t c_end = bisect_right(start, B, chunk + start[c_prev]);
      int64_t c_idx = c_end - 1;
      int64_t chunk_end = chunk + start[c_prev];
      if ((chunk_end - start[c_idx]) < (end[c_idx] - chunk_end)) c_end -= 1;
      c_end = std::min(c_end, b_end);
      if (c_end <= c_prev) c_end = c_prev + 1;  // a fragment larger than the chunk (reference would spin)
      if (cnt < max_out) {
        out[3 * cnt + 0] = i;
        out[3 * cnt + 1] = c_prev;
        out[3 * cnt + 2] = c_end;
      }
      ++cnt;
      c_prev = c_end;
    }
    b_prev = b_end;
  }
  return cnt;
}
