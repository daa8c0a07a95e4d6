/*
 * vsn.h - C ABI of the MI355X-native ViSNet energy+force calculator.
 *
 * This is the drop-in boundary for AI2BMD's per-MD-step hot path.  Every entry
 * point below replaces one interface of the reference (paths relative to
 * /root/reference/src):
 *
 *   vsn_create / vsn_load_weight / vsn_finalize
 *        <- ViSNet/model/visnet.py:73-93  load_model(): build the network from
 *           ckpt["hyper_parameters"], load ckpt["state_dict"] (221 tensors at the
 *           defaults, names listed in SURVEY.md 8a/a3), freeze, move to device.
 *   vsn_forces
 *        <- Calculators/visnet_calculator.py:54-63  ViSNetModel.dl_potential_loader
 *           (= ViSNet.forward, ViSNet/model/visnet.py:135-166: radius graph,
 *           RBF/SH, embeddings, L x ViS-MP, gated equivariant read-out,
 *           per-fragment energy sum, F = -dE/dpos).
 *   vsn_combine_plan_create / vsn_combine
 *        <- Calculators/combiner.py:12-41  DipeptideBondedCombiner
 *           (E = sum E_dip - sum E_ace ; F = scatter_sum(cat[F_dip,-F_ace][select], origin)).
 *   vsn_partition
 *        <- Calculators/device_strategy.py:84-127 _set_combined_work_partitions.
 *
 * Conventions: plain C types only; all `dev_*` pointers are HIP device
 * pointers owned by the caller (e.g. PyTorch-ROCm tensors); `host_*` pointers
 * are host memory.  Every function returns 0 on success or a negative code;
 * vsn_last_error() gives the message (the Python host raises RuntimeError,
 * matching the reference where exceptions propagate to the ASE caller).
 * A handle is bound to one device and must not be used concurrently from two
 * threads; different handles (different devices) may run concurrently, which is
 * how DLBondedCalculator drives them (Calculators/bonded.py:75-77).
 */
#ifndef VSN_H
#define VSN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsn_ctx* vsn_handle;

enum { VSN_VECNORM_NONE = 0, VSN_VECNORM_RMS = 1, VSN_VECNORM_MAXMIN = 2 };
enum { VSN_RBF_EXPNORM = 0, VSN_RBF_GAUSS = 1 };
/* act_class_mapping (ViSNet/model/utils.py:93-116): "silu" and "swish" are the same function */
enum { VSN_ACTIVATION_SILU = 0, VSN_ACTIVATION_SSP = 1, VSN_ACTIVATION_TANH = 2, VSN_ACTIVATION_SIGMOID = 3 };

/* Hyper-parameters read by create_model (ViSNet/model/visnet.py:15-30). */
typedef struct vsn_hparams {
  int32_t hidden;            /* embedding_dimension: a multiple of 64, 64..512 */
  int32_t num_layers;        /* num_layers                                     */
  int32_t num_rbf;           /* num_rbf                                        */
  int32_t num_heads;         /* <= 64, dividing hidden (visnet_block.py:158-166)   */
  int32_t lmax;              /* 1 | 2                                          */
  int32_t max_z;             /* embedding rows                                 */
  int32_t max_num_neighbors; /* radius_graph truncation (incl. the self loop)  */
  int32_t vecnorm_type;      /* VSN_VECNORM_*                                  */
  int32_t has_atomref;       /* prior_model == "Atomref" (table length = prior_args.max_z, taken from the tensor) */
  float cutoff;              /* Angstrom                                       */
  int32_t rbf_type;          /* VSN_RBF_EXPNORM | VSN_RBF_GAUSS   (utils.py:22-90 rbf_class_mapping)      */
  int32_t activation;        /* VSN_ACT_* of "activation": dk/dv/s_proj/f_proj and the read-out MLPs       */
  int32_t attn_activation;   /* VSN_ACT_* of "attn_activation": the attention scores (visnet_block.py:278) */
} vsn_hparams;

int vsn_create(vsn_handle* out, const vsn_hparams* hp, int device_id);
void vsn_destroy(vsn_handle h);
const char* vsn_last_error(vsn_handle h);

/* Copies one state_dict tensor (reference key name without the "model."
 * prefix) from `ptr` (device or host memory, fp32, C-contiguous). */
int vsn_load_weight(vsn_handle h, const char* name, const void* ptr, const int64_t* shape, int ndim);
/* Packs/transposes the loaded tensors into the kernels' layouts; fails if a
 * required tensor is missing. */
int vsn_finalize(vsn_handle h);

/* Options: "max_chunk_edges" (workspace bound, default 1310720 edge slots ~ 105 GB at H=256, L=9), "debug" (1 = keep per-layer
 * snapshots for vsn_debug_read), "profile" (1 = time every GEMM launch, see vsn_profile_read), "overlap" (bit 0 / bit 1 =
 * run the forward / reverse side work on a second HIP stream, default 2), "fuse_fwd", "fuse_bwd" (vertical fusions,
 * default 1), "fuse_head" (one-launch node-local read-out on single-protein sizes, default 1), "split_rev" (K-slices
 * of the long-K reverse products summed by their consumer, default 1), "fuse_panel" (fragment batches at hidden 256:
 * the target-side vector-message / attention adjoints as prologues of panel GEMMs, csrc/fused.hip, default 1). */
int vsn_set_option(vsn_handle h, const char* key, int64_t value);

/*
 * Energy and forces of a batch of B fragments holding N atoms in total.
 *   dev_z        int64 [N]    atomic numbers                (FragmentData.z)
 *   dev_pos      f32   [N,3]  positions, Angstrom           (FragmentData.pos)
 *   host_start   int64 [B]    first atom of each fragment   (FragmentData.start)
 *   host_end     int64 [B]    one-past-last atom            (FragmentData.end)
 *   dev_e_out    f32   [B]    per-fragment energy (empty fragments: `mean`)
 *   dev_f_out    f32   [N,3]  forces = -dE/dpos
 * Asynchronous on `stream` (a hipStream_t); outputs are valid once the stream
 * has drained.  Fragments must be contiguous and ascending (start[b] == end[b-1]).
 */
int vsn_forces(vsn_handle h, const int64_t* dev_z, const float* dev_pos, const int64_t* host_start,
               const int64_t* host_end, int64_t N, int64_t B, float* dev_e_out, float* dev_f_out, void* stream);

/* With option "profile"=1 every GEMM launch is bracketed by HIP events on the launch stream;
 * this returns, per GEMM kernel v in {0: k_gemm<128,128>, 1: k_gemm<64,64>, 2: k_gemm<128,32>,
 * 3: k_gemm_group (64x64 tiles, several products per launch)},
 * out[4v..4v+3] = {launches, total ms, total algorithmic flops, total algorithmic bytes}
 * accumulated since the option was set. */
int vsn_profile_read(vsn_handle h, double* out16);
/* Same profile mode, the scatter path: the forward edge-attention launch (k_edge_attn / k_edge_attn_update, v = 0) and
 * the vector-message aggregation + node update (k_node_update, v = 1) of every layer are bracketed too;
 * out[4v..4v+3] = {launches, total ms, total ALGORITHMIC HBM bytes (every array the launch touches, once), 0}.
 * (replaces reading torch_scatter / MessagePassing.propagate time off a profiler in the reference,
 *  ViSNet/model/visnet_block.py:276-312) */
int vsn_profile_read_scatter(vsn_handle h, double* out8);
/* The same records for ALL timed node walks: v = 0, 1 as above; 2 = k_bwd_hf1 (reverse vector messages, both sides,
 * + the per-edge half and the source side of the edge-update adjoint), 3 = k_bwd_hf2 (reverse attention target side
 * + the per-node half of the edge-update adjoint), 4 = k_bwd_attn_S, 5 = k_bwd_norm_update (LayerNorm adjoint + node
 * update adjoint) - the hand-derived reverse of visnet_block.py:237-312 at single-protein sizes.  Every launch
 * carries two events on its own dispatch packet (begin..end timestamps, no bracket correction).  Writes
 * min(max_kinds, 6) rows of 4 doubles {launches, total ms, total algorithmic bytes, 0}; returns the rows written. */
int vsn_profile_read_walks(vsn_handle h, double* out, int max_kinds);
/* THE byte model behind the figures above (and behind tools/walk_table.py, which calls it through ctypes instead of
 * restating it): algorithmic HBM bytes of ONE launch of node walk `kernel` ("k_edge_attn", "k_edge_attn_update",
 * "k_edge_update", "k_node_update", "k_bwd_hf1", "k_bwd_hf2", "k_bwd_attn_S", "k_bwd_norm_update", "k_bwd_gm_fused",
 * "k_bwd_gf_fused", "k_bwd_edge_update_T", "k_bwd_edge_update_S", "k_bwd_vecmsg_S") over n nodes and e edges at hidden
 * width H, S spherical components, nh heads - every distinct array the launch reads or writes, once.  f0 / f1: the
 * launch's variant flags (csrc/engine.hip).  Pure function, no device; < 0 for an unknown kernel name.
 * (the reference has no counterpart: it reads times off a profiler and never prices a launch against a roofline) */
double vsn_walk_alg_bytes(const char* kernel, int H, int S, int nh, double n, double e, int f0, int f1);
/* Average time (ms) between the two events of an EMPTY bracket on the launch stream, measured in the same profiled
 * calls (8 per chunk): what the bracket itself adds to every per-launch figure above.  0 when nothing was measured. */
double vsn_profile_bracket_ms(vsn_handle h);

/* Device-side edge count of the last chunk processed (synchronises). */
int64_t vsn_last_num_edges(vsn_handle h);

/* Device-side status word of the last chunk (synchronises): 0 = ok, bit 0 = an atomic number was outside
 * [0, min(max_z, atomref rows)) - the reference's nn.Embedding raises IndexError there
 * (ViSNet/model/visnet_block.py:110, priors.py:86-87); vsn_forces is asynchronous, so the kernels clamp the
 * index (no out-of-bounds read) and write NaN to every energy and force of the chunk instead. */
int vsn_last_status(vsn_handle h);

/* Debug/verification: copies a named internal buffer of the LAST chunk to host
 * memory (synchronises).  Returns the number of floats (or int32s) written, or
 * a negative code.  Names: see DESIGN.md "debug taps". */
int64_t vsn_debug_read(vsn_handle h, const char* name, int layer, void* host_out, int64_t max_elems);

/* Stand-alone GEMM tap used by the unit tests and the roofline bench:
 * C[M,Nc] (+)= A[M,K] * Bt[Nc,K]^T (+ bias). flags: 1 = accumulate, 2 = silu(A).
 * K and Nc multiples of 32; lda, ldb, ldc multiples of 4 floats; A, Bt, C and bias 16-byte aligned (the tiles move
 * 16-byte row pieces); anything else returns -22. */
int vsn_gemm(vsn_handle h, const float* dev_A, int lda, const float* dev_Bt, int ldb, float* dev_C, int ldc,
             const float* dev_bias, int M, int Nc, int K, int flags, void* stream);

/* ---- the exchange step of the sharded path as direct peer writes (SURVEY.md 8e "tuned variant") ----
 * Replaces, for one process per GPU, what the reference does with worker processes and a host concatenate
 * (Calculators/bonded.py:65-89, visnet_calculator.py:78-118): every rank's slot (forces of its fragment rows + energies
 * of its fragments, `slot_floats` floats) is gathered on every rank.  One kernel per step STORES the rank's slot into
 * each peer's gather buffer (buffers mapped with hipIpcGetMemHandle / hipIpcOpenMemHandle, xGMI point-to-point), raises
 * a per-peer flag and waits for the peers' flags; double-buffered by step parity (csrc/p2p.hip).  Bitwise the buffer
 * `torch.distributed.all_gather_into_tensor` produces; that call stays the default exchange.
 *   create -> export (VSN_P2P_HANDLE_BYTES per rank) -> [the host all-gathers the handle records once, any transport]
 *   -> connect -> per step: the rank's kernels write vsn_p2p_send_buffer(), then vsn_p2p_allgather(stream, &buf): `buf`
 *   (device, [world][slot_floats]) is complete when the launch retires on `stream`.  All ranks must make the same
 *   sequence of vsn_p2p_allgather calls; destroy only after every rank has finished its last one (host barrier).
 * vsn_p2p_status synchronises `stream` and returns 0, or the step whose wait gave up after the timeout (a peer died). */
typedef struct vsn_p2p* vsn_p2p_handle;
#define VSN_P2P_HANDLE_BYTES 128
int vsn_p2p_create(vsn_p2p_handle* out, int device_id, int rank, int world, int64_t slot_floats);
int vsn_p2p_export(vsn_p2p_handle p, void* handle_bytes /* [VSN_P2P_HANDLE_BYTES] */);
int vsn_p2p_connect(vsn_p2p_handle p, const void* all_handles /* [world][VSN_P2P_HANDLE_BYTES], rank order */);
float* vsn_p2p_send_buffer(vsn_p2p_handle p);                 /* device f32 [slot_floats] */
float* vsn_p2p_gather_buffer(vsn_p2p_handle p, int parity);   /* device f32 [world][slot_floats], parity 0 | 1 */
int vsn_p2p_set_timeout(vsn_p2p_handle p, double seconds);    /* default 5 s */
int vsn_p2p_allgather(vsn_p2p_handle p, void* stream, float** dev_gathered_out);
int vsn_p2p_status(vsn_p2p_handle p, void* stream);
void vsn_p2p_destroy(vsn_p2p_handle p);

/* ---- overlap-force recombination (Calculators/combiner.py:24-41) ---- */
typedef struct vsn_combine_plan* vsn_combine_handle;
/* select/origin as in forces_combine; `n_dip_rows` = rows of the dipeptide
 * block in cat[F_dip, F_ace]; `host_row_of_cat[k]` maps row k of that
 * concatenation to its row in the interleaved fragment-force array. */
int vsn_combine_plan_create(vsn_combine_handle* out, int device_id, int64_t n_prot, int64_t n_cat,
                            int64_t n_dip_rows, const int64_t* host_row_of_cat, const int64_t* host_select,
                            const int64_t* host_origin, int64_t n_select);
void vsn_combine_plan_destroy(vsn_combine_handle p);
/* dev_f_frag f32 [n_cat,3] (interleaved order) -> dev_f_prot f32 [n_prot,3];
 * dev_e_frag f32 [B] with host-provided sign per fragment folded into plan. */
int vsn_combine(vsn_combine_handle p, const float* dev_f_frag, float* dev_f_prot, void* stream);
/* total energy in the same launch (DipeptideBondedCombiner.energy_combine, combiner.py:12-22):
 * E = sum_k host_sign[k] * dev_buf[host_index[k]] over the n per-fragment energies (float offsets into the buffer
 * that also holds the forces; +1 dipeptide, -1 ACE-NME), summed in a fixed order -> dev_e_out f32 [1]. */
int vsn_combine_plan_set_energy(vsn_combine_handle p, int64_t n, const int64_t* host_index, const float* host_sign);
int vsn_combine_with_energy(vsn_combine_handle p, const float* dev_buf, float* dev_f_prot, float* dev_e_out,
                            void* stream);

/* ---- per-step fragment geometry (Fragmentation/distancefrag.py:35-54 + fragments_index gather) ----
 * Row k of the fragment batch is either a copy of protein atom host_src[k] (>= 0) or, when
 * host_src[k] < 0, a cap hydrogen placed at  acceptor + len * unit(toward - acceptor). */
typedef struct vsn_fragplan* vsn_fragplan_handle;
int vsn_fragplan_create(vsn_fragplan_handle* out, int device_id, int64_t n_frag_atoms, const int64_t* host_src,
                        const int64_t* host_acceptor, const int64_t* host_toward, const float* host_len);
void vsn_fragplan_destroy(vsn_fragplan_handle p);
/* dev_prot_pos f32 [n_prot,3] -> dev_frag_pos f32 [n_frag_atoms,3] */
int vsn_build_fragments(vsn_fragplan_handle p, const float* dev_prot_pos, float* dev_frag_pos, void* stream);

/* ---- Langevin integrator step on the device (AIMD/simulator.py:96-116 -> ASE Langevin.step) ----
 * ASE units (eV, Angstrom, amu): dt and friction in ASE time units, kT in eV.  One step = vsn_md_half1, force
 * evaluation at the new positions into dev_F, vsn_md_half2.  Restraints (ASE `Hookean`, simulator.py:139-180) are
 * evaluated by half2 at the new positions and ADDED INTO dev_F, so dev_F = model + restraint forces afterwards
 * (what atoms.get_forces() returns in ASE) and the next half1 reads it as is; call vsn_md_restrain once for the
 * forces of the start geometry.  tether_k > 0 = every atom restrained to host_x0 with threshold 0.
 * PARITY UNPINNED (ASE absent): the coefficients and the Hookean law restate ASE 3.22's published algorithm. */
typedef struct vsn_md* vsn_md_handle;
int vsn_md_create(vsn_md_handle* out, int device_id, int64_t n_atoms, const float* host_mass, float dt, float kT,
                  float friction, uint64_t seed, float tether_k, const float* host_x0);
void vsn_md_destroy(vsn_md_handle p);
int vsn_md_half1(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F, void* stream);
int vsn_md_half2(vsn_md_handle p, const float* dev_x, float* dev_v, float* dev_F, void* stream);
/* The normal draws of the following first halves come from the caller instead of the built-in counter-based generator:
 * dev_xi, dev_eta f32 [n_atoms,3] each (device, borrowed, re-read by every first half - rewrite them between steps), in
 * the order ASE's Langevin.step draws them (xi, then eta).  The reference feeds ASE from its RNGPool of numpy normals
 * (simulator.py:108 `rng=RNGPool(seed, (n, 3), count=2)`, utils/utils.py:28-49): a trajectory can only be laid beside
 * ASE's on the same draws (tests/test_md_vs_ase.py).  (NULL, NULL) returns to the built-in generator. */
int vsn_md_set_noise(vsn_md_handle p, const float* dev_xi, const float* dev_eta);
/* The same two halves with the neighbouring launch of the force evaluation folded in (same arithmetic, same order,
 * bitwise the same trajectory; two launches fewer per step):
 *   vsn_md_half1_build    = vsn_md_half1, then vsn_build_fragments(plan) of the NEW positions into dev_frag_pos
 *                           (the gather DistanceFragment.get_fragments starts with, distancefrag.py:35-54);
 *   vsn_md_combine_half2  = vsn_combine_with_energy(plan) of the (all-gathered) fragment buffer dev_buf into dev_F
 *                           [n_atoms,3] and dev_e_out [1] (combiner.py:12-41), then vsn_md_half2 on that dev_F.
 * The plans must live on the integrator's device; the combine plan needs its energy terms set and n_prot = n_atoms. */
int vsn_md_half1_build(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F, vsn_fragplan_handle plan,
                       float* dev_frag_pos, void* stream);
int vsn_md_combine_half2(vsn_md_handle p, vsn_combine_handle plan, const float* dev_buf, float* dev_F,
                         float* dev_e_out, const float* dev_x, float* dev_v, void* stream);
/*   vsn_md_half1_build_relax = vsn_md_half1_build, then vsn_hopt_run(hopt) on dev_frag_pos - the whole start of a step
 *                           (Langevin half, DistanceFragment.get_fragments: placement distancefrag.py:35-54 + L-BFGS
 *                           relaxation hydrogen/energies.py:211-242) in ONE launch; bitwise the two calls. */
struct vsn_hopt;
int vsn_md_half1_build_relax(vsn_md_handle p, float* dev_x, float* dev_v, const float* dev_F, vsn_fragplan_handle plan,
                             float* dev_frag_pos, struct vsn_hopt* hopt, void* stream);
/* Replaces the restraint set (synchronises the device).  Point springs: atom[t] towards origin3[3t..], force
 * k (r - rt) along the line when r > rt, energy k (r - rt)^2 / 2 - `Hookean(a1=idx, a2=pos, k, rt)`, the
 * pre-equilibration stages (simulator.py:139-166).  Pair springs between atoms a1[t], a2[t] - `Hookean(a1, a2, k, rt)`,
 * the X-H bond restraints of --hydrogen-constraints (simulator.py:168-180).  n_point = n_pair = 0 removes all. */
int vsn_md_set_restraints(vsn_md_handle p, int64_t n_point, const int64_t* host_atom, const float* host_origin3,
                          const float* host_k_point, const float* host_rt_point, int64_t n_pair,
                          const int64_t* host_a1, const int64_t* host_a2, const float* host_k_pair,
                          const float* host_rt_pair);
/* dev_F += restraint forces at dev_x (start geometry; half2 does this itself on every step) */
int vsn_md_restrain(vsn_md_handle p, const float* dev_x, float* dev_F, void* stream);
/* observer hook (utils/utils.py:143-159 printenergy) without a host round trip: dev_out3 = {kinetic energy,
 * restraint energy of the last evaluation, temperature = 2 Ekin / (3 n kB)} */
int vsn_md_observe(vsn_md_handle p, const float* dev_v, float kB, float* dev_out3, void* stream);

/* ---- MM non-bonded term between atoms that do not share a dipeptide (Calculators/nonbonded.py:33-63) ----
 * charges [e], sigma [nm], epsilon [kJ/mol] as OpenMM gives them (AIMD/protein.py:153-175);
 * host_groups4 int32 [n,4]: ids of the dipeptides each atom belongs to, -1 padded (pairs whose id sets
 * intersect are excluded, distancefrag.py:355-363).  Outputs in ASE units (eV, eV/Angstrom). */
typedef struct vsn_mm* vsn_mm_handle;
int vsn_mm_create(vsn_mm_handle* out, int device_id, int64_t n_atoms, const float* host_charge,
                  const float* host_sigma, const float* host_epsilon, const int32_t* host_groups4);
void vsn_mm_destroy(vsn_mm_handle p);
/* dev_pos f32 [n,3] -> dev_e f32 [1], dev_f f32 [n,3] (added to dev_f when accumulate != 0) */
int vsn_mm_forces(vsn_mm_handle p, const float* dev_pos, float* dev_e, float* dev_f, int accumulate, void* stream);

/* ---- cap-hydrogen relaxation (Fragmentation/distancefrag.py:30-32,56-92 get_fragments ->
 * hydrogen/energies.py:211-242 HydrogenOptimizer.optimize_hydrogen; term selection hydrogen/ctable.py:168-244) ----
 * Flat term lists over rows of the fragment position array; every term touches at least one cap hydrogen, and a cap
 * hydrogen is always the first or the last atom of a term.  occ_* is a CSR over cap hydrogens of (type 0 bond /
 * 1 angle / 2 dihedral / 3 non-bonded pair, term index, end 0 first / 1 last, energy share 1/#caps in the term).
 * Energies in kcal/mol as in the AMBER tables (charges pre-scaled by 18.2223). */
typedef struct vsn_hopt_terms {
  int64_t n_rows;                 /* rows of dev_frag_pos */
  int32_t n_cap;
  const int64_t* cap_rows;        /* [n_cap] rows that are optimised */
  const int64_t* alias;           /* [n_rows] row to copy from after the relaxation (ACE-NME rows), -1 = none; or NULL */
  int32_t n_bond;     const int32_t *bond_i, *bond_j;                 const float *bond_k, *bond_r0;
  int32_t n_angle;    const int32_t *angle_i, *angle_j, *angle_k;     const float *angle_kf, *angle_th0;
  int32_t n_dihedral; const int32_t *dih_i, *dih_j, *dih_k, *dih_l;   const float *dih_kf, *dih_per, *dih_phase;
  int32_t n_pair;     const int32_t *pair_i, *pair_j;                 const float *pair_a, *pair_b, *pair_qq;
  const int32_t *occ_ptr, *occ_type, *occ_term, *occ_end;             const float* occ_w;
  int32_t max_iter;               /* 10  (distancefrag.py:30) ; <= 16 */
  float lr, tolerance_grad, tolerance_change;   /* 0.1, 0.1, 0.01 (energies.py:232-238) */
  float scnb, scee;               /* 1.2, 2.0 (energies.py:76-81) */
} vsn_hopt_terms;
typedef struct vsn_hopt* vsn_hopt_handle;
int vsn_hopt_create(vsn_hopt_handle* out, int device_id, const vsn_hopt_terms* host_terms);
void vsn_hopt_destroy(vsn_hopt_handle p);
/* relaxes the cap rows of dev_frag_pos f32 [n_rows,3] in place (one L-BFGS .step), then applies `alias` */
int vsn_hopt_run(vsn_hopt_handle p, float* dev_frag_pos, void* stream);
/* synchronises `stream`; host_iters_evals int32[2] = {L-BFGS iterations, energy evaluations},
 * host_loss_first_last f64[2] (either may be NULL) */
int vsn_hopt_stats(vsn_hopt_handle p, int32_t* host_iters_evals, double* host_loss_first_last, void* stream);

/* ---- work partitions (Calculators/device_strategy.py:84-127) ---- */
/* Writes up to max_out triples (device_idx, frag_begin, frag_end); returns count. */
int vsn_partition(const int64_t* host_start, const int64_t* host_end, int64_t B, int n_devices,
                  int64_t chunk_atoms, int64_t* out_triples, int max_out);

#ifdef __cplusplus
}
#endif
#endif /* VSN_H */
