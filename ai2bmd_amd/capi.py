"""ctypes binding of the C ABI declared in include/vsn.h (libvsn_hip.so).

There is deliberately NO fallback: if the HIP library is missing or fails to
load, importing the product path raises.  (The CPU restatement under oracle/
is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSN_LIB overrides the library (A/B builds during tuning); the default is the in-tree build
LIB_PATH = os.environ.get("VSN_LIB") or os.path.join(_HERE, "libvsn_hip.so")


class VsnHParams(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_rbf", C.c_int32),
        ("num_heads", C.c_int32),
        ("lmax", C.c_int32),
        ("max_z", C.c_int32),
        ("max_num_neighbors", C.c_int32),
        ("vecnorm_type", C.c_int32),
        ("has_atomref", C.c_int32),
        ("cutoff", C.c_float),
        ("rbf_type", C.c_int32),
        ("activation", C.c_int32),
        ("attn_activation", C.c_int32),
    ]


class VsnHoptTerms(C.Structure):
    _i32p, _f32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int64)
    _fields_ = [
        ("n_rows", C.c_int64), ("n_cap", C.c_int32), ("cap_rows", _i64p), ("alias", _i64p),
        ("n_bond", C.c_int32), ("bond_i", _i32p), ("bond_j", _i32p), ("bond_k", _f32p), ("bond_r0", _f32p),
        ("n_angle", C.c_int32), ("angle_i", _i32p), ("angle_j", _i32p), ("angle_k", _i32p), ("angle_kf", _f32p),
        ("angle_th0", _f32p),
        ("n_dihedral", C.c_int32), ("dih_i", _i32p), ("dih_j", _i32p), ("dih_k", _i32p), ("dih_l", _i32p),
        ("dih_kf", _f32p), ("dih_per", _f32p), ("dih_phase", _f32p),
        ("n_pair", C.c_int32), ("pair_i", _i32p), ("pair_j", _i32p), ("pair_a", _f32p), ("pair_b", _f32p),
        ("pair_qq", _f32p),
        ("occ_ptr", _i32p), ("occ_type", _i32p), ("occ_term", _i32p), ("occ_end", _i32p), ("occ_w", _f32p),
        ("max_iter", C.c_int32), ("lr", C.c_float), ("tolerance_grad", C.c_float), ("tolerance_change", C.c_float),
        ("scnb", C.c_float), ("scee", C.c_float),
    ]


VECNORM = {"none": 0, "rms": 1, "max_min": 2}
RBF = {"expnorm": 0, "gauss": 1}
ACTIVATION = {"silu": 0, "swish": 0, "ssp": 1, "tanh": 2, "sigmoid": 3}  # utils.py:93-116 act_class_mapping
P2P_HANDLE_BYTES = 128  # VSN_P2P_HANDLE_BYTES (include/vsn.h)
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime; load it FIRST so this process has exactly
    # one libamdhip64 (loading /opt/rocm's copy before torch's leaves two runtimes in the
    # process and hipSetDevice then fails).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X ViSNet calculator needs its HIP extension "
            "(build it with `python -m ai2bmd_amd.build`); there is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, i64p, f32p = C.c_void_p, C.POINTER(C.c_int64), C.c_void_p
    L.vsn_create.argtypes = [C.POINTER(vp), C.POINTER(VsnHParams), C.c_int]
    L.vsn_create.restype = C.c_int
    L.vsn_destroy.argtypes = [vp]
    L.vsn_destroy.restype = None
    L.vsn_last_error.argtypes = [vp]
    L.vsn_last_error.restype = C.c_char_p
    L.vsn_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, C.c_int]
    L.vsn_load_weight.restype = C.c_int
    L.vsn_finalize.argtypes = [vp]
    L.vsn_finalize.restype = C.c_int
    L.vsn_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.vsn_set_option.restype = C.c_int
    L.vsn_forces.argtypes = [vp, vp, f32p, i64p, i64p, C.c_int64, C.c_int64, f32p, f32p, vp]
    L.vsn_forces.restype = C.c_int
    L.vsn_profile_read.argtypes = [vp, C.POINTER(C.c_double)]
    L.vsn_profile_read.restype = C.c_int
    L.vsn_profile_read_scatter.argtypes = [vp, C.POINTER(C.c_double)]
    L.vsn_profile_read_scatter.restype = C.c_int
    L.vsn_profile_read_walks.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.vsn_profile_read_walks.restype = C.c_int
    L.vsn_walk_alg_bytes.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    L.vsn_walk_alg_bytes.restype = C.c_double
    L.vsn_profile_bracket_ms.argtypes = [vp]
    L.vsn_profile_bracket_ms.restype = C.c_double
    L.vsn_last_num_edges.argtypes = [vp]
    L.vsn_last_num_edges.restype = C.c_int64
    L.vsn_last_status.argtypes = [vp]
    L.vsn_last_status.restype = C.c_int
    L.vsn_debug_read.argtypes = [vp, C.c_char_p, C.c_int, vp, C.c_int64]
    L.vsn_debug_read.restype = C.c_int64
    L.vsn_gemm.argtypes = [vp, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int,
                           C.c_int, vp]
    L.vsn_gemm.restype = C.c_int
    L.vsn_combine_plan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int64, C.c_int64, C.c_int64, i64p, i64p,
                                          i64p, C.c_int64]
    L.vsn_combine_plan_create.restype = C.c_int
    L.vsn_combine_plan_destroy.argtypes = [vp]
    L.vsn_combine_plan_destroy.restype = None
    L.vsn_combine.argtypes = [vp, f32p, f32p, vp]
    L.vsn_combine.restype = C.c_int
    L.vsn_combine_plan_set_energy.argtypes = [vp, C.c_int64, i64p, C.POINTER(C.c_float)]
    L.vsn_combine_plan_set_energy.restype = C.c_int
    L.vsn_combine_with_energy.argtypes = [vp, f32p, f32p, f32p, vp]
    L.vsn_combine_with_energy.restype = C.c_int
    L.vsn_fragplan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int64, i64p, i64p, i64p, C.POINTER(C.c_float)]
    L.vsn_fragplan_create.restype = C.c_int
    L.vsn_fragplan_destroy.argtypes = [vp]
    L.vsn_fragplan_destroy.restype = None
    L.vsn_build_fragments.argtypes = [vp, f32p, f32p, vp]
    L.vsn_build_fragments.restype = C.c_int
    L.vsn_md_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int64, C.POINTER(C.c_float), C.c_float, C.c_float,
                                C.c_float, C.c_uint64, C.c_float, C.POINTER(C.c_float)]
    L.vsn_md_create.restype = C.c_int
    L.vsn_md_destroy.argtypes = [vp]
    L.vsn_md_destroy.restype = None
    L.vsn_md_half1.argtypes = [vp, f32p, f32p, f32p, vp]
    L.vsn_md_half1.restype = C.c_int
    L.vsn_md_half2.argtypes = [vp, f32p, f32p, f32p, vp]
    L.vsn_md_half2.restype = C.c_int
    L.vsn_p2p_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int64]
    L.vsn_p2p_create.restype = C.c_int
    L.vsn_p2p_export.argtypes = [vp, vp]
    L.vsn_p2p_export.restype = C.c_int
    L.vsn_p2p_connect.argtypes = [vp, vp]
    L.vsn_p2p_connect.restype = C.c_int
    L.vsn_p2p_send_buffer.argtypes = [vp]
    L.vsn_p2p_send_buffer.restype = vp
    L.vsn_p2p_gather_buffer.argtypes = [vp, C.c_int]
    L.vsn_p2p_gather_buffer.restype = vp
    L.vsn_p2p_set_timeout.argtypes = [vp, C.c_double]
    L.vsn_p2p_set_timeout.restype = C.c_int
    L.vsn_p2p_allgather.argtypes = [vp, vp, C.POINTER(vp)]
    L.vsn_p2p_allgather.restype = C.c_int
    L.vsn_p2p_status.argtypes = [vp, vp]
    L.vsn_p2p_status.restype = C.c_int
    L.vsn_p2p_destroy.argtypes = [vp]
    L.vsn_p2p_destroy.restype = None
    L.vsn_md_half1_build_relax.argtypes = [vp, f32p, f32p, f32p, vp, f32p, vp, vp]
    L.vsn_md_half1_build_relax.restype = C.c_int
    L.vsn_md_set_noise.argtypes = [vp, vp, vp]
    L.vsn_md_set_noise.restype = C.c_int
    L.vsn_md_half1_build.argtypes = [vp, f32p, f32p, f32p, vp, f32p, vp]
    L.vsn_md_half1_build.restype = C.c_int
    L.vsn_md_combine_half2.argtypes = [vp, vp, f32p, f32p, f32p, f32p, f32p, vp]
    L.vsn_md_combine_half2.restype = C.c_int
    fp_ = C.POINTER(C.c_float)
    L.vsn_md_set_restraints.argtypes = [vp, C.c_int64, i64p, fp_, fp_, fp_, C.c_int64, i64p, i64p, fp_, fp_]
    L.vsn_md_set_restraints.restype = C.c_int
    L.vsn_md_restrain.argtypes = [vp, f32p, f32p, vp]
    L.vsn_md_restrain.restype = C.c_int
    L.vsn_md_observe.argtypes = [vp, f32p, C.c_float, f32p, vp]
    L.vsn_md_observe.restype = C.c_int
    L.vsn_mm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.vsn_mm_create.restype = C.c_int
    L.vsn_mm_destroy.argtypes = [vp]
    L.vsn_mm_destroy.restype = None
    L.vsn_mm_forces.argtypes = [vp, f32p, f32p, f32p, C.c_int, vp]
    L.vsn_mm_forces.restype = C.c_int
    L.vsn_hopt_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(VsnHoptTerms)]
    L.vsn_hopt_create.restype = C.c_int
    L.vsn_hopt_destroy.argtypes = [vp]
    L.vsn_hopt_destroy.restype = None
    L.vsn_hopt_run.argtypes = [vp, f32p, vp]
    L.vsn_hopt_run.restype = C.c_int
    L.vsn_hopt_stats.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double), vp]
    L.vsn_hopt_stats.restype = C.c_int
    L.vsn_partition.argtypes = [i64p, i64p, C.c_int64, C.c_int, C.c_int64, i64p, C.c_int]
    L.vsn_partition.restype = C.c_int
    _lib = L
    return L


def i64_ptr(a):
    """numpy int64 C-contiguous array -> POINTER(c_int64)"""
    return a.ctypes.data_as(C.POINTER(C.c_int64))


EXPORTS = [
    "vsn_create", "vsn_destroy", "vsn_last_error", "vsn_load_weight", "vsn_finalize", "vsn_set_option",
    "vsn_forces", "vsn_profile_read", "vsn_profile_read_scatter", "vsn_profile_read_walks", "vsn_walk_alg_bytes", "vsn_profile_bracket_ms", "vsn_last_num_edges", "vsn_last_status", "vsn_debug_read", "vsn_gemm", "vsn_combine_plan_create",
    "vsn_combine_plan_destroy", "vsn_combine", "vsn_combine_plan_set_energy", "vsn_combine_with_energy", "vsn_partition", "vsn_fragplan_create", "vsn_fragplan_destroy",
    "vsn_p2p_create", "vsn_p2p_export", "vsn_p2p_connect", "vsn_p2p_send_buffer", "vsn_p2p_gather_buffer", "vsn_p2p_set_timeout", "vsn_p2p_allgather", "vsn_p2p_status", "vsn_p2p_destroy",
    "vsn_build_fragments", "vsn_md_create", "vsn_md_destroy", "vsn_md_half1", "vsn_md_half2", "vsn_md_set_noise", "vsn_md_half1_build", "vsn_md_half1_build_relax", "vsn_md_combine_half2", "vsn_md_set_restraints", "vsn_md_restrain", "vsn_md_observe", "vsn_mm_create", "vsn_mm_destroy", "vsn_mm_forces",
    "vsn_hopt_create", "vsn_hopt_destroy", "vsn_hopt_run", "vsn_hopt_stats",
]
