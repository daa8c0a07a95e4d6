"""GPU: the opt-in product mode `gemm_split3` (vsn_set_option): the grouped products of single-protein sizes as
3 x bf16 split MFMA products with fp32 accumulation (csrc/gemm_s3.h) instead of the default fp32 MFMA chain.

The mode has to meet the SAME contract as the default arithmetic (vs the fp64 truth of the reference's
source): per-fragment |dE| <= 1e-5 max(1,|E|), force MAE <= 1e-5 max(1, mean|F|), max|dF| <= 1e-4 max(1, max|F|),
and no worse than 4x the reference's own fp32 error - on every reference-source golden, on the four protein batches
the bench runs, on the seeded fuzz draws and through the device pipeline.  The headline never runs in this mode."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, HIP_CASES, load_golden
from oracle.inputs import random_fragments
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import default_hparams, make_state_dict
from test_gpu_fuzz import draw
from test_gpu_parity import check, frag, model_for

pytestmark = pytest.mark.gpu


def split3(m):
    m.engine.set_option("gemm_split3", 1)
    return m


@pytest.mark.parametrize("name", HIP_CASES)
def test_golden_reference_vectors_in_split3_mode(lib_built, name):
    g = load_golden(name)
    m = split3(model_for(g["hparams"], g["weight_seed"]))
    e, f = m.dl_potential_loader(frag(g["z"], g["pos"], g["start"], g["end"]))
    check(e, f, g["E_ref64"], g["F_ref64"], ref32=(g["E_ref32"], g["F_ref32"]))
    ref_err = max(np.abs(g["F_ref32"] - g["F_ref64"]).max(), 2e-6)
    assert np.abs(f - g["F_ref64"]).max() <= 4 * ref_err + 1e-6 * np.abs(g["F_ref64"]).max()


def test_split3_is_a_different_arithmetic_and_switches_back(lib_built):
    """the option really reroutes the grouped products (bits differ from the fp32 chain, by round-off only), the same
    engine returns to the default arithmetic bit for bit, and repeated calls in the mode are bit-reproducible"""
    g = load_golden("h256_l9_default")
    m = model_for(g["hparams"], g["weight_seed"])
    fd = frag(g["z"], g["pos"], g["start"], g["end"])
    e0, f0 = m.dl_potential_loader(fd)
    m.engine.set_option("gemm_split3", 1)
    e1, f1 = m.dl_potential_loader(fd)
    e1b, f1b = m.dl_potential_loader(fd)
    m.engine.set_option("gemm_split3", 0)
    e2, f2 = m.dl_potential_loader(fd)
    assert np.array_equal(f0, f2) and np.array_equal(e0, e2)
    assert np.array_equal(f1, f1b) and np.array_equal(e1, e1b)
    assert not np.array_equal(f0, f1)
    np.testing.assert_allclose(f1, f0, rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", ["chig", "trpcage", "ww", "abd"])
def test_protein_batches_and_device_pipeline_in_split3_mode(lib_built, name):
    """the bench's own configurations (H=256, L=9, bench weights): the fragment batch through the seam and the
    device-resident step (gather + cap-H relaxation + ViSNet + combine) against the reference-source goldens"""
    import json

    from ai2bmd_amd.amber import load_tables
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan
    from ai2bmd_amd.hydrogen import build_hydrogen_plan
    from ai2bmd_amd.synthetic import default_hparams as dh, make_state_dict as msd
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    d = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
    g = {k: d[k] for k in d.files}
    hp = dh()
    assert json.loads(str(g["hparams"]))["embedding_dimension"] == 256
    m = split3(ViSNetModel(hp, msd(hp, seed=2024), device="cuda:0"))
    e, f = m.dl_potential_loader(frag(g["z"], g["pos_relaxed"], g["start"], g["end"]))
    check(e, f, g["E_ref64_relaxed"], g["F_ref64_relaxed"], ref32=(g["E_ref32_relaxed"], g["F_ref32_relaxed"]))
    p = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    prot = ProteinAtoms(p["names"], p["resnames"], p["resnums"], p["numbers"], p["positions"].astype(np.float64))
    plan = build_plan(prot)
    hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(GOLDEN, "amber_tables.npz")))
    x = torch.as_tensor(prot.positions, dtype=torch.float32, device="cuda:0")
    E, F = ShardedFragmentForces.for_engine(m.engine, plan, hydrogen=hplan).step(x)
    torch.cuda.synchronize()
    Fg, Eg = g["Fprot64_relaxed"], float(g["Eprot64_relaxed"])
    assert np.abs(F.cpu().numpy() - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max())
    assert np.abs(F.cpu().numpy() - Fg).mean() <= 1e-5 * max(1.0, np.abs(Fg).max())
    assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))


@pytest.mark.parametrize("seed", [0, 3, 5, 9, 12, 16, 19, 22])
def test_fuzz_draws_in_split3_mode(lib_built, seed):
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    hp, sizes = draw(seed)
    sd = make_state_dict(hp, seed=500 + seed)
    z, pos, start, end = random_fragments(900 + seed, sizes, cutoff=hp["cutoff"])
    E64, F64, _ = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    E32, F32, _ = ViSNetOracle(hp, sd, torch.float32).energy_forces(z, pos, start, end)
    m = split3(ViSNetModel(hp, sd, device="cuda:0"))
    e, f = m.dl_potential_loader(frag(z, pos, start, end))
    tol_e = np.maximum(1e-5 * np.maximum(1.0, np.abs(E64)), 4 * np.abs(E32 - E64).max())
    tol_f = max(1e-4 * max(1.0, np.abs(F64).max()), 4 * np.abs(F32 - F64).max())
    assert (np.abs(e - E64) <= tol_e).all(), (hp, sizes, np.abs(e - E64).max())
    assert np.abs(f - F64).max() <= tol_f, (hp, sizes, np.abs(f - F64).max(), tol_f)


@pytest.mark.parametrize("fuse_panel", [1, 0])
def test_fragment_batch_in_split3_mode(lib_built, fuse_panel):
    """Batch regime (N >= 4096: one wave per node, 128 x 128 product tiles): in the mode the PLAIN products run the
    128 x 128 split tile (k_gemm3_128); the fused panel products (fuse_panel = 1) keep their fp32 MFMA prologue
    kernels.  Replica 0 of a tiled batch against the fp64 oracle, every replica bit-identical to it."""
    hp = default_hparams(embedding_dimension=256, num_layers=3)
    z1, p1, s1, e1 = random_fragments(6, [27, 12, 33])
    reps, n1 = 80, 72
    z = np.tile(z1, reps)
    pos = np.tile(p1, (reps, 1))
    start = np.concatenate([s1 + r * n1 for r in range(reps)])
    end = np.concatenate([e1 + r * n1 for r in range(reps)])
    assert len(z1) == n1 and len(z) >= 4096
    m = split3(model_for(hp, 4))
    m.engine.set_option("fuse_panel", fuse_panel)
    e, f = m.dl_potential_loader(frag(z, pos, start, end))
    E64, F64, _ = ViSNetOracle(hp, make_state_dict(hp, seed=4), torch.float64).energy_forces(z1, p1, s1, e1)
    check(e.reshape(reps, -1)[0].reshape(-1, 1), f.reshape(reps, n1, 3)[0], E64, F64)
    assert np.array_equal(f.reshape(reps, n1, 3), np.broadcast_to(f.reshape(reps, n1, 3)[0], (reps, n1, 3)))
    m.engine.set_option("gemm_split3", 0)
    e0, f0 = m.dl_potential_loader(frag(z, pos, start, end))
    assert not np.array_equal(f0, f)  # the mode really took the other kernel
    np.testing.assert_allclose(f, f0, rtol=0, atol=2e-5)


def test_reloading_weights_drops_the_split3_planes(lib_built):
    """the packed bf16 planes are cached per weight ADDRESS; vsn_load_weight + vsn_finalize frees and
    re-allocates the arena (very likely at the same addresses), so the cache must be dropped with it - a reload with
    the mode on must give the NEW weights' forces, in split mode and back in fp32."""
    hp = default_hparams(embedding_dimension=256, num_layers=2)
    z, pos, start, end = random_fragments(31, [22, 12, 30, 19])
    m = split3(model_for(hp, 7))
    fd = frag(z, pos, start, end)
    e_a, f_a = m.dl_potential_loader(fd)                       # planes of the seed-7 weights are cached now
    sd_b = make_state_dict(hp, seed=8)
    m.engine.load_state_dict(sd_b)
    e_b, f_b = m.dl_potential_loader(fd)
    E64, F64, _ = ViSNetOracle(hp, sd_b, torch.float64).energy_forces(z, pos, start, end)
    check(e_b, f_b, E64, F64)
    assert np.abs(f_b - f_a).max() > 1e-3                      # (really different weights)
    fresh = split3(model_for(hp, 8))
    e_c, f_c = fresh.dl_potential_loader(fd)
    assert np.array_equal(f_b, f_c) and np.array_equal(e_b, e_c)
    m.engine.set_option("gemm_split3", 0)
    e_d, f_d = m.dl_potential_loader(fd)
    check(e_d, f_d, E64, F64)
