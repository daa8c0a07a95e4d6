"""HIP cap-hydrogen optimiser (csrc/hydrogen.hip through the C ABI) against the reference-generated golden vectors
and against the oracle on displaced geometries that need several L-BFGS iterations."""
import numpy as np
import pytest
import torch

from test_hydrogen import PROTEINS, load_case

from ai2bmd_amd.fragmentation import fragment_positions
from ai2bmd_amd.hydrogen import HydrogenRelaxer
from oracle.hydrogen_oracle import HydrogenOracle

pytestmark = pytest.mark.gpu
TOL = 2e-4  # Angstrom; fp32 optimiser whose reductions are summed in a different order than torch's


def run_hip(hp, pos, max_iter=10):
    dev = torch.device("cuda:0")
    rel = HydrogenRelaxer(hp, len(pos), 0, max_iter=max_iter)
    x = torch.as_tensor(pos, device=dev).contiguous()
    rel.run(x)
    st = rel.stats()
    return x.cpu().numpy(), st


@pytest.mark.parametrize("name", PROTEINS)
def test_matches_reference_golden(name):
    p, plan, hp, gold = load_case(name)
    for tag in ("x0", "x1"):
        pos = fragment_positions(plan, gold[f"{tag}_prot"]).astype(np.float32)
        out, st = run_hip(hp, pos)
        assert np.abs(out[hp.cap_rows] - gold[f"{tag}_caps"]).max() < TOL
        assert abs(st["loss_first"] - gold[f"{tag}_e0"].sum()) < 5e-3
        assert abs(st["loss_last"] - gold[f"{tag}_e1"].sum()) < 5e-3
        # untouched rows: everything that is neither a cap nor an ACE-NME copy of a cap
        ace = hp.alias >= 0
        assert np.array_equal(out[ace], out[hp.alias[ace]])
        fixed = np.ones(len(pos), bool)
        fixed[hp.cap_rows] = False
        fixed[np.flatnonzero(ace)[np.isin(hp.alias[ace], hp.cap_rows)]] = False
        assert np.array_equal(out[fixed], pos[fixed])


@pytest.mark.parametrize("name,seed,ncaps,amp", [("chig", 5, 3, 0.3), ("chig", 6, 2, 0.4), ("trpcage", 7, 4, 0.3),
                                                  ("ww", 8, 1, 0.5), ("abd", 9, 6, 0.3)])
def test_multi_iteration_matches_oracle(name, seed, ncaps, amp):
    p, plan, hp, _ = load_case(name)
    pos = fragment_positions(plan, p.positions).astype(np.float32)
    rng = np.random.default_rng(seed)
    pos[hp.cap_rows[:ncaps]] += amp * rng.standard_normal((ncaps, 3)).astype(np.float32)
    ref, trace = HydrogenOracle(hp).relax(pos, return_trace=True)
    out, st = run_hip(hp, pos)
    assert st["evaluations"] == len(trace), (st, trace)
    assert abs(st["loss_first"] - trace[0]) < 5e-3 * max(1, abs(trace[0]))
    assert abs(st["loss_last"] - trace[-1]) < 5e-3 * max(1, abs(trace[-1]))
    assert np.abs(out[hp.cap_rows] - ref[hp.cap_rows]).max() < 2e-3


def test_gradient_at_entry_matches_autograd():
    """one iteration with lr -> the first step is -t g: recover g from the displacement."""
    p, plan, hp, _ = load_case("trpcage")
    pos = fragment_positions(plan, p.positions).astype(np.float32)
    _, g = HydrogenOracle(hp, torch.float64).energy_grad(pos)
    out, st = run_hip(hp, pos, max_iter=1)
    t = min(1.0, 1.0 / np.abs(g).sum()) * 0.1
    g_hip = -(out[hp.cap_rows].astype(np.float64) - pos[hp.cap_rows]) / t
    assert st["iterations"] == 1 and st["evaluations"] == 1
    assert np.abs(g_hip - g).max() < 2e-2 * np.abs(g).max()  # limited by fp32 rounding of x + t d


def test_bit_reproducible():
    p, plan, hp, _ = load_case("ww")
    pos = fragment_positions(plan, p.positions).astype(np.float32)
    rng = np.random.default_rng(1)
    pos[hp.cap_rows[:2]] += 0.3 * rng.standard_normal((2, 3)).astype(np.float32)
    a, _ = run_hip(hp, pos)
    b, _ = run_hip(hp, pos)
    assert np.array_equal(a, b)


def test_pipeline_with_relaxation_matches_host_composition(lib_built):
    """device pipeline with the HIP relaxation == fragments relaxed by the oracle, then the reference-shaped seam."""
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import combine_host
    from ai2bmd_amd.visnet_calculator import ViSNetModel
    from oracle.weights import default_hparams, make_state_dict

    p, plan, hp, gold = load_case("chig")
    hparams = default_hparams(embedding_dimension=128, num_layers=3)
    model = ViSNetModel(hparams, make_state_dict(hparams, seed=21), device="cuda:0")
    ff = ShardedFragmentForces.for_engine(model.engine, plan, hydrogen=hp)
    prot = gold["x1_prot"]
    E, F = ff.step(torch.as_tensor(prot, dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()
    pos = fragment_positions(plan, prot).astype(np.float32)
    relaxed = HydrogenOracle(hp).relax(pos)
    ace = hp.alias >= 0
    relaxed[ace] = relaxed[hp.alias[ace]]  # distancefrag.py:82 positions[prot.fragments_index]
    assert np.abs(ff.frag_pos.cpu().numpy() - relaxed).max() < TOL
    fd = FragmentData(plan.z, relaxed, plan.start, plan.end, make_batch_index(plan.start, plan.end))
    e_all, f_all = model.dl_potential_loader(fd)
    E_h, F_h = combine_host(plan, e_all, f_all)
    assert abs(float(E) - E_h) <= 2e-4 * max(1.0, abs(E_h))
    np.testing.assert_allclose(F.cpu().numpy(), F_h, rtol=0, atol=5e-4)


@pytest.mark.parametrize("name", ["chig", "chigcyx"])
def test_device_pipeline_matches_reference_pipeline(lib_built, name):
    """the device-resident step (cap-H placement + HIP relaxation + HIP ViSNet + HIP combine) against the golden
    produced by the reference's own fragmenter, hydrogen optimiser, ViSNet source and combiner."""
    import os

    from test_hydrogen import _pipeline_case

    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    p, plan, hp, gold = _pipeline_case(name)  # "chigcyx": a CYX-CYX bridge (44-atom merged fragment + an empty slot)
    hparams = default_hparams(embedding_dimension=128, num_layers=3)
    model = ViSNetModel(hparams, make_state_dict(hparams, seed=int(gold["weight_seed"])), device="cuda:0")
    ff = ShardedFragmentForces.for_engine(model.engine, plan, hydrogen=hp)
    E, F = ff.step(torch.as_tensor(gold["prot_pos"], dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()
    fmax = float(np.abs(gold["F64"]).max())
    assert abs(float(E) - float(gold["E64"])) < 1e-5 * abs(float(gold["E64"]))
    # the relaxed hydrogens agree to 2e-4 A (fp32 optimiser); forces follow to 1e-3 of the largest force
    assert np.abs(F.cpu().numpy() - gold["F64"]).max() < 1e-3 * fmax
    assert np.abs(F.cpu().numpy() - gold["F64"]).mean() < 1e-4 * fmax
