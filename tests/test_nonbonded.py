"""MM non-bonded term: oracle self-checks on CPU, HIP kernel vs oracle on the GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN


def load_protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


def synthetic_params(n, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-0.6, 0.6, n)
    q -= q.mean()
    return q.astype(np.float32), rng.uniform(0.1, 0.35, n).astype(np.float32), rng.uniform(0.05, 0.7, n).astype(
        np.float32)


def dipeptide_atom_lists(plan):
    out = []
    for b in range(len(plan.start)):
        if plan.is_dipeptide[b]:
            a = plan.src[plan.start[b]:plan.end[b]]
            out.append(a[a >= 0].tolist())
    return out


@pytest.mark.parametrize("name", ["chig", "chigcyx"])
def test_group_exclusion_equals_reference_pair_list(name):
    """"chigcyx": the two halves of a CYX pair are ONE dipeptide for the exclusion (distancefrag.py:195-197,355-363)"""
    from ai2bmd_amd.fragmentation import build_plan
    from ai2bmd_amd.nonbonded import dipeptide_groups
    from oracle.nonbonded_oracle import exclude_pairs_from_dipeptides, pair_list

    prot = load_protein(name)
    plan = build_plan(prot)
    ex = exclude_pairs_from_dipeptides(dipeptide_atom_lists(plan))
    src, dst = pair_list(plan.n_prot, ex)
    g = dipeptide_groups(plan)
    share = np.zeros((plan.n_prot, plan.n_prot), bool)
    for u in range(4):
        for v in range(4):
            share |= (g[:, None, u] >= 0) & (g[:, None, u] == g[None, :, v])
    keep = ~share & ~np.eye(plan.n_prot, dtype=bool)
    ref = np.zeros_like(keep)
    ref[src, dst] = True
    assert (keep == ref).all() and keep.sum() > 0


def test_oracle_forces_are_minus_gradient():
    from ai2bmd_amd.fragmentation import build_plan
    from oracle.nonbonded_oracle import K_COULOMB, exclude_pairs_from_dipeptides, mm_nonbonded, pair_list

    assert abs(K_COULOMB - 1389.35) < 0.05  # kJ/mol * Angstrom / e^2 (x10: the reference's 10e6 factor with nm->A)
    prot = load_protein("chig")
    plan = build_plan(prot)
    src, dst = pair_list(plan.n_prot, exclude_pairs_from_dipeptides(dipeptide_atom_lists(plan)))
    q, s, e = synthetic_params(plan.n_prot)
    pos = prot.positions.copy()
    E, F = mm_nonbonded(pos, q, s, e, src, dst)
    rng = np.random.default_rng(1)
    for _ in range(5):
        i, k = rng.integers(plan.n_prot), rng.integers(3)
        h = 1e-5
        p1, p2 = pos.copy(), pos.copy()
        p1[i, k] += h
        p2[i, k] -= h
        g = (mm_nonbonded(p1, q, s, e, src, dst)[0] - mm_nonbonded(p2, q, s, e, src, dst)[0]) / (2 * h)
        assert abs(-g - F[i, k]) <= 1e-5 * max(1.0, abs(F[i, k]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["chig", "ww"])
def test_hip_kernel_matches_oracle(lib_built, name):
    import torch

    from ai2bmd_amd.fragmentation import build_plan
    from ai2bmd_amd.nonbonded import MMNonBondedCalculator
    from oracle.nonbonded_oracle import exclude_pairs_from_dipeptides, mm_nonbonded, pair_list

    prot = load_protein(name)
    plan = build_plan(prot)
    q, s, e = synthetic_params(plan.n_prot, seed=3)
    src, dst = pair_list(plan.n_prot, exclude_pairs_from_dipeptides(dipeptide_atom_lists(plan)))
    E64, F64 = mm_nonbonded(prot.positions.astype(np.float32).astype(np.float64), q.astype(np.float64),
                            s.astype(np.float64), e.astype(np.float64), src, dst)
    calc = MMNonBondedCalculator("cuda:0")
    calc.set_parameters(SimpleNamespace(charges=q, sigmas=s, epsilons=e), plan)
    E, F = calc(SimpleNamespace(positions=prot.positions))
    assert abs(E - E64) <= 2e-5 * max(1.0, abs(E64)), (E, E64)
    assert np.abs(F - F64).max() <= 1e-4 * max(1.0, np.abs(F64).max())
    # accumulate mode adds onto an existing force array
    base = torch.ones(plan.n_prot, 3, device="cuda:0")
    pos = torch.as_tensor(prot.positions.astype(np.float32)).to("cuda:0")
    _, F2 = calc.forces_device(pos, f_out=base, accumulate=True)
    np.testing.assert_allclose(F2.cpu().numpy(), F + 1.0, rtol=0, atol=1e-5 * max(1.0, np.abs(F).max()))


def _amber_params(prot):
    from ai2bmd_amd.amber import load_tables, protein_mm_parameters

    return protein_mm_parameters(prot, load_tables(os.path.join(GOLDEN, "amber_tables.npz")))


@pytest.mark.parametrize("name", ["chig", "trpcage"])
def test_oracle_matches_reference_calculator(name):
    """golden = /root/reference/src/Calculators/nonbonded.py MMNonBondedCalculator run on the same protein
    (oracle/make_nonbonded_golden.py), AMBER-table parameters."""
    from ai2bmd_amd.fragmentation import build_plan
    from oracle.nonbonded_oracle import exclude_pairs_from_dipeptides, mm_nonbonded, pair_list

    prot = load_protein(name)
    plan = build_plan(prot)
    gold = np.load(os.path.join(GOLDEN, f"mm_{name}.npz"))
    q, s, e = _amber_params(prot)
    src, dst = pair_list(plan.n_prot, exclude_pairs_from_dipeptides(dipeptide_atom_lists(plan)))
    assert len(src) == int(gold["n_pairs"])
    E, F = mm_nonbonded(prot.positions, q.astype(np.float64), s.astype(np.float64), e.astype(np.float64), src, dst)
    assert abs(E - float(gold["energy"])) <= 2e-5 * max(1.0, abs(E))            # the reference computes in fp32
    assert np.abs(F - gold["forces"]).max() <= 2e-5 * np.abs(F).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["chig", "trpcage"])
def test_hip_kernel_matches_reference_calculator(lib_built, name):
    from ai2bmd_amd.fragmentation import build_plan
    from ai2bmd_amd.nonbonded import MMNonBondedCalculator

    prot = load_protein(name)
    plan = build_plan(prot)
    gold = np.load(os.path.join(GOLDEN, f"mm_{name}.npz"))
    q, s, e = _amber_params(prot)
    calc = MMNonBondedCalculator("cuda:0")
    calc.set_parameters(SimpleNamespace(charges=q, sigmas=s, epsilons=e), plan)
    E, F = calc(SimpleNamespace(positions=prot.positions))
    assert abs(E - float(gold["energy"])) <= 5e-5 * max(1.0, abs(float(gold["energy"])))
    assert np.abs(F - gold["forces"]).max() <= 1e-4 * np.abs(gold["forces"]).max()
