"""Cap-hydrogen relaxation: AMBER tables, term plan, and the oracle against golden vectors produced by the
reference's own HydrogenOptimizer (oracle/make_hydrogen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from ai2bmd_amd.amber import TOPOLOGY_OF, load_tables
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, fragment_positions
from ai2bmd_amd.hydrogen import build_hydrogen_plan
from oracle.hydrogen_oracle import HydrogenOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PROTEINS = ["chig", "trpcage", "ww", "abd"]


def load_case(name):
    z = np.load(os.path.join(GOLD, f"protein_{name}.npz"))
    p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                     positions=z["positions"])
    plan = build_plan(p)
    hplan = build_hydrogen_plan(p, plan, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    return p, plan, hplan, np.load(os.path.join(GOLD, f"hopt_{name}.npz"))


def test_amber_tables_fixture():
    tables = load_tables(os.path.join(GOLD, "amber_tables.npz"))
    assert set(tables) == set(TOPOLOGY_OF.values())
    aa = tables["AA"]
    assert aa["natom"] == 22 and list(aa["atom_names"][:6]) == ["H1", "CH3", "H2", "H3", "C", "O"]
    assert list(aa["atom_names"][-6:]) == ["N", "H", "CH3", "HH31", "HH32", "HH33"]
    for t in tables.values():
        n = t["natom"]
        assert len(t["charge"]) == n and len(t["atom_type_idx"]) == n and t["number_excluded_atoms"].sum() == len(
            t["excluded_atoms_list"])
        assert t["bonds_inc_hydrogen"][:, :2].max() < n and t["angles_inc_hydrogen"][:, :3].max() < n
        assert abs(t["charge"].sum() / 18.2223 - round(t["charge"].sum() / 18.2223)) < 1e-3  # integer net charge


@pytest.mark.parametrize("name", PROTEINS)
def test_plan_structure(name):
    p, plan, hp, _ = load_case(name)
    caps = np.flatnonzero((plan.src < 0) & np.repeat(plan.is_dipeptide, plan.end - plan.start))
    assert (hp.cap_rows == caps).all()
    # every cap hydrogen has exactly one bond, and is an END atom of every term it appears in
    for c, row in enumerate(hp.cap_rows):
        sl = slice(hp.occ_ptr[c], hp.occ_ptr[c + 1])
        ty, term, end = hp.occ_type[sl], hp.occ_term[sl], hp.occ_end[sl]
        assert (ty == 0).sum() == 1
        for t_, k_, e_ in zip(ty, term, end):
            tab = (hp.bond, hp.angle, hp.dihedral, hp.pair)[t_]
            first, last = tab["i"][k_], tab["jkl"[t_ if t_ < 3 else 0]][k_]
            assert (last if e_ else first) == row
    # energy shares add up to one per term
    for t_, tab in enumerate((hp.bond, hp.angle, hp.dihedral, hp.pair)):
        share = np.zeros(len(tab["i"]))
        np.add.at(share, hp.occ_term[hp.occ_type == t_], hp.occ_w[hp.occ_type == t_])
        assert np.allclose(share, 1.0)
    # ACE-NME rows alias dipeptide rows that carry the same atom: identical before relaxation
    pos = fragment_positions(plan, p.positions)
    ace = hp.alias >= 0
    assert ace.sum() == 12 * (len(plan.start) // 2) and not ace[hp.cap_rows].any()
    assert np.array_equal(pos[ace], pos[hp.alias[ace]])
    assert (plan.z[ace] == plan.z[hp.alias[ace]]).all()
    assert not np.repeat(plan.is_dipeptide, plan.end - plan.start)[ace].any()


@pytest.mark.parametrize("name", PROTEINS)
def test_oracle_matches_reference_optimizer(name):
    """golden = /root/reference HydrogenOptimizer(max_iter=10).optimize_hydrogen on the same geometry."""
    p, plan, hp, gold = load_case(name)
    orc = HydrogenOracle(hp)
    for tag in ("x0", "x1"):
        pos = fragment_positions(plan, gold[f"{tag}_prot"]).astype(np.float32)
        e0 = orc.energy(torch.as_tensor(pos)).numpy()
        assert np.allclose(e0, gold[f"{tag}_e0"], rtol=2e-4, atol=2e-3)  # bond, angle, dihedral, vdw, elec
        out = orc.relax(pos)
        assert np.abs(out[hp.cap_rows] - gold[f"{tag}_caps"]).max() < 1e-5
        not_cap = np.ones(len(pos), bool)
        not_cap[hp.cap_rows] = False
        assert np.array_equal(out[not_cap], pos[not_cap])
        e1 = orc.energy(torch.as_tensor(out)).numpy()
        assert np.allclose(e1, gold[f"{tag}_e1"], rtol=2e-4, atol=2e-3) and e1.sum() < e0.sum()


def test_non_integral_bond_force_constant_survives_the_plan():
    """The bond dict's force constant must stay a float (a custom prmtop may hold non-integral constants)."""
    import copy

    z = np.load(os.path.join(GOLD, "protein_chig.npz"))
    p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                     positions=z["positions"])
    plan = build_plan(p)
    tables = copy.deepcopy(load_tables(os.path.join(GOLD, "amber_tables.npz")))
    for t in tables.values():
        t["bond_force_constant"] = np.asarray(t["bond_force_constant"], np.float64) + 0.37
    hp = build_hydrogen_plan(p, plan, tables)
    assert hp.bond["kf"].dtype == np.float32 and hp.bond["i"].dtype == np.int32
    frac = hp.bond["kf"] - np.floor(hp.bond["kf"])
    assert np.allclose(frac, 0.37, atol=1e-3)
    ref = build_hydrogen_plan(p, plan, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    assert np.allclose(hp.bond["kf"], ref.bond["kf"] + 0.37, atol=1e-3)


def test_oracle_multi_iteration_descends():
    p, plan, hp, _ = load_case("chig")
    pos = fragment_positions(plan, p.positions).astype(np.float32)
    rng = np.random.default_rng(5)
    pos[hp.cap_rows[:3]] += 0.3 * rng.standard_normal((3, 3)).astype(np.float32)
    out, trace = HydrogenOracle(hp).relax(pos, return_trace=True)
    assert len(trace) > 3 and all(b < a for a, b in zip(trace, trace[1:]))


@pytest.mark.parametrize("name", PROTEINS)
def test_mm_parameters_from_amber_tables(name):
    from ai2bmd_amd.amber import protein_mm_parameters

    p, plan, hp, _ = load_case(name)
    q, sig, eps = protein_mm_parameters(p, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    assert abs(q.sum() - round(float(q.sum()))) < 2e-3          # residues carry integer charges
    is_c = (p.numbers == 6) & (p.names == "CA")
    assert np.allclose(sig[is_c], 0.339967, atol=1e-5)          # AMBER CT: Rmin/2 = 1.908 A
    assert np.allclose(eps[is_c], 0.1094 * 4.184, rtol=1e-3)    #           eps = 0.1094 kcal/mol
    assert (eps >= 0).all() and (sig > 0).all()


def _pipeline_case(name):
    z = np.load(os.path.join(GOLD, f"protein_{name}.npz"))
    p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                     positions=z["positions"])
    plan = build_plan(p)
    hp = build_hydrogen_plan(p, plan, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    return p, plan, hp, np.load(os.path.join(GOLD, f"pipeline_{name}.npz"))


@pytest.mark.parametrize("name", ["chig", "chigcyx"])
def test_whole_bonded_path_matches_reference_pipeline(name):
    """golden = the reference's own fragmenter -> hydrogen optimiser -> ViSNet source -> combiner on a displaced
    Chignolin frame (oracle/make_pipeline_golden.py).  Here: our plan + the oracles, host composition."""
    from ai2bmd_amd.fragmentation import combine_host
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict
    from oracle.visnet_oracle import ViSNetOracle

    p, plan, hp, gold = _pipeline_case(name)
    pos = fragment_positions(plan, gold["prot_pos"]).astype(np.float32)
    relaxed = HydrogenOracle(hp).relax(pos)
    ace = hp.alias >= 0
    relaxed[ace] = relaxed[hp.alias[ace]]
    # same atoms at the same (relaxed) places as the reference's fragment batch, fragment by fragment
    for b in range(len(plan.start)):
        a0, a1 = int(plan.start[b]), int(plan.end[b])
        for r in range(a0, a1):
            d = np.abs(relaxed[a0:a1] - gold["frag_pos"][r]).max(axis=1)
            d[plan.z[a0:a1] != gold["frag_z"][r]] = np.inf
            assert d.min() < 3e-5, (b, r, d.min())
    hparams = default_hparams(embedding_dimension=128, num_layers=3)
    sd = make_state_dict(hparams, seed=int(gold["weight_seed"]))
    E, F, _ = ViSNetOracle(hparams, sd, torch.float64).energy_forces(plan.z, relaxed.astype(np.float64), plan.start,
                                                                    plan.end)
    assert len(E) == int(((plan.end - plan.start) > 0).sum())  # energies of non-empty fragments (CYX: one empty slot)
    E_tot, F_prot = combine_host(plan, E.reshape(-1, 1), F)
    assert abs(E_tot - float(gold["E64"])) < 1e-6 * abs(float(gold["E64"]))
    assert np.abs(F_prot - gold["F64"]).max() < 1e-5 * np.abs(gold["F64"]).max()  # measured 1e-6 (fp32 relaxed positions)
