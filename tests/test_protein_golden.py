"""CPU: the oracle restatement (oracle/visnet_oracle.py) reproduces the reference-source goldens of the real
protein fragment batches at the benchmarked size (H=256, L=9; oracle/make_protein_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import make_state_dict


@pytest.mark.parametrize("name,tag", [("chig", "relaxed"), ("trpcage", "placed")])
def test_oracle_matches_reference_on_protein_batches(name, tag):
    d = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
    hp = json.loads(str(d["hparams"]))
    sd = make_state_dict(hp, seed=int(d["weight_seed"]))
    E, F, c = ViSNetOracle(hp, sd, torch.float64).energy_forces(d["z"], d[f"pos_{tag}"], d["start"], d["end"])
    assert np.diff(c["graph"]["rowptr"]).max() == min(int(d[f"max_degree_{tag}"]), hp["max_num_neighbors"])
    np.testing.assert_allclose(E, d[f"E_ref64_{tag}"], rtol=0, atol=1e-9 * max(1.0, np.abs(d[f"E_ref64_{tag}"]).max()))
    np.testing.assert_allclose(F, d[f"F_ref64_{tag}"], rtol=0, atol=1e-9 * max(1.0, np.abs(d[f"F_ref64_{tag}"]).max()))


@pytest.mark.parametrize("name", ["chig", "trpcage", "ww", "abd"])
def test_protein_golden_is_consistent_with_the_plan(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, combine_host, fragment_positions

    d = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
    z = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    p = ProteinAtoms(z["names"], z["resnames"], z["resnums"], z["numbers"], z["positions"].astype(np.float64))
    plan = build_plan(p)
    assert (plan.z == d["z"]).all() and (plan.start == d["start"]).all() and (plan.end == d["end"]).all()
    assert np.array_equal(fragment_positions(plan, p.positions).astype(np.float32), d["pos_placed"])
    for tag in ("relaxed", "placed"):
        E, F = combine_host(plan, d[f"E_ref64_{tag}"], d[f"F_ref64_{tag}"])
        assert abs(E - float(d[f"Eprot64_{tag}"])) < 1e-9 and np.abs(F - d[f"Fprot64_{tag}"]).max() < 1e-12
        # E depends on position differences only: the forces of every fragment sum to zero (the recombined protein
        # forces do not - the cap hydrogens' rows are dropped by select_index, combiner.py:38)
        for a, b in zip(d["start"], d["end"]):
            assert np.abs(d[f"F_ref64_{tag}"][a:b].sum(0)).max() < 1e-9


def test_builder_chain_golden_equals_the_all_reference_chain():
    """tests/golden/refchain_chig.npz (oracle/make_refchain_golden.py) is produced by the reference's OWN fragmenter,
    model, FragmentData split and DipeptideBondedCombiner - no builder arithmetic anywhere.  The protein goldens every
    device-pipeline test uses (`Fprot64_placed` etc., oracle/make_protein_golden.py) recombine the reference model's
    fragment forces with our combine_host on our plan: the two chains agree to fp64 round-off, which pins the latter."""
    a = np.load(os.path.join(GOLDEN, "refchain_chig.npz"))
    b = np.load(os.path.join(GOLDEN, "visnet_prot_chig.npz"))
    assert int(a["weight_seed"]) == int(b["weight_seed"])
    assert abs(float(a["Eprot64"]) - float(b["Eprot64_placed"])) < 1e-10
    assert np.abs(a["Fprot64"] - b["Fprot64_placed"]).max() < 1e-12
    # the same fragment batch row for row (our plan emits the reference's AMBER row order)
    assert np.array_equal(a["start"], b["start"]) and np.array_equal(a["end"], b["end"])
    assert np.array_equal(a["z"], b["z"]) and np.abs(a["pos"] - b["pos_placed"]).max() < 2e-5
    # the fp32 run of the same all-reference chain: the error floor the HIP path is compared with
    assert np.abs(a["Fprot32"] - a["Fprot64"]).max() < 1e-5


def _c1_fragment(tag="relaxed"):
    """BASELINE configs[0] / SURVEY 8(d) C1: the single alanine-dipeptide fragment ACE-ALA-NME (22 atoms), geometry =
    the ALA-3 window of examples/trpcage.pdb with its cap hydrogens, = fragment 2 of the Trp-cage fragment batch; its
    reference-source energy and forces are rows of tests/golden/visnet_prot_trpcage.npz (fragments are independent
    model inputs: visnet.py:135-166 sums per `batch` id)."""
    d = np.load(os.path.join(GOLDEN, "visnet_prot_trpcage.npz"))
    a, b = int(d["start"][2]), int(d["end"][2])
    z = d["z"][a:b]
    assert b - a == 22 and sorted(z.tolist()) == sorted([6, 6, 8, 1, 1, 1] + [7, 1, 6, 1, 6, 1, 1, 1, 6, 8] + [7, 1, 6, 1, 1, 1])
    hp = json.loads(str(d["hparams"]))
    return (hp, int(d["weight_seed"]), z, d[f"pos_{tag}"][a:b], np.array([0]), np.array([22]),
            d[f"E_ref64_{tag}"][2], d[f"F_ref64_{tag}"][a:b], d[f"E_ref32_{tag}"][2], d[f"F_ref32_{tag}"][a:b])


def test_c1_single_alanine_dipeptide_on_cpu():
    """configs[0] (plumbing, no GPU): ONE ACE-ALA-NME fragment, energy + forces by the REFERENCE's own model on CPU
    (source tree here, oracle/_ref elsewhere) and by the oracle restatement, both as a batch of one, against the
    fragment's rows of the Trp-cage reference-source golden."""
    from oracle.ref_import import import_reference_create_model, reference_model_source

    hp, seed, z, pos, start, end, E64, F64, E32, F32 = _c1_fragment()
    sd = make_state_dict(hp, seed=seed)
    E, F, _ = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    np.testing.assert_allclose(E.reshape(-1), np.asarray(E64).reshape(-1), rtol=0, atol=1e-9 * max(1.0, abs(float(E64))))
    np.testing.assert_allclose(F, F64, rtol=0, atol=1e-9)
    if reference_model_source() is None:
        pytest.skip("reference model not present")
    model = import_reference_create_model()(hp)
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    model = model.float().eval()
    Er, Fr = model(dict(z=torch.as_tensor(z), pos=torch.as_tensor(pos), batch=torch.zeros(22, dtype=torch.int64)))
    Er, Fr = Er.detach().numpy().reshape(-1), Fr.detach().numpy()
    # fp32 reference alone vs in the 39-fragment batch: same arithmetic per fragment up to reduction order
    assert abs(float(Er[0]) - float(E32)) <= 2e-5 * max(1.0, abs(float(E32))) and np.abs(Fr - F32).max() <= 5e-6
    assert np.abs(Fr - F64).max() <= 1e-4 * max(1.0, np.abs(F64).max())
