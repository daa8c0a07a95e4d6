"""CPU: pins the oracle restatement (oracle/visnet_oracle.py) to golden vectors
produced by the reference's own source (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import make_state_dict


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_fp64_matches_reference(name):
    g = load_golden(name)
    sd = make_state_dict(g["hparams"], seed=g["weight_seed"])
    o = ViSNetOracle(g["hparams"], sd, torch.float64)
    E, F, _ = o.energy_forces(g["z"], g["pos"], g["start"], g["end"])
    assert E.shape == g["E_ref64"].shape and F.shape == g["F_ref64"].shape
    np.testing.assert_allclose(E, g["E_ref64"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["E_ref64"]).max()))
    np.testing.assert_allclose(F, g["F_ref64"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["F_ref64"]).max()))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_fp32_close_to_reference_fp32(name):
    g = load_golden(name)
    sd = make_state_dict(g["hparams"], seed=g["weight_seed"])
    o = ViSNetOracle(g["hparams"], sd, torch.float32)
    E, F, _ = o.energy_forces(g["z"], g["pos"], g["start"], g["end"])
    # both are fp32 evaluations of the same graph: agree to fp32 round-off
    np.testing.assert_allclose(E, g["E_ref32"], rtol=0, atol=2e-5 * max(1.0, np.abs(g["E_ref64"]).max()))
    np.testing.assert_allclose(F, g["F_ref32"], rtol=0, atol=1e-4 * max(1.0, np.abs(g["F_ref64"]).max()))


@pytest.mark.parametrize("name", [n for n in GOLDEN_CASES if "rms" not in n and "maxmin" not in n])
def test_analytic_reverse_pass_matches_autograd(name):
    g = load_golden(name)
    sd = make_state_dict(g["hparams"], seed=g["weight_seed"])
    o = ViSNetOracle(g["hparams"], sd, torch.float64)
    E, F, _, _ = o.energy_forces_analytic(g["z"], g["pos"], g["start"], g["end"])
    np.testing.assert_allclose(E, g["E_ref64"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["E_ref64"]).max()))
    np.testing.assert_allclose(F, g["F_ref64"], rtol=0, atol=1e-9 * max(1.0, np.abs(g["F_ref64"]).max()))


def test_empty_fragment_dropped_from_energies():
    g = load_golden("h64_l2")  # holds one empty fragment
    assert (g["end"] - g["start"] == 0).sum() == 1
    assert g["E_ref64"].shape[0] == len(g["start"]) - 1
