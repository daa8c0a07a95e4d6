"""CPU, build container only: re-runs the reference's own ViSNet source (through
oracle/shims) against the oracle restatement on fresh seeds.  Skipped where
/root/reference does not exist (the GPU box)."""
import numpy as np
import pytest
import torch

from oracle.inputs import random_fragments
from oracle.ref_import import reference_available
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import default_hparams, make_state_dict

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("vn,lmax,mnb", [("none", 2, 32), ("rms", 2, 32), ("max_min", 1, 32), ("none", 2, 12)])
def test_live_reference(vn, lmax, mnb):
    from oracle.make_golden import run_reference

    hp = default_hparams(embedding_dimension=64, num_layers=2, vecnorm_type=vn, lmax=lmax, max_num_neighbors=mnb)
    sd = make_state_dict(hp, seed=5)
    z, pos, start, end = random_fragments(55, [19, 0, 12, 36])
    E_ref, F_ref = run_reference(hp, sd, z, pos, start, end, torch.float64)
    E, F, c = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    assert np.diff(c["graph"]["rowptr"]).max() <= mnb
    np.testing.assert_allclose(E, E_ref, atol=1e-10)
    np.testing.assert_allclose(F, F_ref, atol=1e-10)
