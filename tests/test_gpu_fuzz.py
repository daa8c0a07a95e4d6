"""GPU: seeded random hyper-parameter / batch combinations against the fp64 oracle - every legal value of the
choices a checkpoint can carry (visnet.py:14-30), crossed at random: hidden width 64..512, layers, lmax, vecnorm,
radial basis and its size, both activations, heads, cutoff, neighbour cap, ragged batches with empty fragments."""
import numpy as np
import pytest
import torch

from oracle.inputs import random_fragments
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import default_hparams, make_state_dict

pytestmark = pytest.mark.gpu

ACTS = ["silu", "swish", "ssp", "tanh", "sigmoid"]


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    H = 64 * int(rng.integers(1, 9))
    # seeds 0..15 draw 1-3 layers (the draws of rounds 1-2, unchanged); seeds >= 16 draw DEEP networks (4-9 layers:
    # the fused vertical passes between layers, the default L = 9 among them)
    L = int(rng.integers(1, 4))
    if seed >= 16:
        L = int(rng.integers(4, 10))
    hp = default_hparams(
        embedding_dimension=H, num_layers=L, lmax=int(rng.integers(1, 3)),
        vecnorm_type=str(rng.choice(["none", "rms", "max_min"])), rbf_type=str(rng.choice(["expnorm", "gauss"])),
        num_rbf=int(rng.choice([8, 20, 32, 50])), activation=str(rng.choice(ACTS)), attn_activation=str(rng.choice(ACTS)),
        num_heads=int(rng.choice([1, 2, 4, 8, 16])), cutoff=float(rng.choice([4.0, 5.0, 6.0])),
        max_num_neighbors=int(rng.choice([12, 32, 64])))
    nfrag = int(rng.integers(1, 6))
    sizes = [int(rng.choice([0, 1, 12, 19, 27, 36, 44, 70])) for _ in range(nfrag)]
    if sum(sizes) == 0:
        sizes[0] = 22
    return hp, sizes


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_matches_oracle(lib_built, seed):
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    hp, sizes = draw(seed)
    sd = make_state_dict(hp, seed=500 + seed)
    z, pos, start, end = random_fragments(900 + seed, sizes, cutoff=hp["cutoff"])
    o32 = ViSNetOracle(hp, sd, torch.float32)
    E64, F64, c = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    E32, F32, _ = o32.energy_forces(z, pos, start, end)
    m = ViSNetModel(hp, sd, device="cuda:0")
    e, f = m.dl_potential_loader(FragmentData(z, pos, start, end, make_batch_index(start, end)))
    assert m.engine.last_num_edges() == len(c["graph"]["src"]), (hp, sizes)
    assert np.isfinite(e).all() and np.isfinite(f).all()
    # SURVEY.md 8c tolerance, or 4x the fp32 evaluation's own distance from the fp64 truth where that is larger
    # (saturating activations / large random weights are ill-conditioned draws)
    tol_e = np.maximum(1e-5 * np.maximum(1.0, np.abs(E64)), 4 * np.abs(E32 - E64).max())
    tol_f = max(1e-4 * max(1.0, np.abs(F64).max()), 4 * np.abs(F32 - F64).max())
    assert (np.abs(e - E64) <= tol_e).all(), (hp, sizes, np.abs(e - E64).max())
    assert np.abs(f - F64).max() <= tol_f, (hp, sizes, np.abs(f - F64).max(), tol_f)
