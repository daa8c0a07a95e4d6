"""Synthetic ViSNet checkpoints: a seeded, torch-version-independent weight generator.

The reference ships no checkpoints (/root/reference/.MISSING_LARGE_BLOBS:1-2)
and reads the hyper-parameters from the checkpoint itself
(src/ViSNet/model/visnet.py:74-76), so parity is pinned on seeded random
weights.  This module generates a state_dict with exactly the keys/shapes the
reference's `create_model(...).state_dict()` has (visnet.py:14-70,
visnet_block.py:22-101, utils.py:279-341, output_modules.py:9-50,
priors.py:48-87) from a numpy Generator, so the same weights can be rebuilt on
the GPU box without shipping 39 MB fixtures.  Used by bench.py (there is no network for
the trained checkpoints) and, through oracle/weights.py, by the parity tests.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np


def default_hparams(**over):
    """Constructor defaults of the reference (visnet_block.py:22-38) in the key
    names `create_model` reads (visnet.py:15-30,33,41,57-69)."""
    hp = dict(
        lmax=2,
        vecnorm_type="none",
        trainable_vecnorm=False,
        num_heads=8,
        num_layers=9,
        embedding_dimension=256,
        num_rbf=32,
        rbf_type="expnorm",
        trainable_rbf=False,
        activation="silu",
        attn_activation="silu",
        max_z=100,
        cutoff=5.0,
        max_num_neighbors=32,
        model="ViSNetBlock",
        prior_model="Atomref",
        prior_args=dict(max_z=100),
        output_model="Scalar",
        reduce_op="add",
        derivative=True,
    )
    hp.update(over)
    if hp.get("prior_model") == "Atomref":
        hp["prior_args"] = dict(max_z=hp["max_z"])
    return hp


def _xavier(rng, out_f, in_f):
    a = math.sqrt(6.0 / (in_f + out_f))
    return rng.uniform(-a, a, size=(out_f, in_f)).astype(np.float32)


BIAS_SCALE = 0.03  # keeps a 9-layer random network well conditioned while exercising every bias path


def _bias(rng, n, trivial):
    if trivial:
        return np.zeros(n, np.float32)
    return rng.uniform(-BIAS_SCALE, BIAS_SCALE, size=n).astype(np.float32)


def rbf_params(hp):
    """ExpNormalSmearing._initial_params (utils.py:40-46) / GaussianSmearing
    (utils.py:75-78), evaluated in float32 like the reference."""
    rc = np.float32(hp["cutoff"])
    R = hp["num_rbf"]
    if hp["rbf_type"] == "expnorm":
        start = np.exp(-rc).astype(np.float32)
        means = np.linspace(start, np.float32(1.0), R, dtype=np.float32)
        beta = np.float32((2.0 / R * (1.0 - float(start))) ** -2)
        betas = np.full(R, beta, np.float32)
        return means, betas
    if hp["rbf_type"] == "gauss":
        offset = np.linspace(np.float32(0.0), rc, R, dtype=np.float32)
        coeff = np.float32(-0.5 / float(offset[1] - offset[0]) ** 2)
        return offset, coeff
    raise NotImplementedError(hp["rbf_type"])


def make_state_dict(hp, seed=0, trivial=False):
    """Seeded weights. trivial=True mimics reset_parameters (zero biases, unit
    norms, mean=0,std=1, atomref=0); trivial=False exercises every bias /
    scale / prior path."""
    rng = np.random.default_rng(seed)
    H = hp["embedding_dimension"]
    R = hp["num_rbf"]
    L = hp["num_layers"]
    Z = hp["max_z"]
    sd = OrderedDict()
    sd["mean"] = np.float32(0.0 if trivial else 0.37)
    sd["std"] = np.float32(1.0 if trivial else 1.9)
    rm = "representation_model."
    sd[rm + "embedding.weight"] = rng.standard_normal((Z, H)).astype(np.float32)
    if hp["rbf_type"] == "gauss":  # GaussianSmearing registers `coeff` (0-d) then `offset` (utils.py:66-73)
        offset, coeff = rbf_params(hp)
        sd[rm + "distance_expansion.coeff"] = coeff
        sd[rm + "distance_expansion.offset"] = offset
    else:
        means, betas = rbf_params(hp)
        sd[rm + "distance_expansion.means"] = means
        sd[rm + "distance_expansion.betas"] = betas
    sd[rm + "neighbor_embedding.embedding.weight"] = rng.standard_normal((Z, H)).astype(np.float32)
    sd[rm + "neighbor_embedding.distance_proj.weight"] = _xavier(rng, H, R)
    sd[rm + "neighbor_embedding.distance_proj.bias"] = _bias(rng, H, trivial)
    sd[rm + "neighbor_embedding.combine.weight"] = _xavier(rng, H, 2 * H)
    sd[rm + "neighbor_embedding.combine.bias"] = _bias(rng, H, trivial)
    sd[rm + "edge_embedding.edge_proj.weight"] = _xavier(rng, H, R)
    sd[rm + "edge_embedding.edge_proj.bias"] = _bias(rng, H, trivial)

    def ln_w(n):
        if trivial:
            return np.ones(n, np.float32)
        return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)

    for l in range(L):
        p = f"{rm}vis_mp_layers.{l}."
        last = l == L - 1
        sd[p + "layernorm.weight"] = ln_w(H)
        sd[p + "layernorm.bias"] = _bias(rng, H, trivial)
        sd[p + "vec_layernorm.weight"] = ln_w(H)
        sd[p + "vec_proj.weight"] = _xavier(rng, 3 * H, H)
        for nm in ("q", "k", "v", "dk", "dv"):
            sd[p + f"{nm}_proj.weight"] = _xavier(rng, H, H)
            sd[p + f"{nm}_proj.bias"] = _bias(rng, H, trivial)
        sd[p + "s_proj.weight"] = _xavier(rng, 2 * H, H)
        sd[p + "s_proj.bias"] = _bias(rng, 2 * H, trivial)
        if not last:
            sd[p + "f_proj.weight"] = _xavier(rng, H, H)
            sd[p + "f_proj.bias"] = _bias(rng, H, trivial)
            sd[p + "w_src_proj.weight"] = _xavier(rng, H, H)
            sd[p + "w_trg_proj.weight"] = _xavier(rng, H, H)
        sd[p + "o_proj.weight"] = _xavier(rng, 3 * H, H)
        sd[p + "o_proj.bias"] = _bias(rng, 3 * H, trivial)
    sd[rm + "out_norm.weight"] = ln_w(H)
    sd[rm + "out_norm.bias"] = _bias(rng, H, trivial)
    sd[rm + "vec_out_norm.weight"] = ln_w(H)
    on = "output_model.output_network."
    h2 = H // 2
    sd[on + "0.vec1_proj.weight"] = _xavier(rng, H, H)
    sd[on + "0.vec2_proj.weight"] = _xavier(rng, h2, H)
    sd[on + "0.update_net.0.weight"] = _xavier(rng, H, 2 * H)
    sd[on + "0.update_net.0.bias"] = _bias(rng, H, trivial)
    sd[on + "0.update_net.2.weight"] = _xavier(rng, 2 * h2, H)
    sd[on + "0.update_net.2.bias"] = _bias(rng, 2 * h2, trivial)
    sd[on + "1.vec1_proj.weight"] = _xavier(rng, h2, h2)
    sd[on + "1.vec2_proj.weight"] = _xavier(rng, 1, h2)
    sd[on + "1.update_net.0.weight"] = _xavier(rng, h2, 2 * h2)
    sd[on + "1.update_net.0.bias"] = _bias(rng, h2, trivial)
    sd[on + "1.update_net.2.weight"] = _xavier(rng, 2, h2)
    sd[on + "1.update_net.2.bias"] = _bias(rng, 2, trivial)
    if hp.get("prior_model") == "Atomref":
        ar = np.zeros((Z, 1), np.float32) if trivial else rng.standard_normal((Z, 1)).astype(np.float32)
        sd["prior_model.initial_atomref"] = ar.copy()
        sd["prior_model.atomref.weight"] = ar
    return sd


def write_lightning_ckpt(path, hp, sd):
    """Writes a Lightning-shaped checkpoint {hyper_parameters, state_dict with
    'model.' prefix} - the on-disk format `load_model` reads
    (visnet.py:74-87)."""
    import torch

    ck = {
        "hyper_parameters": dict(hp),
        "state_dict": {"model." + k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd.items()},
    }
    torch.save(ck, path)
