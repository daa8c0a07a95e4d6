"""TEST INFRASTRUCTURE ONLY - generates tests/golden/*.npz by running the
REFERENCE's own ViSNet source (/root/reference/src/ViSNet/model, imported through
oracle/ref_import.py + oracle/shims) on seeded weights and seeded inputs.

Run in the build container only (the reference tree does not exist on the GPU
box):   python -m oracle.make_golden
Each fixture stores the inputs, the hyper-parameters, the weight seed and the
reference's (E, F) in float64 (truth) and float32 (what the reference computes).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.inputs import random_fragments  # noqa: E402
from oracle.ref_import import import_reference_create_model  # noqa: E402
from oracle.weights import default_hparams, make_state_dict  # noqa: E402

CASES = {
    # name: (hparam overrides, weight seed, input seed, fragment sizes)
    "h64_l2": (dict(embedding_dimension=64, num_layers=2), 11, 101, [22, 12, 0, 30, 19]),
    "h128_l3_lmax1": (dict(embedding_dimension=128, num_layers=3, lmax=1), 12, 102, [12, 26, 33]),
    "h64_l2_trunc": (dict(embedding_dimension=64, num_layers=2, max_num_neighbors=16), 13, 103, [44, 36, 12]),
    "h256_l9_default": (dict(), 14, 104, [22, 12, 28]),
    "h64_l3_rms": (dict(embedding_dimension=64, num_layers=3, vecnorm_type="rms"), 15, 105, [24, 12]),
    "h64_l3_maxmin": (dict(embedding_dimension=64, num_layers=3, vecnorm_type="max_min"), 16, 106, [24, 12]),
    "h64_l2_whole": (dict(embedding_dimension=64, num_layers=2), 17, 107, [120]),
    # hyper-parameters a checkpoint may legally carry (visnet.py:74-76 reads them from the file): the other
    # radial basis (utils.py:60-90), the other activations (utils.py:93-116), any hidden width that is a multiple of 64
    "h64_l2_gauss": (dict(embedding_dimension=64, num_layers=2, rbf_type="gauss"), 18, 108, [22, 12, 30]),
    "h128_l2_gauss_lmax1": (dict(embedding_dimension=128, num_layers=2, rbf_type="gauss", lmax=1, num_rbf=20), 19, 109,
                            [26, 12]),
    "h64_l2_ssp": (dict(embedding_dimension=64, num_layers=2, activation="ssp", attn_activation="ssp"), 20, 110,
                   [22, 12, 19]),
    "h64_l2_tanh_sig": (dict(embedding_dimension=64, num_layers=2, activation="tanh", attn_activation="sigmoid"), 21, 111,
                        [24, 12]),
    "h128_l2_sig_swish": (dict(embedding_dimension=128, num_layers=2, activation="sigmoid", attn_activation="swish"), 22,
                          112, [28, 12]),
    "h192_l2": (dict(embedding_dimension=192, num_layers=2), 23, 113, [22, 12, 27]),
    "h320_l2_rms": (dict(embedding_dimension=320, num_layers=2, vecnorm_type="rms"), 24, 114, [24, 12]),
    "h384_l2_lmax1": (dict(embedding_dimension=384, num_layers=2, lmax=1), 25, 115, [30, 12]),
    "h448_l2": (dict(embedding_dimension=448, num_layers=2, num_heads=4), 26, 116, [19, 12]),
    "h512_l3": (dict(embedding_dimension=512, num_layers=3), 27, 117, [22, 12, 33]),
    # reduce_op = "mean" (visnet.py:146): per-fragment mean of the atomic terms; an empty fragment in the batch
    "h64_l2_mean": (dict(embedding_dimension=64, num_layers=2, reduce_op="mean"), 28, 118, [22, 0, 12, 31]),
    # head counts that do not divide 64 (the reference asks only hidden % num_heads == 0, visnet_block.py:158-166):
    # 3 heads of 64 channels at 3 channels per lane (a lane's channels straddle heads), 5 heads, 12 heads of 16
    "h192_l2_heads3": (dict(embedding_dimension=192, num_layers=2, num_heads=3), 29, 119, [22, 12, 27]),
    "h320_l2_heads5": (dict(embedding_dimension=320, num_layers=2, num_heads=5, activation="ssp"), 30, 120, [24, 12]),
    "h192_l3_heads12": (dict(embedding_dimension=192, num_layers=3, num_heads=12), 31, 121, [30, 12, 19]),
    # the generic-head path with lmax = 1 (S = 3), the mean reduction and an rms VecLayerNorm at once
    "h192_l2_heads6_lmax1_mean_rms": (dict(embedding_dimension=192, num_layers=2, num_heads=6, lmax=1, reduce_op="mean",
                                       vecnorm_type="rms"), 32, 122, [28, 12, 0, 22]),
}


def run_reference(hp, sd, z, pos, start, end, dtype):
    create_model = import_reference_create_model()
    prev = torch.get_default_dtype()
    try:
        # utils.py:271 and visnet_block.py:119 allocate default-dtype zeros
        torch.set_default_dtype(dtype)
        model = create_model(hp)
        model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
        model = model.to(dtype).eval()
        for p in model.parameters():
            p.requires_grad = False
        sizes = end - start
        batch = np.repeat(np.cumsum(sizes > 0) - 1, sizes)
        E, F = model(dict(z=torch.as_tensor(z), pos=torch.as_tensor(pos).to(dtype), batch=torch.as_tensor(batch)))
        return E.detach().numpy(), F.detach().numpy()
    finally:
        torch.set_default_dtype(prev)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (over, wseed, iseed, sizes) in CASES.items():
        if os.path.exists(os.path.join(out_dir, f"visnet_{name}.npz")) and "--all" not in sys.argv:
            continue  # committed fixtures stay byte-identical; `--all` regenerates everything
        hp = default_hparams(**over)
        sd = make_state_dict(hp, seed=wseed)
        z, pos, start, end = random_fragments(iseed, sizes, cutoff=hp["cutoff"])
        E64, F64 = run_reference(hp, sd, z, pos, start, end, torch.float64)
        E32, F32 = run_reference(hp, sd, z, pos, start, end, torch.float32)
        np.savez_compressed(
            os.path.join(out_dir, f"visnet_{name}.npz"),
            hparams=json.dumps(hp), weight_seed=wseed, z=z, pos=pos, start=start, end=end,
            E_ref64=E64, F_ref64=F64, E_ref32=E32.astype(np.float32), F_ref32=F32.astype(np.float32),
        )
        print(f"{name}: N={len(z)} B={len(start)} |F|mean={np.abs(F64).mean():.4f} "
              f"fp32-vs-fp64 dE={np.abs(E32 - E64).max():.2e} dF={np.abs(F32 - F64).max():.2e}")


if __name__ == "__main__":
    main()
