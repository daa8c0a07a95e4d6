"""TEST INFRASTRUCTURE ONLY - one protein golden that NO line of builder arithmetic touches.

    python -m oracle.make_refchain_golden        (build container only; ~2 min)

For Chignolin (the reference's own pre-processed example) every link is the reference's own code:

    Fragmentation/distancefrag.py   DistanceFragment.fragment + get_dipeptide_positions   (oracle/ref_fragmenter.py)
      -> z / pos (first-guess cap hydrogens, AMBER row order) / start / end / select_index / origin_index
    ViSNet/model                    create_model(default hparams) + bench.py's seeded weights, fp64 and fp32
    AIMD/fragment.py                FragmentData.scalar_split / vector_split
    Calculators/combiner.py         DipeptideBondedCombiner.energy_combine / forces_combine (torch_scatter shim)

and the result (protein energy, forces[175, 3]) is stored as tests/golden/refchain_chig.npz.  The other protein
goldens (`Fprot64_*` in visnet_prot_*.npz, oracle/make_protein_golden.py) recombine the reference model's fragment
forces with OUR `combine_host` on OUR plan; tests/test_protein_golden.py lays the two side by side, and the device
pipeline (cap-hydrogen relaxation off = the same "placed" geometry) is checked against this file on the GPU.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ai2bmd_amd.fragmentation import ProteinAtoms  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from oracle.make_golden import run_reference  # noqa: E402
from oracle.ref_caller import load_reference_caller  # noqa: E402
from oracle.ref_fragmenter import run_reference_fragmenter  # noqa: E402

WEIGHT_SEED = 2024
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    d = np.load(os.path.join(GOLD, "protein_chig.npz"))
    p = ProteinAtoms(names=d["names"], resnames=d["resnames"], resnums=d["resnums"], numbers=d["numbers"],
                     positions=d["positions"].astype(np.float64))
    r = run_reference_fragmenter(p)
    hp = default_hparams()
    sd = make_state_dict(hp, seed=WEIGHT_SEED)
    z = np.asarray(r["z"], np.int64)
    pos = np.asarray(r["pos"], np.float32)
    start, end = np.asarray(r["start"], np.int64), np.asarray(r["end"], np.int64)
    ref = load_reference_caller(lambda path, device: None, object, prefer="source")
    fd = ref.FragmentData(z, pos, start, end, np.asarray(r["batch"]))
    comb = ref.DipeptideBondedCombiner()
    sel, org = torch.as_tensor(r["select_index"]), torch.as_tensor(r["origin_index"])
    out = dict(z=z.astype(np.int16), pos=pos, start=start.astype(np.int32), end=end.astype(np.int32),
               select_index=np.asarray(r["select_index"], np.int32), origin_index=np.asarray(r["origin_index"], np.int32),
               weight_seed=WEIGHT_SEED)
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        E, F = run_reference(hp, sd, z, pos, start, end, dt)
        E_t, F_t = torch.as_tensor(E), torch.as_tensor(F)
        e_dip, e_ace = (E_t[s] for s in fd.scalar_split())
        f_dip, f_ace = (F_t[s] for s in fd.vector_split())
        out[f"Eprot{tag}"] = np.asarray(comb.energy_combine(e_dip, e_ace), np.float64)
        out[f"Fprot{tag}"] = np.asarray(comb.forces_combine(len(p), f_dip, f_ace, sel, org), np.float64)
        print(tag, "E", float(out[f"Eprot{tag}"]), "|F|max", float(np.abs(out[f"Fprot{tag}"]).max()), flush=True)
    np.savez_compressed(os.path.join(GOLD, "refchain_chig.npz"), **out)


if __name__ == "__main__":
    main()
