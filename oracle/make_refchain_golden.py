"""TEST INFRASTRUCTURE ONLY - one protein golden that NO line of builder arithmetic touches.

    python -m oracle.make_refchain_golden [case ...]     (build container only; minutes per case)

Every link is the reference's own code:

    Fragmentation/distancefrag.py   DistanceFragment.fragment + get_dipeptide_positions   (oracle/ref_fragmenter.py)
      -> z / pos (first-guess cap hydrogens, AMBER row order) / start / end / select_index / origin_index
    ViSNet/model                    create_model(default hparams) + bench.py's seeded weights, fp64 and fp32
    AIMD/fragment.py                FragmentData.scalar_split / vector_split
    Calculators/combiner.py         DipeptideBondedCombiner.energy_combine / forces_combine (torch_scatter shim)

and the result (protein energy, forces[n_prot, 3]) is stored as tests/golden/refchain_<case>.npz.

Cases (`CASES`):
    chig        Chignolin, default hyper-parameters (max_num_neighbors = 32: no target is truncated)
    chig_nb20   Chignolin, max_num_neighbors = 20: the neighbour lists of more than half of the targets are truncated -
                `radius_graph` keeps the LOWEST-index sources (utils.py:259-266), so the answer depends on the row order
                of the fragment batch; only the reference's own order (AMBER order, distancefrag.py:731-737) gives it
    abd_nb31    ABD (746 atoms, 93 fragments; atoms in the reference's pre-processed order,
                ai2bmd_amd.fragmentation.preprocessed_order), max_num_neighbors = 31: one below its largest in-degree
    abd_nb24    ABD, max_num_neighbors = 24
Each file also stores how many targets were truncated (`n_truncated`, `n_targets`).

The other protein goldens (`Fprot64_*` in visnet_prot_*.npz, oracle/make_protein_golden.py) recombine the reference
model's fragment forces with OUR `combine_host` on OUR plan; tests/test_protein_golden.py lays the two side by side,
and the device pipeline (cap-hydrogen relaxation off = the same "placed" geometry) and the reference's own caller are
checked against these files on the GPU (tests/test_gpu_pipeline.py, tests/test_gpu_reference_caller.py).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ai2bmd_amd.fragmentation import ProteinAtoms, preprocessed_order  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from oracle.make_golden import run_reference  # noqa: E402
from oracle.ref_caller import load_reference_caller  # noqa: E402
from oracle.ref_fragmenter import run_reference_fragmenter  # noqa: E402

WEIGHT_SEED = 2024
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {  # name: (protein fixture, hyper-parameter overrides)
    "chig": ("chig", {}),
    "chig_nb20": ("chig", dict(max_num_neighbors=20)),
    "abd_nb31": ("abd", dict(max_num_neighbors=31)),
    "abd_nb24": ("abd", dict(max_num_neighbors=24)),
}


def truncation_count(pos, start, end, cutoff, max_nb):
    """(targets whose in-degree incl. the self loop exceeds max_nb, all targets)"""
    n_trunc = 0
    for a, b in zip(start, end):
        p = pos[a:b].astype(np.float64)
        if len(p):
            d2 = ((p[:, None] - p[None]) ** 2).sum(-1)
            n_trunc += int(((d2 < cutoff * cutoff).sum(1) > max_nb).sum())
    return n_trunc, int(len(pos))


def load_protein_for_reference(name):
    d = np.load(os.path.join(GOLD, f"protein_{name}.npz"))
    p = ProteinAtoms(names=d["names"], resnames=d["resnames"], resnums=d["resnums"], numbers=d["numbers"],
                     positions=d["positions"].astype(np.float64))
    return p if name.startswith("chig") else preprocessed_order(p)


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for case, (name, over) in CASES.items():
        if only and case not in only:
            continue
        p = load_protein_for_reference(name)
        r = run_reference_fragmenter(p)
        hp = default_hparams(**over)
        sd = make_state_dict(hp, seed=WEIGHT_SEED)
        z = np.asarray(r["z"], np.int64)
        pos = np.asarray(r["pos"], np.float32)
        start, end = np.asarray(r["start"], np.int64), np.asarray(r["end"], np.int64)
        ref = load_reference_caller(lambda path, device: None, object, prefer="source")
        fd = ref.FragmentData(z, pos, start, end, np.asarray(r["batch"]))
        comb = ref.DipeptideBondedCombiner()
        sel, org = torch.as_tensor(r["select_index"]), torch.as_tensor(r["origin_index"])
        n_trunc, n_tgt = truncation_count(pos, start, end, hp["cutoff"], hp["max_num_neighbors"])
        out = dict(z=z.astype(np.int16), pos=pos, start=start.astype(np.int32), end=end.astype(np.int32),
                   select_index=np.asarray(r["select_index"], np.int32),
                   origin_index=np.asarray(r["origin_index"], np.int32), weight_seed=WEIGHT_SEED,
                   max_num_neighbors=hp["max_num_neighbors"], n_truncated=n_trunc, n_targets=n_tgt)
        print(f"{case}: B={len(start)} N={len(z)} max_nb={hp['max_num_neighbors']} truncated targets "
              f"{n_trunc}/{n_tgt} ({100.0 * n_trunc / n_tgt:.1f} %)", flush=True)
        for tag, dt in (("64", torch.float64), ("32", torch.float32)):
            E, F = run_reference(hp, sd, z, pos, start, end, dt)
            E_t, F_t = torch.as_tensor(E), torch.as_tensor(F)
            e_dip, e_ace = (E_t[s] for s in fd.scalar_split())
            f_dip, f_ace = (F_t[s] for s in fd.vector_split())
            out[f"Eprot{tag}"] = np.asarray(comb.energy_combine(e_dip, e_ace), np.float64)
            out[f"Fprot{tag}"] = np.asarray(comb.forces_combine(len(p), f_dip, f_ace, sel, org), np.float64)
            if tag == "64":  # per-fragment truth too: localises a mismatch, and feeds the oracle check on CPU
                out["E_ref64"], out["F_ref64"] = np.asarray(E, np.float64), np.asarray(F, np.float64)
            print(" ", tag, "E", float(out[f"Eprot{tag}"]), "|F|max", float(np.abs(out[f"Fprot{tag}"]).max()), flush=True)
        np.savez_compressed(os.path.join(GOLD, f"refchain_{case}.npz"), **out)


if __name__ == "__main__":
    main()
