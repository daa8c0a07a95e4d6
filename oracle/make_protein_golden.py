"""TEST INFRASTRUCTURE ONLY - golden vectors for the configurations bench.py actually runs: the real fragment
batches of the reference's example proteins (Chignolin B=19 N=391, Trp-cage B=39 N=737, WW B=69 N=1387,
ABD B=93 N=1850) at the reference's DEFAULT hyper-parameters (H=256, L=9) with bench.py's weight seed,
evaluated by the REFERENCE's own ViSNet source (/root/reference/src/ViSNet/model through oracle/shims).

    python -m oracle.make_protein_golden          (build container only)
    python -m oracle.make_protein_golden --small  (Chignolin at H=128, L=6: the small variant of SURVEY 8d)

Two geometries per protein, both from tests/golden/protein_<name>.npz through our fragment plan (which is pinned
row for row on the reference's own DistanceFragment, tests/golden/fragref_chig.npz):
  * "relaxed": cap hydrogens placed and relaxed by oracle/hydrogen_oracle.py (pinned bit-exactly on the reference's
    own HydrogenOptimizer, tests/golden/hopt_*.npz) = what DistanceFragment.get_fragments hands to the calculator
    (/root/reference/src/Fragmentation/distancefrag.py:56-92);
  * "placed": cap hydrogens at their first-guess position only (distancefrag.py:35-54), the input of the
    device pipeline when the relaxation is switched off.
Stored per geometry: fragment positions (fp32), per-fragment E and per-row F of the reference in fp64 (truth) and
fp32 (what the reference computes), and the recombined protein energy / forces (Calculators/combiner.py:12-41).
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

from ai2bmd_amd.amber import load_tables  # noqa: E402
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, combine_host, fragment_positions  # noqa: E402
from ai2bmd_amd.hydrogen import build_hydrogen_plan  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from oracle.hydrogen_oracle import HydrogenOracle  # noqa: E402
from oracle.make_golden import run_reference  # noqa: E402

WEIGHT_SEED = 2024  # bench.py's weights
PROTEINS = ("chig", "trpcage", "ww", "abd")
GOLD = os.path.join(ROOT, "tests", "golden")


def max_degree(pos, start, end, cutoff):
    deg = 0
    for a, b in zip(start, end):
        p = pos[a:b].astype(np.float64)
        if len(p):
            d2 = ((p[:, None] - p[None]) ** 2).sum(-1)
            deg = max(deg, int((d2 < cutoff * cutoff).sum(1).max()))
    return deg


def main():
    # `--small`: the SURVEY 8(d) small variant (H = 128, L = 6, "since the trained values are unknown") on Chignolin
    # only -> tests/golden/visnet_prot_chig_h128l6.npz (bench.py's `secondary` small-variant line is guarded by it)
    small = "--small" in sys.argv
    hp = default_hparams(embedding_dimension=128, num_layers=6) if small else default_hparams()
    suffix = "_h128l6" if small else ""
    sd = make_state_dict(hp, seed=WEIGHT_SEED)
    tables = load_tables(os.path.join(GOLD, "amber_tables.npz"))
    for name in (("chig",) if small else PROTEINS):
        z = np.load(os.path.join(GOLD, f"protein_{name}.npz"))
        p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                         positions=z["positions"].astype(np.float64))
        plan = build_plan(p)
        hplan = build_hydrogen_plan(p, plan, tables)
        placed = fragment_positions(plan, p.positions).astype(np.float32)
        relaxed = HydrogenOracle(hplan).relax(placed.copy())
        ace = hplan.alias >= 0
        relaxed[ace] = relaxed[hplan.alias[ace]]
        out = dict(hparams=json.dumps(hp), weight_seed=WEIGHT_SEED, z=plan.z.astype(np.int64), start=plan.start,
                   end=plan.end)
        for tag, pos in (("relaxed", relaxed), ("placed", placed)):
            E64, F64 = run_reference(hp, sd, plan.z, pos, plan.start, plan.end, torch.float64)
            E32, F32 = run_reference(hp, sd, plan.z, pos, plan.start, plan.end, torch.float32)
            Ep, Fp = combine_host(plan, E64, F64)
            out.update({f"pos_{tag}": pos, f"E_ref64_{tag}": E64, f"F_ref64_{tag}": F64,
                        f"E_ref32_{tag}": E32.astype(np.float32), f"F_ref32_{tag}": F32.astype(np.float32),
                        f"Eprot64_{tag}": np.float64(Ep), f"Fprot64_{tag}": np.asarray(Fp, np.float64),
                        f"max_degree_{tag}": max_degree(pos, plan.start, plan.end, hp["cutoff"])})
            print(f"{name}/{tag}: B={len(plan.start)} N={len(plan.z)} deg_max={out[f'max_degree_{tag}']} "
                  f"|F|max={np.abs(F64).max():.3f} fp32-vs-fp64 dE={np.abs(E32 - E64).max():.2e} "
                  f"dF={np.abs(F32 - F64).max():.2e}", flush=True)
        np.savez_compressed(os.path.join(GOLD, f"visnet_prot_{name}{suffix}.npz"), **out)


if __name__ == "__main__":
    main()
