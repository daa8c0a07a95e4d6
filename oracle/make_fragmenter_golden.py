"""TEST INFRASTRUCTURE / DATA PREP - runs the REFERENCE's own fragmenter (oracle/ref_fragmenter.py) on the example
proteins and stores its fragment batch (atomic numbers, first-guess positions incl. cap hydrogens, offsets, force
recombination indices) as golden vectors for ai2bmd_amd.fragmentation.build_plan.

    python -m oracle.make_fragmenter_golden     (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ai2bmd_amd.fragmentation import ProteinAtoms  # noqa: E402
from oracle.ref_fragmenter import run_reference_fragmenter  # noqa: E402


def main():
    # only the PRE-PROCESSED example: the reference's permutation tables (utils/seq_dict.pkl) assume the atom order its
    # own preprocessing (external AmberTools) writes; on the raw examples/*.pdb its fragmenter pairs atomic numbers
    # with the wrong rows.  Chignolin covers TYR ASP PRO GLU THR GLY TRP incl. the PRO / GLY neighbour special cases.
    for name in ("chig",):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"protein_{name}.npz"))
        p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                         positions=z["positions"])
        r = run_reference_fragmenter(p)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"fragref_{name}.npz"),
                            z=r["z"].astype(np.int16), pos=r["pos"].astype(np.float32),
                            start=r["start"].astype(np.int32), end=r["end"].astype(np.int32),
                            select_index=r["select_index"].astype(np.int32),
                            origin_index=r["origin_index"].astype(np.int32), n_dip_rows=np.int32(r["n_dip_rows"]))
        print(name, "B", len(r["start"]), "rows", len(r["z"]), "selected", len(r["select_index"]))


if __name__ == "__main__":
    main()
