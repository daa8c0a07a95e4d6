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


def chig_with_disulfide(p):
    """Chignolin (pre-processed atom order) with TYR-3 and THR-7 turned into a CYX-CYX bridge (their dipeptides share
    no atom, and neither is the last one: the reference indexes past the end when the emptied slot is the last): backbone and CB kept, OG1 / CG -> SG (1.81 A from CB), HB / HB2 -> HB2, CG2 / HB3 -> HB3 (1.09 A), the
    other side-chain atoms dropped; atoms in the pre-processed order N CA C O H HA CB SG HB2 HB3.  The geometry is not
    physical (the two SG are far apart) - the fixture exists to exercise the reference's CYX index algebra, none of
    its examples has a bridge."""
    rule = {7: {"OG1": ("SG", 16, 1.81), "HB": ("HB2", 1, 1.09), "CG2": ("HB3", 1, 1.09)},
            3: {"CG": ("SG", 16, 1.81), "HB2": ("HB2", 1, 1.09), "HB3": ("HB3", 1, 1.09)}}
    rows = []
    for i in range(len(p.numbers)):
        r, nm = int(p.resnums[i]), str(p.names[i])
        if r in rule:
            if nm in ("N", "CA", "C", "O", "H", "HA", "CB"):
                rows.append((r, nm, "CYX", int(p.numbers[i]), p.positions[i]))
            elif nm in rule[r]:
                cb = p.positions[np.flatnonzero((p.resnums == r) & (p.names == "CB"))[0]]
                u = p.positions[i] - cb
                new, z, ln = rule[r][nm]
                rows.append((r, new, "CYX", z, cb + ln * u / np.linalg.norm(u)))
        else:
            rows.append((r, nm, str(p.resnames[i]), int(p.numbers[i]), p.positions[i]))
    out = []
    for r in sorted({x[0] for x in rows}):
        grp = [x for x in rows if x[0] == r]
        if grp[0][2] == "CYX":
            want = ["N", "CA", "C", "O", "H", "HA", "CB", "SG", "HB2", "HB3"]
            grp = [next(x for x in grp if x[1] == w) for w in want]
        out += grp
    return ProteinAtoms(names=np.array([x[1] for x in out]), resnames=np.array([x[2] for x in out]),
                        resnums=np.array([x[0] for x in out]), numbers=np.array([x[3] for x in out]),
                        positions=np.array([x[4] for x in out], dtype=np.float64))


def main():
    # The reference's permutation tables (utils/seq_dict.pkl) assume the atom order its own preprocessing leaves
    # behind (Tinker xyzpdb + utils/pdb.py:reorder_atoms); on the raw examples/*.pdb (tleap order) its fragmenter
    # pairs atomic numbers with the wrong rows.  "chig" is the reference's pre-processed example as it is; the other
    # three examples are put into that order by ai2bmd_amd.fragmentation.preprocessed_order (which reproduces the
    # pre-processed Chignolin file atom for atom) - the C3 / C4 inputs of BASELINE.json.
    for name in ("chig", "chigcyx", "trpcage", "ww", "abd"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"protein_{'chig' if name.startswith('chig') else name}.npz"))
        p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                         positions=z["positions"])
        if not name.startswith("chig"):
            from ai2bmd_amd.fragmentation import preprocessed_order

            p = preprocessed_order(p)
        if name == "chigcyx":
            p = chig_with_disulfide(p)
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "protein_chigcyx.npz"), names=p.names,
                                resnames=p.resnames, resnums=p.resnums, numbers=p.numbers, positions=p.positions)
        r = run_reference_fragmenter(p)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"fragref_{name}.npz"),
                            z=r["z"].astype(np.int16), pos=r["pos"].astype(np.float32),
                            start=r["start"].astype(np.int32), end=r["end"].astype(np.int32),
                            select_index=r["select_index"].astype(np.int32),
                            origin_index=r["origin_index"].astype(np.int32), n_dip_rows=np.int32(r["n_dip_rows"]))
        print(name, "B", len(r["start"]), "rows", len(r["z"]), "selected", len(r["select_index"]))


if __name__ == "__main__":
    main()
