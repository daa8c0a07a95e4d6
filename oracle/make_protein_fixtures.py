"""TEST INFRASTRUCTURE ONLY - turns the reference's example proteins
(/root/reference/examples/*.pdb) into small fixtures under tests/golden/ so that
bench.py and the GPU tests can run where /root/reference does not exist.

    python -m oracle.make_protein_fixtures

Stores the parsed protein (names, residues, numbers, positions); the fragment plan
is rebuilt from it by ai2bmd_amd.fragmentation.build_plan at run time.  Also checks
here, against the reference's own per-residue templates
(/root/reference/src/utils/reference.py:36-64), that every dipeptide built by our
fragmenter has exactly the template's atoms.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ai2bmd_amd.fragmentation import build_plan, parse_pdb  # noqa: E402

EXAMPLES = {
    "chig": "/root/reference/examples/chig_preprocessed/chig-preeq-nowat.pdb",
    "trpcage": "/root/reference/examples/trpcage.pdb",
    "ww": "/root/reference/examples/ww.pdb",
    "abd": "/root/reference/examples/abd.pdb",
}


def reference_templates():
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_reference", "/root/reference/src/utils/reference.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fragment_atomic_numbers


def main():
    tmpl = reference_templates()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, path in EXAMPLES.items():
        p = parse_pdb(path)
        plan = build_plan(p)
        resname_of = {int(r): str(p.resnames[np.flatnonzero(p.resnums == r)[0]]) for r in set(p.resnums.tolist())}
        b = 0
        for fi in range(len(plan.start)):
            zf = np.sort(plan.z[plan.start[fi]:plan.end[fi]])
            if plan.is_dipeptide[fi]:
                res = resname_of[b + 2]
                assert (zf == np.sort(tmpl[res])).all(), (name, fi, res)
                b += 1
            else:
                assert (zf == np.sort(tmpl["ACENME"])).all(), (name, fi)
        np.savez_compressed(
            os.path.join(out_dir, f"protein_{name}.npz"),
            names=p.names.astype("U4"), resnames=p.resnames.astype("U3"), resnums=p.resnums, numbers=p.numbers,
            positions=p.positions.astype(np.float32),
        )
        print(f"{name}: {len(p)} atoms, {int(p.resnums.max())} residues -> B={len(plan.start)} fragments, "
              f"N={len(plan.z)} fragment atoms, cap-H={int((plan.src < 0).sum())}, "
              f"sizes {int((plan.end - plan.start).min())}..{int((plan.end - plan.start).max())}")


if __name__ == "__main__":
    main()
