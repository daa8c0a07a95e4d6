"""TEST INFRASTRUCTURE / DATA PREP - converts the reference's ACE-X-NME AMBER topologies
(/root/reference/src/Fragmentation/prmtop/*.prmtop) into tests/golden/amber_tables.npz with
ai2bmd_amd.amber.read_prmtop, and cross-checks the reader against the reference's own parser
(Fragmentation/hydrogen/ctable.py, imported from the reference tree) field by field.

    python -m oracle.make_amber_fixtures        (build container only)
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ai2bmd_amd.amber import FIELDS, TOPOLOGY_OF, read_prmtop, save_tables  # noqa: E402

PRMTOP_DIR = "/root/reference/src/Fragmentation/prmtop"


def reference_ctable():
    spec = importlib.util.spec_from_file_location("ref_ctable", "/root/reference/src/Fragmentation/hydrogen/ctable.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CTable


def main():
    CTable = reference_ctable()
    tables = {}
    for code in sorted(set(TOPOLOGY_OF.values())):
        path = os.path.join(PRMTOP_DIR, f"{code}.prmtop")
        t = read_prmtop(path)
        ref = CTable.from_prmtop(path)
        assert t["natom"] == ref.natom and t["ntypes"] == ref.ntypes
        for f in FIELDS:
            r = getattr(ref, f).numpy()
            assert t[f].shape == r.shape and np.allclose(t[f], r, rtol=1e-6, atol=0), (code, f)
        tables[code] = t
        print(f"{code}: {t['natom']} atoms, names {' '.join(t['atom_names'][:8])} ...")
    out = os.path.join(ROOT, "tests", "golden", "amber_tables.npz")
    save_tables(out, tables)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
