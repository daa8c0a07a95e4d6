"""TEST INFRASTRUCTURE / DATA PREP - runs the REFERENCE's own MM non-bonded calculator
(/root/reference/src/Calculators/nonbonded.py `MMNonBondedCalculator.set_parameters/__call__`) on the example
proteins and stores energy + forces as golden vectors for oracle/nonbonded_oracle.py and the HIP kernel.

    python -m oracle.make_nonbonded_golden      (build container only)

The reference module is loaded from the reference tree with its two absent imports stubbed:
  * `ase.units` (ASE is not installed): only the six constants it reads (C, _eps0, kJ, mol, nm, pi), restated from
    ASE >= 3.12's default CODATA-2014 table (oracle/nonbonded_oracle.py header);
  * `AIMD.protein.Protein` (needs ase + openmm): only used as a type annotation; a duck-typed object provides
    `sigmas / epsilons / charges / get_positions() / __len__ / initial_mm_adjmatrix()`.
`initial_mm_adjmatrix` follows AIMD/protein.py:133-151 with `exclude_pair` built as distancefrag.py:355-363 does
(all ordered pairs inside one dipeptide) from OUR fragment plan; per-atom parameters come from the AMBER tables
(ai2bmd_amd.amber.protein_mm_parameters) because OpenMM is absent.
"""
import importlib.util
import itertools
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "shims"))

from ai2bmd_amd.amber import load_tables, protein_mm_parameters  # noqa: E402
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan  # noqa: E402
from oracle import nonbonded_oracle as nbo  # noqa: E402


def reference_calculator():
    ase = types.ModuleType("ase")
    units = types.ModuleType("ase.units")
    units.C, units._eps0, units.kJ, units.mol, units.nm, units.pi = nbo.C, nbo._eps0, nbo.kJ, nbo.mol, nbo.nm, np.pi
    aimd = types.ModuleType("AIMD")
    prot = types.ModuleType("AIMD.protein")
    prot.Protein = object
    sys.modules.update({"ase": ase, "ase.units": units, "AIMD": aimd, "AIMD.protein": prot})
    spec = importlib.util.spec_from_file_location("ref_nonbonded", "/root/reference/src/Calculators/nonbonded.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.MMNonBondedCalculator


class DuckProtein:
    def __init__(self, positions, charges, sigmas, epsilons, dipeptides_index):
        self.positions, self.charges, self.sigmas, self.epsilons = positions, charges, sigmas, epsilons
        self.exclude_pair = set()
        for idx in dipeptides_index:  # distancefrag.py:355-363
            for x, y in itertools.combinations(idx, 2):
                self.exclude_pair.add((x, y))
                self.exclude_pair.add((y, x))

    def __len__(self):
        return len(self.positions)

    def get_positions(self):
        return self.positions

    def initial_mm_adjmatrix(self):  # AIMD/protein.py:133-151
        n = len(self.positions)
        pairs = [(i, j) for i, j in itertools.product(range(n), repeat=2) if i != j]
        return torch.tensor([p for p in pairs if p not in self.exclude_pair], dtype=torch.long).t()


def main():
    Calc = reference_calculator()
    tables = load_tables(os.path.join(ROOT, "tests", "golden", "amber_tables.npz"))
    for name in ("chig", "trpcage"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"protein_{name}.npz"))
        p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                         positions=z["positions"])
        plan = build_plan(p)
        q, sig, eps = protein_mm_parameters(p, tables)
        dip = [[int(a) for a in plan.src[plan.start[b]:plan.end[b]] if a >= 0]
               for b in range(len(plan.start)) if plan.is_dipeptide[b]]
        duck = DuckProtein(p.positions.astype(np.float32), q, sig, eps, dip)
        calc = Calc(device="cpu")
        calc.set_parameters(duck)
        e, f = calc(duck)
        # the restatement on the same pair list must agree with the reference's arithmetic (fp32 there, fp64 here)
        e_o, f_o = nbo.mm_nonbonded(p.positions, q.astype(np.float64), sig.astype(np.float64),
                                    eps.astype(np.float64), calc.src.numpy(), calc.dst.numpy())
        print(f"{name}: pairs {calc.src.numel()}  E ref {e:.6f} oracle {e_o:.6f} eV  max|dF| {np.abs(f - f_o).max():.2e} "
              f"(max|F| {np.abs(f).max():.3f})")
        assert abs(e - e_o) < 2e-4 * max(1.0, abs(e_o)) and np.abs(f - f_o).max() < 2e-4 * np.abs(f_o).max()
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"mm_{name}.npz"), energy=np.float64(e),
                            forces=f.astype(np.float32), n_pairs=np.int64(calc.src.numel()))


if __name__ == "__main__":
    main()
