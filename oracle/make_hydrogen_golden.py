"""TEST INFRASTRUCTURE / DATA PREP - runs the REFERENCE's own hydrogen optimiser
(/root/reference/src/Fragmentation/hydrogen/energies.py `HydrogenOptimizer`, with the reference's `CTable`
term filters, ctable.py:168-244) on the example proteins and stores the relaxed cap-hydrogen positions as
golden vectors for oracle/hydrogen_oracle.py and the HIP optimiser.

    python -m oracle.make_hydrogen_golden       (build container only)

What comes from the reference: prmtop parsing, the bond/angle/dihedral/non-bonded term selection per dipeptide,
the five energy functions, the L-BFGS call.  What is restated here: PyG's batching of `ProteinData`
(topology.py:109-130 `__inc__`: atom indices += natom, parameter indices += numbnd/numang/nptra/ntypes(ntypes+1)/2)
because torch_geometric is not installed, and the AMBER row order (by atom NAME, ai2bmd_amd/hydrogen.py, instead
of utils/seq_dict.pkl).  TorchScript is disabled (PYTORCH_JIT=0) so that the scatter shim can be called.
"""
import importlib.util
import os
import sys
import types

os.environ["PYTORCH_JIT"] = "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "shims"))

from ai2bmd_amd.amber import TOPOLOGY_OF, load_tables  # noqa: E402
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, fragment_positions  # noqa: E402
from ai2bmd_amd.hydrogen import build_hydrogen_plan  # noqa: E402
from oracle.hydrogen_oracle import HydrogenOracle  # noqa: E402

REF = "/root/reference/src/Fragmentation/hydrogen"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_modules():
    ctable = _load("ref_ctable", f"{REF}/ctable.py")
    # energies.py imports ProteinData only for type annotations
    pkg = types.ModuleType("Fragmentation")
    hyd = types.ModuleType("Fragmentation.hydrogen")
    topo = types.ModuleType("Fragmentation.hydrogen.topology")
    topo.ProteinData = object
    sys.modules.update({"Fragmentation": pkg, "Fragmentation.hydrogen": hyd, "Fragmentation.hydrogen.topology": topo})
    energies = _load("ref_energies", f"{REF}/energies.py")
    return ctable.CTable, energies.HydrogenOptimizer


def reference_batch(CTable, p, plan, hplan, frag_pos):
    """The batch object HydrogenOptimizer.optimize_hydrogen consumes, in AMBER row order."""
    B = len(plan.start)
    resname_of = {int(r): str(p.resnames[np.flatnonzero(p.resnums == r)[0]]) for r in set(p.resnums.tolist())}
    cat = {k: [] for k in (
        "pos", "atom_idx", "other_idx", "charge", "bond_force_constant", "bond_equil_value", "angle_force_constant",
        "angle_equil_value", "dihedral_force_constant", "dihedral_periodicity", "dihedral_phase",
        "lennard_jones_acoef", "lennard_jones_bcoef", "bonds_atom_idx_src", "bonds_atom_idx_dst", "bond_idx",
        "angles_atom_idx_i", "angles_atom_idx_j", "angles_atom_idx_k", "angle_idx", "dihedrals_atom_idx_i",
        "dihedrals_atom_idx_j", "dihedrals_atom_idx_k", "dihedrals_atom_idx_l", "dihedral_idx",
        "nonbonded_atom_idx_src", "nonbonded_atom_idx_dst", "lj_idx", "bond_batch", "angle_batch", "dihedral_batch",
        "nonbonded_batch")}
    off = dict(atom=0, bnd=0, ang=0, dih=0, lj=0)
    amber_rows = []  # fragment-batch row of every AMBER-ordered batch atom
    ctables = {}
    for g, b in enumerate(range(0, B, 2)):
        code = TOPOLOGY_OF[resname_of[b // 2 + 2]]
        if code not in ctables:
            ctables[code] = CTable.from_prmtop(f"/root/reference/src/Fragmentation/prmtop/{code}.prmtop")
        ct = ctables[code]
        ti = hplan.tmpl_index[g]
        rows = np.arange(plan.start[b], plan.end[b])
        row_of_tmpl = np.empty(len(rows), dtype=np.int64)
        row_of_tmpl[ti] = rows
        amber_rows.append(row_of_tmpl)
        is_cap_t = np.zeros(len(rows), bool)
        is_cap_t[ti[plan.src[rows] < 0]] = True
        atom_idx = torch.as_tensor(np.flatnonzero(is_cap_t), dtype=torch.long)
        other_idx = torch.as_tensor(np.flatnonzero(~is_cap_t), dtype=torch.long)
        cat["pos"].append(torch.as_tensor(frag_pos[row_of_tmpl], dtype=torch.float32))
        cat["atom_idx"].append(atom_idx + off["atom"])
        cat["other_idx"].append(other_idx + off["atom"])
        for k in ("charge", "bond_force_constant", "bond_equil_value", "angle_force_constant", "angle_equil_value",
                  "dihedral_force_constant", "dihedral_periodicity", "dihedral_phase", "lennard_jones_acoef",
                  "lennard_jones_bcoef"):
            cat[k].append(getattr(ct, k).float())
        bs, bd, bi = ct.filter_bonds(atom_idx)
        ai, aj, ak, aidx = ct.filter_angles(atom_idx)
        di, dj, dk, dl, didx = ct.filter_dihedrals(atom_idx)
        ns, nd = ct.gen_nonbonded_pair(atom_idx)
        lj = ct.generate_lj_idx(ns, nd)
        for k, v in (("bonds_atom_idx_src", bs), ("bonds_atom_idx_dst", bd), ("angles_atom_idx_i", ai),
                     ("angles_atom_idx_j", aj), ("angles_atom_idx_k", ak), ("dihedrals_atom_idx_i", di),
                     ("dihedrals_atom_idx_j", dj), ("dihedrals_atom_idx_k", dk), ("dihedrals_atom_idx_l", dl),
                     ("nonbonded_atom_idx_src", ns), ("nonbonded_atom_idx_dst", nd)):
            cat[k].append(v + off["atom"])
        cat["bond_idx"].append(bi + off["bnd"])
        cat["angle_idx"].append(aidx + off["ang"])
        cat["dihedral_idx"].append(didx + off["dih"])
        cat["lj_idx"].append(lj + off["lj"])
        for k, ref in (("bond_batch", bi), ("angle_batch", aidx), ("dihedral_batch", didx), ("nonbonded_batch", lj)):
            cat[k].append(torch.full_like(ref, g))
        off["atom"] += ct.natom
        off["bnd"] += ct.numbnd
        off["ang"] += ct.numang
        off["dih"] += ct.nptra
        off["lj"] += ct.ntypes * (ct.ntypes + 1) // 2
    batch = types.SimpleNamespace(**{k: torch.cat(v) for k, v in cat.items()})
    return batch, np.concatenate(amber_rows)


def main():
    CTable, HydrogenOptimizer = reference_modules()
    tables = load_tables(os.path.join(ROOT, "tests", "golden", "amber_tables.npz"))
    torch.set_num_threads(8)
    for name in ("chig", "trpcage", "ww", "abd"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"protein_{name}.npz"))
        p = ProteinAtoms(names=z["names"], resnames=z["resnames"], resnums=z["resnums"], numbers=z["numbers"],
                         positions=z["positions"])
        plan = build_plan(p)
        hplan = build_hydrogen_plan(p, plan, tables)
        rng = np.random.default_rng(11)
        outs = {}
        for tag, jitter in (("x0", 0.0), ("x1", 0.05)):  # the PDB geometry and a thermally displaced one
            prot = p.positions + jitter * rng.standard_normal(p.positions.shape)
            frag_pos = fragment_positions(plan, prot).astype(np.float32)
            batch, amber_rows = reference_batch(CTable, p, plan, hplan, frag_pos)
            opt = HydrogenOptimizer(max_iter=10)
            e0 = opt.cal_potential_energy(batch).sum(0).numpy()
            opt.optimize_hydrogen(batch)
            e1 = opt.cal_potential_energy(batch).sum(0).numpy()
            relaxed = frag_pos.copy()
            relaxed[amber_rows] = batch.pos.numpy()
            moved = np.abs(relaxed - frag_pos).max(axis=1) > 0
            assert set(np.flatnonzero(moved)) <= set(hplan.cap_rows.tolist())
            # the restatement (same term lists through our own builder, same optimiser) must agree
            orc = HydrogenOracle(hplan)
            e0_o = orc.energy(torch.as_tensor(frag_pos)).numpy()
            rel_o = orc.relax(frag_pos)
            print(f"{name}/{tag}: caps {len(hplan.cap_rows)}  E0 ref {e0.sum():.4f} oracle {e0_o.sum():.4f}  "
                  f"E1 ref {e1.sum():.4f}  max|dx| {np.abs(relaxed - frag_pos).max():.4f}  "
                  f"oracle-vs-ref {np.abs(rel_o - relaxed).max():.2e}")
            assert np.allclose(e0, e0_o, rtol=2e-4, atol=2e-3), (e0, e0_o)
            outs[f"{tag}_prot"] = prot.astype(np.float64)
            outs[f"{tag}_e0"] = e0.astype(np.float64)
            outs[f"{tag}_e1"] = e1.astype(np.float64)
            outs[f"{tag}_caps"] = relaxed[hplan.cap_rows].astype(np.float32)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"hopt_{name}.npz"), **outs)


if __name__ == "__main__":
    main()
