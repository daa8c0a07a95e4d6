"""AMBER topology tables for the cap-hydrogen relaxation (SURVEY.md 8f "next #1").

Own reader of the fields of an AMBER `.prmtop` that the reference's hydrogen optimiser uses
(/root/reference/src/Fragmentation/hydrogen/ctable.py:58-166, format: https://ambermd.org/FileFormats.php).
The reference ships one ACE-X-NME topology per residue type (src/Fragmentation/prmtop/*.prmtop, chosen by
utils/reference.py:7-34 `fragment_info`); `oracle/make_amber_fixtures.py` converts them once into
tests/golden/amber_tables.npz (the reference tree does not exist on the GPU box).
"""
from __future__ import annotations

import functools
import os

import numpy as np

# residue name -> topology code  (utils/reference.py:7-34)
TOPOLOGY_OF = {
    "ALA": "AA", "ARG": "RR", "ASP": "DD", "CYS": "CC", "CYX": "CYX", "GLN": "QQ", "GLY": "GG", "GLU": "EE",
    "LYS": "KK", "ASN": "NN", "LEU": "LL", "PRO": "PP", "SER": "SS", "THR": "TT", "VAL": "VV", "MET": "MM",
    "HIS": "HH", "HIE": "HH", "HID": "HID", "TRP": "WW", "TYR": "YY", "ILE": "II", "PHE": "FF",
}

FIELDS = ("charge", "atom_type_idx", "nonbonded_parm_index", "bond_force_constant", "bond_equil_value",
          "angle_force_constant", "angle_equil_value", "dihedral_force_constant", "dihedral_periodicity",
          "dihedral_phase", "lennard_jones_acoef", "lennard_jones_bcoef", "bonds_inc_hydrogen",
          "angles_inc_hydrogen", "dihedrals_inc_hydrogen", "number_excluded_atoms", "excluded_atoms_list")


def _floordiv3(a):
    return np.floor_divide(a, 3)


def read_prmtop(path):
    """-> dict of numpy arrays with the reference's conventions (ctable.py:98-166): 0-based type / parameter
    indices, atom indices = coordinate index // 3 (floor), parameter index of every term 0-based."""
    flags = {}
    name = None
    with open(path) as fh:
        for line in fh:
            if line.startswith("%FLAG"):
                name = line.split()[1]
                flags[name] = []
            elif line.startswith("%FORMAT"):
                flags[name] = {"fmt": line.strip()[8:-1], "lines": []}
            elif line.startswith("%"):
                continue
            elif name is not None and isinstance(flags[name], dict):
                flags[name]["lines"].append(line.rstrip("\n"))

    def nums(flag, dtype):
        toks = " ".join(flags[flag]["lines"]).split()
        return np.array([dtype(t) for t in toks])

    ptr = nums("POINTERS", int)
    natom, ntypes = int(ptr[0]), int(ptr[1])
    names_raw = "".join(l.ljust(80) for l in flags["ATOM_NAME"]["lines"])
    names = [names_raw[4 * i:4 * i + 4].strip() for i in range(natom)]
    t = dict(natom=natom, ntypes=ntypes, atom_names=np.array(names))
    t["charge"] = nums("CHARGE", float)
    t["atomic_number"] = nums("ATOMIC_NUMBER", int)
    t["atom_type_idx"] = nums("ATOM_TYPE_INDEX", int) - 1
    t["number_excluded_atoms"] = nums("NUMBER_EXCLUDED_ATOMS", int)
    t["nonbonded_parm_index"] = nums("NONBONDED_PARM_INDEX", int) - 1
    for f, flag in (("bond_force_constant", "BOND_FORCE_CONSTANT"), ("bond_equil_value", "BOND_EQUIL_VALUE"),
                    ("angle_force_constant", "ANGLE_FORCE_CONSTANT"), ("angle_equil_value", "ANGLE_EQUIL_VALUE"),
                    ("dihedral_force_constant", "DIHEDRAL_FORCE_CONSTANT"),
                    ("dihedral_periodicity", "DIHEDRAL_PERIODICITY"), ("dihedral_phase", "DIHEDRAL_PHASE"),
                    ("lennard_jones_acoef", "LENNARD_JONES_ACOEF"), ("lennard_jones_bcoef", "LENNARD_JONES_BCOEF")):
        t[f] = nums(flag, float)
    b = nums("BONDS_INC_HYDROGEN", int).reshape(-1, 3)
    a = nums("ANGLES_INC_HYDROGEN", int).reshape(-1, 4)
    d = nums("DIHEDRALS_INC_HYDROGEN", int).reshape(-1, 5)
    t["bonds_inc_hydrogen"] = np.concatenate([_floordiv3(b[:, :2]), b[:, 2:] - 1], axis=1)
    t["angles_inc_hydrogen"] = np.concatenate([_floordiv3(a[:, :3]), a[:, 3:] - 1], axis=1)
    t["dihedrals_inc_hydrogen"] = np.concatenate([_floordiv3(d[:, :4]), d[:, 4:] - 1], axis=1)
    t["excluded_atoms_list"] = nums("EXCLUDED_ATOMS_LIST", int) - 1
    return t


def save_tables(path, tables):
    flat = {}
    for code, t in tables.items():
        for k, v in t.items():
            flat[f"{code}/{k}"] = np.asarray(v)
    np.savez_compressed(path, **flat)


def load_tables(path):
    z = np.load(path, allow_pickle=False)
    out = {}
    for key in z.files:
        code, k = key.split("/", 1)
        out.setdefault(code, {})[k] = z[key]
    for t in out.values():
        t["natom"] = int(t["natom"])
        t["ntypes"] = int(t["ntypes"])
    return out


_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "amber_tables.npz")


@functools.lru_cache(maxsize=2)
def _tables_from(source):
    if source is None:
        return load_tables(_DATA)
    return {f[:-7]: read_prmtop(os.path.join(source, f)) for f in sorted(os.listdir(source)) if f.endswith(".prmtop")}


def default_tables():
    """ACE-X-NME AMBER tables: read from the AI2BMD tree when AI2BMD_PRMTOP_DIR points at its
    src/Fragmentation/prmtop (own .prmtop reader above), else the packaged conversion of the same files."""
    return _tables_from(os.environ.get("AI2BMD_PRMTOP_DIR") or None)


def protein_mm_parameters(prot, tables):
    """Per-atom charge [e], sigma [nm], epsilon [kJ/mol] in the units the reference takes from OpenMM
    (AIMD/protein.py:153-175 `charges`, `sigmas`, `epsilons`), looked up by (residue, atom name) in the ACE-X-NME
    AMBER tables: charge / 18.2223, and from the diagonal Lennard-Jones coefficients A = 4 eps sigma^12,
    B = 4 eps sigma^6 of the atom's type.  (OpenMM is not installed; the capped-dipeptide topologies carry the
    standard residue charges, so interior residues get their usual AMBER values.)"""
    n = len(prot.numbers)
    q, sig, eps = np.zeros(n, np.float32), np.full(n, 0.1, np.float32), np.zeros(n, np.float32)
    any_t = tables["AA"]

    def lookup(t, idx):
        ti = int(t["atom_type_idx"][idx])
        li = int(t["nonbonded_parm_index"][t["ntypes"] * ti + ti])
        a, b = float(t["lennard_jones_acoef"][li]), float(t["lennard_jones_bcoef"][li])
        if a > 0 and b > 0:
            return float(t["charge"][idx]) / 18.2223, 0.1 * (a / b) ** (1.0 / 6.0), 4.184 * b * b / (4.0 * a)
        return float(t["charge"][idx]) / 18.2223, 0.1, 0.0

    for i in range(n):
        res, name = str(prot.resnames[i]), str(prot.names[i])
        if res == "ACE":
            t, names, lo, hi = any_t, list(any_t["atom_names"][:6]), 0, 6
        elif res == "NME":
            nat = any_t["natom"]
            t, names, lo, hi = any_t, list(any_t["atom_names"][-6:]), nat - 6, nat
        else:
            t = tables[TOPOLOGY_OF[res]]
            nat = t["natom"]
            names, lo, hi = list(t["atom_names"][6:nat - 6]), 6, nat - 6
        if name not in names:
            raise ValueError(f"atom {name} of residue {res} not in the AMBER template")
        q[i], sig[i], eps[i] = lookup(t, lo + names.index(name))
    return q, sig, eps
