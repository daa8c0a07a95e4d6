"""Static fragmentation plan: protein -> interleaved dipeptide / ACE-NME batch.

Host-side (numpy) producer of the inputs of the hot path, built once per
simulation; the per-step part (gather + cap-hydrogen placement) is a HIP kernel
driven by the arrays built here (`vsn_build_fragments`).

Restates the behaviour of the reference's fragmenter
  /root/reference/src/Fragmentation/basefrag.py:45-167     (which protein atoms
        belong to which dipeptide / ACE-NME),
  /root/reference/src/Fragmentation/distancefrag.py:366-498 (which bonds are cut
        and capped: acceptor, removed neighbour, acceptor-H length = sum of the
        covalent radii C 0.76 / N 0.71 / H 0.31),
  /root/reference/src/Fragmentation/distancefrag.py:35-54   (cap H placed on the
        acceptor -> removed-neighbour line),
  /root/reference/src/Fragmentation/distancefrag.py:250-350 (interleaving,
        select/origin indices for the force recombination)
  /root/reference/src/Fragmentation/distancefrag.py:185-240,805-845 (CYX pairs: the two
        dipeptides of a disulfide bridge become ONE fragment in the slot of the
        first, the slot of the second stays empty)
Rows inside a fragment follow the REFERENCE's order by default (`order="amber"`): dipeptides in the atom order of
their AMBER topology - what the reference's permutation with utils/seq_dict.pkl leaves behind
(distancefrag.py:731-737) - and ACE-NME fragments as the first six atoms of the next dipeptide followed by the last
six of the previous one (distancefrag.py:291-302).  `radius_graph` keeps the LOWEST-index `max_num_neighbors`
sources of a target, so under neighbour truncation the forces depend on the row order: with the reference's order
the FragmentData handed to the model is row for row the reference's and so is the truncated graph
(tests/golden/refchain_*_nb*.npz).  Atoms are matched to template slots by NAME (ai2bmd_amd/hydrogen.py), so the
protein itself may come in any atom order.  `order="grouped"` keeps the intermediate layout
[previous-residue part | residue | next-residue part] the AMBER order is derived from (tests only).
Pinned on the reference's own fragmenter (oracle/ref_fragmenter.py, tests/golden/fragref_*.npz).
The per-step relaxation of the cap hydrogens is ai2bmd_amd/hydrogen.py + csrc/hydrogen.hip.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ELEMENT_Z = {"H": 1, "C": 6, "N": 7, "O": 8, "S": 16}
RADIUS = {"H": 0.31, "C": 0.76, "N": 0.71, "O": 0.66}  # distancefrag.py:383-388


@dataclass
class ProteinAtoms:
    names: np.ndarray       # atom names (str)
    resnames: np.ndarray    # residue names (str)
    resnums: np.ndarray     # residue numbers, 1-based, continuous
    numbers: np.ndarray     # atomic numbers int64
    positions: np.ndarray   # float64 [n,3]

    def __len__(self):
        return len(self.numbers)


def parse_pdb(path) -> ProteinAtoms:
    names, resn, resi, zs, xyz = [], [], [], [], []
    with open(path) as fh:
        for line in fh:
            if not line.startswith(("ATOM", "HETATM")):
                continue
            name = line[12:16].strip()
            elem = line[76:78].strip() if len(line) >= 78 else ""
            if not elem:
                elem = name.lstrip("0123456789")[0]
            names.append(name)
            resn.append(line[17:20].strip())
            resi.append(int(line[22:26]))
            zs.append(ELEMENT_Z[elem.upper()[0] if elem.upper() not in ELEMENT_Z else elem.upper()])
            xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return ProteinAtoms(np.array(names), np.array(resn), np.array(resi, dtype=np.int64),
                        np.array(zs, dtype=np.int64), np.array(xyz, dtype=np.float64))


@dataclass
class FragmentPlan:
    n_prot: int
    z: np.ndarray          # int64 [Nf] atomic numbers of the fragment batch
    start: np.ndarray      # int64 [B]
    end: np.ndarray        # int64 [B]
    src: np.ndarray        # int64 [Nf] protein atom copied, or -1 for a cap hydrogen
    acceptor: np.ndarray   # int64 [Nf] cap H: atom it is bonded to   (else -1)
    toward: np.ndarray     # int64 [Nf] cap H: removed neighbour giving the direction (else -1)
    length: np.ndarray     # float32 [Nf] cap H: acceptor-H distance (else 0)
    is_dipeptide: np.ndarray  # bool [B]
    # force recombination in the reference's convention (combiner.py:24-41)
    n_dip_rows: int
    row_of_cat: np.ndarray  # int64 [Nf] row k of cat[F_dip, F_ace] -> row in the interleaved batch
    select_index: np.ndarray  # int64 [K] rows of cat[...] that are original (non-cap) atoms
    origin_index: np.ndarray  # int64 [K] protein atom of every selected row
    energy_sign: np.ndarray   # float32 [B] +1 dipeptide, -1 ACE-NME (combiner.py:19)
    # rows of every ORIGINAL dipeptide d (they differ from the fragment ranges when CYX pairs are merged)
    dip_row_start: np.ndarray = None  # int64 [n_dip]
    dip_row_end: np.ndarray = None    # int64 [n_dip]
    cyx_partner: np.ndarray = None    # int64 [n_dip]: dipeptide merged INTO this one (-1 none, -2 = this one was merged away)
    # AMBER template slot of every dipeptide row (-1 on ACE-NME rows); set when the rows ARE in AMBER order, where it is
    # the position inside the fragment (hydrogen.amber_ordered) - build_hydrogen_plan then takes it instead of names
    tmpl_slot: np.ndarray = None      # int64 [Nf]


def _residue_atom(p: ProteinAtoms, resnum: int, name: str) -> int:
    hit = np.flatnonzero((p.resnums == resnum) & (p.names == name))
    if hit.size == 0:
        raise ValueError(f"no atom {name} in residue {resnum}")
    return int(hit[0])


def build_plan(p: ProteinAtoms, order: str = "amber", tables: dict = None) -> FragmentPlan:
    """The fragment batch of protein `p`.  `order="amber"` (default): rows in the reference's order (module docstring);
    `tables`: ACE-X-NME AMBER tables giving the atom order of every template (default: the packaged conversion of the
    reference's .prmtop files, ai2bmd_amd.amber.default_tables)."""
    if order == "grouped":
        return _grouped_plan(p)
    if order != "amber":
        raise ValueError(f"order must be 'amber' or 'grouped', not {order!r}")
    from .amber import default_tables
    from .hydrogen import amber_ordered

    return amber_ordered(p, _grouped_plan(p), tables if tables is not None else default_tables())


def _grouped_plan(p: ProteinAtoms) -> FragmentPlan:
    nres = int(p.resnums.max())
    if len(set(p.resnums.tolist())) != nres:
        raise ValueError("residue numbers are not continuous")  # basefrag.py:67-69
    n_dip, n_ace = nres - 2, nres - 3
    if n_dip < 2:
        raise NotImplementedError("3 or fewer residues (incl. ACE/NME caps): use whole-molecule mode")
    resname_of = {int(r): str(p.resnames[np.flatnonzero(p.resnums == r)[0]]) for r in range(1, nres + 1)}
    if resname_of[1] != "ACE" or resname_of[nres] != "NME":
        raise ValueError("chain must be capped with ACE ... NME")

    def cap(acc, toward, elem_acc):
        return ("cap", acc, toward, RADIUS[elem_acc] + RADIUS["H"])

    def prev_part(r):
        """residue r seen as the acetyl-like N-terminal cap: CA, HA*, caps, C, O (6 atoms)."""
        if resname_of[r] == "ACE":
            idx = np.flatnonzero(p.resnums == r)
            return [("atom", int(i)) for i in idx]
        ca = _residue_atom(p, r, "CA")
        out = [("atom", ca)]
        out += [("atom", int(i)) for i in np.flatnonzero((p.resnums == r) & np.char.startswith(p.names, "HA"))]
        out.append(cap(ca, _residue_atom(p, r, "N"), "C"))  # CA-N -> CA-H
        if resname_of[r] != "GLY":
            out.append(cap(ca, _residue_atom(p, r, "CB"), "C"))  # CA-CB -> CA-H
        out += [("atom", _residue_atom(p, r, "C")), ("atom", _residue_atom(p, r, "O"))]
        return out

    def next_part(r):
        """residue r seen as the N-methyl-amide-like C-terminal cap: N, H, CA, HA*, caps (6 atoms)."""
        if resname_of[r] == "NME":
            idx = np.flatnonzero(p.resnums == r)
            return [("atom", int(i)) for i in idx]
        n = _residue_atom(p, r, "N")
        ca = _residue_atom(p, r, "CA")
        out = [("atom", n)]
        if resname_of[r] == "PRO":
            out.append(cap(n, _residue_atom(p, r, "CD"), "N"))  # N-CD -> N-H (distancefrag.py:470-477)
        else:
            out.append(("atom", _residue_atom(p, r, "H")))
        out.append(("atom", ca))
        out += [("atom", int(i)) for i in np.flatnonzero((p.resnums == r) & np.char.startswith(p.names, "HA"))]
        out.append(cap(ca, _residue_atom(p, r, "C"), "C"))  # CA-C -> CA-H
        if resname_of[r] != "GLY":
            out.append(cap(ca, _residue_atom(p, r, "CB"), "C"))  # CA-CB -> CA-H
        return out

    dipeptides, acenmes = [], []
    for d in range(n_dip):
        r = d + 2
        own = [("atom", int(i)) for i in np.flatnonzero(p.resnums == r)]
        dipeptides.append(prev_part(r - 1) + own + next_part(r + 1))
    for k in range(n_ace):
        acenmes.append(prev_part(k + 2) + next_part(k + 3))
        assert len(acenmes[-1]) == 12, len(acenmes[-1])

    # disulfide bridges (distancefrag.py:805-845): CYX dipeptides paired by nearest SG-SG distance, first come first
    # served; the pair becomes one fragment in the slot of the first, the second slot stays empty (:185-240)
    cyx_partner = -np.ones(n_dip, dtype=np.int64)
    cyx_dips = [d for d in range(n_dip) if resname_of[d + 2] == "CYX"]
    if cyx_dips:
        if len(cyx_dips) % 2:
            raise ValueError("odd number of CYX residues")
        sg = np.array([p.positions[_residue_atom(p, d + 2, "SG")] for d in cyx_dips], dtype=np.float64)
        dist = np.linalg.norm(sg[None, :] - sg[:, None], axis=-1)
        np.fill_diagonal(dist, np.inf)
        pairs = {}
        for i, j in enumerate(np.argmin(dist, axis=-1)):
            if i in pairs or int(j) in pairs:
                continue
            pairs[i] = int(j)
        for i, j in pairs.items():
            cyx_partner[cyx_dips[i]] = cyx_dips[j]
            cyx_partner[cyx_dips[j]] = -2

    frags, is_dip, dip_of_frag_rows = [], [], []
    for d in range(n_dip):  # interleave dip0, ace0, dip1, ... (distancefrag.py:250-255)
        if cyx_partner[d] == -2:
            frags.append([])
            dip_of_frag_rows.append([])
        elif cyx_partner[d] >= 0:
            frags.append(dipeptides[d] + dipeptides[int(cyx_partner[d])])
            dip_of_frag_rows.append([(d, len(dipeptides[d])), (int(cyx_partner[d]), len(dipeptides[int(cyx_partner[d])]))])
        else:
            frags.append(dipeptides[d])
            dip_of_frag_rows.append([(d, len(dipeptides[d]))])
        is_dip.append(True)
        if d < n_ace:
            frags.append(acenmes[d])
            dip_of_frag_rows.append([])
            is_dip.append(False)
    sizes = np.array([len(f) for f in frags], dtype=np.int64)
    end = np.cumsum(sizes)
    start = end - sizes
    Nf = int(end[-1])
    z = np.zeros(Nf, np.int64)
    src = -np.ones(Nf, np.int64)
    acc = -np.ones(Nf, np.int64)
    tow = -np.ones(Nf, np.int64)
    length = np.zeros(Nf, np.float32)
    row = 0
    for f in frags:
        for item in f:
            if item[0] == "atom":
                src[row] = item[1]
                z[row] = p.numbers[item[1]]
            else:
                _, a, t, ln = item
                acc[row], tow[row], length[row] = a, t, ln
                z[row] = 1
            row += 1
    is_dip = np.array(is_dip)
    dip_row_start = np.zeros(n_dip, dtype=np.int64)
    dip_row_end = np.zeros(n_dip, dtype=np.int64)
    for fi, parts in enumerate(dip_of_frag_rows):
        r0 = int(start[fi])
        for d, n in parts:
            dip_row_start[d], dip_row_end[d] = r0, r0 + n
            r0 += n
    # cat[F_dip, F_ace] row order: all dipeptide rows in order, then all ACE-NME rows
    dip_rows = np.concatenate([np.arange(start[b], end[b]) for b in range(len(frags)) if is_dip[b]]).astype(np.int64)
    ace_rows = np.concatenate([np.arange(start[b], end[b]) for b in range(len(frags)) if not is_dip[b]])
    row_of_cat = np.concatenate([dip_rows, ace_rows]).astype(np.int64)
    keep = src[row_of_cat] >= 0
    select_index = np.flatnonzero(keep).astype(np.int64)
    origin_index = src[row_of_cat][keep].astype(np.int64)
    return FragmentPlan(
        n_prot=len(p), z=z, start=start, end=end, src=src, acceptor=acc, toward=tow, length=length,
        is_dipeptide=is_dip, n_dip_rows=int(len(dip_rows)), row_of_cat=row_of_cat, select_index=select_index,
        origin_index=origin_index, energy_sign=np.where(is_dip, 1.0, -1.0).astype(np.float32),
        dip_row_start=dip_row_start, dip_row_end=dip_row_end, cyx_partner=cyx_partner,
    )


_GREEK = {c: i for i, c in enumerate("ABGDEZH")}


def preprocessed_order(p: ProteinAtoms) -> ProteinAtoms:
    """The protein with the atoms of every residue in the order the reference's OWN fragmenter requires of its input:
    what its preprocessing leaves behind (Tinker `xyzpdb` output passed through `reorder_atoms`,
    /root/reference/src/utils/pdb.py:42-100) - backbone N CA C O H HA[2,3], then the side-chain heavy atoms by
    Greek position (B G D E Z H) and branch number, then the side-chain hydrogens in the same order; caps as
    `CH3 C O H1 H2 H3` / `N CH3 H HH31 HH32 HH33`.  The reference permutes its rows with tables that assume this order
    (utils/seq_dict.pkl, distancefrag.py:507-738); `build_plan` matches atoms by NAME and takes any order, so this
    is only needed to hand a protein to the reference (oracle/ref_fragmenter.py) or to write a PDB it can read.
    The rule reproduces the reference's pre-processed Chignolin example atom for atom and, on Trp-cage / WW / ABD,
    makes the reference's fragmenter agree with `amber_ordered(build_plan(.))` row for row
    (tests/test_fragmentation_and_sharding.py)."""
    def key(name):
        body = name[1:]
        return (name.startswith("H"), _GREEK.get(body[:1], 99), body[1:])

    rows = []
    for r in sorted(set(p.resnums.tolist())):
        idx = np.flatnonzero(p.resnums == r)
        nm = [str(x) for x in p.names[idx]]
        rn = str(p.resnames[idx[0]])
        if rn == "ACE":
            want = ["CH3", "C", "O", "H1", "H2", "H3"]
        elif rn == "NME":
            want = ["N", "CH3", "H", "HH31", "HH32", "HH33"]
        else:
            bb = [x for x in ("N", "CA", "C", "O", "H", "HA", "HA2", "HA3") if x in nm]
            want = bb + sorted((x for x in nm if x not in bb), key=key)
        if sorted(want) != sorted(nm):
            raise ValueError(f"residue {r} ({rn}): unexpected atom names {nm}")
        rows += [int(idx[nm.index(w)]) for w in want]
    rows = np.asarray(rows)
    return ProteinAtoms(p.names[rows], p.resnames[rows], p.resnums[rows], p.numbers[rows], p.positions[rows])


def fragment_positions(plan: FragmentPlan, prot_pos: np.ndarray) -> np.ndarray:
    """Host (numpy) evaluation of the per-step fragment geometry - used by tests and
    fixtures; the product path runs the same arithmetic in `vsn_build_fragments`."""
    pos = np.zeros((len(plan.z), 3), dtype=np.float64)
    m = plan.src >= 0
    pos[m] = prot_pos[plan.src[m]]
    c = ~m
    a = prot_pos[plan.acceptor[c]]
    v = prot_pos[plan.toward[c]] - a
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pos[c] = a + v * plan.length[c][:, None]
    return pos


def combine_host(plan: FragmentPlan, e_frag: np.ndarray, f_frag: np.ndarray):
    """numpy statement of DipeptideBondedCombiner (combiner.py:12-41) for tests."""
    nonempty = (plan.end - plan.start) > 0
    E = float((plan.energy_sign[nonempty] * np.asarray(e_frag).reshape(-1)).sum())
    sign = np.where(np.arange(len(plan.row_of_cat)) < plan.n_dip_rows, 1.0, -1.0)
    cat = f_frag[plan.row_of_cat] * sign[:, None]
    F = np.zeros((plan.n_prot, 3), dtype=np.float64)
    np.add.at(F, plan.origin_index, cat[plan.select_index])
    return E, F
