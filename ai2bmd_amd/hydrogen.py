"""Cap-hydrogen relaxation plan (SURVEY.md 8f "next #1").

Every MD step the reference relaxes the hydrogens it added when cutting dipeptides out of the chain with
<= 10 L-BFGS iterations on an AMBER force field restricted to the terms that touch those hydrogens
(/root/reference/src/Fragmentation/distancefrag.py:56-92 `get_fragments`,
 /root/reference/src/Fragmentation/hydrogen/energies.py:8-61,211-242,
 /root/reference/src/Fragmentation/hydrogen/ctable.py:168-244 term filters,
 /root/reference/src/Fragmentation/hydrogen/topology.py:20-130 batching).

This module builds, once per simulation, the flat term lists the HIP optimiser (`csrc/hydrogen.hip`)
consumes:  which AMBER template atom every dipeptide row corresponds to (by atom NAME - the reference
instead permutes rows into AMBER order with utils/seq_dict.pkl), the bond / angle / dihedral / non-bonded
terms that involve a cap hydrogen with their parameters resolved, and, per cap hydrogen, the list of term
occurrences (so gradients are gathered per atom in a fixed order, no atomics).
ACE-NME fragments copy their atoms from the relaxed dipeptides (`alias`): ACE-NME k = the acetyl-like part of
dipeptide k+1 + the N-methyl-amide-like part of dipeptide k (distancefrag.py:291-302).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .amber import TOPOLOGY_OF
from .fragmentation import FragmentPlan, ProteinAtoms

SCNB, SCEE = 1.2, 2.0  # HydrogenOptimizer defaults (energies.py:76-81); applied to ALL pairs there


@dataclass
class HydrogenPlan:
    cap_rows: np.ndarray    # int64 [ncap] rows of the fragment batch that are optimised (dipeptide cap H)
    alias: np.ndarray       # int64 [Nf] for ACE-NME rows: dipeptide row to copy from; -1 for dipeptide rows
    tmpl_index: list        # per dipeptide: template atom index of every row
    # terms (atom entries are fragment-batch rows)
    bond: dict              # i, j, k, r0
    angle: dict             # i, j, k, kf, th0
    dihedral: dict          # i, j, k, l, kf, per, phase
    pair: dict              # i, j, A, B, qq
    occ_ptr: np.ndarray     # int32 [ncap+1]
    occ_type: np.ndarray    # int32 : 0 bond, 1 angle, 2 dihedral, 3 pair
    occ_term: np.ndarray    # int32 index into that type's arrays
    occ_end: np.ndarray     # int32 : 0 = the cap atom is the FIRST atom of the term, 1 = the LAST
    occ_w: np.ndarray       # float32 energy share of this occurrence (1 / number of cap atoms in the term)


def template_index_of_dipeptide(p: ProteinAtoms, plan: FragmentPlan, d: int, names) -> np.ndarray:
    """Template atom index (within `names`, the atom names of one ACE-X-NME template) of every row of the
    original dipeptide d, matching by atom name."""
    names = [str(n) for n in names]
    nat = len(names)
    rows = np.arange(plan.dip_row_start[d], plan.dip_row_end[d])
    if len(rows) != nat:
        raise ValueError(f"dipeptide {d}: {len(rows)} atoms, template has {nat}")
    src = plan.src[rows]
    b = d
    r = d + 2  # central residue number (1-based, ACE = 1)
    out = -np.ones(nat, dtype=np.int64)
    resnum = np.where(src >= 0, p.resnums[np.maximum(src, 0)], 0)
    # cap hydrogens belong to the residue of their acceptor
    capres = np.where(src < 0, p.resnums[np.maximum(plan.acceptor[rows], 0)], 0)
    res_of_row = np.where(src >= 0, resnum, capres)
    ace_names, nme_names = names[:6], names[-6:]
    mid_names = names[6:-6]

    def name_of(k):
        return str(p.names[src[k]]) if src[k] >= 0 else "cap"

    prev = [k for k in range(nat) if res_of_row[k] == r - 1]
    own = [k for k in range(nat) if res_of_row[k] == r]
    nxt = [k for k in range(nat) if res_of_row[k] == r + 1]
    assert len(prev) == 6 and len(nxt) == 6 and len(own) == nat - 12, (b, len(prev), len(own), len(nxt))
    # --- acetyl-like part: template H1 CH3 H2 H3 C O
    if str(p.resnames[np.flatnonzero(p.resnums == r - 1)[0]]) == "ACE":
        for k in prev:
            out[k] = ace_names.index(name_of(k))
    else:
        # H1, then (H3, H2) - for the C-terminal dipeptide (H2, H3): the slot assignment the reference's permutation
        # ends up with (distancefrag.py:570-620 puts the two new hydrogens in reverse order except at the C-terminus)
        hs = [0, 2, 3] if d == len(plan.dip_row_start) - 1 else [0, 3, 2]
        for k in prev:
            nm = name_of(k)
            if nm == "CA":
                out[k] = 1
            elif nm == "C":
                out[k] = 4
            elif nm == "O":
                out[k] = 5
            else:  # HA / HA2 / HA3 / cap -> the three equivalent methyl hydrogens (same type and charge)
                out[k] = hs.pop(0)
    # --- the residue itself
    for k in own:
        nm = name_of(k)
        if nm not in mid_names:
            raise ValueError(f"atom {nm} of residue {r} not in the AMBER template {mid_names}")
        out[k] = 6 + mid_names.index(nm)
    # --- N-methyl-amide-like part: template N H CH3 HH31 HH32 HH33
    base = nat - 6
    if str(p.resnames[np.flatnonzero(p.resnums == r + 1)[0]]) == "NME":
        for k in nxt:
            out[k] = base + nme_names.index(name_of(k))
    else:
        hs = [3, 4, 5]
        first_h_done = False
        for k in nxt:
            nm = name_of(k)
            if nm == "N":
                out[k] = base + 0
            elif nm == "CA":
                out[k] = base + 2
            elif nm == "H" or (nm == "cap" and not first_h_done and plan.acceptor[rows[k]] >= 0
                               and str(p.names[plan.acceptor[rows[k]]]) == "N"):
                out[k] = base + 1  # amide H (for PRO: the cap placed on N)
                first_h_done = True
            else:
                out[k] = base + hs.pop(0)
    if sorted(out.tolist()) != list(range(nat)):
        raise ValueError(f"dipeptide {b}: name matching is not a bijection")
    return out


def template_slots(p: ProteinAtoms, plan: FragmentPlan, tables: dict) -> list:
    """Per dipeptide FRAGMENT (slot d of the interleaved batch = fragment 2d): the AMBER template atom index of every
    row.  A plan whose rows already are in AMBER order carries them (`plan.tmpl_slot`); otherwise atoms are matched to
    the template by NAME.  Empty for a CYX dipeptide merged into its partner's fragment."""
    resname_of = {int(r): str(p.resnames[np.flatnonzero(p.resnums == r)[0]]) for r in set(p.resnums.tolist())}
    out = []
    for b in range(0, len(plan.start), 2):
        d = b // 2
        if plan.cyx_partner[d] == -2:  # merged into its disulfide partner's fragment
            out.append(np.zeros(0, dtype=np.int64))
            continue
        if plan.tmpl_slot is not None:
            out.append(np.asarray(plan.tmpl_slot[plan.start[b]:plan.end[b]], dtype=np.int64))
            continue
        t = tables[TOPOLOGY_OF[resname_of[d + 2]]]
        if plan.cyx_partner[d] >= 0:
            # CYX pair: one 44-atom AMBER topology = two ACE-CYX-NME halves bridged by the S-S bond
            # (utils/reference.py:41,71; distancefrag.py:185-240); halves in the order (this, partner)
            half = t["natom"] // 2
            nm = t["atom_names"]
            out.append(np.concatenate([template_index_of_dipeptide(p, plan, d, nm[:half]),
                                       half + template_index_of_dipeptide(p, plan, int(plan.cyx_partner[d]), nm[half:])]))
        else:
            out.append(template_index_of_dipeptide(p, plan, d, t["atom_names"]))
    return out


def acenme_alias(plan: FragmentPlan) -> np.ndarray:
    """int64 [Nf]: for every ACE-NME row the dipeptide row that carries the same atom (-1 on dipeptide rows).
    ACE-NME k = acetyl-like part of dipeptide k+1 + N-methyl-amide-like part of dipeptide k (distancefrag.py:291-302)."""
    alias = -np.ones(len(plan.z), dtype=np.int64)
    for b in range(1, len(plan.start), 2):
        k = b // 2
        rows = np.arange(plan.start[b], plan.end[b])
        dn = np.arange(plan.dip_row_start[k + 1], plan.dip_row_end[k + 1])  # dipeptide k+1
        dp = np.arange(plan.dip_row_start[k], plan.dip_row_end[k])          # dipeptide k

        def key(rw):
            return (int(plan.src[rw]), int(plan.acceptor[rw]), int(plan.toward[rw]))

        lut_n = {key(rw): rw for rw in dn}
        lut_p = {key(rw): rw for rw in dp}
        for j, rw in enumerate(rows):  # rows 0..5: acetyl-like part, 6..11: amide-like part (both row orders)
            alias[rw] = (lut_n if j < 6 else lut_p)[key(rw)]
    return alias


def build_hydrogen_plan(p: ProteinAtoms, plan: FragmentPlan, tables: dict) -> HydrogenPlan:
    Nf = len(plan.z)
    B = len(plan.start)
    resname_of = {int(r): str(p.resnames[np.flatnonzero(p.resnums == r)[0]]) for r in set(p.resnums.tolist())}
    bond = dict(i=[], j=[], kf=[], r0=[])
    angle = dict(i=[], j=[], k=[], kf=[], th0=[])
    dih = dict(i=[], j=[], k=[], l=[], kf=[], per=[], phase=[])
    pair = dict(i=[], j=[], A=[], B=[], qq=[])
    tmpl_index = []
    cap_rows = []
    occ = {}  # cap row -> list of (type, term, end, ncap_in_term)

    def add_occ(row, typ, term, end, ncap):
        occ.setdefault(int(row), []).append((typ, term, end, ncap))

    slots = template_slots(p, plan, tables)
    for b in range(0, B, 2):
        d = b // 2
        ti = slots[d]
        if plan.cyx_partner[d] == -2:  # merged into its disulfide partner's fragment
            tmpl_index.append(ti)
            continue
        t = tables[TOPOLOGY_OF[resname_of[d + 2]]]
        tmpl_index.append(ti)
        rows = np.arange(plan.start[b], plan.end[b])
        assert len(rows) == len(ti)
        row_of_tmpl = np.empty(len(rows), dtype=np.int64)
        row_of_tmpl[ti] = rows
        capmask_t = np.zeros(len(rows), dtype=bool)
        capmask_t[ti[plan.src[rows] < 0]] = True  # template indices that are cap hydrogens
        cap_rows.extend(rows[plan.src[rows] < 0].tolist())
        # bonds / angles / dihedrals that touch a cap hydrogen (ctable.py:168-199)
        for a, c, idx in t["bonds_inc_hydrogen"]:
            if capmask_t[a] or capmask_t[c]:
                term = len(bond["i"])
                bond["i"].append(row_of_tmpl[a]); bond["j"].append(row_of_tmpl[c])
                bond["kf"].append(t["bond_force_constant"][idx]); bond["r0"].append(t["bond_equil_value"][idx])
                n = int(capmask_t[a]) + int(capmask_t[c])
                if capmask_t[a]:
                    add_occ(row_of_tmpl[a], 0, term, 0, n)
                if capmask_t[c]:
                    add_occ(row_of_tmpl[c], 0, term, 1, n)
        for a, m, c, idx in t["angles_inc_hydrogen"]:
            if capmask_t[a] or capmask_t[m] or capmask_t[c]:
                assert not capmask_t[m], "a cap hydrogen cannot be the apex of an angle"
                term = len(angle["i"])
                angle["i"].append(row_of_tmpl[a]); angle["j"].append(row_of_tmpl[m]); angle["k"].append(row_of_tmpl[c])
                angle["kf"].append(t["angle_force_constant"][idx]); angle["th0"].append(t["angle_equil_value"][idx])
                n = int(capmask_t[a]) + int(capmask_t[c])
                if capmask_t[a]:
                    add_occ(row_of_tmpl[a], 1, term, 0, n)
                if capmask_t[c]:
                    add_occ(row_of_tmpl[c], 1, term, 1, n)
        for a, m1, m2, c, idx in t["dihedrals_inc_hydrogen"]:
            if m2 < 0 or c < 0:  # ctable.py:198: improper / multi-term markers are dropped
                continue
            if capmask_t[a] or capmask_t[m1] or capmask_t[m2] or capmask_t[c]:
                assert not capmask_t[m1] and not capmask_t[m2]
                term = len(dih["i"])
                dih["i"].append(row_of_tmpl[a]); dih["j"].append(row_of_tmpl[m1])
                dih["k"].append(row_of_tmpl[m2]); dih["l"].append(row_of_tmpl[c])
                dih["kf"].append(t["dihedral_force_constant"][idx]); dih["per"].append(t["dihedral_periodicity"][idx])
                dih["phase"].append(t["dihedral_phase"][idx])
                n = int(capmask_t[a]) + int(capmask_t[c])
                if capmask_t[a]:
                    add_occ(row_of_tmpl[a], 2, term, 0, n)
                if capmask_t[c]:
                    add_occ(row_of_tmpl[c], 2, term, 1, n)
        # non-bonded pairs i < j with a cap hydrogen, minus AMBER's excluded list (ctable.py:201-230)
        nat = t["natom"]
        ptr = np.concatenate([[0], np.cumsum(t["number_excluded_atoms"])])
        excluded = set()
        for a in range(nat):
            for c in t["excluded_atoms_list"][ptr[a]:ptr[a + 1]]:
                excluded.add((a, int(c)))
        for a in range(nat):
            for c in range(a + 1, nat):
                if not (capmask_t[a] or capmask_t[c]) or (a, c) in excluded:
                    continue
                li = t["nonbonded_parm_index"][t["ntypes"] * t["atom_type_idx"][a] + t["atom_type_idx"][c]]
                term = len(pair["i"])
                pair["i"].append(row_of_tmpl[a]); pair["j"].append(row_of_tmpl[c])
                pair["A"].append(t["lennard_jones_acoef"][li]); pair["B"].append(t["lennard_jones_bcoef"][li])
                pair["qq"].append(t["charge"][a] * t["charge"][c])
                n = int(capmask_t[a]) + int(capmask_t[c])
                if capmask_t[a]:
                    add_occ(row_of_tmpl[a], 3, term, 0, n)
                if capmask_t[c]:
                    add_occ(row_of_tmpl[c], 3, term, 1, n)

    cap_rows = np.asarray(cap_rows, dtype=np.int64)
    occ_ptr = [0]
    ot, oterm, oend, ow = [], [], [], []
    for row in cap_rows:
        for typ, term, end, n in occ.get(int(row), []):
            ot.append(typ); oterm.append(term); oend.append(end); ow.append(1.0 / n)
        occ_ptr.append(len(ot))

    alias = acenme_alias(plan)  # ACE-NME rows alias the dipeptide rows they are cut from

    index_keys = ("i", "j", "k", "l")  # atom rows; every other key is a float parameter (force constants: "kf")
    f32 = lambda d_: {k_: np.asarray(v, dtype=np.int32 if k_ in index_keys else np.float32) for k_, v in d_.items()}
    return HydrogenPlan(
        cap_rows=cap_rows, alias=alias, tmpl_index=tmpl_index,
        bond=f32(bond), angle=f32(angle), dihedral=f32(dih), pair=f32(pair),
        occ_ptr=np.asarray(occ_ptr, np.int32), occ_type=np.asarray(ot, np.int32),
        occ_term=np.asarray(oterm, np.int32), occ_end=np.asarray(oend, np.int32), occ_w=np.asarray(ow, np.float32),
    )


def amber_ordered(p: ProteinAtoms, plan: FragmentPlan, tables: dict) -> FragmentPlan:
    """The same plan with the rows of every fragment in the REFERENCE's order: dipeptides in the atom order of their
    AMBER topology (what utils/seq_dict.pkl produces, distancefrag.py:728-737), ACE-NME fragments as the first six
    atoms of the next dipeptide followed by the last six of the previous one (distancefrag.py:291-302), and the
    recombination pairs listed like the reference lists them (fragment by fragment, protein atoms ascending,
    distancefrag.py:338-350,740-802).  A FragmentData built from it is row for row what
    `DistanceFragment.get_fragments` returns (tests/test_fragmentation_and_sharding.py, against the reference's own
    fragmenter).  This is what `fragmentation.build_plan` returns by default."""
    import dataclasses

    if plan.tmpl_slot is not None:
        return plan
    slots = template_slots(p, plan, tables)
    alias = acenme_alias(plan)
    Nf = len(plan.z)
    new_of_old = np.arange(Nf, dtype=np.int64)
    tmpl_slot = -np.ones(Nf, dtype=np.int64)
    for b in range(0, len(plan.start), 2):
        a0, ti = int(plan.start[b]), slots[b // 2]
        if len(ti):
            new_of_old[a0:a0 + len(ti)] = a0 + ti
            tmpl_slot[a0:a0 + len(ti)] = np.arange(len(ti))
    nat_of = plan.dip_row_end - plan.dip_row_start
    for b in range(1, len(plan.start), 2):
        a0 = int(plan.start[b])
        for j in range(12):
            src_row = int(alias[a0 + j])              # the dipeptide row this ACE-NME row is a copy of
            d = int(np.flatnonzero((plan.dip_row_start <= src_row) & (src_row < plan.dip_row_end))[0])
            # template slot of that row inside ITS dipeptide half (CYX pairs: second half is offset by 22)
            frag = 2 * (d if plan.cyx_partner[d] != -2 else int(np.flatnonzero(plan.cyx_partner == d)[0]))
            slot = int(slots[frag // 2][src_row - int(plan.start[frag])])
            half0 = 0 if plan.cyx_partner[d] != -2 else int(nat_of[frag // 2])
            slot -= half0
            new_of_old[a0 + j] = a0 + (slot if j < 6 else 6 + slot - (int(nat_of[d]) - 6))
    assert sorted(new_of_old.tolist()) == list(range(Nf))
    old_of_new = np.argsort(new_of_old)
    fields = {k: getattr(plan, k)[old_of_new] for k in ("z", "src", "acceptor", "toward", "length")}
    # cat[F_dip, F_ace] keeps the same fragment order; inside a fragment the rows follow the new order
    is_dip_row = np.repeat(plan.is_dipeptide, plan.end - plan.start)
    rows = np.arange(Nf)
    row_of_cat = np.concatenate([rows[is_dip_row], rows[~is_dip_row]]).astype(np.int64)
    cat_src = fields["src"][row_of_cat]
    frag_of_cat = np.repeat(np.arange(len(plan.start)), plan.end - plan.start)[row_of_cat]
    sel = np.flatnonzero(cat_src >= 0)
    # fragment order of cat[...] first, protein atom second (a protein atom occurs once per fragment, except the atoms
    # the two halves of a CYX pair share: those keep their row order)
    cat_pos_of_frag = np.empty(len(plan.start), dtype=np.int64)
    cat_pos_of_frag[np.concatenate([np.flatnonzero(plan.is_dipeptide), np.flatnonzero(~plan.is_dipeptide)])] = \
        np.arange(len(plan.start))
    sel = sel[np.lexsort((sel, cat_src[sel], cat_pos_of_frag[frag_of_cat[sel]]))]
    # original dipeptide row ranges move with their rows only through the in-fragment permutation (ranges unchanged)
    return dataclasses.replace(plan, **fields, row_of_cat=row_of_cat, select_index=sel.astype(np.int64),
                               origin_index=cat_src[sel].astype(np.int64), tmpl_slot=tmpl_slot)


class HydrogenRelaxer:
    """Device handle of the HIP optimiser (`vsn_hopt_*`, csrc/hydrogen.hip) for one HydrogenPlan."""

    def __init__(self, hplan: HydrogenPlan, n_rows: int, device_index: int = 0, max_iter: int = 10):
        import ctypes as C

        from . import capi

        self._C, self._L = C, capi.lib()
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        keep = []

        def P(arr, ctype):
            keep.append(arr)
            return arr.ctypes.data_as(C.POINTER(ctype))

        b, a, d, p = hplan.bond, hplan.angle, hplan.dihedral, hplan.pair
        t = capi.VsnHoptTerms()
        t.n_rows, t.n_cap = n_rows, len(hplan.cap_rows)
        t.cap_rows, t.alias = P(i64(hplan.cap_rows), C.c_int64), P(i64(hplan.alias), C.c_int64)
        t.n_bond = len(b["i"])
        t.bond_i, t.bond_j = P(i32(b["i"]), C.c_int32), P(i32(b["j"]), C.c_int32)
        t.bond_k, t.bond_r0 = P(f32(b["kf"]), C.c_float), P(f32(b["r0"]), C.c_float)
        t.n_angle = len(a["i"])
        t.angle_i, t.angle_j, t.angle_k = (P(i32(a[k]), C.c_int32) for k in "ijk")
        t.angle_kf, t.angle_th0 = P(f32(a["kf"]), C.c_float), P(f32(a["th0"]), C.c_float)
        t.n_dihedral = len(d["i"])
        t.dih_i, t.dih_j, t.dih_k, t.dih_l = (P(i32(d[k]), C.c_int32) for k in "ijkl")
        t.dih_kf, t.dih_per, t.dih_phase = (P(f32(d[k]), C.c_float) for k in ("kf", "per", "phase"))
        t.n_pair = len(p["i"])
        t.pair_i, t.pair_j = P(i32(p["i"]), C.c_int32), P(i32(p["j"]), C.c_int32)
        t.pair_a, t.pair_b, t.pair_qq = (P(f32(p[k]), C.c_float) for k in ("A", "B", "qq"))
        t.occ_ptr, t.occ_type = P(i32(hplan.occ_ptr), C.c_int32), P(i32(hplan.occ_type), C.c_int32)
        t.occ_term, t.occ_end = P(i32(hplan.occ_term), C.c_int32), P(i32(hplan.occ_end), C.c_int32)
        t.occ_w = P(f32(hplan.occ_w), C.c_float)
        t.max_iter, t.lr, t.tolerance_grad, t.tolerance_change = max_iter, 0.1, 0.1, 0.01
        t.scnb, t.scee = SCNB, SCEE
        self._h = C.c_void_p()
        rc = self._L.vsn_hopt_create(C.byref(self._h), device_index, C.byref(t))
        if rc:
            raise RuntimeError(f"vsn_hopt_create failed ({rc})")
        self.n_rows = n_rows

    def run(self, frag_pos, stream=None):
        """frag_pos: float32 [n_rows,3] device tensor, relaxed in place."""
        import torch

        C = self._C
        assert frag_pos.is_cuda and frag_pos.dtype == torch.float32 and frag_pos.is_contiguous()
        assert frag_pos.shape[0] == self.n_rows
        st = stream if stream is not None else torch.cuda.current_stream(frag_pos.device)
        rc = self._L.vsn_hopt_run(self._h, C.c_void_p(frag_pos.data_ptr()), C.c_void_p(st.cuda_stream))
        if rc:
            raise RuntimeError(f"vsn_hopt_run failed ({rc})")

    def stats(self, stream=None):
        import torch

        C = self._C
        st = stream if stream is not None else torch.cuda.current_stream()
        ie, ll = (C.c_int32 * 2)(), (C.c_double * 2)()
        rc = self._L.vsn_hopt_stats(self._h, ie, ll, C.c_void_p(st.cuda_stream))
        if rc:
            raise RuntimeError(f"vsn_hopt_stats failed ({rc})")
        return dict(iterations=ie[0], evaluations=ie[1], loss_first=ll[0], loss_last=ll[1])

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.vsn_hopt_destroy(self._h)
                self._h = None
        except Exception:
            pass
