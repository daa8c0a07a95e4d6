"""MM non-bonded calculator on the MI355X (SURVEY.md 8f "next #2").

Mirror of the reference's `MMNonBondedCalculator` (/root/reference/src/Calculators/nonbonded.py:9-63):
`set_parameters(prot)` takes the per-atom OpenMM parameters (`prot.charges`, `prot.sigmas`,
`prot.epsilons`, AIMD/protein.py:153-175 - injected arrays here, OpenMM is not installed) and
`__call__(prot) -> (energy, forces)` returns numpy values in ASE units.  Instead of materialising the
O(N^2) pair list (protein.py:133-151) the kernel tests dipeptide co-membership on the fly.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi
from .fragmentation import FragmentPlan


def dipeptide_groups(plan: FragmentPlan) -> np.ndarray:
    """int32 [n_prot, 4]: ids of the dipeptides every protein atom belongs to (-1 padded);
    two atoms are excluded from the MM term iff their id sets intersect (distancefrag.py:355-363)."""
    g = -np.ones((plan.n_prot, 4), dtype=np.int32)
    fill = np.zeros(plan.n_prot, dtype=np.int64)
    dip = 0
    for b in range(len(plan.start)):
        if not plan.is_dipeptide[b]:
            continue
        atoms = plan.src[plan.start[b]:plan.end[b]]
        for a in atoms[atoms >= 0]:
            if fill[a] >= 4:
                raise ValueError("an atom belongs to more than 4 dipeptides")
            g[a, fill[a]] = dip
            fill[a] += 1
        dip += 1
    return g


class MMNonBondedCalculator:
    def __init__(self, device="cuda:0") -> None:
        if not str(device).startswith("cuda"):
            raise RuntimeError("MMNonBondedCalculator: this package is the MI355X path ('cuda:<k>' devices only)")
        self.device = device
        self._L = capi.lib()
        self._h = None
        self.n = 0

    def set_parameters(self, prot, plan: FragmentPlan = None) -> None:
        """`prot.charges / sigmas / epsilons` like the reference (nonbonded.py:24-31).  The pair exclusions come from
        the fragment plan DistanceFragment.fragment(prot) left on `prot` (the reference reads prot.exclude_pair through
        prot.initial_mm_adjmatrix(), protein.py:133-151); `plan` overrides it."""
        if plan is None:
            plan = getattr(prot, "_vsn_plan", None)
            if plan is None:
                raise RuntimeError("MMNonBondedCalculator.set_parameters: call DistanceFragment.fragment(prot) first "
                                   "(simulator.py:53-57 does), or pass the FragmentPlan")
        q = np.ascontiguousarray(prot.charges, dtype=np.float32)
        s = np.ascontiguousarray(prot.sigmas, dtype=np.float32)
        e = np.ascontiguousarray(prot.epsilons, dtype=np.float32)
        g = np.ascontiguousarray(dipeptide_groups(plan), dtype=np.int32)
        self.n = len(q)
        fp = C.POINTER(C.c_float)
        self._h = C.c_void_p()
        rc = self._L.vsn_mm_create(C.byref(self._h), torch.device(self.device).index or 0, self.n,
                                   q.ctypes.data_as(fp), s.ctypes.data_as(fp), e.ctypes.data_as(fp),
                                   g.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise RuntimeError(f"vsn_mm_create failed ({rc})")
        self._e = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._f = torch.zeros(self.n, 3, dtype=torch.float32, device=self.device)

    def forces_device(self, pos: torch.Tensor, f_out: torch.Tensor = None, accumulate: bool = False):
        """pos [n,3] device tensor -> (E 1-element tensor, F [n,3]); asynchronous on the current stream."""
        f = self._f if f_out is None else f_out
        st = torch.cuda.current_stream(self.device)
        rc = self._L.vsn_mm_forces(self._h, C.c_void_p(pos.data_ptr()), C.c_void_p(self._e.data_ptr()),
                                   C.c_void_p(f.data_ptr()), 1 if accumulate else 0, C.c_void_p(st.cuda_stream))
        if rc:
            raise RuntimeError(f"vsn_mm_forces failed ({rc})")
        return self._e, f

    def __call__(self, prot):
        xyz = prot.get_positions() if hasattr(prot, "get_positions") else prot.positions
        pos = torch.as_tensor(np.ascontiguousarray(xyz, dtype=np.float32)).to(self.device)
        e, f = self.forces_device(pos)
        return float(e.cpu()[0]), f.cpu().numpy()

    def __del__(self):
        try:
            if self._h:
                self._L.vsn_mm_destroy(self._h)
                self._h = None
        except Exception:
            pass
