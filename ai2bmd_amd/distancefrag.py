"""Per-step fragment producer: mirror of the reference's `DistanceFragment`
(/root/reference/src/Fragmentation/distancefrag.py:18-363) for the calls the calculators make:

    fragment(prot)       once      :94-363   index algebra; leaves fragments_z / fragments_start / fragments_end /
                                             fragments_batch / select_index / origin_index on `prot`
    get_fragments(prot)  per step  :56-92    cap-hydrogen placement (:35-54) + HydrogenOptimizer (hydrogen/energies.py:
                                             211-242) on the dipeptides, ACE-NME rows cut out of the relaxed dipeptides,
                                             returns the interleaved FragmentData

The index algebra is ai2bmd_amd.fragmentation.build_plan (atoms matched by NAME, any atom order), placement and
relaxation run on the GPU (`vsn_build_fragments`, `vsn_hopt_run`); the FragmentData handed back is host numpy like
the reference's, so the reference's DLBondedCalculator code drives it unchanged.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .amber import default_tables  # noqa: F401  (re-exported: callers import it from here)
from .fragment import FragmentData, make_batch_index
from .fragmentation import ProteinAtoms, build_plan
from .hydrogen import HydrogenRelaxer, build_hydrogen_plan

def as_protein_atoms(prot) -> ProteinAtoms:
    """`prot`: a ProteinAtoms, or an ase.Atoms-like object read from a PDB (arrays 'atomtypes', 'residuenames',
    'residuenumbers' as ase.io.read gives them - what AIMD/protein.py:15 wraps)."""
    if isinstance(prot, ProteinAtoms):
        return prot
    arr = prot.arrays
    return ProteinAtoms(names=np.asarray([str(s).strip() for s in arr["atomtypes"]]),
                        resnames=np.asarray([str(s).strip() for s in arr["residuenames"]]),
                        resnums=np.asarray(arr["residuenumbers"], dtype=np.int64),
                        numbers=np.asarray(arr["numbers"], dtype=np.int64),
                        positions=np.asarray(arr["positions"], dtype=np.float64))


def _positions(prot):
    if hasattr(prot, "arrays") and "positions" in getattr(prot, "arrays"):
        return np.asarray(prot.arrays["positions"])
    return np.asarray(prot.positions)


class DistanceFragment:
    def __init__(self, max_iter: int = 10, tables=None, device: str = None, relax: bool = True) -> None:
        self.max_iter, self.tables, self.device, self.relax = max_iter, tables, device, relax
        self.plan = self.hplan = self.relaxer = self._fp = None

    def fragment(self, prot) -> None:
        from .device_strategy import DeviceStrategy

        p = as_protein_atoms(prot)
        plan = build_plan(p, tables=self.tables)
        self._release()        # a second protein on the same instance: drop the previous plan's device state
        self.plan = plan
        prot.fragments_z = plan.z
        prot.fragments_start, prot.fragments_end = plan.start, plan.end
        prot.fragments_batch = make_batch_index(plan.start, plan.end)
        prot._vsn_plan = plan  # read by MMNonBondedCalculator.set_parameters (the reference leaves prot.exclude_pair)
        dev = self.device
        if dev is None:
            try:
                dev = DeviceStrategy.get_optimiser_device() if DeviceStrategy._bonded_devices else "cuda:0"
            except Exception:
                dev = "cuda:0"
        self.device = dev
        # like the reference (distancefrag.py:347-353): the recombination indices are torch tensors on the default
        # device - what the reference's own DipeptideBondedCombiner (torch_scatter) is handed by its DLBondedCalculator
        try:
            idx_dev = DeviceStrategy.get_default_device() if DeviceStrategy._bonded_devices else dev
        except Exception:
            idx_dev = dev
        prot.select_index = torch.as_tensor(np.asarray(plan.select_index, dtype=np.int64)).to(idx_dev)
        prot.origin_index = torch.as_tensor(np.asarray(plan.origin_index, dtype=np.int64)).to(idx_dev)
        idx = torch.device(dev).index or 0
        L = capi.lib()
        self._L = L
        self._fp = C.c_void_p()
        rc = L.vsn_fragplan_create(C.byref(self._fp), idx, len(plan.z), capi.i64_ptr(np.ascontiguousarray(plan.src)),
                                   capi.i64_ptr(np.ascontiguousarray(plan.acceptor)),
                                   capi.i64_ptr(np.ascontiguousarray(plan.toward)),
                                   np.ascontiguousarray(plan.length, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(f"vsn_fragplan_create failed ({rc})")
        if self.relax:
            self.hplan = build_hydrogen_plan(p, plan, self.tables if self.tables is not None else default_tables())
            self.relaxer = HydrogenRelaxer(self.hplan, len(plan.z), idx, max_iter=self.max_iter)
        self._pos = torch.empty(len(plan.z), 3, dtype=torch.float32, device=dev)

    def get_fragments(self, prot) -> FragmentData:
        if self.plan is None:
            raise RuntimeError("DistanceFragment.fragment(prot) has not been called")
        st = torch.cuda.current_stream(self.device)
        pos = np.asarray(_positions(prot))
        io = self.__dict__.get("_io")
        if io is None or io[0].shape[0] != len(pos) or io[2].shape != self._pos.shape or io[1].device != self._pos.device:
            # persistent staging (host positions in, fragment positions out, every MD step): pinned buffers, one
            # device copy of the protein, asynchronous copies on the launch stream, ONE synchronisation
            io = self._io = (torch.empty(len(pos), 3, dtype=torch.float32).pin_memory(),
                             torch.empty(len(pos), 3, dtype=torch.float32, device=self.device),
                             torch.empty(self._pos.shape, dtype=torch.float32).pin_memory())
        pin_x, dev_x, pin_out = io
        pin_x.numpy()[...] = pos
        dev_x.copy_(pin_x, non_blocking=True)
        rc = self._L.vsn_build_fragments(self._fp, C.c_void_p(dev_x.data_ptr()), C.c_void_p(self._pos.data_ptr()),
                                         C.c_void_p(st.cuda_stream))
        if rc:
            raise RuntimeError(f"vsn_build_fragments failed ({rc})")
        if self.relaxer is not None:
            self.relaxer.run(self._pos, st)
        pin_out.copy_(self._pos, non_blocking=True)
        st.synchronize()
        return FragmentData(prot.fragments_z, pin_out.numpy().copy(), prot.fragments_start, prot.fragments_end,
                            prot.fragments_batch)

    def _release(self):
        if getattr(self, "_fp", None):
            self._L.vsn_fragplan_destroy(self._fp)
        self._fp = self._io = self.hplan = self.relaxer = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
