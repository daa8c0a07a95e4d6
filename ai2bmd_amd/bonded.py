"""Fragment-batch force field: the caller side of the ViSNet seam.

Mirrors the reference's DLBondedCalculator (/root/reference/src/Calculators/bonded.py:18-123):
fragments -> contiguous atom-balanced partitions (device_strategy.py:84-127) -> one
model per device -> concatenate -> split dipeptide/ACE-NME -> combine
(combiner.py:12-41).  Two hosts are provided:

* `DLBondedCalculator` - the reference's in-process shape: same constructor, `calculate`
  and `__call__(prot)`; one Python thread per device handle (bonded.py:75-77), host numpy
  in/out, same return tuple.  The
  reference reaches GPUs >= 2 through pickle-over-socket worker processes
  (visnet_calculator.py:78-118); here every device is an in-process handle.
* `ShardedFragmentForces` - the MI355X-native shape: one process per GPU
  (torch.distributed, backend "nccl" = RCCL over xGMI), positions stay in HBM,
  each rank evaluates its contiguous fragment range, ONE padded all-gather per
  step moves the shard forces+energies (a few KB: latency-bound, so a single
  fused buffer), then every rank runs the same deterministic combine.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import capi
from .device_strategy import DEFAULT_CHUNK_ATOMS, DeviceStrategy, device_ranges, work_partitions
from .fragment import FragmentData
from .fragmentation import FragmentPlan


class DipeptideBondedCombiner:
    """numpy mirror of Calculators/combiner.py:5-41."""

    @staticmethod
    def energy_combine(dipeptides_energy, ACE_NMEs_energy):
        return np.float32(np.asarray(dipeptides_energy).sum() - np.asarray(ACE_NMEs_energy).sum())

    def __init__(self):
        self._idx_host = {}   # per combiner: (data_ptr, version, length) of an index tensor -> its host copy

    def _index(self, idx):
        """recombination indices as host numpy: the fragmenter leaves torch tensors on the default device like the
        reference's (distancefrag.py:347-353); they are fixed for a simulation, so the host copy is made once per
        (storage, in-place version) and kept by THIS combiner only"""
        if isinstance(idx, np.ndarray):
            return idx
        if not torch.is_tensor(idx):
            return np.asarray(idx)
        key = (idx.device, idx.data_ptr(), idx._version, tuple(idx.shape))
        hit = self._idx_host.get(key)
        if hit is None:
            if len(self._idx_host) >= 4:  # select + origin of one protein; a new protein replaces them
                self._idx_host.clear()
            hit = self._idx_host[key] = idx.detach().cpu().numpy()
        return hit

    def forces_combine(self, prot_len, dipeptides_forces, ACE_NMEs_forces, select_index, origin_index):
        cat = np.concatenate([np.asarray(dipeptides_forces), -np.asarray(ACE_NMEs_forces)])[self._index(select_index)]
        oi = self._index(origin_index)
        # scatter_sum(cat, origin_index) (combiner.py:38-39) as three bincounts: rows are added in index order like
        # np.add.at, in float64 (np.add.at is an order of magnitude slower per call and this runs every MD step)
        return np.stack([np.bincount(oi, weights=cat[:, k], minlength=prot_len) for k in range(3)], 1).astype(np.float32)


class DLBondedCalculator:
    r"""Mirror of the reference's DLBondedCalculator (bonded.py:18-123): same constructor
    (`ckpt_path, ckpt_type`), `calculate(fragments)` and `__call__(prot)`; devices and work partitions come from
    `DeviceStrategy`, the models from `get_visnet_model` (one in-process handle per 'cuda:k')."""

    def __init__(self, ckpt_path: str, ckpt_type: str, **kwargs) -> None:
        import os.path as osp

        from .distancefrag import DistanceFragment
        from .visnet_calculator import get_visnet_model

        self.ckpt_path, self.ckpt_type = ckpt_path, ckpt_type
        self.fragment_method = DistanceFragment()
        self.combiner = DipeptideBondedCombiner()
        model_path = osp.join(self.ckpt_path, f"visnet-uni-{self.ckpt_type}.ckpt")
        self.models = [get_visnet_model(model_path, device) for device in DeviceStrategy.get_bonded_devices()]
        self.chunk_atoms = DeviceStrategy._chunk_size
        self._work = None

    @classmethod
    def from_models(cls, models, chunk_atoms: int = DEFAULT_CHUNK_ATOMS, fragment_method=None):
        """test / embedding aid: wrap already constructed model handles (no checkpoint file, no DeviceStrategy)"""
        if not models:
            raise RuntimeError("No compute resources for bonded calculation")
        self = cls.__new__(cls)
        self.ckpt_path = self.ckpt_type = None
        self.fragment_method, self.combiner = fragment_method, DipeptideBondedCombiner()
        self.models, self.chunk_atoms, self._work = list(models), chunk_atoms, None
        return self

    def set_work_partitions(self, start, end):
        self._work = work_partitions(start, end, len(self.models), self.chunk_atoms)

    @staticmethod
    def _inference_impl(data, model):
        outs = [model.dl_potential_loader(unit) for unit in data]
        return [o[0] for o in outs], [o[1] for o in outs]

    def calculate(self, fragments: FragmentData):
        """-> (dipeptides_energy, dipeptides_forces, ACE_NMEs_energy, ACE_NMEs_forces) numpy."""
        work = self._work
        if work is None:
            work = DeviceStrategy.get_work_partitions() if self.ckpt_path is not None else None
            if not work:
                self.set_work_partitions(fragments.start, fragments.end)
                work = self._work
        parts = [[] for _ in self.models]
        for dev, f0, f1 in work:
            if f1 > f0:
                parts[dev].append(fragments[f0:f1])
        if len(self.models) == 1:  # one device: no hand-off to a worker thread (bonded.py:75-77 starts one per device)
            res = [self._inference_impl(parts[0], self.models[0])]
        else:
            if getattr(self, "_pool", None) is None:
                self._pool = ThreadPoolExecutor(len(self.models))  # kept: a pool per call costs ~0.1 ms of every step
            futs = [self._pool.submit(self._inference_impl, d, m) for d, m in zip(parts, self.models)]
            res = [f.result() for f in futs]
        es = [e for r in res for e in r[0]]
        fs = [f for r in res for f in r[1]]
        energy = es[0] if len(es) == 1 else np.concatenate(es)
        forces = fs[0] if len(fs) == 1 else np.concatenate(fs)
        # the dipeptide / ACE-NME masks depend on the fragment offsets only, which are fixed for a simulation
        key = (np.asarray(fragments.start).tobytes(), np.asarray(fragments.end).tobytes())  # (content, not identity)
        if getattr(self, "_split_key", None) != key:
            self._split_key, self._splits = key, (fragments.scalar_split(), fragments.vector_split())
        (sd, sa), (vd, va) = self._splits
        return energy[sd], forces[vd], energy[sa], forces[va]

    def __call__(self, prot):
        """-> (energy, forces[n_prot,3]) like bonded.py:102-123: fragments of the current positions (cap hydrogens
        placed and relaxed), model evaluation, recombination with prot.select_index / prot.origin_index."""
        fragments = self.fragment_method.get_fragments(prot)
        e_dip, f_dip, e_ace, f_ace = self.calculate(fragments)
        energy = self.combiner.energy_combine(e_dip, e_ace)
        forces = self.combiner.forces_combine(len(prot), f_dip, f_ace, prot.select_index, prot.origin_index)
        return energy, forces


def combine_numpy(n_prot, e_dip, f_dip, e_ace, f_ace, select_index, origin_index):
    """DipeptideBondedCombiner on the host (combiner.py:12-41) for the numpy-shaped path."""
    energy = np.float32(e_dip.sum() - e_ace.sum())
    cat = np.concatenate([f_dip, -f_ace])[select_index]
    out = np.zeros((n_prot, 3), dtype=np.float32)
    np.add.at(out, origin_index, cat)
    return energy, out


# ------------------------------------------------------------------------------------
class _HipTail:
    """The two ends of a sharded step on the device, through the C ABI: fragment gather + cap-hydrogen placement
    (`vsn_build_fragments`) and the deterministic combine (`vsn_combine[_with_energy]`).  `for_engine` talks to the
    device only through this object and the engine, so the CPU tests can run the SAME wiring (exchange-buffer views,
    slot layout, remapped combine plan) over gloo with stand-ins for both (tests/test_fragmentation_and_sharding.py)."""

    def __init__(self, device, index):
        self.device, self.index, self.L = device, index, capi.lib()

    def stream(self):
        return torch.cuda.current_stream(self.device)

    def _sp(self, st):
        return C.c_void_p(st.cuda_stream)

    def fragplan(self, src, acc, tow, ln):
        fp = C.c_void_p()
        rc = self.L.vsn_fragplan_create(C.byref(fp), self.index, len(src), capi.i64_ptr(src), capi.i64_ptr(acc),
                                        capi.i64_ptr(tow), ln.ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(f"vsn_fragplan_create failed ({rc})")
        return fp

    def build(self, fp, prot_pos, pos_geo, st):
        rc = self.L.vsn_build_fragments(fp, C.c_void_p(prot_pos.data_ptr()), C.c_void_p(pos_geo.data_ptr()), self._sp(st))
        if rc:
            raise RuntimeError(f"vsn_build_fragments failed ({rc})")

    def combine_plan(self, n_prot, row_of_cat, n_dip_rows, select_index, origin_index, e_idx, e_sgn):
        cp = C.c_void_p()
        rc = self.L.vsn_combine_plan_create(C.byref(cp), self.index, n_prot, len(row_of_cat), n_dip_rows,
                                            capi.i64_ptr(row_of_cat), capi.i64_ptr(select_index),
                                            capi.i64_ptr(origin_index), len(select_index))
        if rc:
            raise RuntimeError(f"vsn_combine_plan_create failed ({rc})")
        rc = self.L.vsn_combine_plan_set_energy(cp, len(e_idx), capi.i64_ptr(e_idx),
                                                e_sgn.ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(f"vsn_combine_plan_set_energy failed ({rc})")
        return cp

    def combine(self, cp, buf, F_prot, st):
        rc = self.L.vsn_combine(cp, C.c_void_p(buf.data_ptr()), C.c_void_p(F_prot.data_ptr()), self._sp(st))
        if rc:
            raise RuntimeError(f"vsn_combine failed ({rc})")

    def combine_with_energy(self, cp, buf, F_prot, E_tot, st):
        rc = self.L.vsn_combine_with_energy(cp, C.c_void_p(buf.data_ptr()), C.c_void_p(F_prot.data_ptr()),
                                            C.c_void_p(E_tot.data_ptr()), self._sp(st))
        if rc:
            raise RuntimeError(f"vsn_combine_with_energy failed ({rc})")


class _DevView:
    """zero-copy torch view of library-owned device memory (the CUDA array interface torch.as_tensor understands)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f4", data=(int(ptr), False), version=2)


class P2PExchange:
    """The tuned exchange step: every rank STORES its slot into its peers' gather buffers - one HIP
    launch per step, no collective library in between (`vsn_p2p_*`, csrc/p2p.hip; buffers mapped into the peers with
    hipIpcGetMemHandle / hipIpcOpenMemHandle, xGMI point-to-point).  The handle records are exchanged ONCE over the
    process group that already exists (`all_gather_object`: gloo or RCCL); after that the group is not on the step's
    path.  `send` is the rank's slot (its kernels write there), `gather(stream)` returns this step's [world * slot]
    buffer - bitwise what `all_gather_into_tensor(recv, send)` would hold."""

    def __init__(self, device, rank, world, slot, group=None):
        self.L = capi.lib()
        self.device, self.rank, self.world, self.slot, self.group = device, rank, world, int(slot), group
        idx = torch.device(device).index or 0
        self._h = C.c_void_p()
        rc = self.L.vsn_p2p_create(C.byref(self._h), idx, rank, world, self.slot)
        if rc:
            raise RuntimeError(f"vsn_p2p_create failed ({rc})")
        mine = (C.c_char * capi.P2P_HANDLE_BYTES)()
        rc = self.L.vsn_p2p_export(self._h, C.cast(mine, C.c_void_p))
        if rc:
            raise RuntimeError(f"vsn_p2p_export failed ({rc}): hipIpcGetMemHandle refused (HSA_ENABLE_IPC_MODE_LEGACY=0?)")
        records = [bytes(mine)]
        if world > 1:
            import torch.distributed as dist

            records = [None] * world
            dist.all_gather_object(records, bytes(mine), group=group)
        blob = b"".join(records)
        rc = self.L.vsn_p2p_connect(self._h, C.cast(C.c_char_p(blob), C.c_void_p))
        if rc:
            raise RuntimeError(f"vsn_p2p_connect failed ({rc}): hipIpcOpenMemHandle refused a peer's buffer")
        self.send = torch.as_tensor(_DevView(self.L.vsn_p2p_send_buffer(self._h), self.slot), device=device)
        ptrs = [self.L.vsn_p2p_gather_buffer(self._h, k) for k in (0, 1)]
        self._bufs = {int(ptr): torch.as_tensor(_DevView(ptr, world * self.slot), device=device) for ptr in ptrs}
        if world > 1:
            dist.barrier(group=group)  # every rank has mapped every buffer before the first store

    def gather(self, stream):
        out = C.c_void_p()
        rc = self.L.vsn_p2p_allgather(self._h, C.c_void_p(stream.cuda_stream), C.byref(out))
        if rc:
            raise RuntimeError(f"vsn_p2p_allgather failed ({rc})")
        return self._bufs[int(out.value)]  # (the half of the double buffer the LIBRARY chose for this step)

    def check(self, stream=None):
        """synchronises; raises if a wait gave up (a peer never stored its slot)"""
        st = stream or torch.cuda.current_stream(self.device)
        rc = self.L.vsn_p2p_status(self._h, C.c_void_p(st.cuda_stream))
        if rc:
            raise RuntimeError(f"P2P exchange: the wait of step {rc} timed out (a peer never delivered its slot)")

    def close(self):
        if getattr(self, "_h", None):
            if self.world > 1:
                import torch.distributed as dist

                torch.cuda.synchronize(self.device)
                if dist.is_initialized():
                    dist.barrier(group=self.group)  # nobody frees a buffer a slower peer still stores into
            self.L.vsn_p2p_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):  # (no collective from a finaliser: close() is the orderly way)
                self.L.vsn_p2p_destroy(self._h)
                self._h = None
        except Exception:
            pass


class ShardedFragmentForces:
    """Device-resident protein -> (E, F[n_prot,3]) evaluator, one instance per rank.

    `local_fn(pos_frag_local) -> (e_local [B_loc], f_local [N_loc,3])`,
    `build_fn(prot_pos, out)` and `combine_fn(f_all_padded) -> F_prot` are injected so
    the sharding / collective logic is testable on CPU (gloo) without a GPU; the
    product wiring is `ShardedFragmentForces.for_engine(...)`.
    """

    def __init__(self, plan: FragmentPlan, rank: int, world: int, device, group=None, balance: str = "atoms"):
        """balance: "atoms" = the reference's partition rule (device_strategy.py:84-127), "cost" = the same rule on
        the fragments' edge counts (see device_strategy.device_ranges)."""
        self.plan, self.rank, self.world, self.device, self.group = plan, rank, world, device, group
        self.balance = balance
        self.ranges = device_ranges(plan.start, plan.end, world, balance=balance)
        self.f0, self.f1 = self.ranges[rank]
        starts = np.append(plan.start, plan.end[-1])
        self.atom_lo = [int(starts[a]) for a, _ in self.ranges]
        self.atom_hi = [int(starts[b]) for _, b in self.ranges]
        self.rows = [hi - lo for lo, hi in zip(self.atom_lo, self.atom_hi)]
        self.nfrag = [b - a for a, b in self.ranges]
        self.max_rows = max(self.rows + [1])
        self.max_frag = max(self.nfrag + [1])
        # floats per rank in the fused buffer: forces rows then energies, padded to whole rows of 3
        self.slot = self.max_rows * 3 + ((self.max_frag + 2) // 3) * 3
        lo, hi = self.atom_lo[rank], self.atom_hi[rank]
        self.local_start = (plan.start[self.f0:self.f1] - lo).astype(np.int64)
        self.local_end = (plan.end[self.f0:self.f1] - lo).astype(np.int64)
        self.local_rows = hi - lo
        frag_owner = np.zeros(len(plan.start), dtype=np.int64)
        for r, (a, b) in enumerate(self.ranges):
            frag_owner[a:b] = r
        frag_local = np.arange(len(plan.start)) - np.asarray([a for a, _ in self.ranges])[frag_owner]
        self.frag_owner, self.frag_local = frag_owner, frag_local
        self.send = torch.zeros(self.slot, dtype=torch.float32, device=device)
        self.recv = torch.zeros(world * self.slot, dtype=torch.float32, device=device)
        self.local_fn = self.combine_fn = self.combine_energy_fn = None
        self.direct = False  # True when local_fn writes straight into the exchange buffer
        self.emulate = False
        self.force_collective = False  # world == 1: still go through the all-gather (exercises RCCL on a 1-GPU box)
        self.p2p = None                # P2PExchange: direct peer writes instead of the collective (for_engine exchange="p2p")
        self.fused_tail = None
        self.energy_sign = torch.as_tensor(plan.energy_sign, device=device)
        nonempty = (plan.end - plan.start) > 0
        self._e_index = torch.as_tensor(
            (frag_owner * self.slot + self.max_rows * 3 + frag_local)[nonempty], device=device)
        self._e_sign = torch.as_tensor(plan.energy_sign[nonempty], device=device)

    def gathered_force_rows(self):
        """int64 [Nf]: float offset, inside the all-gathered buffer, of every row of the
        interleaved fragment batch (slot % 3 == 0, so offset // 3 is a row index)."""
        owner = np.zeros(len(self.plan.z), dtype=np.int64)
        for r in range(self.world):
            owner[self.atom_lo[r]:self.atom_hi[r]] = r
        local = np.arange(len(self.plan.z)) - np.asarray(self.atom_lo)[owner]
        return owner * self.slot + local * 3

    def step(self, prot_pos):
        """prot_pos [n_prot,3] on self.device -> (E 0-d tensor, F [n_prot,3] tensor)."""
        buf = self.exchange(prot_pos)
        if self.combine_energy_fn is not None:  # product wiring: forces and total energy in one HIP launch
            return self.combine_energy_fn(buf)
        F = self.combine_fn(buf)
        E = (buf[self._e_index] * self._e_sign).sum()
        return E, F

    def exchange(self, prot_pos, prebuilt=False):
        """Everything of `step` in front of the combine: this rank's fragments evaluated, every rank's forces and
        energies gathered -> the padded buffer the combine reads.  prebuilt: the fragment geometry of `prot_pos` is
        already in `self.frag_pos` (the integrator's fused first half wrote it: LangevinHIP, vsn_md_half1_build)."""
        e_loc, f_loc = self.local_fn(prot_pos, prebuilt) if prebuilt else self.local_fn(prot_pos)
        if self.p2p is not None:  # the rank's kernels wrote its slot of the P2P send buffer: one launch gathers all slots
            return self.p2p.gather(torch.cuda.current_stream(self.device))
        stage = self.recv if (self.world == 1 and not self.force_collective) else self.send
        if not self.direct:  # local_fn returned its own tensors: stage them into the exchange buffer
            stage[: self.local_rows * 3] = f_loc.reshape(-1)
            stage[self.max_rows * 3: self.max_rows * 3 + len(self.local_start)] = e_loc
        if self.world > 1 and self.emulate:
            # single-process stand-in for one rank of a `world`-rank job (tuning aid: per-rank step time at that
            # shard size on a 1-GPU box); the other ranks' slots stay zero
            self.recv[self.rank * self.slot:(self.rank + 1) * self.slot] = self.send
        elif self.world > 1 or self.force_collective:
            import torch.distributed as dist

            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    # ---- product wiring: HIP engine + HIP gather/cap-H + HIP combine -------------------
    @classmethod
    def for_engine(cls, engine, plan: FragmentPlan, rank=0, world=1, group=None, hydrogen=None,
                   force_collective=False, balance: str = "atoms", tail=None, exchange: str = "collective"):
        """hydrogen: optional ai2bmd_amd.hydrogen.HydrogenPlan - relax the cap hydrogens every call like
        DistanceFragment.get_fragments (distancefrag.py:76-82).  The relaxation couples all dipeptides, so with
        it every rank builds and relaxes ALL fragment rows and then evaluates only its own shard.
        tail: the device ends (default: the HIP kernels through the C ABI, `_HipTail`); the CPU tests pass a torch
        stand-in together with a stand-in engine to run this very wiring over gloo.
        exchange: "collective" (default) = ONE `all_gather_into_tensor` per step (RCCL); "p2p" = the tuned variant,
        every rank stores its slot straight into its peers' gather buffers (`P2PExchange`, one process per GPU)."""
        dev = engine.device
        self = cls(plan, rank, world, dev, group, balance=balance)
        self.force_collective = bool(force_collective)
        if exchange not in ("collective", "p2p"):
            raise ValueError(f"exchange must be 'collective' or 'p2p', not {exchange!r}")
        if exchange == "p2p":
            self.p2p = P2PExchange(dev, rank, world, self.slot, group)
        tail = tail or _HipTail(dev, engine.index)
        lo, hi = self.atom_lo[rank], self.atom_hi[rank]
        # fragment geometry plan restricted to this rank's rows (all rows when the caps are relaxed)
        g_lo, g_hi = (0, len(plan.z)) if hydrogen is not None else (lo, hi)
        src = np.ascontiguousarray(plan.src[g_lo:g_hi])
        acc = np.ascontiguousarray(plan.acceptor[g_lo:g_hi])
        tow = np.ascontiguousarray(plan.toward[g_lo:g_hi])
        ln = np.ascontiguousarray(plan.length[g_lo:g_hi], dtype=np.float32)
        self._fp = tail.fragplan(src, acc, tow, ln)
        # the combine plan reads straight from the padded all-gather buffer (rows of 3 floats)
        rows = self.gathered_force_rows() // 3
        row_of_cat = np.ascontiguousarray(rows[plan.row_of_cat])
        e_idx = np.ascontiguousarray(self._e_index.cpu().numpy(), dtype=np.int64)
        e_sgn = np.ascontiguousarray(self._e_sign.cpu().numpy(), dtype=np.float32)
        self._cp = tail.combine_plan(plan.n_prot, row_of_cat, plan.n_dip_rows, np.ascontiguousarray(plan.select_index),
                                     np.ascontiguousarray(plan.origin_index), e_idx, e_sgn)
        z_loc = torch.as_tensor(plan.z[lo:hi], dtype=torch.int64).to(dev)
        pos_geo = torch.empty(max(g_hi - g_lo, 1), 3, dtype=torch.float32, device=dev)
        pos_loc = pos_geo[lo - g_lo: max(hi - g_lo, lo - g_lo + 1)]
        self.relaxer = None
        if hydrogen is not None:
            from .hydrogen import HydrogenRelaxer

            self.relaxer = HydrogenRelaxer(hydrogen, len(plan.z), engine.index)
        self.frag_pos = pos_geo
        nloc, bloc = hi - lo, self.f1 - self.f0
        # the kernels write this rank's forces / energies straight into its slot of the exchange buffer
        stage = self.p2p.send if self.p2p is not None else (self.recv if (world == 1 and not force_collective) else self.send)
        f_loc = stage[: max(nloc, 1) * 3].view(-1, 3)
        e_loc = stage[self.max_rows * 3: self.max_rows * 3 + max(bloc, 1)]
        self.direct = True
        F_prot = torch.empty(plan.n_prot, 3, dtype=torch.float32, device=dev)

        def local_fn(prot_pos, prebuilt=0):
            # prebuilt: 1 = the fragment geometry is already in pos_geo, 2 = and its cap hydrogens are relaxed (the
            # integrator's fused first half did both: vsn_md_half1_build[_relax])
            st = tail.stream()
            if nloc:
                if not prebuilt:
                    tail.build(self._fp, prot_pos, pos_geo, st)
                if self.relaxer is not None and prebuilt != 2:
                    self.relaxer.run(pos_geo, st)
                engine.forces_device(z_loc[:nloc], pos_loc[:nloc], self.local_start, self.local_end, e_loc[:bloc],
                                     f_loc[:nloc], stream=st)
            return e_loc[:bloc], f_loc[:nloc]

        def combine_fn(buf):
            tail.combine(self._cp, buf, F_prot, tail.stream())
            return F_prot

        E_tot = torch.zeros(1, dtype=torch.float32, device=dev)

        def combine_energy_fn(buf):
            tail.combine_with_energy(self._cp, buf, F_prot, E_tot, tail.stream())
            return E_tot[0], F_prot

        self.local_fn, self.combine_fn, self.combine_energy_fn = local_fn, combine_fn, combine_energy_fn
        # what the integrator's fused halves need (LangevinHIP: vsn_md_half1_build / vsn_md_combine_half2)
        self.fused_tail = (self._fp, self._cp, pos_geo, F_prot, E_tot) if (nloc and isinstance(tail, _HipTail)) else None
        self._keep = (z_loc, pos_geo, pos_loc, e_loc, f_loc, F_prot, E_tot, tail)
        return self
