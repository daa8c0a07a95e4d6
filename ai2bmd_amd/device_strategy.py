"""Fragment -> device/chunk work partitions.

Mirror of the reference's data-parallel scheduler
(/root/reference/src/Calculators/device_strategy.py:84-127,
`DeviceStrategy._set_combined_work_partitions`): the interleaved fragment list is
cut into `n_devices` contiguous blocks balanced by atom count (a fragment that
straddles a cut goes to the nearer side), each block further cut into chunks of
at most ~`chunk_atoms` atoms.  The arithmetic lives in the C ABI
(`vsn_partition`, ai2bmd_amd/csrc/engine.hip) so non-Python hosts share it.
"""
from __future__ import annotations

import numpy as np

from . import capi

DEFAULT_CHUNK_ATOMS = 9999  # the reference's --chunk-size default (AIMD/arguments.py:189-197)


def work_partitions(start, end, n_devices: int, chunk_atoms: int = DEFAULT_CHUNK_ATOMS):
    """-> list of (device_idx, frag_begin, frag_end), ascending and covering every fragment."""
    start = np.ascontiguousarray(start, dtype=np.int64)
    end = np.ascontiguousarray(end, dtype=np.int64)
    B = len(start)
    if B == 0:
        return []
    cap = B + n_devices + 8
    out = np.zeros(3 * cap, dtype=np.int64)
    n = capi.lib().vsn_partition(capi.i64_ptr(start), capi.i64_ptr(end), B, int(n_devices), int(chunk_atoms),
                                 capi.i64_ptr(out), cap)
    if n < 0:
        raise RuntimeError(f"vsn_partition failed ({n})")
    if n > cap:
        raise RuntimeError("vsn_partition: output overflow")
    return [tuple(int(v) for v in out[3 * i:3 * i + 3]) for i in range(n)]


def fragment_cost(start, end, max_num_neighbors: int = 32):
    """Work of one fragment ~ its edge count: n * min(n, max_num_neighbors + 1) (dipeptides are smaller than the
    cutoff sphere, so nearly every pair is an edge; self loops included) - the per-edge linears and gathers are
    ~ 80 % of an evaluation, SURVEY.md 8e."""
    n = np.asarray(end, dtype=np.int64) - np.asarray(start, dtype=np.int64)
    return n * np.minimum(n, int(max_num_neighbors) + 1)


def device_ranges(start, end, n_devices: int, balance: str = "atoms", max_num_neighbors: int = 32):
    """Per-device contiguous fragment range [(f0, f1)] (chunks merged) - what one
    rank of the multi-GPU calculator owns.

    balance = "atoms": the reference's rule (device_strategy.py:84-127, blocks of equal ATOM count);
    balance = "cost":  the same cutting rule applied to the cumulative EDGE count (n_f * min(n_f, max_nb + 1))
    instead of the cumulative atom count - the step of a rank follows its edges, and the step of the job is the
    slowest rank's (a 36-atom TRP dipeptide costs 9x a 12-atom ACE-NME, not 3x)."""
    if balance == "cost":
        cost = fragment_cost(start, end, max_num_neighbors)
        cum = np.cumsum(cost)
        start, end = cum - cost, cum
    elif balance != "atoms":
        raise ValueError(f"device_ranges: unknown balance {balance!r}")
    parts = work_partitions(start, end, n_devices, chunk_atoms=1 << 40)
    rng = [(0, 0)] * n_devices
    seen = {}
    for d, a, b in parts:
        lo, hi = seen.get(d, (a, b))
        seen[d] = (min(lo, a), max(hi, b))
    last = 0
    for d in range(n_devices):
        if d in seen:
            rng[d] = seen[d]
            last = seen[d][1]
        else:
            rng[d] = (last, last)
    return rng


class DeviceStrategy:
    """Mirror of the reference's device/work registry (/root/reference/src/Calculators/device_strategy.py:11-265)
    for the slots the hot path reads: `get_bonded_devices()`, `get_non_bonded_device()`, `get_default_device()`,
    `get_optimiser_device()`, `set_work_partitions(start, end)`, `get_work_partitions()`.  Same class-level state,
    same `initialize(dev_strategy, work_strategy, mm_method, gpu_count, chunk_size)` slot rules (:143-235); the
    solvent / preprocess slots and the CPU thread policy belong to parts of AI2BMD outside this package.  With
    `gpu_count == 0` the reference falls back to two CPU models (:176); this package has no CPU model, so that
    raises."""

    _gpu_count = 0
    _chunk_size = DEFAULT_CHUNK_ATOMS
    _bonded_devices: list = []
    _non_bonded_device = "cuda:0"
    _default_device = "cuda:0"
    _optimiser_device = "cuda:0"
    _fragment_strategy = False
    _work_partitions: list = []

    @classmethod
    def _check_device(cls, device: str):
        if not device.startswith("cuda"):
            raise Exception("Unrecognized device (this package drives 'cuda:<k>' devices only)")
        tup = device.split(":")
        assert len(tup) == 2, "invalid device syntax"
        n = int(tup[1])
        assert 0 <= n < cls._gpu_count, "invalid device index"

    @classmethod
    def initialize(cls, dev_strategy: str = "small-molecule", work_strategy: str = "combined", mm_method: str = "mm",
                   gpu_count: int = None, chunk_size: int = DEFAULT_CHUNK_ATOMS):
        if gpu_count is None:
            import torch

            gpu_count = torch.cuda.device_count()
        if gpu_count < 1:
            raise RuntimeError("DeviceStrategy: no GPU visible and this package has no CPU model")
        if work_strategy == "combined" and chunk_size == 0:
            raise ValueError(f"chunk-size: {chunk_size} must be non-zero for 'combined' work strategy")
        cls._gpu_count, cls._chunk_size = gpu_count, chunk_size
        last = gpu_count - 1
        if dev_strategy == "excess-compute":
            bonded = ["cuda:0"] if gpu_count == 1 else [f"cuda:{i}" for i in range(gpu_count - 1)]
        elif dev_strategy in ("small-molecule", "large-molecule"):
            bonded = [f"cuda:{i}" for i in range(gpu_count)]
        else:
            raise Exception("Unknown compute strategy")
        cls._bonded_devices = bonded
        cls._non_bonded_device = f"cuda:{last}"
        cls._default_device = "cuda:0"
        # the reference relaxes the cap hydrogens on the CPU unless 'large-molecule' has > 2 GPUs (:215); here the
        # relaxation is a HIP kernel, so it always runs on the first bonded device
        cls._optimiser_device = bonded[0]
        cls._fragment_strategy = dev_strategy == "large-molecule"
        cls._work_partitions = []
        return {"mm-method": mm_method}

    @classmethod
    def get_bonded_devices(cls):
        if len(cls._bonded_devices) < 1:
            raise Exception("No compute resources for bonded calculation")
        for dev in cls._bonded_devices:
            cls._check_device(dev)
        return cls._bonded_devices

    @classmethod
    def get_non_bonded_device(cls):
        cls._check_device(cls._non_bonded_device)
        return cls._non_bonded_device

    @classmethod
    def get_default_device(cls):
        cls._check_device(cls._default_device)
        return cls._default_device

    @classmethod
    def get_optimiser_device(cls):
        return cls._optimiser_device

    @classmethod
    def fragment_strategy(cls):
        return cls._fragment_strategy

    @classmethod
    def set_work_partitions(cls, start, end):
        cls._work_partitions = work_partitions(start, end, len(cls._bonded_devices), cls._chunk_size)

    @classmethod
    def get_work_partitions(cls):
        return cls._work_partitions
