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


def device_ranges(start, end, n_devices: int):
    """Per-device contiguous fragment range [(f0, f1)] (chunks merged) - what one
    rank of the multi-GPU calculator owns."""
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
