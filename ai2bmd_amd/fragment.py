"""FragmentData - the batch struct that crosses the calculator seam.

Mirror of the reference's boundary type (/root/reference/src/AIMD/fragment.py:7-55)
so that objects produced by the reference's DistanceFragment.get_fragments can be
passed in unchanged (duck-typed: only .z/.pos/.start/.end/.batch are read) and so
that the parity tests read like reference code.  No ASE dependency.

Layout: fragments are stored back to back; `start[b]:end[b]` is the atom range
of fragment b, `batch[i]` the 0-based id of the non-empty fragment of atom i.
AI2BMD interleaves dipeptides and ACE-NME caps: even fragment slots are
dipeptides, odd slots ACE-NMEs (distancefrag.py:250-255); empty slots are allowed
(the second half of a CYX pair, reference.py:41-42).
"""
from __future__ import annotations

import numpy as np


class FragmentData:
    def __init__(self, z, pos, start, end, batch):
        self.z = z
        self.pos = pos
        self.start = start
        self.end = end
        self.batch = batch
        self._split_cache = {}

    def __len__(self):
        return len(self.start)

    def __getitem__(self, f_idx):
        """Sub-batch of fragments, offsets rebased to 0 (fragment.py:15-29)."""
        if isinstance(f_idx, (int, np.integer)):
            f_idx = slice(int(f_idx), int(f_idx) + 1)
        lo, hi, step = f_idx.indices(len(self))
        if step != 1 or hi <= lo:
            raise IndexError("FragmentData supports non-empty contiguous fragment ranges only")
        a0, a1 = int(self.start[lo]), int(self.end[hi - 1])
        return FragmentData(
            self.z[a0:a1],
            self.pos[a0:a1],
            self.start[lo:hi] - self.start[lo],
            self.end[lo:hi] - self.start[lo],
            self.batch[a0:a1] - self.batch[a0],
        )

    def scalar_split(self):
        """Masks over the per-fragment energies of NON-EMPTY fragments:
        (is_dipeptide, is_acenme)  (fragment.py:31-38)."""
        if "s" not in self._split_cache:
            nonempty = (np.asarray(self.end) - np.asarray(self.start)) != 0
            is_dip = (np.arange(len(self)) % 2) == 0
            self._split_cache["s"] = (is_dip[nonempty], ~is_dip[nonempty])
        return self._split_cache["s"]

    def vector_split(self):
        """Masks over atoms: (in a dipeptide, in an ACE-NME)  (fragment.py:40-47)."""
        if "v" not in self._split_cache:
            n = int(self.end[-1])
            is_dip = np.zeros(n, dtype=bool)
            for b in range(0, len(self), 2):
                is_dip[int(self.start[b]):int(self.end[b])] = True
            self._split_cache["v"] = (is_dip, ~is_dip)
        return self._split_cache["v"]

    def get_fragment(self, idx: int):
        """(z, pos) of one fragment (the reference returns an ase.Atoms, fragment.py:52-55)."""
        s, e = int(self.start[idx]), int(self.end[idx])
        return self.z[s:e], self.pos[s:e]


def make_batch_index(start, end):
    """0-based ids contiguous over non-empty fragments (what the reference feeds
    the model as `batch`, visnet.py:146)."""
    start = np.asarray(start, dtype=np.int64)
    end = np.asarray(end, dtype=np.int64)
    sizes = end - start
    ids = np.cumsum(sizes > 0) - 1
    return np.repeat(ids, sizes).astype(np.int64)
