"""TEST INFRASTRUCTURE ONLY - synthetic fragment geometries for the parity tests.

Molecule-like random clusters (bonded-distance growth with a minimum separation)
with the element mix of AI2BMD fragments; sizes follow the dipeptide templates of
/root/reference/src/utils/reference.py:36-64 (19..36 atoms, 44 for a CYX pair,
12 for ACE-NME).  Real-protein fixtures come from oracle/fragmenter.py.
"""
from __future__ import annotations

import numpy as np

ELEMENTS = np.array([1, 1, 1, 6, 6, 7, 8, 16])


def random_cluster(rng, n, min_dist=0.95, bond=(1.0, 1.6)):
    pts = [np.zeros(3)]
    while len(pts) < n:
        anchor = pts[rng.integers(len(pts))]
        v = rng.standard_normal(3)
        v /= np.linalg.norm(v)
        cand = anchor + v * rng.uniform(*bond)
        d = np.linalg.norm(np.asarray(pts) - cand, axis=1)
        if d.min() >= min_dist:
            pts.append(cand)
    return np.asarray(pts)


def random_fragments(seed, sizes, cutoff=5.0, margin=2e-3):
    """-> z int64 [N], pos float32 [N,3], start, end (int64 [B]).  Regenerates a
    cluster whenever a pair sits within `margin` of the cutoff, so that the edge
    set does not depend on last-bit rounding of d^2."""
    rng = np.random.default_rng(seed)
    zs, ps, start, end = [], [], [], []
    s = 0
    for n in sizes:
        start.append(s)
        end.append(s + n)
        s += n
        if n == 0:
            continue
        # big clusters have ~n^2/2 pairs: keep the exclusion band small enough to be satisfiable
        band = margin if n <= 64 else margin * (64.0 / n) ** 2
        for attempt in range(200):
            p = random_cluster(rng, n).astype(np.float32)
            p += rng.uniform(-20, 20, size=3).astype(np.float32)
            d = np.linalg.norm(p[:, None, :].astype(np.float64) - p[None, :, :], axis=-1)
            if np.abs(d - cutoff).min() > band:
                break
        else:
            raise RuntimeError("could not place a cluster away from the cutoff")
        ps.append(p)
        zs.append(rng.choice(ELEMENTS, size=n))
    z = np.concatenate(zs).astype(np.int64) if zs else np.zeros(0, np.int64)
    pos = np.concatenate(ps).astype(np.float32) if ps else np.zeros((0, 3), np.float32)
    return z, pos, np.asarray(start, np.int64), np.asarray(end, np.int64)
