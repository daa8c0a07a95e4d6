"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy, float64) of the reference's MM non-bonded term
  /root/reference/src/Calculators/nonbonded.py:33-63   (LJ + Coulomb, analytic forces)
  /root/reference/src/AIMD/protein.py:133-151          (all ordered pairs i != j minus exclude_pair)
  /root/reference/src/Fragmentation/distancefrag.py:355-363 (exclude_pair = pairs inside one dipeptide)

Pinned on a run of the reference's own `MMNonBondedCalculator` (oracle/make_nonbonded_golden.py ->
tests/golden/mm_*.npz): the module is loaded from the reference tree with its two absent imports stubbed -
`ase.units` (ASE is not installed; third-party, version unpinned by the reference) and the `Protein` type
annotation.  The unit constants are restated from ASE's CODATA-2014 table
(ase.units since 3.12): _e = 1.6021766208e-19 C, _Nav = 6.022140857e23, _eps0 = 1/(mu0 c^2) with
mu0 = 4e-7 pi, c = 299792458 -> C = 1/_e, kJ = 1000/_e, mol = _Nav, nm = 10 Angstrom.
The restatement is checked against its own finite-difference gradient and against a brute-force
pair list (tests/test_nonbonded.py).
"""
from __future__ import annotations

import itertools
import math

import numpy as np

_e = 1.6021766208e-19
_Nav = 6.022140857e23
_c = 299792458.0
_mu0 = 4.0e-7 * math.pi
_eps0 = 1.0 / (_mu0 * _c ** 2)
C = 1.0 / _e
kJ = 1000.0 / _e
mol = _Nav
nm = 10.0
K_COULOMB = 1 / (4 * math.pi * _eps0) * 10e6 * mol * C ** (-2)  # nonbonded.py:18
KJ_MOL = kJ / mol


def exclude_pairs_from_dipeptides(dipeptides_index):
    """distancefrag.py:355-363"""
    ex = set()
    for idx in dipeptides_index:
        for x, y in itertools.combinations(idx, 2):
            ex.add((x, y))
            ex.add((y, x))
    return ex


def pair_list(n, exclude):
    """protein.py:133-151: every ordered pair i != j not excluded -> (src, dst)"""
    src, dst = [], []
    for i in range(n):
        for j in range(n):
            if i != j and (i, j) not in exclude:
                src.append(i)
                dst.append(j)
    return np.asarray(src, np.int64), np.asarray(dst, np.int64)


def mm_nonbonded(pos, charges, sigmas, epsilons, src, dst):
    """nonbonded.py:33-63 -> (energy [eV], forces [n,3] eV/Angstrom)"""
    pos = np.asarray(pos, np.float64)
    vec = pos[dst] - pos[src]
    d2 = (vec ** 2).sum(-1)
    d = np.sqrt(d2)
    sig = 0.5 * (sigmas[src] + sigmas[dst]) * nm
    eps = np.sqrt(epsilons[src] * epsilons[dst])
    c6 = (sig ** 2 / d2) ** 3
    c12 = c6 ** 2
    e_lj = 4 * eps * (c12 - c6)
    f_lj = (24 * eps * (2 * c12 - c6) / d2)[:, None] * vec
    e_c = K_COULOMB * charges[src] * charges[dst] / d
    f_c = (e_c / d2)[:, None] * vec
    force = np.zeros_like(pos)
    np.add.at(force, dst, f_lj + f_c)
    return (e_lj.sum() + e_c.sum()) * KJ_MOL / 2, force * KJ_MOL
