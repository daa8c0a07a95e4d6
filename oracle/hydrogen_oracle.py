"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's cap-hydrogen relaxation.

Energies: /root/reference/src/Fragmentation/hydrogen/energies.py:8-61 (bond, angle, dihedral, Lennard-Jones,
Coulomb; every term counted once; LJ / 1.2 and Coulomb / 2.0 on ALL pairs, energies.py:48-60,76-81).
Optimiser: torch.optim.LBFGS(lr=0.1, max_iter=10, tolerance_grad=0.1, tolerance_change=0.01), one `.step`,
over the positions of all cap hydrogens of all dipeptides jointly (energies.py:211-242,
distancefrag.py:30-32,79).

Works on the flat term lists of ai2bmd_amd.hydrogen.HydrogenPlan.  Pinned against the reference's own
HydrogenOptimizer run in the build container (oracle/make_hydrogen_golden.py -> tests/golden/hopt_*.npz).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

SCNB, SCEE = 1.2, 2.0


def _t(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    return t.to(dtype) if dtype is not None else t


class HydrogenOracle:
    def __init__(self, hplan, dtype=torch.float32):
        self.dtype = dtype
        self.cap = _t(hplan.cap_rows, torch.long)
        L, D = torch.long, dtype
        b, a, d, p = hplan.bond, hplan.angle, hplan.dihedral, hplan.pair
        self.b = (_t(b["i"], L), _t(b["j"], L), _t(b["kf"], D), _t(b["r0"], D))
        self.a = (_t(a["i"], L), _t(a["j"], L), _t(a["k"], L), _t(a["kf"], D), _t(a["th0"], D))
        self.d = (_t(d["i"], L), _t(d["j"], L), _t(d["k"], L), _t(d["l"], L), _t(d["kf"], D), _t(d["per"], D), _t(d["phase"], D))
        self.p = (_t(p["i"], L), _t(p["j"], L), _t(p["A"], D), _t(p["B"], D), _t(p["qq"], D))

    def energy(self, pos):
        """pos [Nf,3] -> tensor [5] (bond, angle, dihedral, vdw, elec), kcal/mol."""
        bi, bj, bk, br0 = self.b
        dist = torch.norm(pos[bi] - pos[bj], dim=-1)
        e_b = 0.5 * (bk * (dist - br0).square()).sum()
        ai, aj, ak, akf, ath = self.a
        v0, v1 = pos[ai] - pos[aj], pos[ak] - pos[aj]
        ang = torch.atan2(torch.norm(torch.cross(v0, v1, dim=-1), dim=-1), (v0 * v1).sum(-1))
        e_a = 0.5 * (akf * (ang - ath).square()).sum()
        di, dj, dk, dl, dkf, dper, dph = self.d
        p0, p1, p2, p3 = pos[di], pos[dj], pos[dk], pos[dl]
        w0, w1, w2 = p1 - p2, p1 - p0, p3 - p2
        n1 = F.normalize(torch.cross(w1, w0, dim=-1), dim=-1)
        n2 = F.normalize(torch.cross(w0, w2, dim=-1), dim=-1)
        m1 = torch.cross(n1, F.normalize(w0, dim=-1), dim=-1)
        phi = torch.atan2((m1 * n2).sum(-1), (n1 * n2).sum(-1))
        e_d = 0.5 * (dkf * (1 + torch.cos(dper * phi - dph))).sum()
        pi, pj, pA, pB, pqq = self.p
        r = torch.norm(pos[pi] - pos[pj], dim=-1)
        r6 = r ** 6
        e_v = (pA / (r6 * r6) - pB / r6).sum() / SCNB
        e_e = (pqq / r).sum() / SCEE
        return torch.stack([e_b, e_a, e_d, e_v, e_e])

    def energy_grad(self, pos_np):
        pos = _t(pos_np, self.dtype).clone().requires_grad_(True)
        e = self.energy(pos).sum()
        (g,) = torch.autograd.grad(e, pos)
        return float(e.detach()), g[self.cap].numpy()

    def relax(self, pos_np, max_iter=10, return_trace=False):
        """-> positions [Nf,3] with the cap-hydrogen rows relaxed (other rows untouched)."""
        base = _t(pos_np, self.dtype).clone()
        x = torch.nn.Parameter(base[self.cap].clone())
        opt = torch.optim.LBFGS([x], lr=0.1, max_iter=max_iter, tolerance_grad=0.1, tolerance_change=0.01)
        trace = []

        def closure():
            opt.zero_grad()
            pos = base.index_put((self.cap,), x)
            e = self.energy(pos).sum()
            e.backward()
            trace.append(float(e.detach()))
            return e

        opt.step(closure)
        out = base.clone()
        out[self.cap] = x.detach()
        return (out.numpy(), trace) if return_trace else out.numpy()
