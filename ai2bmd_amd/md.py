"""Minimal device-resident Langevin loop used by bench.py / tests to drive the force
path the way the reference's simulator does (one force evaluation per step).

ASE is not installed here, so the ASE 3.22 `Langevin.step` algorithm that
/root/reference/src/AIMD/simulator.py:96-116 configures (1 fs, 300 K, friction
0.001 / fs, fixcm, two normal draws per atom per step - RNGPool(count=2),
simulator.py:108) is restated: this is plumbing around the hot path (SURVEY.md 8f
"next #4"), plain torch elementwise ops on tensors that never leave HBM.
"""
from __future__ import annotations

import math

import numpy as np
import torch

# ASE unit system: eV, Angstrom, amu
FS = 0.09822694788464063      # ase.units.fs
KB = 8.617330337217213e-05    # ase.units.kB  (eV/K)
MASSES = {1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 16: 32.06}


class Langevin:
    def __init__(self, numbers, positions, force_fn, device, timestep_fs=1.0, temperature_K=300.0,
                 friction_per_fs=0.001, seed=0, tether_k=0.0):
        """force_fn(pos[n,3] device tensor) -> (E 0-d tensor, F[n,3] tensor).
        tether_k > 0 adds a harmonic restraint to the start geometry (eV/A^2), the same
        device used by the reference's restrained pre-equilibration (simulator.py:139-166);
        bench.py uses it because seeded random weights are not a physical potential."""
        self.device = device
        self.n = len(numbers)
        m = np.array([MASSES[int(z)] for z in numbers], dtype=np.float64)
        self.m = torch.as_tensor(m, dtype=torch.float32, device=device)[:, None]
        self.x = torch.as_tensor(np.asarray(positions), dtype=torch.float32, device=device).contiguous()
        self.x0 = self.x.clone()
        self.force_fn = force_fn
        self.tether_k = float(tether_k)
        dt = timestep_fs * FS
        T = temperature_K * KB
        fr = friction_per_fs / FS
        sigma = torch.sqrt(2.0 * T * fr / self.m)
        self.dt = dt
        self.c1 = dt / 2.0 - dt * dt * fr / 8.0
        self.c2 = dt * fr / 2.0 - dt * dt * fr * fr / 8.0
        self.c3 = math.sqrt(dt) * sigma / 2.0 - dt ** 1.5 * fr * sigma / 8.0
        self.c5 = dt ** 1.5 * sigma / (2.0 * math.sqrt(3.0))
        self.c4 = fr / 2.0 * self.c5
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        # Maxwell-Boltzmann start (simulator.py:96)
        self.v = torch.randn(self.n, 3, generator=self.gen, device=device) * torch.sqrt(T / self.m)
        self.E, self.F = self._forces()
        self.steps = 0

    def _forces(self):
        E, F = self.force_fn(self.x)
        if self.tether_k:
            dx = self.x - self.x0
            F = F - self.tether_k * dx
            E = E + 0.5 * self.tether_k * (dx * dx).sum()
        return E, F

    def step(self):
        xi = torch.randn(self.n, 3, generator=self.gen, device=self.device)
        eta = torch.randn(self.n, 3, generator=self.gen, device=self.device)
        rnd_pos = self.c5 * eta
        rnd_vel = self.c3 * xi - self.c4 * eta
        rnd_pos = rnd_pos - rnd_pos.sum(0, keepdim=True) / self.n  # fixcm
        rnd_vel = rnd_vel - (rnd_vel * self.m).sum(0, keepdim=True) / (self.m * self.n)
        v = self.v + (self.c1 * self.F / self.m - self.c2 * self.v + rnd_vel)
        self.x = (self.x + self.dt * v + rnd_pos).contiguous()
        self.E, self.F = self._forces()
        self.v = v + (self.c1 * self.F / self.m - self.c2 * v + rnd_vel)
        self.steps += 1

    def kinetic_energy(self):
        return 0.5 * (self.m * self.v * self.v).sum()


class LangevinHIP:
    """Same algorithm as `Langevin`, two HIP launches per step (`vsn_md_half1/half2`, csrc/md.hip)
    instead of ~25 torch elementwise kernels; normal deviates come from a counter-based Philox
    generator keyed by (seed, step, atom), so trajectories are reproducible but differ from the
    torch-generator ones."""

    def __init__(self, numbers, positions, force_fn, device, timestep_fs=1.0, temperature_K=300.0,
                 friction_per_fs=0.001, seed=0, tether_k=0.0):
        import ctypes as C

        from . import capi

        self._C, self._L = C, capi.lib()
        self.device = device
        self.n = len(numbers)
        m = np.array([MASSES[int(z)] for z in numbers], dtype=np.float32)
        self.m = torch.as_tensor(m, device=device)[:, None]
        x0 = np.ascontiguousarray(positions, dtype=np.float32)
        self.x = torch.as_tensor(x0, device=device).contiguous()
        self.force_fn = force_fn
        dt, kT, fr = timestep_fs * FS, temperature_K * KB, friction_per_fs / FS
        self._h = C.c_void_p()
        idx = torch.device(device).index or 0
        rc = self._L.vsn_md_create(C.byref(self._h), idx, self.n, m.ctypes.data_as(C.POINTER(C.c_float)),
                                   C.c_float(dt), C.c_float(kT), C.c_float(fr), C.c_uint64(seed),
                                   C.c_float(tether_k), x0.ctypes.data_as(C.POINTER(C.c_float)))
        if rc:
            raise RuntimeError(f"vsn_md_create failed ({rc})")
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.v = (torch.randn(self.n, 3, generator=gen, device=device) * torch.sqrt(kT / self.m)).contiguous()
        self.E, F = self.force_fn(self.x)
        self.F = F.contiguous()
        self.steps = 0

    def step(self):
        C = self._C
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._L.vsn_md_half1(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                  C.c_void_p(self.F.data_ptr()), st)
        if rc:
            raise RuntimeError(f"vsn_md_half1 failed ({rc})")
        self.E, F = self.force_fn(self.x)
        self.F = F if F.is_contiguous() else F.contiguous()
        rc = self._L.vsn_md_half2(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                  C.c_void_p(self.F.data_ptr()), st)
        if rc:
            raise RuntimeError(f"vsn_md_half2 failed ({rc})")
        self.steps += 1

    def kinetic_energy(self):
        return 0.5 * (self.m * self.v * self.v).sum()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.vsn_md_destroy(self._h)
                self._h = None
        except Exception:
            pass
