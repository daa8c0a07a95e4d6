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

import os

import numpy as np
import torch

# ASE unit system: eV, Angstrom, amu
FS = 0.09822694788464063      # ase.units.fs
KB = 8.617330337217213e-05    # ase.units.kB  (eV/K)
MASSES = {1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 16: 32.06}


# (units.kcal / units.mol) / units.eV = 4184 / (_e * _Nav) with the CODATA 2014 table FS and KB above come from (ASE's
# default since 3.17); the literature value 0.0433641153 (older CODATA) differs in the 7th digit
KCALMOL2EV = 0.04336410390059322


class TemperatureRunawayError(RuntimeError):
    """utils/utils.py:108-111: raised by the energy observer when T > 1.5 x the thermostat temperature"""

    def __init__(self, temp_k, *args):
        self.temp = temp_k
        super().__init__(*args)


class Hookean:
    """Data of an ASE `Hookean(a1, a2, k, rt)` restraint as the reference builds them (simulator.py:139-180):
    `a2` an atom index -> spring between two atoms; `a2` a 3-vector -> spring to a fixed point.  Active beyond the
    threshold distance rt: force k (r - rt) along the line, energy k (r - rt)^2 / 2."""

    def __init__(self, a1, a2, k, rt=None):
        self.a1, self.k, self.rt = int(a1), float(k), 0.0 if rt is None else float(rt)
        if np.ndim(a2) == 0:
            self.a2, self.origin = int(a2), None
        else:
            self.a2, self.origin = None, np.asarray(a2, dtype=np.float64).reshape(3)


def hookean_forces(x, constraints):
    """torch restatement of the restraint force / energy on positions x [n,3] (double-checks the HIP kernel)."""
    F = torch.zeros_like(x)
    E = torch.zeros((), dtype=x.dtype, device=x.device)
    for c in constraints:
        p2 = x[c.a2] if c.a2 is not None else torch.as_tensor(c.origin, dtype=x.dtype, device=x.device)
        d = p2 - x[c.a1]
        r = torch.linalg.norm(d)
        if float(r.detach()) > c.rt and float(r.detach()) > 0:
            f = c.k * (r - c.rt) * d / r
            F[c.a1] += f
            if c.a2 is not None:
                F[c.a2] -= f
            E = E + 0.5 * c.k * (r - c.rt) ** 2
    return E, F


class _MDBase:
    """What the reference's simulate() does around `MolDyn.run` (simulator.py:118-193): observers attached with
    an interval, restrained pre-equilibration stages, hydrogen bond restraints."""

    def _init_observers(self, temperature_K):
        self.temp_k = float(temperature_K)
        self.observers = []
        self.nsteps = 0

    def attach(self, fn, interval=1):
        """ASE `MolecularDynamics.attach`: fn() is called after every `interval`-th step"""
        self.observers.append((fn, int(interval)))

    def run(self, steps):
        # ASE 3.22 Dynamics.irun calls the attached observers once on the starting state (nsteps == 0) before the
        # first step - the reference's MolDyn.run inherits that: initial frame + energy line, runaway guard included
        if self.nsteps == 0:
            for fn, _iv in self.observers:
                fn()
        for _ in range(int(steps)):
            self.step()
            self.nsteps += 1
            for fn, iv in self.observers:
                if self.nsteps % iv == 0:
                    fn()

    def printenergy(self, quiet=False):
        """utils/utils.py:143-159 MDObserver.printenergy: Epot / Ekin / Etot and the temperature-runaway guard"""
        epot, ekin, temp = self.observe()
        if temp > 1.5 * self.temp_k:
            raise TemperatureRunawayError(temp, "temperature runaway")
        if not quiet:
            print(f"Step {self.nsteps:d}: Epot = {epot:.3f}eV Ekin = {ekin:.3f}eV Etot = {epot + ekin:.3f}eV")
        return epot, ekin, temp

    def pre_equilibrate(self, indices, preeq_steps, restraints=(10, 5, 1, 0.5, 0.1)):
        """simulator.py:139-166: for each stage, every atom of `indices` on a Hookean spring of `restraint`
        kcal/mol/A^2 (k = restraint * kcal/mol in eV, rt = 0) to its position at the start of the stage."""
        keep = list(self.constraints)
        for restraint in restraints:
            ref = self.x.detach().cpu().numpy().astype(np.float64)
            self.set_constraints(keep + [Hookean(a1=i, a2=ref[i], k=restraint * KCALMOL2EV, rt=0) for i in indices])
            self.run(preeq_steps)
        self.set_constraints(keep)


def _draw(rng, n, device):
    """(xi, eta) of one step from a caller-supplied source with numpy's `standard_normal(size)` (the reference hands
    ASE `rng=RNGPool(seed, (n, 3), count=2)`, simulator.py:108; ASE draws xi, then eta)"""
    xi = np.asarray(rng.standard_normal(size=(n, 3)), dtype=np.float32)
    eta = np.asarray(rng.standard_normal(size=(n, 3)), dtype=np.float32)
    return torch.as_tensor(xi, device=device), torch.as_tensor(eta, device=device)


class Langevin(_MDBase):
    def __init__(self, numbers, positions, force_fn, device, timestep_fs=1.0, temperature_K=300.0,
                 friction_per_fs=0.001, seed=0, tether_k=0.0, rng=None, velocities=None):
        """force_fn(pos[n,3] device tensor) -> (E 0-d tensor, F[n,3] tensor).
        rng: object with numpy's `standard_normal(size=(n, 3))` supplying the two draws of every step (the reference's
        RNGPool); velocities: start velocities [n,3] in ASE units instead of the built-in Maxwell-Boltzmann draw.
        tether_k > 0 restrains every atom to its start position (eV/A^2, threshold 0) - the reference's
        pre-equilibration device (simulator.py:139-166); bench.py uses it because seeded random weights are not a
        physical potential.  self.E / self.F are model + restraints (what ASE's atoms.get_forces() returns with
        constraints attached)."""
        self.device = device
        self.n = len(numbers)
        m = np.array([MASSES[int(z)] for z in numbers], dtype=np.float64)
        self.m = torch.as_tensor(m, dtype=torch.float32, device=device)[:, None]
        self.x = torch.as_tensor(np.asarray(positions), dtype=torch.float32, device=device).contiguous()
        self.x0 = self.x.clone()
        self.force_fn = force_fn
        self.tether_k = float(tether_k)
        self.constraints = []
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
        self.rng = rng
        # Maxwell-Boltzmann start (simulator.py:96)
        if velocities is not None:
            self.v = torch.as_tensor(np.asarray(velocities), dtype=torch.float32, device=device).contiguous()
        else:
            self.v = torch.randn(self.n, 3, generator=self.gen, device=device) * torch.sqrt(T / self.m)
        self._init_observers(temperature_K)
        self.E, self.F = self._forces()
        self.steps = 0

    def set_constraints(self, constraints):
        self.constraints = list(constraints)
        self.E, self.F = self._forces()

    def _forces(self):
        E, F = self.force_fn(self.x)
        F = F.clone()
        self.E_restraint = torch.zeros((), device=self.x.device)
        if self.tether_k:
            dx = self.x - self.x0
            F = F - self.tether_k * dx
            self.E_restraint = self.E_restraint + 0.5 * self.tether_k * (dx * dx).sum()
        if self.constraints:
            e_c, f_c = hookean_forces(self.x, self.constraints)
            F = F + f_c
            self.E_restraint = self.E_restraint + e_c
        return E + self.E_restraint, F

    def step(self):
        if self.rng is not None:
            xi, eta = _draw(self.rng, self.n, self.device)
        else:
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

    def observe(self):
        ekin = float(self.kinetic_energy())
        return float(self.E), ekin, 2.0 * ekin / (3.0 * self.n * KB)


class LangevinHIP(_MDBase):
    """Same algorithm as `Langevin`, two HIP launches per step (`vsn_md_half1/half2`, csrc/md.hip)
    instead of ~25 torch elementwise kernels; normal deviates come from a counter-based Philox
    generator keyed by (seed, step, atom), so trajectories are reproducible but differ from the
    torch-generator ones.  Restraints (tether / Hookean lists) are evaluated inside half2; self.F is model +
    restraints and self.E the model energy plus the restraint energy of the same evaluation (read lazily: the
    restraint part is reduced on the device only when an observer asks).

    In-place contract (`inplace_forces=True`, the default): half2 ADDS the restraint forces into the tensor
    `force_fn` returned, so `force_fn` must hand back a buffer it rewrites on every call (the calculators of this
    package do).  A `force_fn` that returns a cached / constant tensor, or keeps using its result, needs
    `inplace_forces=False`: the integrator then works on its own copy (one extra copy kernel per step).

    Fused ends (`fuse_tail=True`, the default): when `force_fn` is the `step` of a `ShardedFragmentForces` wired to the
    HIP engine, the first half also gathers the fragment geometry of the new positions and the second half also
    combines the fragment forces (`vsn_md_half1_build`, `vsn_md_combine_half2`): the same arithmetic in the same order
    - bitwise the same trajectory - in two launches fewer per step.  Any other `force_fn` is called as is."""

    def __init__(self, numbers, positions, force_fn, device, timestep_fs=1.0, temperature_K=300.0,
                 friction_per_fs=0.001, seed=0, tether_k=0.0, inplace_forces=True, fuse_tail=True, rng=None,
                 velocities=None):
        """rng / velocities: as in `Langevin` - with `rng` the two draws of every step are uploaded and the first half
        reads them instead of its counter-based generator (`vsn_md_set_noise`)."""
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
        self.inplace_forces = bool(inplace_forces)
        self._ff = self._fusable_evaluator(force_fn, device) if (fuse_tail and inplace_forces) else None
        self._fuse_relax = os.environ.get("VSN_MD_FUSE_RELAX", "1") != "0"  # A/B switch
        self.tether_k = float(tether_k)
        self._x0 = x0.astype(np.float64)
        self.constraints = []
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
        if velocities is not None:
            self.v = torch.as_tensor(np.asarray(velocities), dtype=torch.float32, device=device).contiguous()
        else:
            self.v = (torch.randn(self.n, 3, generator=gen, device=device) * torch.sqrt(kT / self.m)).contiguous()
        self.rng = rng
        if rng is not None:
            self._noise = torch.empty(2, self.n, 3, dtype=torch.float32, device=device)
            rc = self._L.vsn_md_set_noise(self._h, C.c_void_p(self._noise[0].data_ptr()),
                                          C.c_void_p(self._noise[1].data_ptr()))
            if rc:
                raise RuntimeError(f"vsn_md_set_noise failed ({rc})")
        self._obs = torch.zeros(4, dtype=torch.float32, device=device)
        self._init_observers(temperature_K)
        self._start_forces()
        self.steps = 0

    def _fusable_evaluator(self, force_fn, device):
        """the ShardedFragmentForces whose bound `step` is `force_fn`, if its HIP wiring (gather plan, combine plan,
        buffers on this device, one row per atom of this system) allows the fused ends; else None"""
        from .bonded import ShardedFragmentForces

        if os.environ.get("VSN_MD_FUSE", "1") == "0":  # A/B switch
            return None
        ff = getattr(force_fn, "__self__", None)
        if not isinstance(ff, ShardedFragmentForces) or getattr(force_fn, "__func__", None) is not ShardedFragmentForces.step:
            return None
        tail = ff.fused_tail
        if tail is None or tail[3].shape[0] != self.n or tail[3].device != torch.device(device):
            return None
        return ff

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _start_forces(self):
        C = self._C
        self.E_model, F = self.force_fn(self.x)
        if not self.inplace_forces:
            F = F.clone()  # the restraint kernel adds into F: never into a tensor the caller may own / cache
        self.F = F if F.is_contiguous() else F.contiguous()
        rc = self._L.vsn_md_restrain(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.F.data_ptr()), self._stream())
        if rc:
            raise RuntimeError(f"vsn_md_restrain failed ({rc})")

    def set_constraints(self, constraints):
        """Hookean list (plus the creation-time tether, if any) -> the device restraint lists"""
        C = self._C
        cons = list(constraints)
        self.constraints = cons
        pts = [c for c in cons if c.a2 is None]
        prs = [c for c in cons if c.a2 is not None]
        atom = [c.a1 for c in pts]
        org = [c.origin for c in pts]
        kp, rp = [c.k for c in pts], [c.rt for c in pts]
        if self.tether_k:
            atom += list(range(self.n))
            org += list(self._x0)
            kp += [self.tether_k] * self.n
            rp += [0.0] * self.n
        a = np.ascontiguousarray(atom, dtype=np.int64)
        o = np.ascontiguousarray(np.asarray(org, dtype=np.float32).reshape(-1, 3))
        k1, r1 = np.ascontiguousarray(kp, dtype=np.float32), np.ascontiguousarray(rp, dtype=np.float32)
        i1 = np.ascontiguousarray([c.a1 for c in prs], dtype=np.int64)
        i2 = np.ascontiguousarray([c.a2 for c in prs], dtype=np.int64)
        k2 = np.ascontiguousarray([c.k for c in prs], dtype=np.float32)
        r2 = np.ascontiguousarray([c.rt for c in prs], dtype=np.float32)
        fp = C.POINTER(C.c_float)
        from . import capi

        rc = self._L.vsn_md_set_restraints(self._h, len(a), capi.i64_ptr(a), o.ctypes.data_as(fp), k1.ctypes.data_as(fp),
                                           r1.ctypes.data_as(fp), len(i1), capi.i64_ptr(i1), capi.i64_ptr(i2),
                                           k2.ctypes.data_as(fp), r2.ctypes.data_as(fp))
        if rc:
            raise RuntimeError(f"vsn_md_set_restraints failed ({rc})")
        self._start_forces()  # forces of the current geometry under the new restraint set

    def _upload_noise(self):
        xi, eta = _draw(self.rng, self.n, "cpu")
        self._noise.copy_(torch.stack([xi, eta]))  # stream-ordered: lands before the first half reads it

    def _step_fused(self):
        """half1 + fragment gather | local evaluation + exchange | combine + half2"""
        C = self._C
        st = self._stream()
        if self.rng is not None:
            self._upload_noise()
        fp, cp, frag_pos, F_prot, E_tot = self._ff.fused_tail
        relaxer = getattr(self._ff, "relaxer", None)
        if relaxer is not None and self._fuse_relax:
            # + the cap-hydrogen relaxation of the gathered fragments: the whole start of the step in one launch
            rc = self._L.vsn_md_half1_build_relax(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                                  C.c_void_p(self.F.data_ptr()), fp, C.c_void_p(frag_pos.data_ptr()),
                                                  relaxer._h, st)
            built = 2
        else:
            rc = self._L.vsn_md_half1_build(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                            C.c_void_p(self.F.data_ptr()), fp, C.c_void_p(frag_pos.data_ptr()), st)
            built = 1
        if rc:
            raise RuntimeError(f"vsn_md_half1_build[_relax] failed ({rc})")
        buf = self._ff.exchange(self.x, prebuilt=built)
        rc = self._L.vsn_md_combine_half2(self._h, cp, C.c_void_p(buf.data_ptr()), C.c_void_p(F_prot.data_ptr()),
                                          C.c_void_p(E_tot.data_ptr()), C.c_void_p(self.x.data_ptr()),
                                          C.c_void_p(self.v.data_ptr()), st)
        if rc:
            raise RuntimeError(f"vsn_md_combine_half2 failed ({rc})")
        self.E_model, self.F = E_tot[0], F_prot
        self.steps += 1

    def step(self):
        if self._ff is not None:
            return self._step_fused()
        C = self._C
        st = self._stream()
        if self.rng is not None:
            self._upload_noise()
        rc = self._L.vsn_md_half1(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                  C.c_void_p(self.F.data_ptr()), st)
        if rc:
            raise RuntimeError(f"vsn_md_half1 failed ({rc})")
        self.E_model, F = self.force_fn(self.x)
        if not self.inplace_forces:
            F = F.clone()
        self.F = F if F.is_contiguous() else F.contiguous()
        rc = self._L.vsn_md_half2(self._h, C.c_void_p(self.x.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                  C.c_void_p(self.F.data_ptr()), st)
        if rc:
            raise RuntimeError(f"vsn_md_half2 failed ({rc})")
        self.steps += 1

    def _observe_device(self):
        """{Ekin, E_restraint, T} reduced on the device into self._obs (no host round trip)"""
        C = self._C
        rc = self._L.vsn_md_observe(self._h, C.c_void_p(self.v.data_ptr()), C.c_float(KB),
                                    C.c_void_p(self._obs.data_ptr()), self._stream())
        if rc:
            raise RuntimeError(f"vsn_md_observe failed ({rc})")
        return self._obs

    @property
    def E(self):
        """potential energy of the last evaluation = model + restraints (a 0-d device tensor)"""
        return self.E_model + self._observe_device()[1]

    def kinetic_energy(self):
        # a copy: _obs is rewritten by every later observe / E / kinetic_energy call
        return self._observe_device()[0].clone()

    def observe(self):
        o = self._observe_device()
        epot = float(self.E_model) + float(o[1])
        return epot, float(o[0]), float(o[2])

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.vsn_md_destroy(self._h)
                self._h = None
        except Exception:
            pass
