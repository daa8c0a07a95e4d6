"""f4 (device integrator) against ASE ITSELF - STAGED: ASE is not in this image, every test here is behind
`pytest.importorskip("ase")` and switches on the first time the suite runs where ASE is installed.

What the reference configures (/root/reference/src/AIMD/simulator.py):
    :96        MaxwellBoltzmannDistribution(atoms, temperature_K=T, rng=np.random.default_rng(seed))
    :108-116   Langevin(atoms, timestep=1 fs, temperature_K=T, friction=0.001 / fs, fixcm=True,
                        rng=RNGPool(seed, (n, 3), count=2))            (utils/utils.py:28-49: numpy normals)
    :139-180   Hookean(a1, a2, k, rt) restraints
ASE draws `xi`, then `eta` from `rng.standard_normal(size=(n, 3))` every step; both integrators of this package take the
same source (`rng=`, `vsn_md_set_noise` for the HIP one) and the same start velocities (`velocities=`), so trajectories
can be laid side by side step for step.  Without ASE the coefficients stay pinned only on our restatement of ASE's
published algorithm ("parity unpinned" in DESIGN.md); what IS pinned on the reference's code - RNGPool, draw count,
pre-equilibration ladder, friction, runaway threshold - is tests/test_md_host.py.  The HIP integrator against the torch
restatement on the same injected draws runs today (tests/test_gpu_md.py).
"""
import numpy as np
import pytest
import torch

ase = pytest.importorskip("ase", reason="ASE is not installed: the f4 pin against ase.md.langevin is staged")

K_SPRING = 3.0  # eV / A^2


def _system(n=24, seed=7):
    rng = np.random.default_rng(seed)
    numbers = rng.choice([1, 6, 7, 8], size=n)
    x0 = rng.standard_normal((n, 3)) * 2.0
    x = x0 + 0.05 * rng.standard_normal((n, 3))
    return numbers, x0, x


def _rng_pool(seed, n):
    """the reference's own noise source where its tree / oracle/_ref is present, else the same class restated inline"""
    try:
        from oracle.ref_caller import caller_source, load_reference_caller

        if caller_source() is not None:
            return load_reference_caller(lambda *a: None, object).utils.RNGPool(seed, (n, 3), 2)
    except Exception:
        pass
    return np.random.default_rng(seed)


def _ase_run(numbers, x0, x, v0, steps, seed, constraints=()):
    from ase import Atoms, units
    from ase.calculators.calculator import Calculator, all_changes
    from ase.md.langevin import Langevin

    class Harmonic(Calculator):
        implemented_properties = ["energy", "forces"]

        def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
            super().calculate(atoms, properties, system_changes)
            d = self.atoms.get_positions() - x0
            self.results = {"energy": 0.5 * K_SPRING * float((d * d).sum()), "forces": -K_SPRING * d}

    atoms = Atoms(numbers=numbers, positions=x)
    atoms.calc = Harmonic()
    atoms.set_velocities(v0)
    if constraints:
        atoms.set_constraint(list(constraints))
    dyn = Langevin(atoms, timestep=1.0 * units.fs, temperature_K=300.0, friction=0.001 / units.fs, fixcm=True,
                   rng=_rng_pool(seed, len(numbers)))
    traj = []
    for _ in range(steps):
        dyn.run(1)
        traj.append((atoms.get_positions().copy(), atoms.get_velocities().copy()))
    return traj


def _harmonic_force_fn(x0, device):
    x0_t = torch.as_tensor(x0, dtype=torch.float32, device=device)
    buf = {}

    def fn(x):
        d = x - x0_t
        buf["F"] = -K_SPRING * d
        return 0.5 * K_SPRING * (d * d).sum(), buf["F"]

    return fn


def test_unit_constants_are_ases():
    from ase import units

    from ai2bmd_amd import md

    assert md.FS == pytest.approx(units.fs, rel=1e-12) and md.KB == pytest.approx(units.kB, rel=1e-12)
    assert md.KCALMOL2EV == pytest.approx(units.kcal / units.mol, rel=1e-12)
    from ase.data import atomic_masses

    for z, m in md.MASSES.items():
        assert m == pytest.approx(float(atomic_masses[z]), rel=1e-6)


def test_maxwell_boltzmann_start_is_normals_times_sqrt_kT_over_m():
    """simulator.py:96: ASE's draw = rng.standard_normal((n, 3)) * sqrt(kT / m) - the law both integrators of this
    package use for their own start velocities (their generator differs; `velocities=` takes ASE's)"""
    from ase import Atoms, units
    from ase.md.velocitydistribution import MaxwellBoltzmannDistribution

    numbers, x0, x = _system()
    atoms = Atoms(numbers=numbers, positions=x)
    MaxwellBoltzmannDistribution(atoms, temperature_K=300.0, rng=np.random.default_rng(5))
    xi = np.random.default_rng(5).standard_normal((len(numbers), 3))
    np.testing.assert_allclose(atoms.get_velocities(), xi * np.sqrt(300.0 * units.kB / atoms.get_masses())[:, None],
                               rtol=1e-12)


def test_torch_langevin_follows_ase_langevin_on_the_same_draws():
    from ai2bmd_amd.md import KB, MASSES, Langevin

    numbers, x0, x = _system()
    m = np.array([MASSES[int(z)] for z in numbers])
    v0 = np.random.default_rng(2).standard_normal((len(numbers), 3)) * np.sqrt(300.0 * KB / m)[:, None]
    ref = _ase_run(numbers, x0, x, v0, steps=50, seed=11)
    md = Langevin(numbers, x, _harmonic_force_fn(x0, "cpu"), "cpu", seed=0, rng=_rng_pool(11, len(numbers)),
                  velocities=v0)
    for k, (xr, vr) in enumerate(ref):
        md.step()
        # fp32 state against ASE's fp64: round-off grows slowly over 50 steps of a stiff harmonic well
        np.testing.assert_allclose(md.x.numpy(), xr, rtol=0, atol=2e-5, err_msg=f"x, step {k}")
        np.testing.assert_allclose(md.v.numpy(), vr, rtol=0, atol=2e-4 * np.abs(vr).max(), err_msg=f"v, step {k}")


def test_hookean_law_is_ases():
    from ase import Atoms
    from ase.constraints import Hookean as AseHookean

    from ai2bmd_amd.md import Hookean, hookean_forces

    numbers, x0, x = _system(n=8)
    cases = [dict(a1=0, a2=1, k=2.0, rt=0.5), dict(a1=2, a2=(0.3, -1.0, 2.0), k=1.5, rt=0.0),
             dict(a1=3, a2=4, k=7.0, rt=100.0)]
    for c in cases:
        atoms = Atoms(numbers=numbers, positions=x)
        con = AseHookean(**c)
        f = np.zeros((len(numbers), 3))
        con.adjust_forces(atoms, f)
        e = con.adjust_potential_energy(atoms)
        E, F = hookean_forces(torch.as_tensor(x, dtype=torch.float64), [Hookean(c["a1"], c["a2"], c["k"], c["rt"])])
        np.testing.assert_allclose(F.numpy(), f, atol=1e-12)
        assert float(E) == pytest.approx(float(e), abs=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("with_restraint", [False, True])
def test_hip_langevin_follows_ase_langevin_on_the_same_draws(lib_built, with_restraint):
    """LangevinHIP (csrc/md.hip: two launches per step) beside ase.md.langevin.Langevin: the reference's RNGPool feeds
    both, same start velocities, harmonic forces; optionally one Hookean pair spring on both sides."""
    from ai2bmd_amd.md import KB, MASSES, Hookean, LangevinHIP

    numbers, x0, x = _system()
    m = np.array([MASSES[int(z)] for z in numbers])
    v0 = np.random.default_rng(2).standard_normal((len(numbers), 3)) * np.sqrt(300.0 * KB / m)[:, None]
    cons_ase, cons = (), []
    if with_restraint:
        from ase.constraints import Hookean as AseHookean

        cons_ase, cons = (AseHookean(a1=0, a2=5, k=3.0, rt=1.0),), [Hookean(0, 5, 3.0, rt=1.0)]
    ref = _ase_run(numbers, x0, x, v0, steps=50, seed=11, constraints=cons_ase)
    md = LangevinHIP(numbers, x, _harmonic_force_fn(x0, "cuda:0"), "cuda:0", seed=0, rng=_rng_pool(11, len(numbers)),
                     velocities=v0, inplace_forces=False)
    if cons:
        md.set_constraints(cons)
    for k, (xr, vr) in enumerate(ref):
        md.step()
        np.testing.assert_allclose(md.x.cpu().numpy(), xr, rtol=0, atol=2e-5, err_msg=f"x, step {k}")
        np.testing.assert_allclose(md.v.cpu().numpy(), vr, rtol=0, atol=2e-4 * np.abs(vr).max(), err_msg=f"v, step {k}")
