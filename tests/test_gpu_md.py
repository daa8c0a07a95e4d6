"""GPU: the device Langevin step (csrc/md.hip) against the torch restatement of ASE's Langevin
(ai2bmd_amd/md.py) - exactly in the noise-free limit, statistically with noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def harmonic(k):
    def fn(x):
        return 0.5 * k * (x * x).sum(), (-k * x).contiguous()

    return fn


def test_noise_free_limit_matches_torch_langevin(lib_built):
    from ai2bmd_amd.md import Langevin, LangevinHIP

    rng = np.random.default_rng(0)
    numbers = rng.choice([1, 6, 7, 8, 16], size=175)
    pos = rng.standard_normal((175, 3)).astype(np.float32)
    a = Langevin(numbers, pos, harmonic(2.0), "cuda:0", temperature_K=0.0, seed=1, tether_k=0.7)
    b = LangevinHIP(numbers, pos, harmonic(2.0), "cuda:0", temperature_K=0.0, seed=1, tether_k=0.7)
    # T = 0: no noise, zero initial velocities in both; give both the same kick
    v0 = torch.randn(175, 3, generator=torch.Generator().manual_seed(3)).to("cuda:0") * 0.05
    a.v, b.v = v0.clone(), v0.clone()
    for _ in range(50):
        a.step()
        b.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(b.x.cpu().numpy(), a.x.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(b.v.cpu().numpy(), a.v.cpu().numpy(), rtol=0, atol=2e-5)


def test_thermostat_reaches_target_temperature_and_keeps_com(lib_built):
    from ai2bmd_amd.md import KB, LangevinHIP

    rng = np.random.default_rng(1)
    n = 746
    numbers = rng.choice([1, 6, 7, 8], size=n)
    pos = (rng.standard_normal((n, 3)) * 5).astype(np.float32)
    md = LangevinHIP(numbers, pos, harmonic(0.0), "cuda:0", temperature_K=300.0, friction_per_fs=0.05, seed=5)
    com0 = (md.m * md.x).sum(0) / md.m.sum()
    temps = []
    for s in range(3000):
        md.step()
        if s >= 1000 and s % 10 == 0:
            temps.append(float(2.0 * md.kinetic_energy() / (3 * n * KB)))
    torch.cuda.synchronize()
    T = np.mean(temps)
    assert abs(T - 300.0) < 12.0, T  # free particles: <T> = 300 K, sigma ~ 300*sqrt(2/(3n))/sqrt(samples)
    # fixcm: the random displacements are centre-of-mass free (mean position drift only through v)
    assert torch.isfinite(md.x).all()
    # reproducible: same seed -> same trajectory
    md2 = LangevinHIP(numbers, pos, harmonic(0.0), "cuda:0", temperature_K=300.0, friction_per_fs=0.05, seed=5)
    for _ in range(20):
        md2.step()
    md3 = LangevinHIP(numbers, pos, harmonic(0.0), "cuda:0", temperature_K=300.0, friction_per_fs=0.05, seed=5)
    for _ in range(20):
        md3.step()
    assert (md2.x == md3.x).all()
    _ = com0


def _restraints(rng, n, pos):
    from ai2bmd_amd.md import Hookean

    cons = []
    for i in rng.choice(n, size=40, replace=False):          # position restraints, some with a dead zone
        cons.append(Hookean(a1=int(i), a2=pos[i] + rng.normal(0, 0.3, 3), k=float(rng.uniform(0.1, 3.0)),
                            rt=float(rng.choice([0.0, 0.2]))))
    heavy = rng.choice(n, size=25, replace=False)
    for i in heavy:                                           # bond restraints; atom `hub` carries several
        j = int((i + 1 + rng.integers(n - 1)) % n)
        cons.append(Hookean(a1=int(i), a2=j, k=float(rng.uniform(0.5, 5.0)), rt=float(rng.uniform(0.5, 2.5))))
    hub = int(heavy[0])
    for j in rng.choice([a for a in range(n) if a != hub], size=3, replace=False):
        cons.append(Hookean(a1=int(j), a2=hub, k=1.3, rt=0.9))
    return cons


def test_hookean_restraints_match_torch_restatement(lib_built):
    """Hookean point / pair springs with thresholds (simulator.py:139-180): forces and energy of the HIP per-atom
    lists against the torch loop, on the start geometry (vsn_md_restrain) and after steps (inside half2)."""
    from ai2bmd_amd.md import Langevin, LangevinHIP, hookean_forces

    rng = np.random.default_rng(4)
    n = 175
    numbers = rng.choice([1, 6, 7, 8, 16], size=n)
    pos = (rng.standard_normal((n, 3)) * 3).astype(np.float32)
    cons = _restraints(rng, n, pos.astype(np.float64))

    def zero(x):
        return torch.zeros((), device=x.device), torch.zeros_like(x)

    b = LangevinHIP(numbers, pos, zero, "cuda:0", temperature_K=0.0, seed=1)
    b.set_constraints(cons)
    torch.cuda.synchronize()
    E_t, F_t = hookean_forces(torch.as_tensor(pos, dtype=torch.float64), cons)
    np.testing.assert_allclose(b.F.cpu().numpy(), F_t.numpy(), rtol=0, atol=2e-5)
    assert abs(float(b.E) - float(E_t)) < 1e-4 * max(1.0, float(E_t))
    # the same restraints drive both integrators to the same trajectory (T = 0: no noise)
    a = Langevin(numbers, pos, zero, "cuda:0", temperature_K=0.0, seed=1)
    a.set_constraints(cons)
    for _ in range(40):
        a.step()
        b.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(b.x.cpu().numpy(), a.x.cpu().numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(b.F.cpu().numpy(), a.F.cpu().numpy(), rtol=0, atol=5e-5)
    assert abs(float(b.E) - float(a.E)) < 1e-4 * max(1.0, abs(float(a.E)))
    # bit-reproducible (per-atom lists, no atomics)
    c = LangevinHIP(numbers, pos, zero, "cuda:0", temperature_K=0.0, seed=1)
    c.set_constraints(cons)
    for _ in range(40):
        c.step()
    assert torch.equal(c.x, b.x) and torch.equal(c.F, b.F)


def test_energy_and_forces_include_the_tether_in_both_integrators(lib_built):
    from ai2bmd_amd.md import Langevin, LangevinHIP

    rng = np.random.default_rng(5)
    numbers = rng.choice([1, 6, 8], size=60)
    pos = rng.standard_normal((60, 3)).astype(np.float32)
    a = Langevin(numbers, pos, harmonic(1.5), "cuda:0", temperature_K=0.0, seed=2, tether_k=0.9)
    b = LangevinHIP(numbers, pos, harmonic(1.5), "cuda:0", temperature_K=0.0, seed=2, tether_k=0.9)
    v0 = torch.randn(60, 3, generator=torch.Generator().manual_seed(1)).to("cuda:0") * 0.05
    a.v, b.v = v0.clone(), v0.clone()
    for _ in range(10):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert abs(float(a.E) - float(b.E)) < 1e-4 * max(1.0, abs(float(a.E)))
    np.testing.assert_allclose(b.F.cpu().numpy(), a.F.cpu().numpy(), rtol=0, atol=2e-5)
    assert float(b.E) > float(b.E_model)  # the restraint energy is in E


def test_observers_preequilibration_and_runaway_guard(lib_built):
    """attach(fn, interval) / run(steps) like ASE's MolecularDynamics; the five restrained pre-equilibration stages
    (simulator.py:139-166); the temperature-runaway guard of the energy observer (utils/utils.py:143-159)."""
    from ai2bmd_amd.md import KB, LangevinHIP, TemperatureRunawayError

    rng = np.random.default_rng(6)
    n = 120
    numbers = rng.choice([1, 6, 7, 8], size=n)
    pos = (rng.standard_normal((n, 3)) * 4).astype(np.float32)
    md = LangevinHIP(numbers, pos, harmonic(0.0), "cuda:0", temperature_K=300.0, friction_per_fs=0.02, seed=9)
    seen, snaps = [], []
    md.attach(lambda: seen.append(md.nsteps), interval=5)
    md.attach(lambda: snaps.append(md.x.clone()), interval=10)      # trajectory frames stay in HBM
    md.attach(lambda: md.printenergy(quiet=True), interval=5)
    md.run(30)
    # ASE 3.22 Dynamics.irun: every observer also sees the starting state (nsteps == 0) once
    assert seen == [0, 5, 10, 15, 20, 25, 30] and len(snaps) == 4 and snaps[0].is_cuda
    epot, ekin, temp = md.observe()
    assert abs(temp - 2.0 * ekin / (3 * n * KB)) < 1e-3 * temp and abs(ekin - float(0.5 * (md.m * md.v ** 2).sum())) < 1e-3
    x_before = md.x.clone()
    md.pre_equilibrate(list(range(n)), preeq_steps=4)
    assert md.nsteps == 30 + 5 * 4 and md.constraints == [] and not torch.equal(md.x, x_before)
    hot = LangevinHIP(numbers, pos, harmonic(0.0), "cuda:0", temperature_K=300.0, seed=9)
    hot.v = hot.v * 3.0                                               # 9x the kinetic energy
    with pytest.raises(TemperatureRunawayError):
        hot.printenergy(quiet=True)


def test_constant_force_tensor_is_never_written_with_inplace_forces_off(lib_built):
    """`inplace_forces=False` (md.py): a force_fn that hands back a CACHED tensor keeps it untouched - at
    construction, through repeated set_constraints (each re-evaluates the start forces and adds the restraint
    forces) and through steps - and the integrator's own forces are cache + restraints, not an accumulation."""
    from ai2bmd_amd.md import Hookean, LangevinHIP, hookean_forces

    rng = np.random.default_rng(11)
    n = 64
    numbers = rng.choice([1, 6, 7, 8], size=n)
    pos = (rng.standard_normal((n, 3)) * 2).astype(np.float32)
    cached = torch.as_tensor(rng.standard_normal((n, 3)).astype(np.float32), device="cuda:0")
    keep = cached.clone()

    def fn(x):
        return torch.zeros((), device=x.device), cached

    md = LangevinHIP(numbers, pos, fn, "cuda:0", temperature_K=0.0, seed=1, tether_k=0.8, inplace_forces=False)
    assert torch.equal(cached, keep)
    cons = [Hookean(a1=int(i), a2=np.asarray(pos[i] + 0.3, dtype=np.float64), k=2.0, rt=0.0) for i in range(0, n, 4)]
    for _ in range(3):
        md.set_constraints(cons)
        torch.cuda.synchronize()
        assert torch.equal(cached, keep)
    # forces the integrator holds = cached + restraints of the start geometry (tether at x0 gives zero there)
    _, F_r = hookean_forces(torch.as_tensor(pos, dtype=torch.float64), cons)
    np.testing.assert_allclose(md.F.cpu().numpy(), keep.cpu().numpy() + F_r.numpy(), rtol=0, atol=2e-5)
    for _ in range(5):
        md.step()
    torch.cuda.synchronize()
    assert torch.equal(cached, keep)


@pytest.mark.parametrize("fused", [False, True])
def test_injected_draws_hip_integrator_follows_the_torch_restatement_step_for_step(lib_built, fused):
    """With noise: both integrators take the step's (xi, eta) from the SAME source - the reference's own RNGPool
    (utils/utils.py:28-49, executed from the reference tree / oracle/_ref; simulator.py:108 hands it to ASE) - and the
    same start velocities, so the trajectories can be compared step for step instead of statistically: the HIP
    kernels (`vsn_md_set_noise` -> k_md_half1[_build] / k_md_[combine_]half2) against ~25 torch elementwise ops.
    `fused`: through the fused ends on a real ShardedFragmentForces (Chignolin, H = 64) instead of a harmonic well.
    (The same comparison against ase.md.langevin.Langevin itself is staged in tests/test_md_vs_ase.py.)"""
    import os

    from ai2bmd_amd.md import KB, MASSES, Hookean, Langevin, LangevinHIP
    from oracle.ref_caller import caller_source, load_reference_caller

    assert caller_source() is not None
    utils = load_reference_caller(lambda *a: None, object).utils
    if fused:
        from ai2bmd_amd.bonded import ShardedFragmentForces
        from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan
        from ai2bmd_amd.synthetic import default_hparams, make_state_dict
        from ai2bmd_amd.visnet_calculator import ViSNetEngine
        from conftest import GOLDEN

        d = np.load(os.path.join(GOLDEN, "protein_chig.npz"))
        prot = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))
        hp = default_hparams(embedding_dimension=64, num_layers=2)
        eng = ViSNetEngine(hp, make_state_dict(hp, seed=9), "cuda:0")
        numbers, pos = prot.numbers, prot.positions.astype(np.float32)
        plan = build_plan(prot)
        fa = ShardedFragmentForces.for_engine(eng, plan).step
        fb = ShardedFragmentForces.for_engine(eng, plan).step
    else:
        rng = np.random.default_rng(0)
        numbers = rng.choice([1, 6, 7, 8, 16], size=175)
        pos = rng.standard_normal((175, 3)).astype(np.float32)
        fa = fb = harmonic(2.0)
    n = len(numbers)
    m = np.array([MASSES[int(z)] for z in numbers])
    v0 = np.random.default_rng(4).standard_normal((n, 3)) * np.sqrt(300.0 * KB / m)[:, None]
    a = Langevin(numbers, pos, fa, "cuda:0", seed=1, tether_k=0.7, rng=utils.RNGPool(21, (n, 3), 2), velocities=v0)
    b = LangevinHIP(numbers, pos, fb, "cuda:0", seed=1, tether_k=0.7, rng=utils.RNGPool(21, (n, 3), 2), velocities=v0,
                    inplace_forces=fused)
    assert (b._ff is not None) == fused
    cons = [Hookean(0, 5, 3.0, rt=1.0)]
    a.set_constraints(cons)
    b.set_constraints(cons)
    # harmonic well: round-off only; the network's forces amplify the integrators' last-bit differences a little
    tol = 1e-4 if fused else 2e-5
    for k in range(40):
        a.step()
        b.step()
        np.testing.assert_allclose(b.x.cpu().numpy(), a.x.cpu().numpy(), rtol=0, atol=tol, err_msg=f"x, step {k}")
        np.testing.assert_allclose(b.v.cpu().numpy(), a.v.cpu().numpy(), rtol=0, atol=tol, err_msg=f"v, step {k}")
    # the draws really entered: the same run on the built-in generator differs
    c = LangevinHIP(numbers, pos, fb, "cuda:0", seed=1, tether_k=0.7, velocities=v0, inplace_forces=fused)
    c.set_constraints(cons)
    for _ in range(40):
        c.step()
    assert np.abs(c.x.cpu().numpy() - b.x.cpu().numpy()).max() > 1e-3
