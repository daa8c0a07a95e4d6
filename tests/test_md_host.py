"""CPU: the torch restatement of the MD loop pieces (ai2bmd_amd/md.py) - Hookean restraint law, observers,
pre-equilibration schedule.  PARITY UNPINNED: ASE is absent, these restate ASE 3.22's published behaviour; the tests
check self-consistency (force = -dE/dx, analytic cases), not ASE itself."""
import numpy as np
import pytest
import torch

from ai2bmd_amd.md import KB, KCALMOL2EV, Hookean, Langevin, TemperatureRunawayError, hookean_forces


def test_hookean_force_is_minus_gradient_of_its_energy_and_respects_the_threshold():
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((12, 3)) * 2, dtype=torch.float64, requires_grad=True)
    cons = [Hookean(0, 1, k=2.0, rt=0.5), Hookean(2, np.array([0.3, -1.0, 2.0]), k=1.5, rt=0.0),
            Hookean(3, 4, k=7.0, rt=100.0), Hookean(5, x[5].detach().numpy() + 0.05, k=3.0, rt=0.2)]
    E, F = hookean_forces(x, cons)
    (g,) = torch.autograd.grad(E, x)
    np.testing.assert_allclose(F.detach().numpy(), -g.numpy(), atol=1e-12)
    assert F[3].abs().max() == 0 and F[4].abs().max() == 0 and F[5].abs().max() == 0   # inside their thresholds
    # pair spring: equal and opposite; point spring pulls towards the point
    assert torch.allclose(F[0], -F[1]) and float((F[2] * (torch.tensor([0.3, -1.0, 2.0], dtype=torch.float64) - x[2])).sum()) > 0
    r = float(torch.linalg.norm(x[1] - x[0]))
    assert abs(float(torch.linalg.norm(F[0])) - 2.0 * (r - 0.5)) < 1e-12


def test_langevin_observers_and_preequilibration_schedule():
    rng = np.random.default_rng(1)
    n = 20
    numbers = rng.choice([1, 6, 8], size=n)
    pos = rng.standard_normal((n, 3)).astype(np.float32)

    def free(x):
        return torch.zeros(()), torch.zeros_like(x)

    md = Langevin(numbers, pos, free, "cpu", temperature_K=300.0, friction_per_fs=0.01, seed=3)
    stages = []
    md.attach(lambda: stages.append((md.nsteps, len(md.constraints), md.constraints[0].k if md.constraints else 0.0)), 1)
    md.pre_equilibrate(list(range(n)), preeq_steps=2)
    # ASE 3.22 Dynamics.irun: the observers also see the starting state once (nsteps == 0), then every step
    assert [s[0] for s in stages] == list(range(0, 11)) and all(s[1] == n for s in stages)
    ks = [round(s[2] / KCALMOL2EV, 6) for s in stages[1::2]]
    assert ks == [10, 5, 1, 0.5, 0.1] and md.constraints == []      # simulator.py:143
    epot, ekin, temp = md.observe()
    assert abs(temp - 2 * ekin / (3 * n * KB)) < 1e-9
    md.v = md.v * 4
    with pytest.raises(TemperatureRunawayError):
        md.printenergy(quiet=True)


def test_energy_includes_tether_and_constraints():
    numbers = [6, 1, 1]
    pos = np.array([[0, 0, 0], [1.1, 0, 0], [0, 1.2, 0]], np.float32)

    def zero(x):
        return torch.zeros(()), torch.zeros_like(x)

    md = Langevin(numbers, pos, zero, "cpu", temperature_K=0.0, tether_k=2.0)
    md.x = md.x + 0.1
    md.set_constraints([Hookean(1, 0, k=4.0, rt=1.0)])
    r = float(torch.linalg.norm(md.x[1] - md.x[0]))
    expect = 0.5 * 2.0 * 9 * 0.01 + 0.5 * 4.0 * (r - 1.0) ** 2
    assert abs(float(md.E) - expect) < 1e-6


def test_schedule_constants_and_noise_draws_match_the_reference_simulator():
    """f4 (integrator) is PARITY-UNPINNED on the Langevin coefficients (ASE absent); what CAN be pinned on the
    reference is pinned here, once: the noise source (utils/utils.py:28-49 `RNGPool`, executed from the reference tree
    or oracle/_ref), the number of normal draws per step (simulator.py:108: count=2), the pre-equilibration ladder and
    its unit conversion (simulator.py:139-156), the friction (simulator.py:114) and the temperature-runaway threshold
    (utils/utils.py:153-155) - the last four read out of the reference's code objects (the modules themselves import
    ASE and cannot be executed here)."""
    import pytest
    import torch

    from ai2bmd_amd import md
    from oracle.ref_caller import caller_source, load_reference_caller
    from oracle.ref_import import REFERENCE_SRC, reference_available

    if caller_source() is None:
        pytest.skip("reference not present")
    ref = load_reference_caller(lambda p, d: None, object)
    # ---- the noise source: a pool of `count` pre-drawn arrays, refilled by an observer after every step ----
    n = 7
    pool = ref.utils.RNGPool(seed=11, shape=(n, 3), count=2)
    plain = np.random.default_rng(11)
    for _step in range(3):
        xi, eta = pool.standard_normal((n, 3)), pool.standard_normal((n, 3))   # what ase Langevin.step asks for
        assert np.array_equal(xi, plain.standard_normal((n, 3))) and np.array_equal(eta, plain.standard_normal((n, 3)))
        assert len(pool.pool) == 0
        pool.fill()                                                           # MDObserver.fill_rng_pool, interval 1
        assert len(pool.pool) == 2
    # -> the reference's noise is the default_rng(seed) stream, two (n, 3) arrays per step, in order.  Ours:
    calls = []
    real = torch.randn

    def counting(*shape, **kw):
        calls.append(tuple(shape))
        return real(*shape, **kw)

    sim = md.Langevin(np.full(n, 6), np.random.default_rng(0).normal(size=(n, 3)),
                      lambda x: (torch.zeros(()), torch.zeros_like(x)), "cpu", seed=11)
    torch.randn = counting
    try:
        sim.step()
        sim.step()
    finally:
        torch.randn = real
    assert calls == [(n, 3)] * 4                                              # two draws of (n, 3) per step
    # ---- constants of simulate() / printenergy(), from the reference's code objects ----
    if reference_available():
        def consts_of(path, fn_name):
            def walk(co):
                if co.co_name == fn_name:
                    return co
                for c in co.co_consts:
                    if hasattr(c, "co_consts"):
                        r = walk(c)
                        if r is not None:
                            return r
                return None
            co = walk(compile(open(path).read(), path, "exec"))
            assert co is not None, fn_name
            return list(co.co_consts)

        import os

        cs = consts_of(os.path.join(REFERENCE_SRC, "AIMD", "simulator.py"), "simulate")
        ladder = [c for c in cs if isinstance(c, (int, float)) and not isinstance(c, bool) and c in (10, 5, 1, 0.5, 0.1)]
        assert [10, 5, 1, 0.5, 0.1] == ladder[:5] or (10, 5, 1, 0.5, 0.1) in cs
        assert 0.001 in cs and ("seed", "shape", "count") in cs and 2 in cs
        import inspect

        assert inspect.signature(md._MDBase.pre_equilibrate).parameters["restraints"].default == (10, 5, 1, 0.5, 0.1)
        assert inspect.signature(md.Langevin.__init__).parameters["friction_per_fs"].default == 0.001
        assert 1.5 in consts_of(os.path.join(REFERENCE_SRC, "utils", "utils.py"), "printenergy")
    # kcal/mol in eV with the CODATA 2014 values ASE 3.22 uses: the k of every pre-equilibration spring
    e_, nav, kb, amu = 1.6021766208e-19, 6.022140857e23, 1.38064852e-23, 1.660539040e-27
    assert abs(md.KCALMOL2EV - 4184.0 / nav / e_) < 1e-15
    assert abs(md.KB - kb / e_) < 1e-18 and abs(md.FS - 1e-5 * (e_ / amu) ** 0.5) < 1e-14   # one table for all three
    sim2 = md.Langevin(np.full(2, 6), np.zeros((2, 3)), lambda x: (torch.zeros(()), torch.zeros_like(x)), "cpu", seed=1)
    sim2.temp_k = 300.0
    sim2.observe = lambda: (0.0, 0.0, 450.1)
    with pytest.raises(md.TemperatureRunawayError):
        sim2.printenergy(quiet=True)
    sim2.observe = lambda: (0.0, 0.0, 449.9)
    sim2.printenergy(quiet=True)
