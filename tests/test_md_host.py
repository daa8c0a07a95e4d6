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
