"""GPU: the device-resident per-step pipeline (fragment gather + cap-H kernel ->
vsn_forces -> combine kernel) against the host statement of the same steps built
from the reference-shaped pieces (FragmentData + dl_potential_loader + the
numpy combiner), on the reference's Chignolin example."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.weights import default_hparams, make_state_dict

pytestmark = pytest.mark.gpu


def load_protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


@pytest.fixture(scope="module")
def setup(lib_built):
    from ai2bmd_amd.fragmentation import build_plan
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    hp = default_hparams(embedding_dimension=128, num_layers=3)
    sd = make_state_dict(hp, seed=21)
    prot = load_protein("chig")
    plan = build_plan(prot)
    model = ViSNetModel(hp, sd, device="cuda:0")
    return hp, sd, prot, plan, model


def test_chignolin_pipeline_matches_host_composition(setup):
    from ai2bmd_amd.bonded import DLBondedCalculator, ShardedFragmentForces, combine_numpy
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import combine_host, fragment_positions

    hp, sd, prot, plan, model = setup
    assert len(plan.start) == 19 and len(plan.z) == 391 and plan.n_prot == 175
    ff = ShardedFragmentForces.for_engine(model.engine, plan)
    x = torch.as_tensor(prot.positions, dtype=torch.float32, device="cuda:0")
    E, F = ff.step(x)
    torch.cuda.synchronize()
    # host composition, reference-shaped: FragmentData -> DLBondedCalculator.calculate -> combiner
    pos = fragment_positions(plan, prot.positions).astype(np.float32)
    fd = FragmentData(plan.z, pos, plan.start, plan.end, make_batch_index(plan.start, plan.end))
    calc = DLBondedCalculator.from_models([model])
    e_dip, f_dip, e_ace, f_ace = calc.calculate(fd)
    E_h, F_h = combine_numpy(plan.n_prot, e_dip, f_dip, e_ace, f_ace, plan.select_index, plan.origin_index)
    e_all, f_all = model.dl_potential_loader(fd)
    E_h2, F_h2 = combine_host(plan, e_all, f_all)
    assert abs(float(E) - float(E_h)) <= 1e-4 * max(1.0, abs(float(E_h)))
    assert abs(float(E_h) - E_h2) <= 1e-4 * max(1.0, abs(E_h2))
    np.testing.assert_allclose(F.cpu().numpy(), F_h, rtol=0, atol=2e-5)
    np.testing.assert_allclose(F_h, F_h2, rtol=0, atol=2e-5)


def test_chignolin_against_fp64_oracle(setup):
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import fragment_positions
    from oracle.visnet_oracle import ViSNetOracle

    hp, sd, prot, plan, model = setup
    pos = fragment_positions(plan, prot.positions).astype(np.float32)
    E64, F64, c = ViSNetOracle(hp, sd, torch.float64).energy_forces(plan.z, pos, plan.start, plan.end)
    e, f = model.dl_potential_loader(FragmentData(plan.z, pos, plan.start, plan.end,
                                                  make_batch_index(plan.start, plan.end)))
    assert np.abs(e - E64).max() <= 1e-5 * max(1.0, np.abs(E64).max())
    assert np.abs(f - F64).mean() <= 1e-5 * max(1.0, np.abs(F64).mean())
    assert np.abs(f - F64).max() <= 1e-4 * max(1.0, np.abs(F64).max())


def test_short_md_is_finite_and_reproducible(setup):
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.md import Langevin

    hp, sd, prot, plan, model = setup
    ff = ShardedFragmentForces.for_engine(model.engine, plan)

    def run():
        md = Langevin(prot.numbers, prot.positions, ff.step, "cuda:0", seed=3, tether_k=5.0)
        for _ in range(10):
            md.step()
        torch.cuda.synchronize()
        return md.x.cpu().numpy().copy(), float(md.E)

    x1, e1 = run()
    x2, e2 = run()
    assert np.isfinite(x1).all() and (x1 == x2).all() and e1 == e2
    assert np.abs(x1 - prot.positions).max() < 1.0  # tethered: stays near the start geometry


@pytest.mark.parametrize("relax,collective", [(False, False), (True, False), (True, True)])
def test_fused_integrator_ends_give_the_same_trajectory_bit_for_bit(setup, relax, collective):
    """LangevinHIP folds the fragment gather into its first half and the combine into its second
    (vsn_md_half1_build / vsn_md_combine_half2) when it drives a ShardedFragmentForces: two launches fewer, the same
    arithmetic in the same order - positions, velocities, forces, energies and observables equal bit for bit to the
    unfused sequence, with and without the cap-hydrogen relaxation and through the (one-rank) all-gather."""
    import torch.distributed as dist

    from ai2bmd_amd.amber import load_tables
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.hydrogen import build_hydrogen_plan
    from ai2bmd_amd.md import Hookean, LangevinHIP

    hp, sd, prot, plan, model = setup
    hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(GOLDEN, "amber_tables.npz"))) if relax else None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = collective and not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    out = {}
    try:
        for fuse in (True, False):
            ff = ShardedFragmentForces.for_engine(model.engine, plan, hydrogen=hplan, force_collective=collective)
            md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=11, tether_k=2.0, fuse_tail=fuse)
            assert (md._ff is not None) == fuse
            md.set_constraints([Hookean(0, 5, 3.0, rt=1.0)])
            rec = []
            for _ in range(12):
                md.step()
                rec.append((md.x.cpu().numpy().copy(), md.v.cpu().numpy().copy(), md.F.cpu().numpy().copy(),
                            float(md.E_model), md.observe()))
            out[fuse] = rec
    finally:
        if created:
            dist.destroy_process_group()
    for a, b in zip(out[True], out[False]):
        for u, w in zip(a[:3], b[:3]):
            assert np.isfinite(u).all() and np.array_equal(u, w)
        assert a[3] == b[3] and a[4] == b[4]


def test_two_handles_driven_from_two_threads(setup):
    """DLBondedCalculator drives one model per device from a thread pool (bonded.py:75-77).  Two handles on
    the same GPU stand in for two devices: partitions are evaluated concurrently and concatenated."""
    from ai2bmd_amd.bonded import DLBondedCalculator
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import fragment_positions
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    hp, sd, prot, plan, model = setup
    other = ViSNetModel(hp, sd, device="cuda:0")
    pos = fragment_positions(plan, prot.positions).astype(np.float32)
    fd = FragmentData(plan.z, pos, plan.start, plan.end, make_batch_index(plan.start, plan.end))
    one = DLBondedCalculator.from_models([model]).calculate(fd)
    two = DLBondedCalculator.from_models([model, other], chunk_atoms=120)
    two.set_work_partitions(fd.start, fd.end)
    assert {d for d, _, _ in two._work} == {0, 1} and len(two._work) >= 3
    for _ in range(3):
        res = two.calculate(fd)
        for a, b in zip(one, res):
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-5)


def test_rccl_all_gather_path_on_one_gpu(setup):
    """world_size-1 RCCL process group: the sharded evaluator goes through dist.all_gather_into_tensor on the GPU
    (same code path as N > 1, exercised here because the round's GPU box has one device)."""
    import torch.distributed as dist

    from ai2bmd_amd.bonded import ShardedFragmentForces

    hp, sd, prot, plan, model = setup
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        x = torch.as_tensor(prot.positions, dtype=torch.float32, device="cuda:0")
        E0, F0 = ShardedFragmentForces.for_engine(model.engine, plan).step(x)
        F0 = F0.clone()
        ff = ShardedFragmentForces.for_engine(model.engine, plan, force_collective=True)
        E1, F1 = ff.step(x)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(E0) == float(E1) and torch.equal(F0, F1)
    finally:
        if created:
            dist.destroy_process_group()


def test_streamed_conformations_equal_their_one_by_one_evaluation(lib_built):
    """bench.py's configs[4] pipe (run_frag_stream: pinned host batches, double-buffered H2D on a copy stream, evaluation,
    D2H, host-side checksums, all overlapped) against the same batches evaluated one at a time with nothing in flight:
    the float64 checksums of E and F agree to the last bit (same kernels, same inputs - a buffer reused too early or a
    result read too soon would show), every batch is consumed exactly once, and the golden block is re-checked in
    stream."""
    import argparse

    import bench
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict
    from ai2bmd_amd.visnet_calculator import ViSNetEngine

    hp = default_hparams()
    eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
    args = bench.parse_args(["--workload", "frag_stream", "--frags-per-gpu", "330", "--conformations", "2310"])
    ctx = argparse.Namespace(dev="cuda:0", rank=0, world=1, barrier=torch.cuda.synchronize, max_over_ranks=lambda v: v)
    keep = {}
    r = bench.run_frag_stream(ctx, eng, hp, args, golden_every=3, keep=keep)
    nb = r["steps"]
    assert nb == 7 and r["config"]["conformations"] == 7 * 330 and r["parity"]["golden_blocks_checked"] == 3
    assert r["parity"]["max_dF"] <= 1e-4
    z = torch.as_tensor(keep["z"], dtype=torch.int64).cuda()
    sE = sF = sF2 = 0.0
    for b in range(nb):
        pos = torch.as_tensor(keep["pos"][b]).cuda()
        e = torch.empty(len(keep["start"]), device="cuda:0")
        f = torch.empty(len(keep["z"]), 3, device="cuda:0")
        eng.forces_device(z, pos, keep["start"], keep["end"], e, f)
        torch.cuda.synchronize()
        sE += float(e.cpu().numpy().sum(dtype=np.float64))
        sF += float(np.abs(f.cpu().numpy()).sum(dtype=np.float64))
        sF2 += float(np.square(f.cpu().numpy(), dtype=np.float64).sum())
    c = r["config"]["checksum"]
    assert c["sum_E"] == sE and c["sum_absF"] == sF and c["sum_F2"] == sF2, (c, sE, sF, sF2)


def test_profile_mode_times_every_node_walk_with_its_byte_model(lib_built):
    """vsn_profile_read_walks (bench.py's roofline.hbm / roofline.reverse_walks): in profile mode every node walk of a
    single-protein evaluation - forward and reverse - carries an event pair on its own dispatch packet; launches per
    evaluation follow the launch sequence of DESIGN.md section 5, every record has a positive time and the
    algorithmic bytes of the formulas in csrc/engine.hip (= tools/walk_table.py), and the results are those of the
    unprofiled run bit for bit."""
    import sys

    from ai2bmd_amd.synthetic import default_hparams as dh, make_state_dict as msd
    from ai2bmd_amd.visnet_calculator import ViSNetEngine
    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from walk_table import alg_floats

    g = np.load(os.path.join(GOLDEN, "visnet_prot_chig.npz"))
    hp = dh()
    eng = ViSNetEngine(hp, msd(hp, seed=2024), "cuda:0")
    z = torch.as_tensor(g["z"], dtype=torch.int64).cuda()
    pos = torch.as_tensor(g["pos_relaxed"]).cuda()
    e0 = torch.empty(len(g["start"]), device="cuda:0")
    f0 = torch.empty(len(g["z"]), 3, device="cuda:0")
    eng.forces_device(z, pos, g["start"], g["end"], e0, f0)
    torch.cuda.synchronize()
    E = eng.last_num_edges()
    eng.set_option("profile", 1)
    e1, f1 = torch.empty_like(e0), torch.empty_like(f0)
    eng.forces_device(z, pos, g["start"], g["end"], e1, f1)
    torch.cuda.synchronize()
    w = eng.profile_read_walks()
    eng.set_option("profile", 0)
    assert torch.equal(e0, e1) and torch.equal(f0, f1)
    L, n = hp["num_layers"], len(g["z"])
    want = dict(k_edge_attn=L, k_node_update=L, k_bwd_hf1=L - 1, k_bwd_hf2=L - 2, k_bwd_attn_S=L, k_bwd_norm_update=L)
    assert {k: int(v["launches"]) for k, v in w.items()} == want
    assert all(v["ms"] > 0 and v["bytes"] > 0 for v in w.values())
    for k in ("k_bwd_hf2", "k_bwd_attn_S", "k_node_update"):       # launches of one kind only: the model to the byte
        assert w[k]["bytes"] == pytest.approx(4.0 * alg_floats(k, n, E) * want[k], rel=1e-12), k
    # k_bwd_hf1: L - 2 launches with the edge update, one (the last layer) without
    full = 4.0 * alg_floats("k_bwd_hf1", n, E)
    assert (L - 2) * full < w["k_bwd_hf1"]["bytes"] < (L - 1) * full


def test_p2p_exchange_with_one_rank_is_the_collective_free_step(setup):
    """`exchange="p2p"` at world == 1 (the rank stores its slot into its OWN gather buffer; csrc/p2p.hip): the same
    forces and energy as the collective-free step, bit for bit, over several steps (both halves of the double buffer),
    through the fused integrator ends too; no wait gives up."""
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.md import LangevinHIP

    hp, sd, prot, plan, model = setup
    x = torch.as_tensor(prot.positions, dtype=torch.float32, device="cuda:0")
    ref = ShardedFragmentForces.for_engine(model.engine, plan)
    p2p = ShardedFragmentForces.for_engine(model.engine, plan, exchange="p2p")
    assert p2p.p2p is not None and ref.p2p is None
    for k in range(5):
        xk = x + 0.01 * k
        E0, F0 = ref.step(xk)
        E0, F0 = float(E0), F0.clone()
        E1, F1 = p2p.step(xk)
        torch.cuda.synchronize()
        assert E0 == float(E1) and torch.equal(F0, F1), k
    p2p.p2p.check()
    out = {}
    for ex in ("collective", "p2p"):
        ff = ShardedFragmentForces.for_engine(model.engine, plan, exchange=ex)
        md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=11, tether_k=2.0)
        assert md._ff is not None
        for _ in range(9):
            md.step()
        torch.cuda.synchronize()
        out[ex] = (md.x.cpu().numpy().copy(), md.v.cpu().numpy().copy(), md.F.cpu().numpy().copy())
        if ff.p2p is not None:
            ff.p2p.check()
            ff.p2p.close()
    for a, b in zip(out["collective"], out["p2p"]):
        assert np.isfinite(a).all() and np.array_equal(a, b)
