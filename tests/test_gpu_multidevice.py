"""GPU, STAGED FOR HARDWARE THE TEST BOXES DO NOT HAVE: these tests need >= 2 MI355X in one node and skip cleanly on a
1-GPU box (`torch.cuda.device_count() < 2`).  They switch on by themselves the first time `pytest -m gpu` runs on a
multi-GPU node and cover the two things that have never executed:

  (i)   RCCL with >= 2 ranks: `bench.py --gpus N` (one process per GPU, backend "nccl" = RCCL over xGMI), the sharded
        WW-domain / Chignolin MD step with ONE all-gather per step - golden parity on every rank before the clock and
        bit-identical trajectories on all ranks after the loop (bench.py asserts both and refuses to print otherwise);
  (ii)  the in-process multi-device mode the reference actually uses (Calculators/bonded.py:39-44 one model per device,
        :75-77 one thread per device): the REFERENCE's own DLBondedCalculator with `_bonded_devices = ["cuda:0",
        "cuda:1", ...]` on the HIP seam - one handle, stream, workspace and LDS attribute set per REAL device;
  (iii) the mirror class ai2bmd_amd.bonded.DLBondedCalculator on all devices of the node.

What already runs on one GPU: the same sharded path with 2 / 4 / 8 real ranks over gloo (tests/test_gpu_multirank.py),
two handles on `cuda:0` driven from the reference's thread pool (tests/test_gpu_reference_caller.py), a world-1 RCCL
all-gather (tests/test_gpu_pipeline.py).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
need2 = pytest.mark.skipif(NDEV < 2, reason=f"needs >= 2 GPUs in one node (this box has {NDEV})")


def _world_sizes():
    out = [2]
    if NDEV >= 4:
        out.append(4)
    if NDEV > 2 and NDEV not in out:
        out.append(NDEV)
    return out


def _run_ranks(world, workload, exchange):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", workload,
                        "--steps", "50", "--warmup", "5", "--no-secondary", "--no-cpu-baseline", "--exchange", exchange,
                        "--dump-state"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines  # the result line ALONE on stdout (RCCL's banner goes to stderr: bench.StdoutGuard)
    return json.loads(lines[-1])


@need2
@pytest.mark.parametrize("workload", ["ww_md", "chig_md"])
@pytest.mark.parametrize("world", _world_sizes())
def test_p2p_exchange_over_xgmi_is_the_rccl_collective_bit_for_bit(lib_built, world, workload):
    """the tuned exchange step across REAL devices: every rank stores its slot into its peers' IPC-mapped buffers over
    xGMI (csrc/p2p.hip) - same final state as the RCCL all-gather run, to the last bit"""
    if world > NDEV:
        pytest.skip(f"{world} ranks need {world} GPUs")
    a = _run_ranks(world, workload, "collective")
    b = _run_ranks(world, workload, "p2p")
    assert b["config"]["exchange"].startswith("p2p") and b["backend"] == "nccl"
    assert a["config"]["state_checksum"] == b["config"]["state_checksum"]
    p = b["parity"]
    assert p["pipeline_max_dF"] <= 1e-4 * max(1.0, p["max_abs_F"])


@need2
@pytest.mark.parametrize("workload", ["ww_md", "chig_md"])
@pytest.mark.parametrize("world", _world_sizes())
def test_sharded_md_over_rccl(lib_built, world, workload):
    """bench.py launches its own ranks (torch.distributed.run, 127.0.0.1) when started without RANK in the environment"""
    if world > NDEV:
        pytest.skip(f"{world} ranks need {world} GPUs")
    out = _run_ranks(world, workload, "collective")
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"] == "nccl"
    assert out["steps"] == 50 and out["scaling"] == "strong" and out["value"] > 0
    p = out["parity"]
    assert p["pipeline_max_dF"] <= 1e-4 * max(1.0, p["max_abs_F"]) and p["max_dF_over_ranks"] <= 1e-4
    full = dict(chig_md=391, ww_md=1387)[workload]
    assert 0 < out["config"]["frag_atoms_local"] < full


def _protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


@pytest.fixture(scope="module")
def ckpt_dir(lib_built, tmp_path_factory):
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict, write_lightning_ckpt

    td = tmp_path_factory.mktemp("ckpt_md")
    hp = default_hparams()
    write_lightning_ckpt(str(td / "visnet-uni-bench.ckpt"), hp, make_state_dict(hp, seed=2024))
    return str(td)


@need2
@pytest.mark.parametrize("name", ["chig", "ww"])
def test_reference_caller_with_one_handle_per_real_device(ckpt_dir, name):
    """the reference's DLBondedCalculator, unchanged, with every GPU of the node in DeviceStrategy._bonded_devices"""
    import threading

    from ai2bmd_amd.distancefrag import DistanceFragment
    from ai2bmd_amd.visnet_calculator import get_visnet_model
    from oracle.ref_caller import caller_source, load_reference_caller

    assert caller_source() is not None
    calls = []

    class Traced:
        def __init__(self, inner):
            self.inner, self.device = inner, inner.device

        def dl_potential_loader(self, frag_data):
            calls.append((self.device, threading.get_ident(), len(frag_data)))
            return self.inner.dl_potential_loader(frag_data)

    ref = load_reference_caller(lambda path, device: Traced(get_visnet_model(path, device)), DistanceFragment)
    DS = ref.DeviceStrategy
    devices = [f"cuda:{k}" for k in range(NDEV)]
    DS._gpu_count, DS._bonded_devices, DS._default_device, DS._chunk_size = NDEV, devices, "cuda:0", 120
    DS._optimiser_device = "cuda:0"
    calc = ref.DLBondedCalculator(ckpt_dir, "bench")
    assert [m.device for m in calc.models] == devices
    prot = _protein(name)
    calc.fragment_method.fragment(prot)
    DS.set_work_partitions(prot.fragments_start.tolist(), prot.fragments_end.tolist())
    E, F = calc(prot)
    used = {c[0] for c in calls}
    assert len(used) >= 2 and len({c[1] for c in calls}) >= 2  # several real devices, several executor threads
    g = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
    Fg, Eg = g["Fprot64_relaxed"], float(g["Eprot64_relaxed"])
    assert np.abs(F - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max()), np.abs(F - Fg).max()
    assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))
    # a second call (MD step 2): same handles, same answer bit for bit
    E2, F2 = calc(prot)
    assert np.array_equal(F, F2) and float(E) == float(E2)


@need2
def test_mirror_calculator_on_all_devices(ckpt_dir):
    """ai2bmd_amd.bonded.DLBondedCalculator(ckpt_path, ckpt_type)(prot) with DeviceStrategy holding every GPU"""
    from ai2bmd_amd.bonded import DLBondedCalculator
    from ai2bmd_amd.device_strategy import DeviceStrategy

    saved = (DeviceStrategy._gpu_count, list(DeviceStrategy._bonded_devices or []), DeviceStrategy._default_device,
             DeviceStrategy._chunk_size, getattr(DeviceStrategy, "_optimiser_device", None))
    try:
        DeviceStrategy._gpu_count = NDEV
        DeviceStrategy._bonded_devices = [f"cuda:{k}" for k in range(NDEV)]
        DeviceStrategy._default_device = DeviceStrategy._optimiser_device = "cuda:0"
        DeviceStrategy._chunk_size = 120
        calc = DLBondedCalculator(ckpt_dir, "bench")
        assert len(calc.models) == NDEV and len({m.device for m in calc.models}) == NDEV
        for name in ("chig", "ww"):
            prot = _protein(name)
            calc.fragment_method.fragment(prot)
            calc._work = None
            DeviceStrategy.set_work_partitions(prot.fragments_start.tolist(), prot.fragments_end.tolist())
            E, F = calc(prot)
            g = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
            Fg, Eg = g["Fprot64_relaxed"], float(g["Eprot64_relaxed"])
            assert np.abs(F - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max())
            assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))
    finally:
        (DeviceStrategy._gpu_count, DeviceStrategy._bonded_devices, DeviceStrategy._default_device,
         DeviceStrategy._chunk_size, DeviceStrategy._optimiser_device) = saved


@need2
def test_engine_on_device_one_matches_device_zero(lib_built):
    """per-device state (hipSetDevice at every C-ABI entry, LDS attributes per (device, kernel), stream / workspace owned
    by the handle): the same batch on cuda:0 and cuda:1 gives the same bits"""
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict
    from ai2bmd_amd.visnet_calculator import ViSNetModel

    g = np.load(os.path.join(GOLDEN, "visnet_prot_chig.npz"))
    hp = default_hparams()
    sd = make_state_dict(hp, seed=2024)
    fd = FragmentData(g["z"], g["pos_relaxed"], g["start"], g["end"], make_batch_index(g["start"], g["end"]))
    outs = [ViSNetModel(hp, sd, device=f"cuda:{k}").dl_potential_loader(fd) for k in range(2)]
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.abs(outs[1][1] - g["F_ref64_relaxed"]).max() <= 1e-4 * max(1.0, np.abs(g["F_ref64_relaxed"]).max())
