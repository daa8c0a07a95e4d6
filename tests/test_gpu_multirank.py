"""GPU: the N > 1 product path with N REAL ranks on the one GPU of the test box.

RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the boxes have one GPU - so `bench.py --gpus N
--share-gpu` runs the ranks over gloo with DEVICE tensors (gloo stages through the host): everything of the sharded
path is the product's own - one process and one HIP engine per rank, the reference's partition rule, the fragment
gather + cap-hydrogen relaxation on every rank, each rank's shard evaluated by the HIP kernels straight into its slot
of the exchange buffer, ONE all-gather per step, the remapped combine, the fused integrator halves - only the
transport under `all_gather_into_tensor` is not RCCL.  Before its clock starts bench.py checks the recombined protein
forces of step 0 against the reference-source golden on EVERY rank (a wrong slot, offset or remap fails there), and
after the loop that all ranks still hold bit-identical trajectories."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_shared(world, workload, steps, exchange):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--workload",
                        workload, "--steps", str(steps), "--warmup", "3", "--no-secondary", "--no-cpu-baseline",
                        "--exchange", exchange, "--dump-state"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("world,workload,steps", [(2, "chig_md", 40), (8, "chig_md", 25), (4, "ww_md", 20)])
def test_sharded_md_with_real_ranks_on_one_gpu(lib_built, world, workload, steps):
    out = _run_shared(world, workload, steps, "collective")
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"] == "gloo"
    assert out["data"].startswith("SHARED-GPU VALIDATION RUN") and out["steps"] == steps and out["scaling"] == "strong"
    p = out["parity"]
    # step 0 through gather + cap-H + shard evaluation + all-gather + combine, against the reference-source golden
    assert p["pipeline_max_dF"] <= 1e-4 * max(1.0, p["max_abs_F"]) and p["max_dF_over_ranks"] <= 1e-4
    # this rank's shard really is a shard
    full = dict(chig_md=391, ww_md=1387)[workload]
    assert 0 < out["config"]["frag_atoms_local"] < full


@pytest.mark.parametrize("world,workload,steps", [(2, "chig_md", 40), (8, "chig_md", 25), (4, "ww_md", 20)])
def test_p2p_exchange_is_the_collective_bit_for_bit(lib_built, world, workload, steps):
    """The tuned exchange step (csrc/p2p.hip): every rank stores its slot straight into its peers' gather
    buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle mappings between the rank processes - here all on one device),
    flags + waits in the same launch, double-buffered by step parity.  Same sharded MD run as above with
    `--exchange p2p`: golden parity on every rank at step 0, all ranks bit-identical after the loop (bench.py asserts
    both, and that no wait gave up), AND the final state equals the gloo-collective run's to the last bit - a copy is a
    copy, whichever transport carried it."""
    a = _run_shared(world, workload, steps, "collective")
    b = _run_shared(world, workload, steps, "p2p")
    assert b["config"]["exchange"].startswith("p2p") and a["config"]["exchange"] == "all_gather_into_tensor"
    assert b["n_gpus"] == world and b["steps"] == steps
    p = b["parity"]
    assert p["pipeline_max_dF"] <= 1e-4 * max(1.0, p["max_abs_F"])
    assert a["config"]["state_checksum"] == b["config"]["state_checksum"], (a["config"]["state_checksum"],
                                                                             b["config"]["state_checksum"])


WORKER_EMPTY = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ai2bmd_amd.amber import load_tables
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan
from ai2bmd_amd.hydrogen import build_hydrogen_plan
from ai2bmd_amd.md import LangevinHIP
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
gold = os.path.join(sys.argv[1], "tests", "golden")
d = np.load(os.path.join(gold, "protein_chig.npz"))
m = (d["resnums"] <= 4) | (d["resnums"] == 12)      # ACE-TYR-TYR-ASP-NME: 5 fragments for 8 ranks
rn = d["resnums"][m].copy(); rn[rn == 12] = 5
pos0 = d["positions"][m].astype(np.float64).copy()
pos0[rn == 5] += pos0[(rn == 4) & (d["names"][m] == "C")][0] - pos0[(rn == 5) & (d["names"][m] == "N")][0] + 1.3
prot = ProteinAtoms(d["names"][m], d["resnames"][m], rn, d["numbers"][m], pos0)
plan = build_plan(prot)
hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(gold, "amber_tables.npz")))
hp = default_hparams(embedding_dimension=128, num_layers=3)
eng = ViSNetEngine(hp, make_state_dict(hp, seed=5), "cuda:0")
x = torch.as_tensor(prot.positions, dtype=torch.float32, device="cuda:0")
for hyd in (None, hplan):
    ff = ShardedFragmentForces.for_engine(eng, plan, rank=rank, world=world, hydrogen=hyd)
    one = ShardedFragmentForces.for_engine(eng, plan, rank=0, world=1, hydrogen=hyd)
    E, F = ff.step(x)
    E1, F1 = one.step(x)
    torch.cuda.synchronize()
    assert torch.isfinite(F).all() and torch.allclose(F, F1, rtol=0, atol=2e-5), (rank, float((F - F1).abs().max()))
    assert abs(float(E) - float(E1)) <= 1e-4 * max(1.0, abs(float(E1)))
    owns = ff.f1 - ff.f0
    assert (ff.fused_tail is None) == (owns == 0)
# MD: ranks that own nothing integrate through the unfused halves, the others through the fused ones - same bits
ff = ShardedFragmentForces.for_engine(eng, plan, rank=rank, world=world, hydrogen=hplan)
md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=3, tether_k=5.0)
for _ in range(25):
    md.step()
torch.cuda.synchronize()
chk = torch.stack([md.x.double().sum(), md.v.double().sum(), md.F.double().sum()])
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi), (rank, (hi - lo).tolist())
print(f"rank {rank} ok frags={ff.f1 - ff.f0} fused={md._ff is not None}", flush=True)
dist.destroy_process_group()
'''


def test_ranks_that_own_nothing_with_real_engines(lib_built, tmp_path):
    """Five fragments on eight real ranks (one GPU, gloo): three ranks own no fragment - zero-row plans and views, no
    engine call - and still enter the all-gather; every rank's recombined forces equal the single-rank evaluation, and
    after 25 Langevin steps (fused halves on the owning ranks, unfused on the empty ones) all eight trajectories are
    bit-identical."""
    script = tmp_path / "worker_empty.py"
    script.write_text(WORKER_EMPTY)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), ROOT],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(8):
        assert f"rank {k} ok" in r.stdout
    assert r.stdout.count("frags=0 fused=False") == 3 and r.stdout.count("fused=True") == 5


WORKER_BATCH = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine
G = os.path.join(sys.argv[1], "tests", "golden")
pool = []
for name in ("chig", "trpcage", "ww", "abd"):
    g = np.load(os.path.join(G, f"visnet_prot_{name}.npz"))
    for a, b in zip(g["start"], g["end"]):
        pool.append((g["z"][a:b], g["pos_relaxed"][a:b]))
rng = np.random.default_rng(7)
zs, ps, sizes = [], [], []
for i in range(2048):
    z, p = pool[i % len(pool)]
    zs.append(z); sizes.append(len(z)); ps.append(p if i < len(pool) else p + rng.normal(0, 0.05, size=p.shape))
end = np.cumsum(sizes); start = end - np.asarray(sizes)
z = torch.as_tensor(np.concatenate(zs), dtype=torch.int64).cuda()
pos = torch.as_tensor(np.concatenate(ps).astype(np.float32)).cuda()
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
open(sys.argv[2] + ".ready", "w").close()          # both processes evaluate at the same time
t0 = time.time()
while not os.path.exists(sys.argv[3] + ".ready") and time.time() - t0 < 300:
    time.sleep(0.05)
outs = []
for r in range(int(sys.argv[4])):
    e = torch.empty(len(start), device="cuda:0"); f = torch.empty(len(z), 3, device="cuda:0")
    eng.forces_device(z, pos, start, end, e, f)
    torch.cuda.synchronize()
    outs.append((e.cpu().numpy(), f.cpu().numpy()))
bad = [r for r in range(1, len(outs)) if not (np.array_equal(outs[r][0], outs[0][0]) and np.array_equal(outs[r][1], outs[0][1]))]
np.savez(sys.argv[2] + ".npz", e=outs[0][0], f=outs[0][1], bad=np.asarray(bad, dtype=np.int64))
'''


def test_batch_evaluation_is_bit_reproducible_next_to_a_second_process_on_the_gpu(lib_built, tmp_path):
    """Two PROCESSES on one GPU, each evaluating the same 2048-fragment batch again and again while the other does the
    same: every evaluation of both must be the same bits.  Found in round 6 (LAB_NOTES section 15,
    tools/lab/pk_micro.hip): on this hardware `v_pk_mul_f32 ... op_sel:[0,1]` returns a wrong low half in lanes 48..63
    now and then while ANOTHER process runs kernels on the device; a build with packed fp32 VALU instructions had 7 of 8
    such evaluations differ (max |dF| 2e-2 eV/A, zeroed channels in the attention messages).  The library is compiled
    without them (ai2bmd_amd/build.py NO_PACKED_FP32)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER_BATCH)
    tags = [str(tmp_path / "a"), str(tmp_path / "b")]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, tags[i], tags[1 - i], "8"], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    import numpy as np
    a, b = (np.load(t + ".npz") for t in tags)
    assert a["bad"].size == 0 and b["bad"].size == 0, (a["bad"], b["bad"])
    assert np.array_equal(a["e"], b["e"]) and np.array_equal(a["f"], b["f"])  # and the two processes agree
