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


@pytest.mark.parametrize("world,workload,steps", [(2, "chig_md", 40), (8, "chig_md", 25), (4, "ww_md", 20)])
def test_sharded_md_with_real_ranks_on_one_gpu(lib_built, world, workload, steps):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--workload",
                        workload, "--steps", str(steps), "--warmup", "3", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"] == "gloo"
    assert out["data"].startswith("SHARED-GPU VALIDATION RUN") and out["steps"] == steps and out["scaling"] == "strong"
    p = out["parity"]
    # step 0 through gather + cap-H + shard evaluation + all-gather + combine, against the reference-source golden
    assert p["pipeline_max_dF"] <= 1e-4 * max(1.0, p["max_abs_F"]) and p["max_dF_over_ranks"] <= 1e-4
    # this rank's shard really is a shard
    full = dict(chig_md=391, ww_md=1387)[workload]
    assert 0 < out["config"]["frag_atoms_local"] < full
