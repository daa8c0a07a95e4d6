"""Build container only (/root/reference present): time ratio of the oracle port (oracle/visnet_oracle.py) to the
REFERENCE's own ViSNet source (oracle/ref_import.py + shims) on the Chignolin fragment batch, fp32, same threads -
the factor between a `cpu_baseline.kind = "port"` figure (what the GPU box can time) and the reference's CPU path."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ai2bmd_amd.fragmentation import build_plan, fragment_positions  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from oracle.visnet_oracle import ViSNetOracle  # noqa: E402

hp = default_hparams()
sd = make_state_dict(hp, seed=2024)
prot = bench.load_protein("chig")
plan = build_plan(prot)
pos = fragment_positions(plan, prot.positions).astype(np.float32)
kind, ref = bench._cpu_evaluator(hp, sd)
assert kind == "reference", "needs /root/reference"
port = ViSNetOracle(hp, sd, torch.float32).energy_forces
nt = min(8, os.cpu_count() or 1)
torch.set_num_threads(nt)
res = {}
for name, fn in (("reference", ref), ("port", port)):
    fn(plan.z, pos, plan.start, plan.end)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 12:
        fn(plan.z, pos, plan.start, plan.end)
        n += 1
    res[name] = (time.perf_counter() - t0) / n
print(f"threads={nt} reference {res['reference']:.3f} s/eval, port {res['port']:.3f} s/eval, "
      f"port/reference = {res['port'] / res['reference']:.3f}")
