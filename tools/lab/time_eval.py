"""Lab tool (NOT a bench line): wall time of repeated ViSNet force evaluations of one protein's golden fragment batch,
for A/B and ablation builds selected with VSN_LIB.  No parity check - ablation builds compute wrong numbers on purpose.

    [VSN_LIB=...] python tools/lab/time_eval.py [chig|trpcage|ww|abd] [reps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "chig"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = np.load(os.path.join(ROOT, "tests", "golden", f"visnet_prot_{name}.npz"))
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
z = torch.as_tensor(g["z"], dtype=torch.int64).cuda()
p = torch.as_tensor(g["pos_relaxed"], dtype=torch.float32).cuda()
e = torch.empty(len(g["start"]), device="cuda")
f = torch.empty(len(g["z"]), 3, device="cuda")
for _ in range(20):
    eng.forces_device(z, p, g["start"], g["end"], e, f)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.forces_device(z, p, g["start"], g["end"], e, f)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{name}: {1e3 * dt:.4f} ms per evaluation ({1 / dt:.1f}/s), lib={os.environ.get('VSN_LIB', 'default')}")
