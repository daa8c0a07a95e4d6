import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import load_protein
from ai2bmd_amd.fragmentation import build_plan, fragment_positions
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import default_hparams, make_state_dict
hp = default_hparams(); sd = make_state_dict(hp, seed=2024)
prot = load_protein("chig"); plan = build_plan(prot)
pos = fragment_positions(plan, prot.positions).astype(np.float32)
o = ViSNetOracle(hp, sd, torch.float32)
print("lscpu:", os.popen("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket'").read())
for nt in [8, 16, 32, 64, 128]:
    torch.set_num_threads(nt)
    o.energy_forces(plan.z, pos, plan.start, plan.end)
    t = time.perf_counter(); o.energy_forces(plan.z, pos, plan.start, plan.end); print(nt, "threads:", time.perf_counter() - t, "s")
