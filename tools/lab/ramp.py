"""per-segment MD step rate over a long run (clock / power ramp of the GPU under this workload)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import load_protein
from ai2bmd_amd.amber import load_tables
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import build_plan
from ai2bmd_amd.hydrogen import build_hydrogen_plan
from ai2bmd_amd.md import LangevinHIP
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine
dev = "cuda:0"
hp = default_hparams(); eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), dev)
prot = load_protein("chig"); plan = build_plan(prot)
hplan = build_hydrogen_plan(prot, plan, load_tables("tests/golden/amber_tables.npz"))
ff = ShardedFragmentForces.for_engine(eng, plan, hydrogen=hplan)
md = LangevinHIP(prot.numbers, prot.positions, ff.step, dev, seed=0, tether_k=5.0)
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(seg): md.step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"steps {s*seg:5d}-{(s+1)*seg:5d}: {seg/el:7.1f} steps/s  (t = {time.perf_counter():.1f})", flush=True)
