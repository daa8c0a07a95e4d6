"""LAB: one MD step captured in a HIP graph (torch.cuda.CUDAGraph) against plain stream launches, for the whole protein
and for one rank's share of an 8-rank job (emulated).  The integrator's noise counter is a kernel ARGUMENT, so a
replayed graph repeats the same noise: a TIMING experiment only.    python tools/lab/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from ai2bmd_amd.amber import load_tables
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import build_plan
from ai2bmd_amd.hydrogen import build_hydrogen_plan
from ai2bmd_amd.md import LangevinHIP
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
for pname in ("chig", "ww"):
    prot = bench.load_protein(pname)
    plan = build_plan(prot)
    hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(bench.GOLD, "amber_tables.npz")))
    for r, w in ((0, 1), (0, 8)):
        ff = ShardedFragmentForces.for_engine(eng, plan, rank=r, world=w, hydrogen=hplan)
        ff.emulate = w > 1
        md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=0, tether_k=5.0)
        for _ in range(30):
            md.step()
        torch.cuda.synchronize()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            md.step()
        torch.cuda.synchronize()
        eager = 1e3 * (time.perf_counter() - t0) / n
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(s):
                for _ in range(3):
                    md.step()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                md.step()
            torch.cuda.synchronize()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            graph = 1e3 * (time.perf_counter() - t0) / n
            ok = bool(torch.isfinite(md.x).all())
        except Exception as e:
            print(f"{pname} rank {r}/{w}: eager {eager:.3f} ms/step; capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
            raise SystemExit(1)
        print(f"{pname} rank {r}/{w}: eager {eager:.3f} ms/step, graph replay {graph:.3f} ms/step ({ok})", flush=True)
        del md, ff
