"""LAB: which earlier phase of the default bench makes the '+ MM non-bonded' secondary crawl at N = 2 on a shared GPU
(342 ms per step in gpurun_out/r05s).   torchrun --nproc-per-node 2 tools/lab/n2_mm_probe.py"""
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

args = bench.parse_args(["--gpus", os.environ["WORLD_SIZE"], "--share-gpu", "--steps", "20", "--warmup", "3"])
ctx = bench.Ctx(args)
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), ctx.dev)

def mm_run(tag):
    t0 = time.perf_counter()
    r, keep = bench.run_md(ctx, eng, hp, "chig", args, 40, 5, mm=True, tether_k=50.0)
    del keep
    if ctx.rank == 0:
        print(f"{tag}: MM secondary {r['ms_per_step']:.2f} ms/step (whole call {time.perf_counter() - t0:.1f} s)", flush=True)

mm_run("fresh")
r = bench.run_md(ctx, eng, hp, "trpcage", args, 40, 5)[0]
if ctx.rank == 0: print("trpcage", round(r["ms_per_step"], 2), flush=True)
mm_run("after trpcage")
r = bench.run_frag_batch(ctx, eng, hp, args, 2, 1)
if ctx.rank == 0: print("batch", round(r["ms_per_step"], 1), "mem GB", torch.cuda.mem_get_info(), flush=True)
mm_run("after frag_batch")
r = bench.run_frag_stream(ctx, eng, hp, args, conformations=8 * 4096 * ctx.world)
if ctx.rank == 0: print("stream", round(r["ms_per_step"], 1), flush=True)
mm_run("after frag_stream")
ctx.close()
