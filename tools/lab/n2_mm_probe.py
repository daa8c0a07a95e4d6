"""LAB: which earlier phase of the default bench makes the '+ MM non-bonded' secondary crawl at N = 2 on a shared GPU
(342 ms per step in gpurun_out/r05s).   torchrun --nproc-per-node 2 tools/lab/n2_mm_probe.py

FOUND (round 5): not the MM term.  After run_frag_stream - the first phase that opens a third HIP stream per process
(the copy stream, next to the launch stream and the engine's side stream, plus gloo's pool streams) - EVERY MD step of
both ranks takes 0.3-1.4 s, fused or unfused integrator, with or without MM, and emptying the caches does not help;
before it every phase runs at its normal rate (4.6 ms per step).  Two processes on ONE device then hold more HIP
streams than the device has hardware queues and the queues are time-sliced between the processes.  An artefact of
the --share-gpu validation mode (one process per device, the product layout, never shares a device's queues; N = 1
runs the same phases at full rate), recorded here so that nobody reads a shared-GPU timing as anything."""
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
r = bench.run_frag_stream(ctx, eng, hp, args, conformations=8 * 4096 * ctx.world)
if ctx.rank == 0: print("stream", round(r["ms_per_step"], 1), flush=True)
r = bench.run_md(ctx, eng, hp, "chig", args, 40, 5)[0]
if ctx.rank == 0: print("plain chig (fused integrator) after frag_stream", round(r["ms_per_step"], 2), flush=True)
os.environ["VSN_MD_FUSE"] = "0"
r = bench.run_md(ctx, eng, hp, "chig", args, 40, 5)[0]
os.environ["VSN_MD_FUSE"] = "1"
if ctx.rank == 0: print("plain chig (unfused integrator) after frag_stream", round(r["ms_per_step"], 2), flush=True)
mm_run("after frag_stream")
import gc
gc.collect(); torch.cuda.empty_cache()
try:
    torch._C._host_emptyCache()
except Exception as e:
    if ctx.rank == 0: print("no _host_emptyCache", e)
mm_run("after gc + empty caches")
ctx.close()
