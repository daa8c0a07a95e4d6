"""Lab: one protein's fragment batch split over two engine handles on two streams with a fork / join per evaluation
(the GEMMs of one half next to the gather kernels of the other) against the single-lane evaluation.

    python tools/lab/lanes_md.py [chig|trpcage|ww|abd] [reps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ai2bmd_amd.device_strategy import device_ranges  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "chig"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = np.load(os.path.join(ROOT, "tests", "golden", f"visnet_prot_{name}.npz"))
hp = default_hparams()
sd = make_state_dict(hp, seed=2024)
dev = "cuda:0"
start, end = g["start"].astype(np.int64), g["end"].astype(np.int64)
z_all = torch.as_tensor(g["z"], dtype=torch.int64).to(dev)
p_all = torch.as_tensor(g["pos_relaxed"], dtype=torch.float32).to(dev)
e_all = torch.empty(len(start), device=dev)
f_all = torch.empty(len(g["z"]), 3, device=dev)


def bench(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


eng = ViSNetEngine(hp, sd, dev)
one = bench(lambda: eng.forces_device(z_all, p_all, start, end, e_all, f_all))
f_ref = f_all.clone()
print(f"{name}: one lane  {1e3 * one:.4f} ms per evaluation")
for lanes in (2, 3):
    engs = [eng] + [ViSNetEngine(hp, sd, dev) for _ in range(lanes - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    parts = []
    for f0, f1 in device_ranges(start, end, lanes):
        a0, a1 = int(start[f0]), int(end[f1 - 1])
        parts.append((z_all[a0:a1], p_all[a0:a1], start[f0:f1] - a0, end[f0:f1] - a0, e_all[f0:f1], f_all[a0:a1]))
    main = torch.cuda.current_stream(dev)
    ev_fork = torch.cuda.Event()
    ev_join = [torch.cuda.Event() for _ in range(lanes)]

    def step():
        ev_fork.record(main)
        for l in range(lanes):
            streams[l].wait_event(ev_fork)
            z, p, s, e_, eo, fo = parts[l]
            engs[l].forces_device(z, p, s, e_, eo, fo, stream=streams[l])
            ev_join[l].record(streams[l])
        for l in range(lanes):
            main.wait_event(ev_join[l])

    f_all.zero_()
    t = bench(step)
    torch.cuda.synchronize()
    print(f"{name}: {lanes} lanes {1e3 * t:.4f} ms per evaluation ({one / t:.3f}x), max|dF| vs one lane "
          f"{(f_all - f_ref).abs().max().item():.2e}")
