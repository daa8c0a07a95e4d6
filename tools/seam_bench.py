import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from ai2bmd_amd.bonded import DLBondedCalculator
from ai2bmd_amd.device_strategy import DeviceStrategy
from ai2bmd_amd.synthetic import default_hparams, make_state_dict, write_lightning_ckpt
hp = default_hparams()
td = tempfile.mkdtemp()
write_lightning_ckpt(os.path.join(td, "visnet-uni-bench.ckpt"), hp, make_state_dict(hp, seed=2024))
DeviceStrategy.initialize("small-molecule", "combined", "mm", gpu_count=1, chunk_size=9999)
calc = DLBondedCalculator(td, "bench")
for name in ("chig", "ww"):
    prot = bench.load_protein(name)
    calc.fragment_method.fragment(prot)
    DeviceStrategy.set_work_partitions(prot.fragments_start, prot.fragments_end)
    calc._work = None
    for _ in range(10): calc(prot)
    n = 200 if name == "chig" else 80
    t0 = time.perf_counter()
    for _ in range(n): E, F = calc(prot)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(n): fd = calc.fragment_method.get_fragments(prot)
    dt_f = time.perf_counter() - t1
    print(name, "DLBondedCalculator(prot) calls/s", round(n / dt, 1), "ms", round(1e3 * dt / n, 3), "| get_fragments ms", round(1e3 * dt_f / n, 3))
