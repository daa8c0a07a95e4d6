"""Per-rank step time of an N-rank MD job, measured on ONE GPU: every rank r of N in {2, 4, 8} is emulated in turn
(`ShardedFragmentForces.emulate`: rank r's shard, the collective replaced by a copy) and the job's step is the MAX
over its ranks + the exchange step.  The exchange is no longer a guess: `--exchange-us` (default: the MEASURED software
cost of one `all_gather_into_tensor` per step through torch.distributed's RCCL process group, bench.py
`config.rccl1_allgather`, 9.8 us on Chignolin and on the WW domain; the tuned direct-write exchange csrc/p2p.hip is
measured beside it) is added to every multi-rank job step - the wire time of 2-3 KB over xGMI is not in it (no
multi-GPU box).  Both partition rules: "atoms" (the reference's, device_strategy.py:84-127) and "cost"
(edge-balanced).   python tools/shard_table.py [chig ww] [--exchange-us 9.8] > table.md"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ai2bmd_amd.amber import load_tables  # noqa: E402
from ai2bmd_amd.bonded import ShardedFragmentForces  # noqa: E402
from ai2bmd_amd.fragmentation import build_plan  # noqa: E402
from ai2bmd_amd.hydrogen import build_hydrogen_plan  # noqa: E402
from ai2bmd_amd.md import LangevinHIP  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402


def step_ms(eng, prot, plan, hplan, r, w, balance, steps):
    ff = ShardedFragmentForces.for_engine(eng, plan, rank=r, world=w, hydrogen=hplan, balance=balance)
    ff.emulate = w > 1
    md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=0, tether_k=5.0)
    for _ in range(20):
        md.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        md.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, ff.local_rows, int(ff.f1 - ff.f0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("proteins", nargs="*", default=["chig", "ww"])
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--exchange-us", type=float, default=9.8,
                    help="measured per-step cost of the exchange step (bench.py config.rccl1_allgather.*.delta_us)")
    a = ap.parse_args()
    hp = default_hparams()
    eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
    tables = load_tables(os.path.join(bench.GOLD, "amber_tables.npz"))
    print(f"| protein | N | rule | per-rank step ms (rank 0..N-1) | fragment atoms per rank | max over ranks ms | "
          f"+ exchange ({a.exchange_us:g} us, measured) = job step ms | speed-up |")
    print("|---|---|---|---|---|---|---|---|")
    for pname in a.proteins:
        prot = bench.load_protein(pname)
        plan = build_plan(prot)
        hplan = build_hydrogen_plan(prot, plan, tables)
        t1, _, _ = step_ms(eng, prot, plan, hplan, 0, 1, "atoms", a.steps)
        print(f"| {pname} | 1 | - | {t1:.3f} | {len(plan.z)} | {t1:.3f} | {t1:.3f} | 1.00 |", flush=True)
        for w in (2, 4, 8):
            for rule in ("atoms", "cost"):
                ts, rows = [], []
                for r in range(w):
                    t, n, _ = step_ms(eng, prot, plan, hplan, r, w, rule, a.steps)
                    ts.append(t)
                    rows.append(n)
                job = max(ts) + a.exchange_us * 1e-3
                print(f"| {pname} | {w} | {rule} | {' '.join(f'{t:.2f}' for t in ts)} | {' '.join(map(str, rows))} | "
                      f"{max(ts):.3f} | {job:.3f} | {t1 / job:.2f} |", flush=True)


if __name__ == "__main__":
    main()
