#!/usr/bin/env python
"""bench.py - MD steps/s on Chignolin through the MI355X ViSNet force path.

    python bench.py --gpus N --steps K --warmup W [--workload chig_md|frag_batch|ww_md|trpcage_md]

One "step" of the default workload (BASELINE.json configs[1]) = one MD step of
capped Chignolin (175 atoms -> 19 fragments, 391 fragment atoms): fragment
gather + cap-hydrogen placement, ViSNet energy+forces of the fragment batch
(neighbour list, embeddings, 9 ViS-MP layers, read-out, hand-written reverse
pass), overlap-force recombination, one Langevin update.  Everything stays in
HBM during the timed region.  With N > 1 (one process per GPU, launched by
torch.distributed.run) the fragments are sharded over the ranks and one fused
RCCL all-gather per step recombines shard forces/energies (strong scaling).
`--workload frag_batch` measures pure fragment-batch force throughput instead
(independent units, no collective, weak scaling).

Weights are seeded random at the reference's default hyper-parameters (the
checkpoints are not in the reference tree); the input geometry is the reference's
own Chignolin example, shipped as tests/golden/protein_chig.npz.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak


def load_protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(ROOT, "tests", "golden", f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


def fwd_flops(N, E, H, L, S, R):
    """Algorithmic forward FLOPs of one evaluation (minimal formulation), minus the
    layer-0 vector projections (10*S*N*H^2) and f_proj (2*E*H^2): vec == 0 entering layer 0
    (visnet_block.py:119-121) makes them identically zero, so they are never computed."""
    full = ((L - 1) * ((12 + 10 * S) * N * H * H + 10 * E * H * H) + ((12 + 6 * S) * N * H * H + 8 * E * H * H)
            + (3.5 * S + 7) * N * H * H + 4 * N * H * H + 2 * R * H * (2 * E - N))
    return full - (10 * S * N * H * H + 2 * E * H * H if L > 1 else 6 * S * N * H * H)


def cpu_baseline_md(plan, prot, hp, sd, budget_s=20.0):
    """Oracle ("port" of the reference algorithm, plain torch fp32 + autograd) timed on the host cores on
    the same Chignolin fragment batch - force evaluation only.  torch's intra-op threading saturates
    early on these small tensors (2x EPYC 9575F: 16 threads 2.4 s, 128 threads 10 s per evaluation), so
    the thread count is probed and the fastest one is used and reported as `cores`."""
    from ai2bmd_amd.fragmentation import fragment_positions
    from oracle.visnet_oracle import ViSNetOracle

    pos = fragment_positions(plan, prot.positions).astype(np.float32)
    o = ViSNetOracle(hp, sd, torch.float32)
    ncpu = os.cpu_count() or 1
    best_nt, best_t = None, None
    for nt in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
        torch.set_num_threads(nt)
        o.energy_forces(plan.z, pos, plan.start, plan.end)  # warm-up at this thread count
        t0 = time.perf_counter()
        o.energy_forces(plan.z, pos, plan.start, plan.end)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    n, t0 = 0, time.perf_counter()
    while True:
        o.energy_forces(plan.z, pos, plan.start, plan.end)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 8:
            break
    return dict(value=n / el, unit="MD steps/s", cores=best_nt, kind="port",
                sample=f"{n} energy+force evaluations of the Chignolin fragment batch (B={len(plan.start)}, "
                       f"N={len(plan.z)}) by oracle/visnet_oracle.py (fp32 torch + autograd, best of 8/16/32 "
                       f"intra-op threads on a {ncpu}-hardware-thread host), integrator excluded")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="chig_md", choices=["chig_md", "trpcage_md", "ww_md", "abd_md",
                                                               "frag_batch"])
    ap.add_argument("--frags-per-gpu", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--integrator", default="hip", choices=["hip", "torch"],
                    help="Langevin update: two HIP launches per step (default) or ~25 torch elementwise kernels")
    ap.add_argument("--no-relax-caps", action="store_true",
                    help="skip the per-step cap-hydrogen L-BFGS relaxation (reference: DistanceFragment.get_fragments)")
    ap.add_argument("--mm", action="store_true",
                    help="add the MM non-bonded term (reference: MMNonBondedCalculator on top of the fragment forces; "
                         "< 1 %% of the step).  Off by default: with seeded random ViSNet weights nothing but the "
                         "tether holds polar hydrogens (AMBER gives them no LJ core), so over thousands of steps the "
                         "Coulomb term tears the structure apart and the workload would change under the clock")
    ap.add_argument("--emulate-shard", default="", help="tuning aid, single process: 'r/w' = time rank r's share of a "
                    "w-rank MD job without the collective (not a valid bench line)")
    ap.add_argument("--chunk-edges", type=int, default=0, help="override vsn max_chunk_edges (workspace bound)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    group = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragmentation import build_plan, fragment_positions
    from ai2bmd_amd.md import Langevin, LangevinHIP
    from ai2bmd_amd.visnet_calculator import ViSNetEngine
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # seeded weight generator

    hp = default_hparams()
    sd = make_state_dict(hp, seed=2024)
    eng = ViSNetEngine(hp, sd, dev)
    if args.chunk_edges:
        eng.set_option("max_chunk_edges", args.chunk_edges)
    for kv in filter(None, os.environ.get("VSN_OPTS", "").split(",")):  # tuning aid: VSN_OPTS=fuse_fwd=0,overlap=0
        k_, v_ = kv.split("=")
        eng.set_option(k_, int(v_))
    H, L, S, R = hp["embedding_dimension"], hp["num_layers"], 8, hp["num_rbf"]

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    if args.workload.endswith("_md"):
        pname = args.workload[:-3]
        prot = load_protein(pname)
        plan = build_plan(prot)
        hplan = None
        if not args.no_relax_caps:
            from ai2bmd_amd.amber import load_tables
            from ai2bmd_amd.hydrogen import build_hydrogen_plan

            hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(ROOT, "tests", "golden",
                                                                             "amber_tables.npz")))
        if args.emulate_shard:
            er, ew = (int(v) for v in args.emulate_shard.split("/"))
            ff = ShardedFragmentForces.for_engine(eng, plan, rank=er, world=ew, hydrogen=hplan)
            ff.emulate = True
        else:
            ff = ShardedFragmentForces.for_engine(eng, plan, rank=rank, world=world, group=group, hydrogen=hplan)
        force_fn = ff.step
        if args.mm:
            # full AI2BMD potential = fragment (ViSNet) forces + MM Lennard-Jones/Coulomb between atoms that never
            # share a dipeptide (Calculators/nonbonded.py:33-63); charges / sigma / epsilon from the AMBER tables
            from types import SimpleNamespace

            from ai2bmd_amd.amber import load_tables, protein_mm_parameters
            from ai2bmd_amd.nonbonded import MMNonBondedCalculator

            q_, s_, e_ = protein_mm_parameters(prot, load_tables(os.path.join(ROOT, "tests", "golden",
                                                                                "amber_tables.npz")))
            mm = MMNonBondedCalculator(dev)
            mm.set_parameters(SimpleNamespace(charges=q_, sigmas=s_, epsilons=e_), plan)

            def force_fn(pos):
                E, F = ff.step(pos)
                e_mm, _ = mm.forces_device(pos, f_out=F, accumulate=True)
                return E + e_mm[0], F

        Integ = LangevinHIP if args.integrator == "hip" else Langevin
        md = Integ(prot.numbers, prot.positions, force_fn, dev, seed=0, tether_k=5.0)
        for _ in range(args.warmup):
            md.step()
        edges_before = eng.last_num_edges()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            md.step()
        barrier()
        el = time.perf_counter() - t0
        edges_after = eng.last_num_edges()
        # the workload must not change under the clock (a structure that flies apart has fewer edges = less work)
        assert abs(edges_after - edges_before) <= 0.1 * max(edges_before, 1), (edges_before, edges_after)
        if world > 1:
            import torch.distributed as dist

            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(md.x).all() and torch.isfinite(md.F).all(), "non-finite MD state"
        value, unit, metric = args.steps / el, "steps/s", f"MD steps/sec on {pname}"
        E_edges = eng.last_num_edges()
        n_loc = ff.local_rows
        workload = (f"{pname} AIMD loop: {len(prot)} atoms, B={len(plan.start)} fragments, N={len(plan.z)} fragment "
                    f"atoms, {'cap-H L-BFGS relaxation every step, ' if hplan is not None else ''}"
                    f"{'+ MM non-bonded (LJ + Coulomb, AMBER parameters), ' if args.mm else ''}Langevin 1 fs 300 K "
                    f"friction 0.001/fs, harmonic tether 5 eV/A^2 (random weights), "
                    f"ViSNet H={H} L={L} rbf={R} lmax=2 heads=8 cutoff=5")
        scaling = "strong"
        units_per_step = 1
        # ---- instrumented pass: HIP events around every GEMM launch (same stream) ----
        eng.set_option("profile", 1)
        nprof = 5
        for _ in range(nprof):
            md.step()
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.set_option("profile", 0)
        flops_eval = 2.0 * fwd_flops(n_loc, E_edges, H, L, S, R)
        extra = dict(edges_local=E_edges, edges_at_start_of_timed_region=edges_before, frag_atoms_local=n_loc, algorithmic_gflop_per_step_local=flops_eval / 1e9)
        if world == 1:
            # the reference-shaped seam (host numpy in, host numpy out => H2D + D2H over PCIe every call);
            # reported for information, never as `value`
            from ai2bmd_amd.fragment import FragmentData, make_batch_index
            from ai2bmd_amd.visnet_calculator import ViSNetModel

            seam = ViSNetModel.__new__(ViSNetModel)
            seam.device, seam.engine, seam.stream = dev, eng, torch.cuda.Stream(device=dev)
            fpos = fragment_positions(plan, prot.positions).astype(np.float32)
            fd = FragmentData(plan.z, fpos, plan.start, plan.end, make_batch_index(plan.start, plan.end))
            for _ in range(5):
                seam.dl_potential_loader(fd)
            t1 = time.perf_counter()
            for _ in range(50):
                seam.dl_potential_loader(fd)
            extra["host_seam_evals_per_s_pcie_inclusive"] = 50 / (time.perf_counter() - t1)
    else:
        # pure fragment-batch throughput: per-GPU batch built from the example proteins' fragments
        rng = np.random.default_rng(1234 + rank)
        zs, ps, sizes = [], [], []
        pool = []
        for pname in ("chig", "trpcage", "ww", "abd"):
            pr = load_protein(pname)
            pl = build_plan(pr)
            fp = fragment_positions(pl, pr.positions)
            for b in range(len(pl.start)):
                pool.append((pl.z[pl.start[b]:pl.end[b]], fp[pl.start[b]:pl.end[b]]))
        for i in range(args.frags_per_gpu):
            zf, pf = pool[i % len(pool)]
            zs.append(zf)
            ps.append(pf - pf.mean(0) + rng.normal(0, 0.05, size=pf.shape))
            sizes.append(len(zf))
        end = np.cumsum(sizes)
        start = end - np.asarray(sizes)
        z = torch.as_tensor(np.concatenate(zs), dtype=torch.int64).to(dev)
        pos = torch.as_tensor(np.concatenate(ps), dtype=torch.float32).to(dev)
        e = torch.empty(len(start), device=dev)
        f = torch.empty(len(z), 3, device=dev)
        for _ in range(args.warmup):
            eng.forces_device(z, pos, start, end, e, f)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.forces_device(z, pos, start, end, e, f)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist

            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(f).all()
        value = args.steps * args.frags_per_gpu * world / el
        unit, metric = "fragments/s", "fragment-batch forces/sec"
        E_edges = eng.last_num_edges()
        workload = (f"synthetic dipeptide/ACE-NME batch: {args.frags_per_gpu} fragments per GPU "
                    f"({len(z)} atoms) harvested from the example proteins with 0.05 A jitter, pure energy+force "
                    f"evaluation, ViSNet H={H} L={L}")
        scaling = "weak"
        eng.set_option("profile", 1)
        nprof = 2
        for _ in range(nprof):
            eng.forces_device(z, pos, start, end, e, f)
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.set_option("profile", 0)
        extra = dict(atoms_per_gpu=int(len(z)), edges_last_chunk=E_edges)

    # dominant kernel = the GEMM tile variant with the most device time
    dom = max(prof, key=lambda k: prof[k]["ms"])
    pd = prof[dom]
    gemm_ms_step = sum(v["ms"] for v in prof.values()) / nprof
    # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes, so the per-launch
    # average of the committed run of THIS command is read back from profiles/ (null when absent)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            traffic = json.load(fh).get(args.workload, {}).get(dom)
    except Exception:
        traffic = None
    roof = dict(
        bound="mfma", kernel=f"vsn::{dom}",
        achieved=(pd["flops"] / (pd["ms"] * 1e-3)) / 1e12 if pd["ms"] > 0 else 0.0,
        peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
        frac=((pd["flops"] / (pd["ms"] * 1e-3)) / 1e12 / MFMA_F32_PEAK_TFLOPS) if pd["ms"] > 0 else 0.0,
        traffic=traffic,
        launches_per_step=pd["launches"] / nprof,
        avg_launch_us=1e3 * pd["ms"] / max(pd["launches"], 1),
        algorithmic_bytes_per_launch=pd["bytes"] / max(pd["launches"], 1),
        all_gemm_ms_per_step=gemm_ms_step,
        all_gemm_tflops=(sum(v["flops"] for v in prof.values()) / max(sum(v["ms"] for v in prof.values()), 1e-9)) / 1e9,
    )
    out = dict(
        metric=metric, value=value, unit=unit, n_gpus=world, steps=args.steps, warmup=args.warmup,
        ms_per_step=1e3 * el / args.steps, higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32",
        data="synthetic (seeded random weights at the reference's default hyper-parameters; "
             "geometry = reference examples/*.pdb fixtures)",
        config=dict(workload=workload, **extra), roofline=roof,
    )
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload.endswith("_md"):
        out["cpu_baseline"] = cpu_baseline_md(plan, prot, hp, sd)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
