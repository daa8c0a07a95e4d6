#!/usr/bin/env python
"""bench.py - MD steps/s on Chignolin (+ fragment-batch forces/s) through the MI355X ViSNet force path.

    python bench.py --gpus N --steps K --warmup W [--workload chig_md|trpcage_md|ww_md|abd_md|frag_batch]

One "step" of the default workload (BASELINE.json configs[1]) = one MD step of capped Chignolin (175 atoms -> 19
fragments, 391 fragment atoms): fragment gather + cap-hydrogen placement and L-BFGS relaxation, ViSNet
energy+forces of the fragment batch (neighbour list, embeddings, 9 ViS-MP layers, read-out, hand-written reverse
pass), overlap-force recombination, one Langevin update.  Everything stays in HBM during the timed region.

The ONE JSON line (rank 0) carries the whole BASELINE metric: `value` = Chignolin MD steps/s, and `secondary` =
fragment-batch forces/s at 4096 fragments per GPU (independent units, no collective, weak scaling), Trp-cage MD
steps/s (configs[2]) and, for N > 1, WW-domain MD steps/s (configs[3]); each with its own roofline block.

With N > 1 the fragments of the MD workloads are sharded over the ranks (one process per GPU) and one fused RCCL
all-gather per step recombines shard forces/energies (strong scaling).  `python bench.py --gpus N` starts its own
ranks (re-executes itself under torch.distributed.run) when it was not launched by one.

Before any clock starts, the forces of the workload's step-0 fragment batch are compared with golden vectors
computed by the reference's own ViSNet source (tests/golden/visnet_prot_*.npz); a mismatch aborts the run
(`parity` block).  `value` is timed over EXACTLY the K steps asked for; chig_md IS the 1000-step loop of configs[1], so
with K < 1000 that loop is timed right behind the K steps as well (config.c2_loop; the roofline blocks are quoted on
it) and without --steps K = 1000.

Weights are seeded random at the reference's default hyper-parameters (the checkpoints are not in the reference
tree); the input geometry is the reference's own examples, shipped as tests/golden/protein_*.npz.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s reached by a float4 copy)
C2_STEPS = 1000               # BASELINE.json configs[1]: "Chignolin ... full AIMD loop, 1000 steps, 1xMI355X"


def _traffic_profile():
    """newest profiles/r*_pmc_traffic.json (written by tools/profile_round.sh)"""
    import glob

    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return os.path.relpath(fs[-1], ROOT) if fs else os.path.join("profiles", "none")


TRAFFIC_PROFILE = _traffic_profile()
PRETTY = dict(chig="Chignolin", trpcage="Trp-cage", ww="WW domain", abd="ABD")


# ------------------------------------------------------------------------------------------------------------
# launch plumbing
# ------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps: `value` is quoted on EXACTLY K.  chig_md (BASELINE configs[1] = the 1000-step "
                         "Chignolin loop) defaults to K = 1000 and, when a smaller K is asked for, times the 1000-step "
                         "loop as well, right behind the K steps (config.c2_loop); other workloads default to 200 MD "
                         "steps / 2 batches")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="chig_md", choices=["chig_md", "trpcage_md", "ww_md", "abd_md",
                                                               "frag_batch", "frag_stream"])
    ap.add_argument("--frags-per-gpu", type=int, default=4096)
    ap.add_argument("--conformations", type=int, default=1_000_000,
                    help="frag_stream: total NEW conformations over all ranks (BASELINE configs[4]: 1M)")
    ap.add_argument("--min-seconds", type=float, default=0.0,
                    help="tuning aid (default off): stretch the timed region to at least this long, "
                         "steps = max(--steps, ceil(min_seconds / step time))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="primary workload only (profiling runs)")
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
    ap.add_argument("--balance", default="atoms", choices=["atoms", "cost"],
                    help="N > 1: contiguous fragment shards balanced by atoms (the reference's rule, "
                         "device_strategy.py:84-127; default) or by edge count (measured: within 1 %% of each other, "
                         "DESIGN.md section 5 - the per-rank step is dominated by its size-independent part)")
    ap.add_argument("--chunk-edges", type=int, default=0, help="override vsn max_chunk_edges (workspace bound)")
    ap.add_argument("--exchange", choices=("collective", "p2p"), default="collective",
                    help="the exchange step of the sharded MD path: ONE all_gather_into_tensor per step (RCCL; default) "
                         "or the tuned variant - every rank stores its slot straight into its peers' gather buffers "
                         "(hipIpc-mapped, one HIP launch per step: csrc/p2p.hip)")
    ap.add_argument("--dump-state", action="store_true",
                    help="MD workloads: put float64 checksums of the final positions / velocities / forces into "
                         "config.state_checksum (tests compare runs that must agree to the last bit)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="VALIDATION aid, not a measurement: N ranks on ONE GPU (device = LOCAL_RANK %% device count) "
                         "over gloo with device tensors - the real sharded product path (one HIP engine per rank, fused "
                         "integrator halves, the all-gather, the remapped combine) where no multi-GPU node exists; "
                         "RCCL itself refuses two ranks on one device")
    ap.add_argument("--stub", action="store_true",
                    help="launch-logic self-test on CPU (gloo, sleep-based fake step): NOT a measurement")
    a = ap.parse_args(argv)
    a.steps_requested = a.steps
    if a.steps is None:
        a.steps = 2 if a.workload in ("frag_batch", "frag_stream") else (C2_STEPS if a.workload == "chig_md" else 200)
    return a


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks, one per GPU, like the reference starts its
    extra GPU workers itself (/root/reference/src/Calculators/visnet_calculator.py:78-118, bonded.py:75-77)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


class Ctx:
    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.stub = args.stub
        self.group = None
        self.backend = None
        self.share_gpu = bool(getattr(args, "share_gpu", False))
        if self.stub:
            self.dev = "cpu"
        else:
            idx = self.local_rank % max(torch.cuda.device_count(), 1) if self.share_gpu else self.local_rank
            torch.cuda.set_device(idx)
            self.dev = f"cuda:{idx}"
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend = "gloo" if (self.stub or self.share_gpu) else "nccl"  # "nccl" IS RCCL on ROCm
            kw = {} if self.backend == "gloo" else dict(device_id=torch.device(self.dev))
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.stub:
            torch.cuda.synchronize()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed_region(ctx, step_fn, steps, warmup, min_seconds=0.0):
    """W untimed warm-up steps, then EXACTLY `k` timed steps bracketed by barrier + device synchronisation on both
    sides; k = steps (with the tuning aid --min-seconds > 0: max(steps, ceil(min_seconds / step time)) from a short
    calibration, agreed over the ranks).  -> (k, seconds = max over ranks)."""
    for _ in range(warmup):
        step_fn()
    ctx.barrier()
    k = int(steps)
    if min_seconds > 0:
        ncal = max(1, min(steps, 5))
        t0 = time.perf_counter()
        for _ in range(ncal):
            step_fn()
        ctx.barrier()
        est = ctx.max_over_ranks((time.perf_counter() - t0) / ncal)
        k = int(max(steps, math.ceil(min_seconds / max(est, 1e-9))))
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        step_fn()
    ctx.barrier()
    return k, ctx.max_over_ranks(time.perf_counter() - t0)


# ------------------------------------------------------------------------------------------------------------
# workload helpers
# ------------------------------------------------------------------------------------------------------------
def load_protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLD, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


def load_golden(name):
    d = np.load(os.path.join(GOLD, f"visnet_prot_{name}.npz"))
    return {k: d[k] for k in d.files}


def fwd_flops(N, E, H, L, S, R):
    """Algorithmic forward FLOPs of one evaluation (minimal formulation), minus the
    layer-0 vector projections (10*S*N*H^2) and f_proj (2*E*H^2): vec == 0 entering layer 0
    (visnet_block.py:119-121) makes them identically zero, so they are never computed."""
    full = ((L - 1) * ((12 + 10 * S) * N * H * H + 10 * E * H * H) + ((12 + 6 * S) * N * H * H + 8 * E * H * H)
            + (3.5 * S + 7) * N * H * H + 4 * N * H * H + 2 * R * H * (2 * E - N))
    return full - (10 * S * N * H * H + 2 * E * H * H if L > 1 else 6 * S * N * H * H)


def count_edges(pos, start, end, cutoff, max_nb):
    """host count of the radius-graph edges (self loops included, <= max_nb sources per target)"""
    tot = 0
    for a, b in zip(start, end):
        p = np.asarray(pos[a:b], np.float32)
        if len(p):
            d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
            tot += int(np.minimum((d2 < np.float32(cutoff * cutoff)).sum(1), max_nb).sum())
    return tot


class ParityFailure(SystemExit):
    """a workload failed its parity guard.  Raised on EVERY rank at the same point (the verdict is agreed over the ranks
    before anybody raises), so a secondary can be dropped by all ranks together; fatal for the primary workload."""


_CTX = None  # the run's Ctx (set by main): parity verdicts are agreed over its ranks


def _agreed(ok: bool) -> bool:
    """True only if `ok` on every rank (one all-reduce where a process group exists; outside every timed region)"""
    if _CTX is None or _CTX.world == 1:
        return bool(ok)
    return _CTX.max_over_ranks(0.0 if ok else 1.0) == 0.0


def parity_check(what, E, F, E64, F64, tol_f=1e-4, tol_e=1e-5, collective=True):
    """SURVEY.md 8c tolerance (fp32 contract vs the reference's fp64 result); raises ParityFailure on mismatch - on all
    ranks when any rank mismatches (`collective`; the in-stream check of run_frag_stream passes False and settles its
    verdict after the loop)."""
    E, F = np.asarray(E, np.float64).reshape(-1), np.asarray(F, np.float64)
    E64, F64 = np.asarray(E64, np.float64).reshape(-1), np.asarray(F64, np.float64)
    ok = E.shape == E64.shape and F.shape == F64.shape and np.isfinite(E).all() and np.isfinite(F).all()
    de = float(np.abs(E - E64).max()) if ok else float("nan")
    df = float(np.abs(F - F64).max()) if ok else float("nan")
    mae = float(np.abs(F - F64).mean()) if ok else float("nan")
    ok = ok and (np.abs(E - E64) <= tol_e * np.maximum(1.0, np.abs(E64))).all() \
        and df <= tol_f * max(1.0, float(np.abs(F64).max())) and mae <= 0.1 * tol_f * max(1.0, float(np.abs(F64).mean()))
    if not ok and os.environ.get("VSN_LAB_NO_PARITY"):  # lab ablations compute wrong numbers on purpose (tools/lab)
        print(f"LAB RUN, NOT A MEASUREMENT: parity check skipped ({what})", file=sys.stderr)
        return dict(max_dE=de, max_dF=df, force_mae=mae, max_abs_F=float("nan"), LAB_NO_PARITY=True)
    ok_all = _agreed(ok) if collective else ok
    if not ok_all:
        where = "" if not ok else " (this rank passed; another rank did not)"
        raise ParityFailure(f"PARITY FAILURE before the timed region ({what}): max|dE|={de:.3e} max|dF|={df:.3e} "
                            f"MAE={mae:.3e} against the reference-source golden{where}")
    return dict(max_dE=de, max_dF=df, force_mae=mae, max_abs_F=float(np.abs(F64).max()))


def traffic_from_profile(workload, kernel):
    """HBM traffic per launch of `kernel`: PMC counters need their own rocprofv3 passes, so the per-launch average
    of the committed run of THIS command is read back from profiles/.  -> (bytes | None, note).  The file carries the
    digest of the kernel sources it was measured on (ai2bmd_amd.build._digest): when the library has changed since,
    the figure is NOT reported (it would be another build's traffic)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_PROFILE)) as fh:
            d = json.load(fh)
    except Exception:
        return None, f"{TRAFFIC_PROFILE}: absent"
    try:
        from ai2bmd_amd.build import _digest

        dig = _digest()
    except Exception:
        dig = None
    if d.get("build_digest") != dig:
        return None, (f"{TRAFFIC_PROFILE} was measured on another build of csrc/ (digest "
                      f"{str(d.get('build_digest'))[:12]} != {str(dig)[:12]}): stale, not reported - re-run "
                      "tools/profile_round.sh")
    val = d.get(workload, {}).get(kernel)
    if val is None:
        return None, f"{TRAFFIC_PROFILE} holds no PMC passes of this workload / kernel ({workload}, {kernel})"
    return val, (f"{TRAFFIC_PROFILE} (rocprofv3 PMC passes of this command on this build, per launch; not "
                 "re-measured in this run)")


def roofline_block(eng, nprof, workload, flops_step, ms_per_step):
    """Dominant GEMM kernel from the instrumented pass: HIP events on the launch stream around every launch,
    minus the measured cost of an empty event bracket."""
    prof = eng.profile_read()
    br_ms = eng.profile_bracket_ms()
    dom = max(prof, key=lambda k: prof[k]["ms"])
    pd = prof[dom]
    n = max(pd["launches"], 1)
    raw_us = 1e3 * pd["ms"] / n
    net_us = max(raw_us - 1e3 * br_ms, 1e-3)
    ach = (pd["flops"] / n) / (net_us * 1e-6) / 1e12
    all_ms = sum(v["ms"] - v["launches"] * br_ms for v in prof.values())
    traffic, tnote = traffic_from_profile(workload, dom)
    return dict(
        bound="mfma", kernel=f"vsn::{dom}", achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
        frac=ach / MFMA_F32_PEAK_TFLOPS,
        traffic=traffic, traffic_source=tnote,
        launches_per_step=pd["launches"] / nprof, avg_launch_us=net_us, avg_launch_us_with_event_bracket=raw_us,
        event_bracket_us=1e3 * br_ms,
        algorithmic_flop_per_launch=pd["flops"] / n, algorithmic_bytes_per_launch=pd["bytes"] / n,
        all_gemm_ms_per_step=all_ms / nprof,
        all_gemm_tflops=(sum(v["flops"] for v in prof.values()) / max(all_ms, 1e-9)) / 1e9,
        # whole step against the same peak: algorithmic FLOPs of one evaluation / wall time of one step
        step_frac=(flops_step / (ms_per_step * 1e-3)) / 1e12 / MFMA_F32_PEAK_TFLOPS,
        step_tflops=(flops_step / (ms_per_step * 1e-3)) / 1e12,
    )


MFMA_F32_SUSTAINED_TFLOPS = 116.0  # measured: what the fp32 matrix pipes sustain on real operands (power-limited clock;
                                   # the 128 x 128 batch tile and the GEMM library alike, LAB_NOTES 4.1b)
HBM_SUSTAINED_GBPS = 6300.0        # measured: a float4 device copy


def step_bound(eng, nprof):
    """what the step would take with every GEMM launch at the sustained fp32 matrix rate and every node walk at the
    sustained HBM rate: per GEMM kernel kind, launches x max(average flop / 116 TFLOP/s, average algorithmic bytes /
    6.3 TB/s) (averages over the launches of the kind: launches of one kind with different members make this a lower
    bound of the per-launch sum), plus the node walks' algorithmic bytes / 6.3 TB/s - beside the time those launches
    actually took (the once-per-step launches - graph build, embeddings,
    read-out, cap-hydrogen relaxation, integrator: ~20 launches at the 5-9 us floor of a dependent kernel - are neither
    in the bound nor in `covered_ms`)."""
    prof = eng.profile_read()
    br_ms = eng.profile_bracket_ms()
    walks = eng.profile_read_walks()
    bound = meas = 0.0
    for v in prof.values():
        if v["launches"] > 0:
            # per launch: the group's flops and bytes (averaged over the launches of the kernel)
            fl, by, n = v["flops"] / v["launches"], v["bytes"] / v["launches"], v["launches"]
            bound += n * max(fl / (MFMA_F32_SUSTAINED_TFLOPS * 1e12), by / (HBM_SUSTAINED_GBPS * 1e9)) * 1e3
            meas += v["ms"] - n * br_ms
    for v in walks.values():
        if v["launches"] > 0:
            bound += (v["bytes"] / (HBM_SUSTAINED_GBPS * 1e9)) * 1e3
            meas += v["ms"]
    return dict(step_bound_ms=bound / nprof, covered_ms=meas / nprof,
                rule="per GEMM kernel kind: launches x max(avg flop / 116 TFLOP/s, avg bytes / 6.3 TB/s); "
                     "node walks: algorithmic bytes / 6.3 TB/s")


def rocprof_from_profile(workload, kernel):
    """avg / min duration of `kernel` in the committed rocprofv3 kernel trace of this command on THIS build
    (profiles/r*_pmc_traffic.json, section "kernel_ns"; None when the library is another build)"""
    try:
        with open(os.path.join(ROOT, TRAFFIC_PROFILE)) as fh:
            d = json.load(fh)
        from ai2bmd_amd.build import _digest

        if d.get("build_digest") != _digest():
            return None
        return d.get("kernel_ns", {}).get(workload, {}).get(kernel)
    except Exception:
        return None


def roofline_hbm_block(eng, nprof, workload):
    """The scatter path against the HBM roofline: the forward edge-attention and vector-message aggregation / node
    update launches of every layer.  In the instrumented pass each of these launches carries two events ON ITS
    DISPATCH PACKET (hipExtLaunchKernelGGL, csrc/layer_fwd.hip): their elapsed time is the kernel's own begin..end
    timestamp pair - what rocprofv3's kernel trace records - so no stream-bracket correction enters.
    `achieved` = ALGORITHMIC bytes per launch (every array the launch touches, once: DESIGN.md 4.2) / average launch
    time; `traffic` = the PMC bytes of the same kernel from profiles/ (above the algorithmic bytes = re-reads);
    `rocprof` = the same kernel's avg / min in the committed kernel trace of this build, with the live / trace ratio."""
    walks = eng.profile_read_walks()
    walks = {k: v for k, v in walks.items() if v["launches"] > 0}
    sp = {k: v for k, v in walks.items() if k in ("k_edge_attn", "k_node_update")}
    if not sp:
        return None
    dom = max(sp, key=lambda k: sp[k]["ms"])
    blocks = {}
    for k, v in walks.items():
        n = v["launches"]
        us = max(1e3 * v["ms"] / n, 1e-3)
        gbps = (v["bytes"] / n) / (us * 1e-6) / 1e9
        blocks[k] = dict(launches_per_step=n / nprof, avg_launch_us=us, algorithmic_bytes_per_launch=v["bytes"] / n,
                         achieved_GBps=gbps, frac=gbps / HBM_PEAK_GBPS)
        rp = rocprof_from_profile(workload, k)
        if rp:
            blocks[k]["rocprof"] = dict(avg_us=rp["avg_ns"] / 1e3, min_us=rp["min_ns"] / 1e3,
                                        live_over_trace=us / (rp["avg_ns"] / 1e3),
                                        frac_from_trace=(v["bytes"] / n) / (rp["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS)
        tr, _ = traffic_from_profile(workload, k)
        if tr:
            blocks[k]["traffic"] = tr
            blocks[k]["traffic_over_algorithmic"] = tr / (v["bytes"] / n)
    reverse = {k: blocks.pop(k) for k in list(blocks) if k.startswith("k_bwd_")}
    traffic, tnote = traffic_from_profile(workload, dom)
    d = blocks[dom]
    out = dict(bound="hbm", kernel=f"vsn::{dom}", achieved=d["achieved_GBps"], peak=HBM_PEAK_GBPS, unit="GB/s",
               frac=d["frac"], traffic=traffic, traffic_source=tnote, launches_per_step=d["launches_per_step"],
               avg_launch_us=d["avg_launch_us"], algorithmic_bytes_per_launch=d["algorithmic_bytes_per_launch"],
               timing="dispatch begin..end timestamps (events attached to the kernel's own packet), no bracket correction",
               all_scatter_kernels=blocks)
    if "rocprof" in d:
        out["rocprof"] = d["rocprof"]
    if reverse:
        # the reverse node walks of single-protein sizes (hand-derived adjoints of visnet_block.py:237-312): same timing,
        # algorithmic bytes per launch (every distinct array once, csrc/engine.hip), counter bytes beside them
        out["reverse_walks_detail"] = reverse
        out["reverse_walks"] = {k: dict(us=v["avg_launch_us"], per_step=v["launches_per_step"],
                                        alg_MB=v["algorithmic_bytes_per_launch"] / 1e6,
                                        pmc_MB=(v["traffic"] / 1e6 if "traffic" in v else None),
                                        frac=v["frac"]) for k, v in reverse.items()}
    return out


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
def run_md(ctx, eng, hp, pname, args, steps, warmup, gold_suffix="", mm=None, tether_k=5.0, pre_steps=None):
    """gold_suffix: which reference-source golden guards the run ("" = H=256/L=9, "_h128l6" = the small variant);
    mm: override of --mm for a `secondary` line; tether_k: harmonic tether of every atom to its start position;
    pre_steps: time exactly this many steps first (the driver's --steps K when K < the workload's own step count),
    reported as config.requested_run, then the `steps` of the workload"""
    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragmentation import build_plan, fragment_positions
    from ai2bmd_amd.md import Langevin, LangevinHIP

    dev = ctx.dev
    H, L, S, R = hp["embedding_dimension"], hp["num_layers"], 8, hp["num_rbf"]
    use_mm = args.mm if mm is None else mm
    prot = load_protein(pname)
    plan = build_plan(prot)
    gold = load_golden(pname + gold_suffix)
    hplan = None
    if not args.no_relax_caps:
        from ai2bmd_amd.amber import load_tables
        from ai2bmd_amd.hydrogen import build_hydrogen_plan

        hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    # ---- parity guard 1: the engine on the golden's own fragment batch (same inputs as the reference run) ----
    tag = "relaxed"
    z_t = torch.as_tensor(gold["z"], dtype=torch.int64).to(dev)
    p_t = torch.as_tensor(gold[f"pos_{tag}"], dtype=torch.float32).to(dev)
    e_t = torch.empty(len(gold["start"]), device=dev)
    f_t = torch.empty(len(gold["z"]), 3, device=dev)
    eng.forces_device(z_t, p_t, gold["start"], gold["end"], e_t, f_t)
    torch.cuda.synchronize()
    nonempty = (gold["end"] - gold["start"]) > 0
    par = parity_check(f"{pname} fragment batch, H={H} L={L}", e_t.cpu().numpy()[nonempty], f_t.cpu().numpy(),
                       gold[f"E_ref64_{tag}"], gold[f"F_ref64_{tag}"])
    if args.emulate_shard:
        er, ew = (int(v) for v in args.emulate_shard.split("/"))
        ff = ShardedFragmentForces.for_engine(eng, plan, rank=er, world=ew, hydrogen=hplan, balance=args.balance)
        ff.emulate = True
    else:
        ff = ShardedFragmentForces.for_engine(eng, plan, rank=ctx.rank, world=ctx.world, group=ctx.group,
                                              hydrogen=hplan, balance=args.balance,
                                              exchange=getattr(args, "exchange", "collective"))
    # ---- parity guard 2: the device pipeline (gather + cap-H + ViSNet shard + all-gather + combine) at step 0 ----
    if not args.emulate_shard:
        x0 = torch.as_tensor(prot.positions, dtype=torch.float32, device=dev)
        E0, F0 = ff.step(x0)
        torch.cuda.synchronize()
        ptag = "relaxed" if hplan is not None else "placed"
        # SURVEY 8c contract (1e-4 max|F|) for the relaxed path too: the fp32 L-BFGS on the device leaves the cap
        # hydrogens within 2e-4 A of the reference optimiser's, which moves the forces by ~2e-6 (measured)
        pp = parity_check(f"{pname} protein forces after recombination", [float(E0)], F0.cpu().numpy(),
                          [float(gold[f"Eprot64_{ptag}"])], gold[f"Fprot64_{ptag}"], tol_f=1e-4, tol_e=1e-4)
        par.update(pipeline_max_dF=pp["max_dF"], pipeline_dE=pp["max_dE"])
    par["max_dF_over_ranks"] = ctx.max_over_ranks(par["max_dF"])
    force_fn = ff.step
    if use_mm:
        # full AI2BMD potential = fragment (ViSNet) forces + MM Lennard-Jones/Coulomb between atoms that never
        # share a dipeptide (Calculators/nonbonded.py:33-63); charges / sigma / epsilon from the AMBER tables
        from types import SimpleNamespace

        from ai2bmd_amd.amber import load_tables, protein_mm_parameters
        from ai2bmd_amd.nonbonded import MMNonBondedCalculator

        q_, s_, e_ = protein_mm_parameters(prot, load_tables(os.path.join(GOLD, "amber_tables.npz")))
        mm = MMNonBondedCalculator(dev)
        mm.set_parameters(SimpleNamespace(charges=q_, sigmas=s_, epsilons=e_), plan)

        def force_fn(pos):
            E, F = ff.step(pos)
            e_mm, _ = mm.forces_device(pos, f_out=F, accumulate=True)
            return E + e_mm[0], F

    Integ = LangevinHIP if args.integrator == "hip" else Langevin
    md = Integ(prot.numbers, prot.positions, force_fn, dev, seed=0, tether_k=tether_k)
    for _ in range(min(warmup, 3)):
        md.step()
    edges_before = eng.last_num_edges()
    requested_run = None
    w_left = max(warmup - 3, 0)
    if pre_steps:
        kp, elp = timed_region(ctx, md.step, pre_steps, w_left)
        requested_run = dict(steps=kp, ms_per_step=1e3 * elp / kp, value=kp / elp, unit="steps/s",
                             note=f"exactly the --steps {pre_steps} asked for, timed first (same barrier / "
                                  f"synchronise bracket); `value` is the {steps}-step run the workload is defined on")
        w_left = 0
    k, el = timed_region(ctx, md.step, steps, w_left, args.min_seconds)
    edges_after = eng.last_num_edges()
    # the workload must not change under the clock (a structure that flies apart has fewer edges = less work)
    assert abs(edges_after - edges_before) <= 0.1 * max(edges_before, 1), (edges_before, edges_after)
    assert torch.isfinite(md.x).all() and torch.isfinite(md.F).all(), "non-finite MD state"
    # every rank integrates the whole protein from the same gathered forces: the trajectories must be THE SAME bits
    chk = float(md.x.double().sum().item()) + float(md.v.double().sum().item()) if hasattr(md, "v") else float(md.x.double().sum().item())
    spread = ctx.max_over_ranks(chk) + ctx.max_over_ranks(-chk)  # max - min over the ranks
    assert spread == 0.0, f"ranks have diverged: checksum spread {spread!r} after {k} steps"
    if ff.p2p is not None:
        ff.p2p.check()  # every wait of the direct-write exchange completed (none gave up on a peer)
    state_checksum = None
    if getattr(args, "dump_state", False):
        # (hex strings: the compact line rounds floats to 6 digits, these must survive to the last bit)
        state_checksum = [float(t.double().sum().item()).hex() for t in (md.x, md.v, md.F)] + [
            float(t.double().abs().sum().item()).hex() for t in (md.x, md.v, md.F)]
    ms = 1e3 * el / k
    n_loc = ff.local_rows
    # ---- instrumented pass: HIP events around every GEMM launch (same stream) ----
    eng.set_option("profile", 1)
    nprof = 5
    for _ in range(nprof):
        md.step()
    torch.cuda.synchronize()
    flops_step = 2.0 * fwd_flops(n_loc, edges_after, H, L, S, R)
    # (the PMC passes under profiles/ are of the H=256 / L=9 network: another network has no counter figure)
    wkey = f"{pname}_md" if (H, L) == (256, 9) else f"{pname}_md_h{H}l{L}"
    roof = roofline_block(eng, nprof, wkey, flops_step, ms)
    roof_hbm = roofline_hbm_block(eng, nprof, wkey)
    bound = step_bound(eng, nprof)
    eng.set_option("profile", 0)
    workload = (f"{pname} AIMD loop: {len(prot)} atoms, B={len(plan.start)} fragments, N={len(plan.z)} fragment "
                f"atoms, {'cap-H L-BFGS relaxation every step, ' if hplan is not None else ''}"
                f"{'+ MM non-bonded (LJ + Coulomb, AMBER parameters), ' if use_mm else ''}Langevin 1 fs 300 K "
                f"friction 0.001/fs, harmonic tether {tether_k:g} eV/A^2 (random weights), "
                f"ViSNet H={H} L={L} rbf={R} lmax=2 heads=8 cutoff=5")
    res = dict(metric=f"MD steps/sec on {PRETTY[pname]}", value=k / el, unit="steps/s", steps=k, ms_per_step=ms,
               scaling="strong", config=dict(workload=workload, exchange=("p2p direct peer writes (csrc/p2p.hip)"
                                                                          if ff.p2p is not None else
                                                                          "all_gather_into_tensor") if ctx.world > 1 or
                                             ff.p2p is not None else "none (single rank)", edges_local=edges_after,
                                             edges_at_start_of_timed_region=edges_before, frag_atoms_local=n_loc,
                                             algorithmic_gflop_per_step_local=flops_step / 1e9),
               parity=par, roofline=roof)
    if state_checksum is not None:
        res["config"]["state_checksum"] = state_checksum
    if requested_run:
        # the driver's contract: "time EXACTLY K steps" - `value` / `steps` / `ms_per_step` are the K it asked for; the
        # 1000-step loop BASELINE configs[1] is defined on is measured right behind it, same state, and rides in config
        res["config"]["c2_loop"] = dict(steps=k, ms_per_step=ms, value=k / el, unit="steps/s",
                                        note="BASELINE configs[1]: the 1000-step Chignolin loop, timed after the "
                                             "requested steps (same barrier / synchronise bracket); the roofline "
                                             "blocks are quoted on it")
        res.update(value=requested_run["value"], steps=requested_run["steps"], ms_per_step=requested_run["ms_per_step"])
    if roof_hbm:
        res["roofline"]["reverse_walks"] = roof_hbm.pop("reverse_walks", None)
        res["roofline"]["reverse_walks_detail"] = roof_hbm.pop("reverse_walks_detail", None)
        res["roofline"]["hbm"] = roof_hbm
    res["roofline"]["step_bound"] = bound
    return res, (plan, prot, md)


def collective_delta(ctx, eng, hp, pname, args, steps=300, rounds=3):
    """What ONE `all_gather_into_tensor` per MD step costs on this software stack, MEASURED: the same sharded step with
    `force_collective=True` (a world-1 RCCL process group: ProcessGroupNCCL host enqueue, the event hand-off between
    the compute stream and RCCL's stream, RCCL's own launch) against the collective-free loop, same protein, same
    state, `rounds` alternating blocks of `steps` steps, median of the per-block differences.  A single rank moves no
    bytes over xGMI: this is the software floor of the exchange step every rank of an N-rank job pays, not the wire.
    Only where no process group exists yet (N = 1) and a GPU is present; never part of `value`."""
    import torch.distributed as dist

    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragmentation import build_plan
    from ai2bmd_amd.md import LangevinHIP

    if ctx.world != 1 or dist.is_initialized():
        return None
    prot = load_protein(pname)
    plan = build_plan(prot)
    hplan = None
    if not args.no_relax_caps:
        from ai2bmd_amd.amber import load_tables
        from ai2bmd_amd.hydrogen import build_hydrogen_plan

        hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(GOLD, "amber_tables.npz")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(ctx.dev))
    try:
        mds, ffs = {}, {}
        for coll in (False, True, "p2p"):
            ff = ShardedFragmentForces.for_engine(eng, plan, hydrogen=hplan, force_collective=(coll is True),
                                                  exchange="p2p" if coll == "p2p" else "collective")
            ffs[coll] = ff
            mds[coll] = LangevinHIP(prot.numbers, prot.positions, ff.step, ctx.dev, seed=0, tether_k=5.0)
            for _ in range(10):
                mds[coll].step()
        torch.cuda.synchronize()
        per = {False: [], True: [], "p2p": []}
        for _ in range(rounds):
            for coll in (False, True, "p2p"):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    mds[coll].step()
                torch.cuda.synchronize()
                per[coll].append(1e3 * (time.perf_counter() - t0) / steps)
        # same draws, same arithmetic: the all-gather of one rank is a copy of its own slot
        same = bool(torch.equal(mds[False].x, mds[True].x)) and bool(torch.equal(mds[False].x, mds["p2p"].x))
        ffs["p2p"].p2p.check()
        d = sorted(a - b for a, b in zip(per[True], per[False]))
        dp = sorted(a - b for a, b in zip(per["p2p"], per[False]))
        return dict(workload=f"{pname}_md", steps_per_block=steps, blocks=rounds,
                    ms_per_step_without=float(np.median(per[False])), ms_per_step_with=float(np.median(per[True])),
                    delta_us=1e3 * d[len(d) // 2], delta_us_min=1e3 * d[0], delta_us_max=1e3 * d[-1],
                    # the tuned exchange (csrc/p2p.hip: one launch that stores the slot, raises and awaits the flags),
                    # same protocol: a single rank stores into its own buffer - the launch and its flag round trip
                    p2p_ms_per_step=float(np.median(per["p2p"])), p2p_delta_us=1e3 * dp[len(dp) // 2],
                    p2p_delta_us_min=1e3 * dp[0], p2p_delta_us_max=1e3 * dp[-1],
                    trajectories_bit_identical=same, slot_bytes=int(ff.slot * 4),
                    what="world-1 RCCL all_gather_into_tensor every step (torch.distributed ProcessGroupNCCL) minus the "
                         "collective-free loop: the software cost of the exchange step, no xGMI traffic")
    finally:
        dist.destroy_process_group()


def run_frag_batch(ctx, eng, hp, args, steps, warmup):
    """pure fragment-batch force throughput: the per-GPU batch cycles through the 220 fragments of the four example
    proteins; the first pass keeps the golden geometry (parity guard), later passes add 0.05 A of jitter."""
    from ai2bmd_amd.fragmentation import build_plan

    dev = ctx.dev
    H, L, S, R = hp["embedding_dimension"], hp["num_layers"], 8, hp["num_rbf"]
    rng = np.random.default_rng(1234 + ctx.rank)
    hv = harvested_pool()
    pool = [(z_, p_) for z_, p_, _, _ in hv]
    ref_e = [np.asarray([e_]) for _, _, e_, _ in hv]
    ref_f = [f_ for _, _, _, f_ in hv]
    zs, ps, sizes = [], [], []
    for i in range(args.frags_per_gpu):
        zf, pf = pool[i % len(pool)]
        zs.append(zf)
        ps.append(pf if i < len(pool) else pf - pf.mean(0) + rng.normal(0, 0.05, size=pf.shape))
        sizes.append(len(zf))
    end = np.cumsum(sizes)
    start = end - np.asarray(sizes)
    pos_h = np.concatenate(ps).astype(np.float32)
    z = torch.as_tensor(np.concatenate(zs), dtype=torch.int64).to(dev)
    pos = torch.as_tensor(pos_h).to(dev)
    e = torch.empty(len(start), device=dev)
    f = torch.empty(len(z), 3, device=dev)
    eng.forces_device(z, pos, start, end, e, f)
    torch.cuda.synchronize()
    ng = min(len(pool), args.frags_per_gpu)
    par = parity_check(f"fragment batch of {args.frags_per_gpu}, first {ng} fragments", e.cpu().numpy()[:ng],
                       f.cpu().numpy()[: int(end[ng - 1])], np.concatenate(ref_e[:ng]), np.concatenate(ref_f[:ng]))
    par["max_dF_over_ranks"] = ctx.max_over_ranks(par["max_dF"])

    def step():
        eng.forces_device(z, pos, start, end, e, f)

    k, el = timed_region(ctx, step, steps, warmup, args.min_seconds)
    assert torch.isfinite(f).all()
    ms = 1e3 * el / k
    E_tot = count_edges(pos_h, start, end, hp["cutoff"], hp["max_num_neighbors"])
    flops_step = 2.0 * fwd_flops(len(z), E_tot, H, L, S, R)
    eng.set_option("profile", 1)
    nprof = 2
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    roof = roofline_block(eng, nprof, "frag_batch", flops_step, ms)
    roof_hbm = roofline_hbm_block(eng, nprof, "frag_batch")
    eng.set_option("profile", 0)
    workload = (f"dipeptide/ACE-NME batch: {args.frags_per_gpu} fragments per GPU ({len(z)} atoms, {E_tot} edges) "
                f"harvested from the example proteins, 0.05 A jitter after the first pass, pure energy+force "
                f"evaluation, ViSNet H={H} L={L}")
    return dict(metric="fragment-batch forces/sec", value=k * args.frags_per_gpu * ctx.world / el,
                unit="fragments/s", steps=k, ms_per_step=ms, scaling="weak",
                config=dict(workload=workload, atoms_per_gpu=int(len(z)), edges_per_gpu=E_tot,
                            algorithmic_gflop_per_step_local=flops_step / 1e9,
                            atoms_per_s=k * len(z) * ctx.world / el),
                parity=par, roofline=dict(roof, **({"hbm": roof_hbm} if roof_hbm else {})))


def harvested_pool():
    """the 220 non-empty dipeptide / ACE-NME fragments of the four example proteins with the reference-source golden
    of each: [(z, pos, E64, F64)]"""
    pool = []
    for pname in ("chig", "trpcage", "ww", "abd"):
        g = load_golden(pname)
        ib = 0
        for b in range(len(g["start"])):
            a0, a1 = int(g["start"][b]), int(g["end"][b])
            if a1 > a0:
                pool.append((g["z"][a0:a1], g["pos_placed"][a0:a1], float(np.asarray(g["E_ref64_placed"][ib]).reshape(-1)[0]),
                             g["F_ref64_placed"][a0:a1]))
                ib += 1
    return pool


def run_frag_stream(ctx, eng, hp, args, conformations=None, golden_every=16, keep=None):
    """BASELINE configs[4] as stated - "Protein Unit Dataset throughput: 1M dipeptide conformations batched across
    the GPUs (pure force eval, no integrator)": `conformations` (default 1 000 000 over all ranks) NEW conformations,
    every one evaluated exactly once, in batches of --frags-per-gpu per rank.  Conformation = a fragment of the
    harvested pool (220 dipeptides / ACE-NMEs of the four example proteins) centred, + N(0, 0.05 A) jitter drawn from
    default_rng(1234 + rank).  All batches sit in pinned host memory; per step the next batch's z / pos go H2D on a
    copy stream into the other of two device buffers while the current batch is evaluated, and its energies / forces
    come back D2H into pinned buffers and are CONSUMED on the host (running float64 checksums of E and F) - all inside
    the timed region, PCIe-inclusive.  Every `golden_every`-th batch keeps the golden geometry in its first 220
    fragments and those results are compared with the reference-source golden (parity inside the stream).
    Reported as a `secondary`; never `value`."""
    dev = ctx.dev
    H, L, S, R = hp["embedding_dimension"], hp["num_layers"], 8, hp["num_rbf"]
    rng = np.random.default_rng(1234 + ctx.rank)
    pool = harvested_pool()
    nf = args.frags_per_gpu
    total = int(conformations if conformations is not None else args.conformations)
    nbatch = max(2, -(-total // (nf * ctx.world)))
    frs = [pool[i % len(pool)] for i in range(nf)]
    sizes = np.asarray([len(fr[0]) for fr in frs])
    end = np.cumsum(sizes)
    start = end - sizes
    natoms = int(end[-1])
    ng = min(len(pool), nf)
    ngat = int(end[ng - 1])
    z_all = np.concatenate([fr[0] for fr in frs])
    p_ctr = np.concatenate([fr[1] - fr[1].mean(0) for fr in frs]).astype(np.float32)
    p_gold = np.concatenate([fr[1] for fr in frs[:ng]]).astype(np.float32)
    E_gold = np.asarray([fr[2] for fr in frs[:ng]])
    F_gold = np.concatenate([fr[3] for fr in frs[:ng]])
    z_h = torch.as_tensor(z_all).pin_memory()   # (same fragment layout every batch: one host copy, uploaded every step)
    p_h = torch.empty(nbatch, natoms, 3, dtype=torch.float32).pin_memory()
    pn = p_h.numpy()
    for bi in range(nbatch):
        pn[bi] = p_ctr + np.float32(0.05) * rng.standard_normal(p_ctr.shape, dtype=np.float32)
        if bi % golden_every == 0:
            pn[bi, :ngat] = p_gold
    e_h = torch.empty(2, nf, dtype=torch.float32).pin_memory()
    f_h = torch.empty(2, natoms, 3, dtype=torch.float32).pin_memory()
    zd = [torch.empty(natoms, dtype=torch.int64, device=dev) for _ in range(2)]
    pd = [torch.empty(natoms, 3, dtype=torch.float32, device=dev) for _ in range(2)]
    ed = [torch.empty(nf, device=dev) for _ in range(2)]
    fd = [torch.empty(natoms, 3, device=dev) for _ in range(2)]
    main = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(device=dev)
    up = [torch.cuda.Event() for _ in range(2)]      # H2D of buffer b complete
    free = [torch.cuda.Event() for _ in range(2)]    # evaluation that read buffer b complete
    down = [torch.cuda.Event() for _ in range(2)]    # D2H of result buffer b complete
    state = dict(i=0, sumE=0.0, sumF=0.0, sumF2=0.0, consumed=0, gold=0, gold_dF=0.0, gold_dE=0.0, held=[None, None])

    def upload(bi, b):
        with torch.cuda.stream(copy):
            copy.wait_event(free[b])
            zd[b].copy_(z_h, non_blocking=True)
            pd[b].copy_(p_h[bi % nbatch], non_blocking=True)
            up[b].record(copy)

    def consume(b):
        """host side of the stream: the results of the batch held in result buffer b have landed"""
        bi = state["held"][b]
        if bi is None:
            return
        down[b].synchronize()
        e_, f_ = e_h[b].numpy(), f_h[b].numpy()
        state["sumE"] += float(e_.sum(dtype=np.float64))
        state["sumF"] += float(np.abs(f_).sum(dtype=np.float64))
        state["sumF2"] += float(np.square(f_, dtype=np.float64).sum())
        state["consumed"] += 1
        if bi % golden_every == 0 and bi < nbatch:
            try:
                pr = parity_check(f"streamed batch {bi}, golden block", e_[:ng], f_[:ngat], E_gold, F_gold,
                                  collective=False)
                state["gold_dF"] = max(state["gold_dF"], pr["max_dF"])
                state["gold_dE"] = max(state["gold_dE"], pr["max_dE"])
            except ParityFailure as exc:  # settled over the ranks after the loop (no collective inside the clock)
                state["gold_fail"] = state.get("gold_fail") or str(exc)
            state["gold"] += 1
        state["held"][b] = None

    for b in range(2):
        free[b].record(main)
        down[b].record(main)
    upload(0, 0)

    def step():
        i = state["i"]
        b = i & 1
        upload(i + 1, b ^ 1)                      # next batch flies while this one is evaluated
        main.wait_event(up[b])
        consume(b)                                # result buffer b (batch i - 2) is read before it is overwritten
        eng.forces_device(zd[b], pd[b], start, end, ed[b], fd[b])
        free[b].record(main)
        with torch.cuda.stream(copy):
            copy.wait_event(free[b])
            e_h[b].copy_(ed[b], non_blocking=True)
            f_h[b].copy_(fd[b], non_blocking=True)
            down[b].record(copy)
        state["held"][b] = i
        state["i"] = i + 1

    # warm-up outside the clock: two batches through the same pipe, their results discarded, counters reset
    step()
    step()
    copy.synchronize()
    torch.cuda.synchronize()
    state.update(i=0, sumE=0.0, sumF=0.0, sumF2=0.0, consumed=0, gold=0, gold_dF=0.0, gold_dE=0.0, held=[None, None])
    upload(0, 0)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(nbatch):
        step()
    # the two batches still in flight, in BATCH order (the float64 checksums are sums in the order 0, 1, 2, ...)
    for b in sorted((0, 1), key=lambda b_: -1 if state["held"][b_] is None else state["held"][b_]):
        consume(b)
    copy.synchronize()
    ctx.barrier()
    el = ctx.max_over_ranks(time.perf_counter() - t0)
    k = nbatch
    if not _agreed(not state.get("gold_fail")):
        raise ParityFailure(state.get("gold_fail") or "PARITY FAILURE in the streamed golden blocks of another rank")
    assert state["consumed"] == nbatch and math.isfinite(state["sumE"]) and math.isfinite(state["sumF"])
    if keep is not None:  # tests: the inputs of every batch, to re-evaluate them one by one outside the pipe
        keep.update(z=z_all, pos=pn, start=start, end=end)
    ms = 1e3 * el / k
    E_tot = count_edges(pn[0], start, end, hp["cutoff"], hp["max_num_neighbors"])
    flops_batch = 2.0 * fwd_flops(natoms, E_tot, H, L, S, R)
    return dict(metric="fragment-batch forces/sec, streamed conformations (Protein Unit Dataset throughput)",
                value=k * nf * ctx.world / el, unit="fragments/s", steps=k, ms_per_step=ms, scaling="weak",
                parity=dict(max_dF=state["gold_dF"], max_dE=state["gold_dE"], golden_blocks_checked=state["gold"],
                            max_dF_over_ranks=ctx.max_over_ranks(state["gold_dF"])),
                config=dict(workload=(f"{k * nf * ctx.world} NEW dipeptide/ACE-NME conformations, each evaluated once: {k} "
                                      f"batches of {nf} per GPU ({natoms} atoms, ~{E_tot} edges each), default_rng(1234 + "
                                      f"rank), 0.05 A jitter; pinned host buffers, double-buffered H2D of z/pos on a copy "
                                      f"stream, D2H of E/F consumed on the host (checksums), all inside the timed region "
                                      f"(PCIe-inclusive); golden block re-checked every {golden_every}th batch"),
                            conformations=k * nf * ctx.world, seconds=el,
                            atoms_per_gpu=natoms, atoms_per_s=k * natoms * ctx.world / el,
                            gflops=k * flops_batch * ctx.world / el / 1e9,
                            checksum=dict(sum_E=state["sumE"], sum_absF=state["sumF"], sum_F2=state["sumF2"],
                                          note="rank 0's batches, float64 accumulation on the host"),
                            h2d_bytes_per_step=natoms * 20, d2h_bytes_per_step=natoms * 12 + nf * 4))


def _cpu_evaluator(hp, sd):
    """-> (kind, origin, fn(z, pos, start, end)): the REFERENCE's own ViSNet model (kind "reference") - imported from
    /root/reference/src where that tree exists, else from oracle/_ref, the same modules byte-compiled from that tree by
    oracle/make_ref.py (a build output that travels to the GPU box like the .so) - through oracle/ref_import.py + the
    shims for its un-vendored wheels.  Only when neither is there: the oracle port (kind "port")."""
    from oracle.ref_import import import_reference_create_model, reference_model_source

    origin = reference_model_source()
    if origin is not None:
        create_model = import_reference_create_model()
        model = create_model(hp)
        model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
        model = model.float().eval()
        for p_ in model.parameters():
            p_.requires_grad = False

        def fn(z, pos, start, end):
            sizes = np.asarray(end) - np.asarray(start)
            batch = np.repeat(np.cumsum(sizes > 0) - 1, sizes)
            E, F = model(dict(z=torch.as_tensor(z), pos=torch.as_tensor(pos), batch=torch.as_tensor(batch)))
            return E.detach(), F.detach()

        return "reference", ("/root/reference/src/ViSNet/model" if origin == "source"
                             else "oracle/_ref (the reference's ViSNet/model package byte-compiled by oracle/make_ref.py)"), fn
    from oracle.visnet_oracle import ViSNetOracle

    o = ViSNetOracle(hp, sd, torch.float32)
    return "port", "oracle/visnet_oracle.py (fp32 torch + autograd)", o.energy_forces


def physical_core_count():
    """sockets x cores per socket from `lscpu`, as the reference counts them (src/utils/system.py:28-45)"""
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, check=True).stdout.splitlines()
        cps = next(x for x in out if "Core(s) per socket:" in x).split(":")[1].strip()
        soc = next(x for x in out if "Socket(s):" in x).split(":")[1].strip()
        return int(cps) * int(soc)
    except Exception:
        return None


def _median_rate(fn, warm=2, timed=5, budget_s=40.0):
    """BASELINE.md section 3 protocol: `warm` warm-up evaluations, then `timed` (>= 5) timed ones, median.  A layout
    so slow that this would blow the bench's budget keeps >= 3 timed evaluations and says so (n is reported)."""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    ts = []
    while len(ts) < timed:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if len(ts) >= 3 and time.perf_counter() - t_start > budget_s:
            break
    return 1.0 / float(np.median(ts)), len(ts)


def cpu_baseline_md(plan, prot, hp, sd):
    """The reference's CPU path timed on this node's host cores, same Chignolin fragment batch, same seeded weights,
    same process - force evaluation only (no cap-hydrogen relaxation, no integrator).  Layouts:
    (i)  the reference's LITERAL CPU configuration: two fragment partitions evaluated by two Python threads on one
         shared model with torch.set_num_threads(physical_cores // 2) (device_strategy.py:176,252-263;
         physical cores from lscpu as utils/system.py:28-45 counts them);
    (ii) one partition at the best of 8 / 16 / 32 intra-op threads
         (torch's intra-op threading saturates early on these small tensors: a 128-core host runs the literal
         layout several times SLOWER than 16 threads).
    Protocol: 2 warm-up, 5 timed evaluations, median.  `value` = the fastest layout (the most
    favourable figure for the CPU), `cores` = the threads it used, `physical_cores` = what the node has."""
    from concurrent.futures import ThreadPoolExecutor

    from ai2bmd_amd.device_strategy import device_ranges
    from ai2bmd_amd.fragmentation import fragment_positions

    pos = fragment_positions(plan, prot.positions).astype(np.float32)
    kind, origin, evaluate = _cpu_evaluator(hp, sd)
    ncpu = os.cpu_count() or 1
    phys = physical_core_count() or ncpu
    parts = []
    for f0, f1 in device_ranges(plan.start, plan.end, 2):
        a0, a1 = int(plan.start[f0]), int(plan.end[f1 - 1])
        parts.append((plan.z[a0:a1], pos[a0:a1], plan.start[f0:f1] - a0, plan.end[f0:f1] - a0))
    pool = ThreadPoolExecutor(2)

    def one():
        evaluate(plan.z, pos, plan.start, plan.end)

    def both():
        list(pool.map(lambda p_: evaluate(*p_), parts))

    layouts = {}
    # (i) the reference's literal layout
    nt_ref = max(1, phys // 2)
    torch.set_num_threads(nt_ref)
    r, n = _median_rate(both)
    layouts["reference_layout"] = dict(partitions=2, threads_per_partition=nt_ref, threads=2 * nt_ref,
                                       evals_per_s=r, timed_evaluations=n)
    # (ii) best-of sweep, one partition
    best_nt, best_t = None, None
    for nt in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
        torch.set_num_threads(nt)
        one()
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    r, n = _median_rate(one, budget_s=25.0)
    layouts["single_partition_best_of_8_16_32"] = dict(partitions=1, threads_per_partition=best_nt, threads=best_nt,
                                                       evals_per_s=r, timed_evaluations=n)
    pool.shutdown()
    win = max(layouts, key=lambda k_: layouts[k_]["evals_per_s"])
    return dict(value=layouts[win]["evals_per_s"], unit="force evaluations/s", cores=layouts[win]["threads"], kind=kind,
                physical_cores=phys, host_hw_threads=ncpu, layout_of_value=win, layouts=layouts,
                reference_layout_evals_per_s=layouts["reference_layout"]["evals_per_s"], source=origin,
                protocol="2 warm-up + 5 timed evaluations per layout, median (BASELINE.md section 3)",
                sample_short=(f"{kind} ViSNet model, fp32, energy+force evaluations of the Chignolin fragment batch "
                              f"(B={len(plan.start)}, N={len(plan.z)}); 2 warm-up + 5 timed, median; force evaluation only "
                              f"(GPU value = full MD step); {phys} physical cores"),
                sample=f"energy+force evaluations of the Chignolin fragment batch (B={len(plan.start)}, "
                       f"N={len(plan.z)}), fp32, by {origin}; layouts: the reference's literal CPU configuration "
                       f"(2 partitions x {nt_ref} = physical_cores // 2 threads, device_strategy.py:176,252-263), one "
                       f"partition at {best_nt} threads (best of 8/16/32); on a host with "
                       f"{phys} physical cores / {ncpu} hardware threads; force evaluation only (cap-hydrogen "
                       f"relaxation and integrator excluded)")


def run_stub(ctx, args):
    """launch-logic self-test (tests/test_capi_and_host.py): every rank 'steps' by sleeping; exercises the self
    launch, the barrier / max-over-ranks timing and the one-line report without a GPU."""
    def step():
        time.sleep(0.002 * (1 + ctx.rank))

    k, el = timed_region(ctx, step, args.steps, args.warmup, min(args.min_seconds, 0.05))
    return dict(metric="stub", value=k / el, unit="steps/s", steps=k, ms_per_step=1e3 * el / k, scaling="strong",
                config=dict(workload="STUB: sleep-based fake step on CPU over gloo - launch-logic test, NOT a measurement"))


# ------------------------------------------------------------------------------------------------------------
class StdoutGuard:
    """Only the result line reaches the real stdout.  RCCL prints a version banner through C stdio when a communicator is
    created; with stdout a pipe that text sits in a libc buffer until the process exits - AFTER the JSON line, which
    then is no longer the last line the driver reads (seen on the first run that created a communicator).  From the
    first line of main() file descriptor 1 is stderr for every library and every rank; `result_line` writes to the
    saved descriptor."""

    def __init__(self):
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def result_line(self, line: str):
        sys.stdout.flush()
        os.write(self.real, (line + "\n").encode())


_GUARD = None


def main():
    global _GUARD
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    _GUARD = StdoutGuard()
    ctx = Ctx(args)
    global _CTX
    _CTX = ctx
    if ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}")
    secondary = []
    dropped = []
    cpu = None

    def guarded(label, fn):
        """A SECONDARY that fails its parity guard must not take the headline with it (seen once: the opt-in split-3
        batch on rank 1 of a two-rank shared-GPU run).  ParityFailure is raised on all ranks at the same point, so all
        ranks drop the secondary together; it is named, with the reason, in config.secondary_dropped."""
        try:
            return fn()
        except ParityFailure as exc:
            dropped.append(dict(secondary=label, reason=str(exc)[:300]))
            print(f"SECONDARY DROPPED ({label}): {exc}", file=sys.stderr, flush=True)
            if not ctx.stub:
                torch.cuda.synchronize()
            return None

    if args.stub:
        res = run_stub(ctx, args)
        data = "STUB"
    else:
        from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # seeded weight generator
        from ai2bmd_amd.visnet_calculator import ViSNetEngine

        hp = default_hparams()
        sd = make_state_dict(hp, seed=2024)
        eng = ViSNetEngine(hp, sd, ctx.dev)
        if args.chunk_edges:
            eng.set_option("max_chunk_edges", args.chunk_edges)
        for kv in filter(None, os.environ.get("VSN_OPTS", "").split(",")):  # tuning aid: VSN_OPTS=fuse_fwd=0,overlap=0
            k_, v_ = kv.split("=")
            eng.set_option(k_, int(v_))
        data = ("synthetic (seeded random weights at the reference's default hyper-parameters; "
                "geometry = reference examples/*.pdb fixtures)")
        if ctx.share_gpu:
            data = "SHARED-GPU VALIDATION RUN (all ranks on one device over gloo): NOT a measurement; " + data
        if args.workload.endswith("_md"):
            n_main, pre = args.steps, None
            if args.workload == "chig_md" and not args.emulate_shard and not args.share_gpu:
                # BASELINE configs[1] IS the 1000-step loop: it is always timed; a shorter --steps K is timed first and is
                # what `value` is quoted on (the driver's contract: EXACTLY K steps), the loop rides in config.c2_loop
                n_main = max(args.steps, C2_STEPS)
                pre = args.steps if args.steps < C2_STEPS else None
            res, (plan, prot, md) = run_md(ctx, eng, hp, args.workload[:-3], args, n_main, args.warmup, pre_steps=pre)
            if ctx.world == 1 and not args.emulate_shard:
                # the reference-shaped seam (host numpy in, host numpy out => H2D + D2H over PCIe every call);
                # reported for information, never as `value`
                from ai2bmd_amd.fragment import FragmentData, make_batch_index
                from ai2bmd_amd.fragmentation import fragment_positions
                from ai2bmd_amd.visnet_calculator import ViSNetModel

                seam = ViSNetModel.__new__(ViSNetModel)
                seam.device, seam.engine, seam.stream = ctx.dev, eng, torch.cuda.Stream(device=ctx.dev)
                fpos = fragment_positions(plan, prot.positions).astype(np.float32)
                fd = FragmentData(plan.z, fpos, plan.start, plan.end, make_batch_index(plan.start, plan.end))
                for _ in range(5):
                    seam.dl_potential_loader(fd)
                t1 = time.perf_counter()
                for _ in range(50):
                    seam.dl_potential_loader(fd)
                res["config"]["host_seam_evals_per_s_pcie_inclusive"] = 50 / (time.perf_counter() - t1)
                # what north_star literally describes: the reference-shaped calculator call, host numpy in and out -
                # DLBondedCalculator(prot) = DistanceFragment.get_fragments (cap-H placement + L-BFGS on the device,
                # fragments back to the host) + the seam + DipeptideBondedCombiner on the host (bonded.py:102-123)
                from ai2bmd_amd.bonded import DLBondedCalculator
                from ai2bmd_amd.distancefrag import DistanceFragment

                calc = DLBondedCalculator.from_models([seam], fragment_method=DistanceFragment(device=ctx.dev))
                calc.fragment_method.fragment(prot)
                for _ in range(5):
                    calc(prot)
                t1 = time.perf_counter()
                for _ in range(50):
                    calc(prot)
                res["config"]["reference_shaped_step_calls_per_s"] = 50 / (time.perf_counter() - t1)
                del calc
            if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
                cpu = cpu_baseline_md(plan, prot, hp, sd)
            del md
            if ctx.world == 1 and not args.emulate_shard and not args.no_secondary:
                # the all-gather of the sharded path, measured instead of assumed
                res["config"]["rccl1_allgather"] = {p_: collective_delta(ctx, eng, hp, p_, args) for p_ in ("chig", "ww")}
        elif args.workload == "frag_stream":
            res = run_frag_stream(ctx, eng, hp, args)
        else:
            res = run_frag_batch(ctx, eng, hp, args, args.steps, args.warmup)
        if not args.no_secondary and not args.emulate_shard and args.workload == "chig_md":
            # the rest of BASELINE.json's metric in the same line: fragment-batch forces/s (weak scaling, no
            # collective), Trp-cage (configs[2]) and, sharded over N > 1 GPUs, the WW domain (configs[3])
            extra_md = ["trpcage", "ww"]  # (WW at N = 1 as well: the 1-GPU anchor of the sharded curve)
            for pname in extra_md:
                got = guarded(f"{pname}_md", lambda: run_md(ctx, eng, hp, pname, args, 400 if pname == "trpcage" else 300, 10))
                if got is not None:
                    secondary.append(got[0])
                del got
            r_fb = guarded("frag_batch", lambda: run_frag_batch(ctx, eng, hp, args, 6, 1))
            if r_fb is not None:
                secondary.append(r_fb)
            # configs[4] as stated: NEW conformations every step, H2D / D2H inside the timed region
            r_fs = guarded("frag_stream_pcie", lambda: run_frag_stream(ctx, eng, hp, args))
            if r_fs is not None:
                secondary.append(r_fs)
            # configs[1] (ii): + MM non-bonded.  Short, and on a 10x stiffer tether: with random ViSNet weights nothing
            # but the tether holds polar hydrogens against the Coulomb term (AMBER gives them no LJ core), and at
            # 5 eV/A^2 the structure loses 20 % of its edges within 300 steps - the edge-count assertion of run_md
            # refuses such a run
            a_mm = argparse.Namespace(**vars(args))
            got = guarded("chig_md_mm", lambda: run_md(ctx, eng, hp, "chig", a_mm, 200, 10, mm=True, tether_k=50.0))
            if got is not None:
                r_mm = got[0]
                r_mm["metric"] += " + MM non-bonded"
                secondary.append(r_mm)
            del got
            # SURVEY 8(d) small variant (H = 128, L = 6), its own engine and its own reference-source golden
            hp_s = default_hparams(embedding_dimension=128, num_layers=6)
            eng_s = ViSNetEngine(hp_s, make_state_dict(hp_s, seed=2024), ctx.dev)
            a_s = argparse.Namespace(**vars(args))
            got = guarded("chig_md_h128l6", lambda: run_md(ctx, eng_s, hp_s, "chig", a_s, 600, 10, gold_suffix="_h128l6"))
            if got is not None:
                r_s = got[0]
                r_s["metric"] += " (small variant H=128 L=6)"
                secondary.append(r_s)
            del got, eng_s
            # opt-in arithmetic mode, NEVER the headline: the grouped products as 3 x bf16 split MFMA products with fp32
            # accumulation (csrc/gemm_s3.h).  Same workload, same goldens, same tolerance; its own parity block.
            split3_dtype = "f32 operands as 3 x bf16 split terms (six bf16 MFMA products per k-block), f32 accumulate"
            split3_note = ("fp32-EQUIVALENT FLOP/s of the split products against the fp32 matrix peak (the bf16 pipe does "
                           "6/16 of the fp32 form's matrix cycles): a mode label, not an fp32 MFMA utilisation")
            # (round 6 dropped these two from shared-GPU runs: a neighbour PROCESS on the device made packed fp32 VALU
            #  instructions return wrong halves - LAB_NOTES section 15.  The library is built without those instructions
            #  now, so the mode runs in every configuration again.)
            run_split3 = True
            got = None
            if run_split3:
                eng.set_option("gemm_split3", 1)
                try:
                    got = guarded("chig_md_split3_optin", lambda: run_md(ctx, eng, hp, "chig", args, C2_STEPS, 10))
                finally:
                    eng.set_option("gemm_split3", 0)
            if got is not None:
                r_3 = got[0]
                r_3["metric"] += " (opt-in mode gemm_split3)"
                r_3["dtype"] = split3_dtype
                r_3["roofline"]["note"] = split3_note
                r_3["keep_parity"] = True
                secondary.append(r_3)
            del got
            # the same mode on the fragment batch (plain products on the 128 x 128 split tile; the fused panel
            # products keep their fp32 MFMA kernels)
            r_b3 = None
            if run_split3:
                eng.set_option("gemm_split3", 1)
                try:
                    r_b3 = guarded("frag_batch_split3_optin", lambda: run_frag_batch(ctx, eng, hp, args, 6, 1))
                finally:
                    eng.set_option("gemm_split3", 0)
            if r_b3 is not None:
                r_b3["metric"] += " (opt-in mode gemm_split3)"
                r_b3["dtype"] = split3_dtype
                r_b3["roofline"]["note"] = split3_note
                r_b3["keep_parity"] = True
                secondary.append(r_b3)
    if dropped:
        res["config"]["secondary_dropped"] = dropped
    out = dict(
        metric=res["metric"], value=res["value"], unit=res["unit"], n_gpus=ctx.world, steps=res["steps"],
        steps_requested=args.steps_requested, warmup=args.warmup, ms_per_step=res["ms_per_step"], higher_is_better=True,
        scaling=res["scaling"], vs_baseline=None, dtype="f32", data=data, config=res["config"],
        rccl_ranks=ctx.world, backend=ctx.backend or "none (single process)",
    )
    for key in ("parity", "roofline"):
        if res.get(key):
            out[key] = res[key]
    if "parity" in res:
        out["parity_max_dF"] = res["parity"]["max_dF_over_ranks"]
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if secondary:
        out["secondary"] = [dict(metric=r["metric"], value=r["value"], unit=r["unit"], n_gpus=ctx.world,
                                 steps=r["steps"], ms_per_step=r["ms_per_step"], scaling=r["scaling"],
                                 config=r["config"],
                                 **({"parity_max_dF": r["parity"]["max_dF_over_ranks"]} if "parity" in r else {}),
                                 **({"parity": r["parity"], "dtype": r["dtype"]} if r.get("keep_parity") else {}),
                                 **({k_: r[k_] for k_ in ("roofline",) if r.get(k_)})) for r in secondary]
    if ctx.rank == 0:
        emit(out, args)
    ctx.close()


# ------------------------------------------------------------------------------------------------------------
# the ONE line the driver parses: compact by construction (a 23 KB line was not parsed)
# ------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000  # bytes; the driver keeps an 8 KB tail of stdout


def _short(v, sig=6, smax=240):
    """floats to `sig` significant digits, long strings cut, recursively"""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if math.isfinite(v) else None
    if isinstance(v, str):
        return v if len(v) <= smax else v[: smax - 1] + "~"
    if isinstance(v, dict):
        return {k: _short(x, sig, smax) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x, sig, smax) for x in v]
    return _short(float(v), sig, smax) if hasattr(v, "__float__") else str(v)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


SECONDARY_KEYS = {  # metric of a secondary -> short key in config.secondary_summary
    "MD steps/sec on Trp-cage": "trpcage_md",
    "MD steps/sec on WW domain": "ww_md",
    "fragment-batch forces/sec": "frag_batch",
    "fragment-batch forces/sec, streamed conformations (Protein Unit Dataset throughput)": "frag_stream_pcie",
    "MD steps/sec on Chignolin + MM non-bonded": "chig_md_mm",
    "MD steps/sec on Chignolin (small variant H=128 L=6)": "chig_md_h128l6",
    "MD steps/sec on Chignolin (opt-in mode gemm_split3)": "chig_md_split3_optin",
    "fragment-batch forces/sec (opt-in mode gemm_split3)": "frag_batch_split3_optin",
}


def compact_line(full: dict, detail_path: str | None = None, limit: int = LINE_LIMIT) -> dict:
    """The result line: everything the measurement contract names, nothing nested deeper than it has to be.  The full
    record (every secondary's own roofline, every CPU layout, every scatter kernel) goes to `detail_path`."""
    c = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "steps_requested", "warmup", "ms_per_step",
                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "rccl_ranks", "backend"))
    cfg = full.get("config", {})
    cc = _pick(cfg, ("workload", "exchange", "state_checksum", "edges_local", "frag_atoms_local",
                     "algorithmic_gflop_per_step_local",
                     "host_seam_evals_per_s_pcie_inclusive", "reference_shaped_step_calls_per_s",
                     "reference_caller_on_hip_seam_calls_per_s", "step_bound_ms"))
    if "c2_loop" in cfg:
        cc["c2_loop"] = _pick(cfg["c2_loop"], ("steps", "ms_per_step", "value", "unit"))
    if cfg.get("secondary_dropped"):
        cc["secondary_dropped"] = cfg["secondary_dropped"]
    if isinstance(cfg.get("rccl1_allgather"), dict):  # the measured software cost of the one collective per step
        cc["rccl1_allgather_delta_us"] = {k: (v or {}).get("delta_us") for k, v in cfg["rccl1_allgather"].items()}
        cc["p2p1_exchange_delta_us"] = {k: (v or {}).get("p2p_delta_us") for k, v in cfg["rccl1_allgather"].items()}
    if "requested_run" in cfg:  # (records of rounds <= 5a: value on the 1000-step loop, the requested K beside it)
        cc["requested_run"] = _pick(cfg["requested_run"], ("steps", "ms_per_step", "value", "unit"))
    summ = {}
    for r in full.get("secondary", []):
        key = SECONDARY_KEYS.get(r["metric"], r["metric"])
        e = _pick(r, ("value", "unit", "steps", "ms_per_step"))
        rf = r.get("roofline") or {}
        if "frac" in rf:
            e["mfma_frac"] = rf["frac"]
        if isinstance(rf.get("hbm"), dict) and "frac" in rf["hbm"]:
            e["hbm_frac"] = rf["hbm"]["frac"]
        if "parity_max_dF" in r:
            e["max_dF"] = r["parity_max_dF"]
        e.update(_pick(r.get("config", {}), ("conformations", "atoms_per_s", "gflops", "checksum")))
        summ[key] = e
    if summ:
        cc["secondary_summary"] = summ
    c["config"] = cc
    if "roofline" in full:
        r = full["roofline"]
        cr = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step",
                       "avg_launch_us", "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch", "step_frac",
                       "step_tflops"))
        if isinstance(r.get("hbm"), dict):
            h = r["hbm"]
            ch = _pick(h, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step",
                           "avg_launch_us", "algorithmic_bytes_per_launch"))
            if "rocprof" in h:
                ch["rocprof"] = _pick(h["rocprof"], ("avg_us", "frac_from_trace", "live_over_trace"))
            cr["hbm"] = ch
        if isinstance(r.get("reverse_walks"), dict):
            cr["reverse_walks"] = _short(r["reverse_walks"], sig=4)
        if isinstance(r.get("step_bound"), dict):
            cr["step_bound"] = _pick(r["step_bound"], ("step_bound_ms", "covered_ms"))
        c["roofline"] = cr
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        c["cpu_baseline"] = _pick(b, ("value", "unit", "cores", "kind", "physical_cores", "host_hw_threads",
                                      "layout_of_value", "reference_layout_evals_per_s", "source", "sample_short"))
        c["cpu_baseline"]["sample"] = c["cpu_baseline"].pop("sample_short", None) or b.get("sample", "")
    if "parity" in full:
        c["parity"] = _pick(full["parity"], ("max_dE", "max_dF", "force_mae", "max_abs_F", "pipeline_max_dF",
                                             "max_dF_over_ranks"))
    if detail_path:
        c["full_detail"] = detail_path
    c = _short(c)
    # belt and braces: never exceed the limit - drop the least essential pieces in order until it fits
    for drop in (("config", "secondary_summary", "frag_batch_split3_optin"), ("roofline", "reverse_walks"),
                 ("config", "secondary_summary", "chig_md_split3_optin"), ("data",), ("config", "secondary_summary")):
        if len(json.dumps(c, separators=(",", ":"))) <= limit:
            break
        d = c
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    return c


def emit(full: dict, args) -> None:
    """full record -> gpurun_out/ (scratch; copied into profiles/ by tools/profile_round.sh) and ONE stderr line;
    compact line -> the LAST line of stdout."""
    path = None
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join("gpurun_out", f"bench_full_{args.workload}_n{full['n_gpus']}.json")
        with open(os.path.join(ROOT, path), "w") as fh:
            json.dump(full, fh)
    except OSError:
        path = None
    print("BENCH_FULL_DETAIL (not the result line) " + json.dumps(full), file=sys.stderr, flush=True)
    line = json.dumps(compact_line(full, path), separators=(",", ":"))
    assert len(line) <= LINE_LIMIT and "\n" not in line, len(line)
    if _GUARD is not None:
        _GUARD.result_line(line)
    else:
        print(line, flush=True)


if __name__ == "__main__":
    main()
