"""The node walks against the HBM roofline: one table per workload, written to profiles/<tag>_node_walks.md.

    python tools/walk_table.py <tag>        (reads profiles/<tag>_*; run after tools/profile_round.sh on the same build)

Per kernel: launches per step / batch, average duration from the rocprofv3 KERNEL TRACE, ALGORITHMIC bytes per launch
(every distinct array the launch reads or writes, once: `vsn_walk_alg_bytes` of csrc/engine.hip, the same function the
live figures of bench.py come from; where the bench record of the round holds the live per-launch average of a walk -
launch variants mixed as they occur in a step - that figure is used), the counter bytes per launch (2 FETCH_SIZE + WRITE_SIZE, separate PMC passes), their
ratio, the achieved rate on algorithmic bytes against 8 TB/s, and the launch's bound at the sustained 6.3 TB/s.
`step_bound_ms` = sum over the GEMM launches of flop / 116 TFLOP/s + over these launches of bytes / 6.3 TB/s.
"""
from __future__ import annotations

import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, S, NH = 256, 8, 8
HBM_PEAK, HBM_SUST, MFMA_SUST = 8000e9, 6300e9, 116e12


def alg_floats(kernel, n, e, f0=1, f1=0):
    """distinct floats one launch touches = vsn_walk_alg_bytes / 4: the byte model lives in csrc/engine.hip ONLY (the
    live figures of bench.py come from the same function); this is a ctypes call, not a restatement.  Default flags =
    the common launch of a Chignolin step (next-layer LayerNorm fused, edge update present, accumulating adjoint);
    k_bwd_hf2 sums 2 K-slices of g_m and 3 of g_A there."""
    sys.path.insert(0, ROOT)
    from ai2bmd_amd import capi

    if kernel == "k_bwd_hf2" and f1 == 0:
        f0, f1 = 2, 3
    by = capi.lib().vsn_walk_alg_bytes(kernel.encode(), H, S, NH, float(n), float(e), int(f0), int(f1))
    return None if by < 0 else by / 4.0


def load_csv(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def pooled(rows, name):
    hit = [r for r in rows if name in r["kernel"]]
    if name == "k_node_update":
        hit = [r for r in hit if "k_node_update" in r["kernel"] and "bwd" not in r["kernel"]]
    if name == "k_edge_attn":
        hit = [r for r in hit if "k_edge_attn_update" not in r["kernel"]]
    if name == "k_edge_update":
        hit = [r for r in hit if "bwd" not in r["kernel"]]
    if not hit:
        return None
    calls = sum(int(float(r["calls"])) for r in hit)
    return calls, sum(float(r["total_ns"]) for r in hit) / calls


def pmc_bytes(rows, name):
    for r in rows:
        k = r["kernel"]
        if name in k and (name != "k_node_update" or "bwd" not in k) and (name != "k_edge_attn" or "update" not in k) \
                and (name != "k_edge_update" or "bwd" not in k):
            try:
                return float(r["hbm_MB"]) * 1e6
            except (KeyError, ValueError):
                return None
    return None


def table(tag, wl, n, e, per, kernels, launches_of, live=None):
    """live: {kernel: algorithmic MB per launch} as the engine reported them in the bench run of this round (launch
    variants averaged as they occur in a step); used where present so that profiles/ and the bench line agree"""
    live = live or {}
    ks = load_csv(os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
    pm = load_csv(os.path.join(ROOT, "profiles", f"{tag}_{wl}_pmc.csv"))
    out = [f"| kernel | launches per {per} | avg µs (trace) | algorithmic MB | counter MB | counter / algorithmic | "
           f"achieved TB/s (frac of 8) | bound µs at 6.3 TB/s |", "|---|---|---|---|---|---|---|---|"]
    tot_meas = tot_bound = 0.0
    for k in kernels:
        p = pooled(ks, k)
        if p is None:
            continue
        calls, avg_ns = p
        lp = launches_of(k)
        by = live[k] * 1e6 if k in live else 4.0 * alg_floats(k, n / lp["chunks"], e / lp["chunks"])
        cb = pmc_bytes(pm, k)
        rate = by / (avg_ns * 1e-9)
        bound_us = by / HBM_SUST * 1e6
        tot_meas += lp["per"] * avg_ns * 1e-3
        tot_bound += lp["per"] * bound_us
        out.append(f"| `{k}` | {lp['per']} | {avg_ns / 1e3:.2f} | {by / 1e6:.1f} | "
                   f"{(cb / 1e6 if cb else float('nan')):.1f} | {(cb / by if cb else float('nan')):.2f} | "
                   f"{rate / 1e12:.2f} ({rate / HBM_PEAK:.2f}) | {bound_us:.1f} |")
    return out, tot_meas, tot_bound


def main():
    tag = sys.argv[1]
    full = None
    for cand in (os.path.join(ROOT, "profiles", f"{tag}_bench_full.json"),):
        if os.path.exists(cand):
            full = json.load(open(cand))
    n, e = 391, 6657
    gemm = None
    if full:
        n = full["config"].get("frag_atoms_local", n)
        e = full["config"].get("edges_local", e)
        gemm = full["roofline"]
    L = 9
    chig = dict(k_edge_attn_update=L - 1, k_edge_attn=1, k_node_update=L, k_bwd_hf1=L - 1, k_bwd_hf2=L - 2,
                k_bwd_attn_S=L, k_bwd_norm_update=L)
    lines = [f"# Node walks against the HBM roofline ({tag})", "",
             "Produced by `tools/walk_table.py` from this round's kernel trace and PMC passes (same build; "
             "`tools/profile_round.sh`).  Algorithmic bytes = every distinct array of the launch once "
             "(`csrc/engine.hip`, `tools/walk_table.py::alg_floats`).", "",
             f"## Chignolin MD step (N = {n} fragment atoms, E = {e} edges, H = 256, L = 9)", ""]
    live = {}
    if gemm:
        for k, v in (gemm.get("reverse_walks") or {}).items():
            live[k] = v["alg_MB"]
        for k, v in ((gemm.get("hbm") or {}).get("all_scatter_kernels") or {}).items():
            live[k] = v["algorithmic_bytes_per_launch"] / 1e6
        live.pop("k_edge_attn", None)  # (the live record pools k_edge_attn and k_edge_attn_update: the trace separates them)
    t, meas, bound = table(tag, "chig_md", n, e, "step", list(chig), lambda k: dict(per=chig[k], chunks=1), live=live)
    lines += t
    lines += ["", f"Node walks: {meas:.0f} µs of the step measured, {bound:.0f} µs at 6.3 TB/s on algorithmic bytes."]
    if gemm:
        g_meas = gemm["all_gemm_ms_per_step"] * 1e3
        g_bound = 0.0
        sb = gemm.get("step_bound") or {}
        lines += [f"GEMM launches: {g_meas:.0f} µs measured (HIP events, bracket cost removed).  "
                  f"`step_bound_ms` (live, bench.py: GEMM launches at 116 TFLOP/s + timed walks at 6.3 TB/s) = "
                  f"{sb.get('step_bound_ms', float('nan')):.3f} ms against {sb.get('covered_ms', float('nan')):.3f} ms "
                  f"measured for the same launches; whole step {full['ms_per_step']:.3f} ms."]
    # batch
    bfull = os.path.join(ROOT, "profiles", f"{tag}_bench_full_frag_batch.json")
    nb, eb = 86016, 1372000
    if os.path.exists(bfull):
        b = json.load(open(bfull))
        nb, eb = b["config"]["atoms_per_gpu"], b["config"]["edges_per_gpu"]
    ks = load_csv(os.path.join(ROOT, "profiles", f"{tag}_frag_batch4096_kernel_stats.csv"))
    p = pooled(ks, "k_bwd_gm_fused")
    if p:
        # evaluations in the traced run: parity + warm-up + timed + 2 instrumented; chunks per evaluation from the calls
        evals = 1 + 1 + 2 + 2
        chunks = max(1, round(p[0] / (L * evals)))
        batch = dict(k_edge_attn=L, k_edge_update=L - 1, k_node_update=L, k_bwd_gm_fused=L, k_bwd_gf_fused=L,
                     k_bwd_edge_update_T=L - 2, k_bwd_edge_update_S=L - 2, k_bwd_vecmsg_S=L - 1, k_bwd_attn_S=L,
                     k_bwd_norm_update=L)
        lines += ["", f"## Fragment batch (4096 fragments: N = {nb}, E = {eb}, {chunks} chunks per evaluation)", ""]
        t, meas, bound = table(tag, "frag_batch4096", nb, eb, "layer sweep and chunk", list(batch),
                               lambda k: dict(per=batch[k], chunks=chunks))
        lines += t
        lines += ["", "(`k_bwd_gm_fused` / `k_bwd_gf_fused` are MFMA products with a gather prologue: their bound is "
                      "the product's FLOPs, not these bytes - LAB_NOTES section 13 has their phase table.)"]
    open(os.path.join(ROOT, "profiles", f"{tag}_node_walks.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
