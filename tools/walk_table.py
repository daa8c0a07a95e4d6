"""The node walks against the HBM roofline: one table per workload, written to profiles/<tag>_node_walks.md.

    python tools/walk_table.py <tag>        (reads profiles/<tag>_*; run after tools/profile_round.sh on the same build)

Per kernel: launches per step / batch, average duration from the rocprofv3 KERNEL TRACE, ALGORITHMIC bytes per launch
(every distinct array the launch reads or writes, once: the formulas below = the ones csrc/engine.hip evaluates for
the live figures of bench.py), the counter bytes per launch (2 FETCH_SIZE + WRITE_SIZE, separate PMC passes), their
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


def alg_floats(kernel, n, e, **f):
    """distinct floats (4-byte indices counted as floats) one launch touches; n nodes, e edges of the launch"""
    if kernel == "k_edge_attn_update":
        return e * (2 * H + 2 + H) + n * (3 * H + H + 1) + e * (3 * H + 8) + n * S * 2 * H
    if kernel == "k_edge_attn":
        return e * (2 * H + 2 + H) + n * (3 * H + H + 1)
    if kernel == "k_edge_update":  # pe[f], f r/w, d, src | vp[wt|ws]
        return e * (3 * H + 8 + 1) + n * S * 2 * H
    if kernel == "k_node_update":
        return e * (2 * H + 8 + 1) + n * (S * H * 6 + 3 * H + 2 * H + 1) + n * (2 * H + 1 + S * H)
    if kernel == "k_bwd_hf1":  # with the edge update (7 of its 8 launches per step)
        return e * (2 * H + 8 + 3 + 2 * S + 2 * H) + n * (3 * S * H + 2) + e * (2 * H + 2 * S + H) + n * 3 * S * H
    if kernel == "k_bwd_hf2":
        return (e * (2 * H + 2 * H + 1 + 2 + 1 + H + 2 * H + 2 * NH) + n * (3 * H + 3 * H + H + 2)
                + e * (2 * H + 8) + n * 2 * S * H)
    if kernel == "k_bwd_attn_S":
        return e * (3 * H + 2 * NH + 2) + n * (3 * H + 1)
    if kernel == "k_bwd_norm_update":
        return n * (2 * H + 1 + 2 * H + S * H + 3 * S * H + H + S * H + 3 * S * H + 3 * H) + n * (H + S * H)
    if kernel == "k_bwd_gm_fused":  # tpre, d, g_geo r/w, src|tgt | vh, g_vec; write g_m
        return e * (2 * H + 8 + 2 * S + 2 + H) + n * 2 * S * H
    if kernel == "k_bwd_gf_fused":  # pe[dk|dv], g_m r/w, g_pe[f], C, g_geo r/w, ids, sat_tmp, g_f r/w | qkv, g_A
        return e * (2 * H + 2 * H + H + 1 + 2 + 2 + 2 * NH + 2 * H) + n * 4 * H
    if kernel == "k_bwd_edge_update_T":
        return e * (3 * H + 8 + 4 * S + 1) + n * 3 * S * H
    if kernel == "k_bwd_edge_update_S":
        return e * (2 * H + 10) + n * 2 * S * H
    if kernel == "k_bwd_vecmsg_S":
        return e * (H + 2) + n * 2 * S * H
    return None


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


def table(tag, wl, n, e, per, kernels, launches_of):
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
        fl = alg_floats(k, n / lp["chunks"], e / lp["chunks"])
        by = 4.0 * fl
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
    t, meas, bound = table(tag, "chig_md", n, e, "step", list(chig), lambda k: dict(per=chig[k], chunks=1))
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
