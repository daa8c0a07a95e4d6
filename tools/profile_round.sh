#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box and leaves small summaries under gpurun_out/<tag>/
# (copy the ones to be judged into profiles/).   usage:  bash tools/profile_round.sh <tag>
# Kernel traces and PMC passes are separate runs; PMC passes carry --kernel-trace only (no other trace domain).
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-round}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CH="python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0"
BA="python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0 --workload frag_batch --frags-per-gpu 4096"

timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt_chig" -o c -- $CH --steps 400 --warmup 10 > "$OUT/kt_chig.log" 2>&1
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt_batch" -o c -- $BA --steps 2 --warmup 1 > "$OUT/kt_batch.log" 2>&1
for W in chig batch; do
  DB=$(find "$OUT/kt_$W" -name "*.db" | head -1)
  python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/${W}_kernel_stats.csv"
  python "$R/tools/rocpd_stats.py" "$DB" --busy 0.3 0.6 > "$OUT/${W}_busy.txt"
  # one step of the TIMED region as a kernel sequence (the last five steps of a bench run are the instrumented pass)
  [ "$W" = chig ] && python "$R/tools/rocpd_stats.py" "$DB" --timeline k_md_half1_build -20 > "$OUT/chig_step_timeline.csv"
  rm -rf "$OUT/kt_$W"
done
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_chig_$i" -o b -- $CH --steps 20 --warmup 3 > "$OUT/pmc_chig_$i.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_batch_$i" -o b -- $BA --steps 1 --warmup 1 > "$OUT/pmc_batch_$i.log" 2>&1
done
for W in chig batch; do
  python "$R/tools/pmc_summary.py" "$OUT/pmc_${W}_1" "$OUT/pmc_${W}_2" "$OUT/pmc_${W}_3" > "$OUT/${W}_pmc.csv"
  rm -rf "$OUT"/pmc_${W}_?
done
# per-launch HBM bytes of the dominant GEMM kernels (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes; FETCH doubled on gfx950),
# read back by bench.py as roofline.traffic
python - "$OUT" "$R" <<'PY'
import csv, json, sys
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from ai2bmd_amd.build import _digest
res = {}
want = {
    "chig_md": ("chig", [("k_gemm_group", "k_gemm_group"), ("k_node_update<", "k_node_update"),
                         ("k_edge_attn", "k_edge_attn"), ("k_bwd_hf1", "k_bwd_hf1"), ("k_bwd_hf2", "k_bwd_hf2"),
                         ("k_bwd_attn_S", "k_bwd_attn_S"), ("k_bwd_norm_update", "k_bwd_norm_update")]),
    "frag_batch": ("batch", [("k_gemm<128; 128", "k_gemm<128,128>"), ("k_node_update<", "k_node_update"),
                             ("k_edge_attn<", "k_edge_attn"), ("k_bwd_gm_fused", "k_bwd_gm_fused"),
                             ("k_bwd_gf_fused", "k_bwd_gf_fused"), ("k_bwd_edge_update_T", "k_bwd_edge_update_T"),
                             ("k_bwd_edge_update_S", "k_bwd_edge_update_S"), ("k_bwd_norm_update", "k_bwd_norm_update"),
                             ("k_bwd_attn_S", "k_bwd_attn_S"), ("k_bwd_vecmsg_S", "k_bwd_vecmsg_S"),
                             ("k_edge_update<", "k_edge_update")]),
}
for wl, (tag, pats) in want.items():
    rows = list(csv.DictReader(open(f"{out}/{tag}_pmc.csv")))
    for pat, key in pats:
        for r in rows:
            if pat in r["kernel"] and r.get("hbm_MB") not in (None, "", "nan"):
                res.setdefault(wl, {})[key] = float(r["hbm_MB"]) * 1e6
                break
# avg / min duration of the same kernels in the kernel trace of the same command (bench.py lays its live
# dispatch-timestamp figures beside them: roofline.hbm.rocprof); all instantiations of a name pooled by calls
kn = {}
for wl, (tag, pats) in want.items():
    rows = list(csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")))
    for pat, key in pats:
        pat = pat.replace("; ", ", ").rstrip("<")
        hit = [r for r in rows if pat.split("<")[0] in r["kernel"] and (key != "k_node_update" or "k_node_update" in r["kernel"])]
        if key in ("k_gemm<128,128>",):
            hit = [r for r in rows if "k_gemmILi128ELi128E" in r["kernel"]]
        if hit:
            calls = sum(int(r["calls"]) for r in hit)
            kn.setdefault(wl, {})[key] = dict(calls=calls, avg_ns=sum(int(r["total_ns"]) for r in hit) / calls,
                                              min_ns=min(int(r["min_ns"]) for r in hit))
res["kernel_ns"] = kn
res["build_digest"] = _digest()
res["_note"] = ("HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB*1024) averaged over all launches of the kernel in "
                "the <tag>_pmc.csv of this directory (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, "
                "tools/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section); build_digest = "
                "ai2bmd_amd.build._digest() of the kernel sources measured: bench.py reports these figures only while "
                "the library it runs is that build")
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(res)
PY
tail -n 1 "$OUT/kt_chig.log" | cut -c1-300
ls -la "$OUT"
