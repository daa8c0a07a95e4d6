#!/bin/bash
# End-of-round evidence on one box: the -m gpu suite, the round's profiles (kernel traces + PMC passes + traffic json),
# then the default bench line with the traffic json of THIS build in place.   usage: bash tools/gpu_final.sh r04
set -u
R=$PWD
TAG=${1:-round}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 3 "$OUT/pytest.log"
bash tools/profile_round.sh "$TAG" > "$OUT/profile_round.log" 2>&1
cd "$R"
cp "$OUT/pmc_traffic.json" "profiles/${TAG}_pmc_traffic.json"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cp gpurun_out/bench_full_chig_md_n1.json "$OUT/bench_full.json" 2>/dev/null
# the fragment batch alone (its own full record: atoms / edges per GPU for tools/walk_table.py)
timeout 600 python bench.py --workload frag_batch --steps 6 --warmup 1 --no-cpu-baseline > "$OUT/bench_line_frag_batch.json" 2> "$OUT/bench_frag_batch.err"
cp gpurun_out/bench_full_frag_batch_n1.json "$OUT/bench_full_frag_batch.json" 2>/dev/null
python - "$OUT/bench_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "steps", d["steps"], "c2 loop", d["config"].get("c2_loop", {}).get("value"))
r = d["roofline"]; print("gemm", r["kernel"], round(r["frac"], 4), round(r["avg_launch_us"], 2), "traffic", r["traffic"])
h = r["hbm"]; print("hbm", h["kernel"], round(h["frac"], 4), round(h["avg_launch_us"], 2), "traffic", h["traffic"], h.get("rocprof"))
c = d["cpu_baseline"]; print("cpu", {k: c[k] for k in ("value", "cores", "kind", "physical_cores", "layout_of_value", "reference_layout_evals_per_s")})
for k, v in d["config"].get("secondary_summary", {}).items():
    print("  ", k, round(v["value"], 1), v["unit"], v["steps"])
PY
# the round's files under their profiles/ names (here, on the box, for walk_table; gpurun merges only gpurun_out/ back:
# tools/collect_profiles.sh <tag> does the same copy in the build container)
bash tools/collect_profiles.sh "$TAG" > /dev/null 2>&1 || true
python tools/walk_table.py "$TAG" > "$OUT/walk_table.log" 2>&1 || true
cp "profiles/${TAG}_node_walks.md" "$OUT/node_walks.md" 2>/dev/null
XUS=$(python - "$OUT/bench_full.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{d['config']['rccl1_allgather']['chig']['delta_us']:.1f}")
except Exception:
    print("9.8")
PY
)
timeout 900 python tools/shard_table.py --exchange-us "$XUS" > "$OUT/shard_table.md" 2> "$OUT/shard_table.err" || true
ls "$OUT"
