#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-call2}
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 4 "$OUT/pytest.log"
Q="--no-cpu-baseline --no-secondary --steps 1000 --warmup 20"
for ss in 0 1 0 1; do
  VSN_NU_SS=$ss timeout 300 python bench.py $Q > "$OUT/ab_ss${ss}.json" 2> "$OUT/ab_ss${ss}.err"
  python - "$OUT/ab_ss${ss}.json" "ss=$ss" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["roofline"]["hbm"]["all_scatter_kernels"]
print(sys.argv[2], "steps/s", round(d["value"], 2), "node_update us", round(h["k_node_update"]["avg_launch_us"], 2), "edge_attn us", round(h["k_edge_attn"]["avg_launch_us"], 2), "gemm_group us", round(d["roofline"]["avg_launch_us"], 2), "dF", d["parity_max_dF"])
PY
done
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
python - "$OUT/bench_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "steps", d["steps"])
for k, v in d["config"].get("secondary_summary", {}).items():
    print("  ", k, round(v["value"], 1), v["unit"], v["steps"])
s3 = [r for r in d.get("secondary", []) if "split3" in r["metric"]]
if s3:
    print("split3 parity", s3[0].get("parity"), "gemm us", s3[0]["roofline"]["avg_launch_us"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt_chig" -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > "$OUT/kt_chig.log" 2>&1
DB=$(find "$OUT/kt_chig" -name "*.db" | head -1)
python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/chig_kernel_stats.csv"
rm -rf "$OUT/kt_chig"
python - "$OUT/chig_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["kernel"][4:42], r["calls"], r["avg_ns"], r["min_ns"], r["percent"])
PY
