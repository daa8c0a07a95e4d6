#!/bin/bash
# generic cheap A/B on the Chignolin loop:  bash tools/gpu_ab.sh <tag> "<ENV...>" "<ENV...>" ...   (each config run twice)
set -u
R=$PWD
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
Q="${BENCH_Q:---no-cpu-baseline --no-secondary --steps 1000 --warmup 20}"
i=0
for rep in 1 2; do
  for cfg in "$@"; do
    i=$((i + 1))
    ( export $cfg; timeout 300 python bench.py $Q ) > "$OUT/run$i.json" 2> "$OUT/run$i.err"
    python - "$OUT/run$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d["roofline"].get("hbm", {}).get("all_scatter_kernels", {})
    print("%-60s %8.2f %s | gemm %6.2f us | node_upd %5.2f | dF %.3e" % (sys.argv[2], d["value"], d["unit"], d["roofline"]["avg_launch_us"], h.get("k_node_update", {}).get("avg_launch_us", 0), d["parity_max_dF"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
