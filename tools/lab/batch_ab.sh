#!/bin/bash
# cheap A/B on the 4096-fragment batch:  bash tools/lab/batch_ab.sh <tag> "<ENV...>" "<ENV...>" ...   (each config twice)
set -u
R=$PWD
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
Q="${BENCH_Q:---no-cpu-baseline --no-secondary --workload frag_batch --frags-per-gpu 4096 --steps 6 --warmup 1}"
i=0
for rep in 1 2; do
  for cfg in "$@"; do
    i=$((i + 1))
    ( export $cfg; timeout 300 python bench.py $Q ) > "$OUT/run$i.json" 2> "$OUT/run$i.err"
    python - "$OUT/run$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-50s %9.1f %s | %s %7.1f us frac %.3f | hbm %s frac %.3f | dF %.3e" % (
        sys.argv[2], d["value"], d["unit"], r["kernel"][5:], r["avg_launch_us"], r["frac"],
        r.get("hbm", {}).get("kernel", "?")[5:], r.get("hbm", {}).get("frac", 0), d["parity"]["max_dF"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
  done
done
