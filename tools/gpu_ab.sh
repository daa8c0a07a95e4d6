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
    r = d["roofline"]
    w = r.get("reverse_walks") or {}
    us = lambda k: (w.get(k) or {}).get("us", 0)
    print("%-44s %8.2f %s | gemm %6.2f us | %s %5.2f | hf1 %5.2f hf2 %5.2f attn_S %5.2f norm_upd %5.2f | dF %.3e" % (
        sys.argv[2], d["value"], d["unit"], r["avg_launch_us"], r.get("hbm", {}).get("kernel", "?")[5:],
        r.get("hbm", {}).get("avg_launch_us", 0), us("k_bwd_hf1"), us("k_bwd_hf2"), us("k_bwd_attn_S"),
        us("k_bwd_norm_update"), d["parity"]["max_dF"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
