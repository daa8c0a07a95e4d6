#!/bin/bash
# A/B sweep on the GPU box: one short Chignolin bench (parity guards included) per variant.
#   bash tools/ab_sweep.sh <outdir> "<name>|<env assignments>" ...
# prints "<name> <steps/s> <k_gemm_group us> <parity max|dF|>" per variant into <outdir>/ab.txt
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-ab}
shift
mkdir -p "$OUT"
: > "$OUT/ab.txt"
for spec in "$@"; do
  name=${spec%%|*}
  envs=${spec#*|}
  log="$OUT/$name.log"
  ( export $envs; timeout 300 python "$R/bench.py" ${BENCH_ARGS:---no-cpu-baseline --no-secondary --steps 400 --warmup 20 --min-seconds 1.5} ) > "$log" 2>&1
  python - "$name" "$log" >> "$OUT/ab.txt" <<'PY'
import json, sys
name, log = sys.argv[1:3]
line = [l for l in open(log) if l.startswith("{")]
if not line:
    print(name, "FAILED", open(log).read()[-400:].replace("\n", " | "))
else:
    r = json.loads(line[-1])
    print(name, f"{r['value']:.1f}", f"{r['ms_per_step']:.4f}", f"gemm_us={r['roofline']['avg_launch_us']:.2f}",
          f"frac={r['roofline']['frac']:.3f}", f"dF={r['parity_max_dF']:.2e}", f"pipe_dF={r['parity'].get('pipeline_max_dF', float('nan')):.2e}")
PY
done
cat "$OUT/ab.txt"
