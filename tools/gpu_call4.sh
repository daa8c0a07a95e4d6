#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-call4}
mkdir -p "$OUT"
Q="--no-cpu-baseline --no-secondary --steps 1000 --warmup 20"
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["metric"][:28], round(d["value"], 2), d["unit"], "gemm us", round(d["roofline"]["avg_launch_us"], 2), "dF", d["parity_max_dF"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2; do
  VSN_OPTS=gemm_split3=1 timeout 300 python bench.py $Q > "$OUT/s3_w4_$i.json" 2> "$OUT/s3_w4_$i.err"; show "$OUT/s3_w4_$i.json" "split3 minwaves=4"
  VSN_LIB=$R/ai2bmd_amd/_ab/libvsn_s3w5.so VSN_OPTS=gemm_split3=1 timeout 300 python bench.py $Q > "$OUT/s3_w5_$i.json" 2> "$OUT/s3_w5_$i.err"; show "$OUT/s3_w5_$i.json" "split3 minwaves=5"
done
timeout 300 python bench.py $Q > "$OUT/f32.json" 2> "$OUT/f32.err"; show "$OUT/f32.json" "fp32 default"
