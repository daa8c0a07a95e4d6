#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-call3}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_split3.py "tests/test_gpu_parity.py::test_batch_path_other_head_counts" -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 3 "$OUT/pytest.log"
Q="--no-cpu-baseline --no-secondary --steps 1000 --warmup 20"
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["metric"][:28], round(d["value"], 2), d["unit"], "gemm us", round(d["roofline"]["avg_launch_us"], 2), d["roofline"]["kernel"], "dF", d["parity_max_dF"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for thr in 100000 64 32 1 100000 32; do
  VSN_S3_BM128=$thr VSN_OPTS=gemm_split3=1 timeout 300 python bench.py $Q > "$OUT/s3_bm$thr.json" 2> "$OUT/s3_bm$thr.err"
  show "$OUT/s3_bm$thr.json" "split3 BM128 thr=$thr"
done
QB="--no-cpu-baseline --no-secondary --workload frag_batch --frags-per-gpu 4096 --steps 6 --warmup 1"
timeout 300 python bench.py $QB > "$OUT/batch_f32.json" 2> "$OUT/batch_f32.err"; show "$OUT/batch_f32.json" "batch fp32"
VSN_OPTS=gemm_split3=1 timeout 300 python bench.py $QB > "$OUT/batch_s3.json" 2> "$OUT/batch_s3.err"; show "$OUT/batch_s3.json" "batch split3"
