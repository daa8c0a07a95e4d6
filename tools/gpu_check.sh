#!/bin/bash
# One GPU-box visit: the -m gpu suite, the default bench line, and a kernel trace of the same command so that the
# live dispatch-timestamp figures of roofline.hbm can be laid beside rocprofv3's.   usage: bash tools/gpu_check.sh <tag>
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-check}
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 4 "$OUT/pytest.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
python - "$OUT/bench_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "steps", d["steps"], d["config"].get("requested_run"))
r = d["roofline"]; print("gemm", r["kernel"], r["frac"], r["avg_launch_us"])
h = r.get("hbm"); print("hbm", {k: (v["avg_launch_us"], v["frac"]) for k, v in h["all_scatter_kernels"].items()} if h else None)
c = d.get("cpu_baseline"); print("cpu", {k: c[k] for k in ("value", "cores", "kind", "physical_cores", "layout_of_value", "reference_layout_evals_per_s")} if c else None)
print("secondary", d["config"].get("secondary_summary"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt_chig" -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > "$OUT/kt_chig.log" 2>&1
DB=$(find "$OUT/kt_chig" -name "*.db" | head -1)
python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/chig_kernel_stats.csv"
rm -rf "$OUT/kt_chig"
head -n 12 "$OUT/chig_kernel_stats.csv" | cut -c1-60,140-
tail -n 1 "$OUT/kt_chig.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
h=d['roofline']['hbm']; print('under rocprof: live', {k:v['avg_launch_us'] for k,v in h['all_scatter_kernels'].items()}, 'value', d['value'])"
