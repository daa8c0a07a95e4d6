R=$PWD; OUT=$R/gpurun_out/r03; mkdir -p $OUT
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/kt_chig -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0 --steps 400 --warmup 10 > $OUT/kt_chig.log 2>&1
  DB=$(find $OUT/kt_chig -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB --timeline k_md_half1_build -20 > $OUT/chig_step_timeline.csv; rm -rf $OUT/kt_chig )
timeout 500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_stdout.log 2> $OUT/bench_stderr.log; echo "bench rc=$?"
tail -1 $OUT/bench_stdout.log > $OUT/bench_line.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('launches_per_step'))
for k,v in d.get('secondary',{}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ('value','unit','steps','max_dF','pipeline_max_dF')} if isinstance(v, dict) else v)
print(d['cpu_baseline'])
P
tail -3 $OUT/chig_step_timeline.csv | cut -c1-80
