# lab: the grouped products of a Chignolin step as 3 x bf16 split MFMA products (variant library, see build_s3.py)
export VSN_LIB=$PWD/ai2bmd_amd/_ab/libvsn_${1:-s3}.so
echo "library $VSN_LIB"
mkdir -p gpurun_out/r3s3
for m in 0 1 0 1; do
  VSN_SPLIT3=$m python bench.py --no-cpu-baseline --no-secondary --steps 600 --warmup 50 > gpurun_out/r3s3/bench_$m.json 2> gpurun_out/r3s3/bench_$m.err
  python - $m <<'P'
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3s3/bench_{m}.json").read().strip().splitlines()[-1])
    print(f"VSN_SPLIT3={m}: {d['value']:.1f} steps/s  {d['ms_per_step']:.4f} ms  parity {d.get('parity')}  gemm_us {d['roofline']['avg_launch_us']:.2f}")
except Exception as e:
    print("VSN_SPLIT3=%s failed: %r" % (m, e)); print(open(f"gpurun_out/r3s3/bench_{m}.err").read()[-1500:])
P
done
VSN_SPLIT3=1 timeout 300 python -m pytest tests/test_gpu_proteins.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r3s3/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r3s3/pytest.log | tail -2
