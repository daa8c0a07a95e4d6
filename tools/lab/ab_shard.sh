# steps/s at N = 1 and on rank 0 of an 8-rank job (one GPU; tools/lab, not part of the product)
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-secondary --steps 600 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N1', d['value'])"
  for sh in 0/8 0/4 0/2; do
  python bench.py --no-cpu-baseline --no-secondary --emulate-shard $sh --steps 600 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard $sh', d['value'], d['ms_per_step'])"
  done
done
