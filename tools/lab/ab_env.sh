# steps/s of the default bench under a list of environment settings: bash tools/lab/ab_env.sh "A=1" "A=2 B=3" ...
for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline --no-secondary --steps 800 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$v -> $r"
done
