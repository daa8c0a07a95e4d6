#!/bin/bash
# lab: two concurrent processes of the determinism probe (given modes) on one GPU, N times
N=${1:-3}; shift
for i in $(seq $N); do
  echo "=== round $i: $@"
  (timeout 500 python tools/lab/determinism_probe.py 8 "$@" 2>&1 | grep "pid" | cut -c1-200) &
  (timeout 500 python tools/lab/determinism_probe.py 8 "$@" 2>&1 | grep "pid" | cut -c1-200)
  wait
done
