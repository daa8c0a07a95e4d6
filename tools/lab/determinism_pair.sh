#!/bin/bash
# lab: two concurrent processes of the determinism probe on one GPU, for each option set given (fp32 mode only)
for o in "$@"; do
  echo "=== VSN_OPTS=$o"
  (VSN_OPTS=$o timeout 400 python tools/lab/determinism_probe.py 6 fp32 2>&1 | grep "pid") &
  (VSN_OPTS=$o timeout 400 python tools/lab/determinism_probe.py 6 fp32 2>&1 | grep "pid")
  wait
done
