#!/bin/bash
# lab: two concurrent probe processes in split3 mode, one round per option set
for o in "$@"; do
  echo "=== VSN_OPTS=$o"
  (VSN_OPTS=$o timeout 400 python tools/lab/determinism_probe.py 5 split3 2>&1 | grep "pid" | cut -c1-150) &
  (VSN_OPTS=$o timeout 400 python tools/lab/determinism_probe.py 5 split3 2>&1 | grep "pid" | cut -c1-150)
  wait
done
