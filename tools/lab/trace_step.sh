#!/bin/bash
# lab: kernel-trace one bench run of the Chignolin step and print the once-per-step head / tail of the step timeline
# usage: bash tools/lab/trace_step.sh <tag> [ENV=VAL ...]
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0 --steps 300 --warmup 10 > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/chig_kernel_stats.csv
python $R/tools/rocpd_stats.py $DB --timeline k_md_half1_build -20 > $OUT/chig_step_timeline.csv
rm -rf $OUT/kt
echo "== $TAG $@"; head -8 $OUT/chig_step_timeline.csv | cut -c1-60; tail -1 $OUT/chig_step_timeline.csv
