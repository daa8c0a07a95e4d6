#!/bin/bash
# LAB: kernel trace of ONE rank's share of an N-rank Chignolin job (emulated on one GPU): what a small shard's step is
# made of.   usage: bash tools/lab/shard_trace.sh <tag> <r/w>
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-shard_trace}
SH=${2:-0/8}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt" -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --emulate-shard $SH --steps 400 --warmup 10 > "$OUT/run.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/kernel_stats.csv"
python "$R/tools/rocpd_stats.py" "$DB" --busy 0.3 0.6 > "$OUT/busy.txt"
python "$R/tools/rocpd_stats.py" "$DB" --timeline k_md_half1_build -20 > "$OUT/step_timeline.csv"
cat "$OUT/busy.txt"; head -25 "$OUT/kernel_stats.csv" | cut -c1-150; tail -n 1 "$OUT/run.log" | cut -c1-200
rm -rf "$OUT/kt"
