#!/bin/bash
# LAB: every reverse node walk of a Chignolin step as its OWN kernel on one stream (VSN_FUSE_SIDE=0, overlap off):
# what each part of k_bwd_hf1 / k_bwd_hf2 costs standing alone.   usage: bash tools/lab/parts_alone.sh <tag>
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-parts_alone}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CH="python $R/bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 10"
( export VSN_FUSE_SIDE=0 VSN_OPTS=overlap=0; timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt" -o c -- $CH > "$OUT/run.log" 2>&1 )
DB=$(find "$OUT/kt" -name "*.db" | head -1)
python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/parts_alone_kernel_stats.csv"
grep -E "kernel,|k_bwd|k_node_update|k_edge" "$OUT/parts_alone_kernel_stats.csv" | cut -c1-220
tail -n 1 "$OUT/run.log" | cut -c1-200
rm -rf "$OUT/kt"
