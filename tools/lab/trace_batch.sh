#!/bin/bash
# lab: kernel-trace one fragment-batch bench run, print the node-walk kernels' average durations
# usage: bash tools/lab/trace_batch.sh <tag> [ENV=VAL ...]
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0 --workload frag_batch --frags-per-gpu 4096 --steps 2 --warmup 1 > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/batch_kernel_stats.csv
rm -rf $OUT/kt
echo "== $TAG $@"; tail -1 $OUT/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fragments/s', d['value'])"
grep -E "k_edge_updateI|k_bwd_edge_update_S|k_bwd_vecmsg_S|k_bwd_edge_update_T|k_edge_attnI|k_node_updateI|k_bwd_attn_S|k_bwd_norm_update" $OUT/batch_kernel_stats.csv | awk -F, '{printf "%-40s calls %5d avg %8.1f us  %5.2f%%\n", substr($1,1,40), $2, $4/1000, $7}'
