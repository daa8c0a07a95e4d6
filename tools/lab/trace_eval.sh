#!/bin/bash
# lab: per-kernel average durations of tools/lab/time_eval.py for a list of builds (kernel trace, no counters)
#   bash tools/lab/trace_eval.sh <outdir> "<name>|<env assignments>" ...
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-trace}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}
  envs=${spec#*|}
  ( export $envs; timeout 300 rocprofv3 --kernel-trace -d "$OUT/kt_$name" -o t -- python "$R/tools/lab/time_eval.py" ${PROT:-chig} 200 ) > "$OUT/$name.log" 2>&1
  DB=$(find "$OUT/kt_$name" -name "*.db" | head -1)
  python "$R/tools/rocpd_stats.py" "$DB" | cut -c1-60,60- | awk -F, 'NR==1 || $2 >= 200 {print}' | sed -E 's/"_ZN3vsn[0-9]+//; s/(I[A-Za-z0-9]*E)?Ev.*",/,/' > "$OUT/${name}_stats.csv"
  rm -rf "$OUT/kt_$name"
  grep "ms per evaluation" "$OUT/$name.log"
done
