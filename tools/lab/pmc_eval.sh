#!/bin/bash
# lab: PMC passes (one counter set per pass, --kernel-trace only) of tools/lab/time_eval.py, merged per kernel
#   bash tools/lab/pmc_eval.sh <outdir> "<counters of pass 1>" "<counters of pass 2>" ...
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-pmc}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
dirs=""
for C in "$@"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/p$i" -o b -- python "$R/tools/lab/time_eval.py" ${PROT:-chig} 20 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "b_counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$(dirname "$f")" != "$OUT/p$i" ] && mv "$f" "$OUT/p$i/b_counter_collection.csv"
  dirs="$dirs $OUT/p$i"
done
python "$R/tools/pmc_summary.py" $dirs > "$OUT/pmc.csv"
for d in $dirs; do rm -rf "$d"; done
head -c 3000 "$OUT/pmc.csv"
