#!/bin/bash
# LAB: what the two phases of the fused panel products (gather prologue / MFMA slices) cost alone, and the products
# alone (no side stream).  usage: bash tools/lab/fused_phases.sh <tag>      (wrong numbers on purpose in ABL runs)
# Needs the lab build first (here, where hipcc is):  python tools/build_variant.py fusedabl fused.hip -DVSN_LAB_ABL=1
# - the product library has no ablation switch; the script picks the variant up through VSN_LIB.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-fused_phases}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VSN_LIB=$R/ai2bmd_amd/_ab/libvsn_fusedabl.so
[ -f "$VSN_LIB" ] || { echo "build the lab variant first (see header)"; exit 1; }
BA="python $R/bench.py --no-cpu-baseline --no-secondary --workload frag_batch --frags-per-gpu 4096 --steps 2 --warmup 1"
for cfg in "VSN_OPTS=overlap=0" "VSN_OPTS=overlap=0 VSN_LAB_FUSED_ABL=1 VSN_LAB_NO_PARITY=1" "VSN_OPTS=overlap=0 VSN_LAB_FUSED_ABL=2 VSN_LAB_NO_PARITY=1" "VSN_OPTS=overlap=2"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( export $cfg; timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt_$tag" -o c -- $BA > "$OUT/$tag.log" 2>&1 )
  DB=$(find "$OUT/kt_$tag" -name "*.db" | head -1)
  echo "== $cfg"
  python "$R/tools/rocpd_stats.py" "$DB" | grep -E "kernel,|fused|k_gemmILi128" | cut -c1-200
  tail -n 1 "$OUT/$tag.log" | cut -c1-160
  rm -rf "$OUT/kt_$tag"
done
