#!/bin/bash
# Copies the summaries tools/gpu_final.sh <tag> left under gpurun_out/<tag>/ into profiles/ under their judged names.
# usage: bash tools/collect_profiles.sh r06
set -u
T=${1:?tag}; O=gpurun_out/$T; P=profiles
cp "$O/chig_kernel_stats.csv"   "$P/${T}_chig_md_kernel_stats.csv"
cp "$O/batch_kernel_stats.csv"  "$P/${T}_frag_batch4096_kernel_stats.csv"
cp "$O/chig_pmc.csv"            "$P/${T}_chig_md_pmc.csv"
cp "$O/batch_pmc.csv"           "$P/${T}_frag_batch4096_pmc.csv"
cp "$O/chig_step_timeline.csv"  "$P/${T}_chig_md_step_timeline.csv"
cat "$O/chig_busy.txt" "$O/batch_busy.txt" > "$P/${T}_busy.txt"
cp "$O/pmc_traffic.json"        "$P/${T}_pmc_traffic.json"
[ -f "$O/bench_full.json" ] && cp "$O/bench_full.json" "$P/${T}_bench_full.json"
[ -f "$O/bench_full_frag_batch.json" ] && cp "$O/bench_full_frag_batch.json" "$P/${T}_bench_full_frag_batch.json"
[ -f "$O/bench_line.json" ] && tail -n 1 "$O/bench_line.json" > "$P/${T}_bench_line.json"
[ -f "$O/node_walks.md" ] && cp "$O/node_walks.md" "$P/${T}_node_walks.md"
[ -f "$O/shard_table.md" ] && cp "$O/shard_table.md" "$P/${T}_shard_table.md"
ls -la $P/${T}_*
