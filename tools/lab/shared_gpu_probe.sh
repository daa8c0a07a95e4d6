#!/bin/bash
# lab: run VICTIM commands on a GPU that a second PROCESS (the aggressor) keeps busy - the harness behind LAB_NOTES
# section 15 (is a batch evaluation bit-reproducible next to a neighbour process, and which kernels disturb which?).
#
#   tools/lab/shared_gpu_probe.sh AGGRESSOR 'victim command' ['victim command' ...]
#
# AGGRESSOR  none          victims run alone
#            gemm3         nothing but the split-3 GEMM k_gemm3_128, back to back        (gemm_determinism.py, GD_ONLY)
#            gemm3-zero    the same on all-zero operands                                  (GD_ZERO)
#            batch-split3  whole batch evaluations with gemm_split3 on                    (determinism_probe.py split3)
#            batch-fp32    whole fp32 batch evaluations of OTHER positions (PROBE_SEED=99)
#            lib-bf16 | lib-fp16 | lib-fp32   library matmuls of that type                       (burner_dtype.py)
#            self          every victim command runs twice, concurrently (two equal processes)
# AGGRESSOR_LIB=path   a variant library for the aggressor only (tools/build_variant.py), e.g. MFMA without AccVGPRs
#
# victims used in section 15:   'python tools/lab/determinism_probe.py 10 fp32'     whole evaluation, checksum per run
#                               'python tools/lab/tap_probe.py 8 400'               first differing buffer
#                               'python tools/lab/torch_victim.py 300'              library kernels
#                               'python tools/lab/tap_pattern.py 5 400'             WHAT differs in that buffer
#                               'tools/lab/pk_micro 300'                            packed-fp32 instructions alone
#                               'tools/lab/victim_micro 400', 'tools/lab/stale_micro 3000 256'   other micro-kernels
#   environment of a victim goes in its command: 'AMD_SERIALIZE_KERNEL=3 python tools/lab/determinism_probe.py 8 fp32',
#   'VSN_OPTS=fuse_panel=0 python ...', 'HIP_FORCE_DEV_KERNARG=0 python ...', 'VSN_LIB=$PWD/ai2bmd_amd/_ab/x.so python ...'
set -u
agg=${1:?aggressor}; shift
python -c "import torch; torch.zeros(1).cuda()"   # page the image in before anything is timed against a sleep
filt() { grep -v amdgpu.ids | cut -c1-${PROBE_COLS:-400}; }
AGG_LOG=$(mktemp); AGG=
agg_start() { env "$@" > "$AGG_LOG" 2>&1 & AGG=$!; }   # (the PID is `timeout`'s: a TERM to it ends the aggressor itself)
case $agg in
  none|self)    ;;
  gemm3)        agg_start VSN_LIB=${AGGRESSOR_LIB:-} GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 ;;
  gemm3-zero)   agg_start VSN_LIB=${AGGRESSOR_LIB:-} GD_ZERO=1 GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 ;;
  batch-split3) agg_start timeout 900 python tools/lab/determinism_probe.py 1500 split3 ;;
  batch-fp32)   agg_start PROBE_SEED=99 timeout 900 python tools/lab/determinism_probe.py 1500 fp32 ;;
  lib-bf16|lib-fp16|lib-fp32) agg_start timeout 900 python tools/lab/burner_dtype.py 600 ${agg#lib-} ;;
  *) echo "unknown aggressor $agg" >&2; exit 2 ;;
esac
[ "$agg" = none ] || [ "$agg" = self ] || sleep 14
for v in "$@"; do
  echo "--- [$agg] $v"
  if [ "$agg" = self ]; then (timeout 500 bash -c "$v" 2>&1 | filt) & fi
  timeout 500 bash -c "$v" 2>&1 | filt
  [ "$agg" = self ] && wait
done
if [ -n "$AGG" ]; then kill $AGG 2>/dev/null; wait $AGG 2>/dev/null; echo "--- aggressor:"; filt < "$AGG_LOG"; fi
rm -f "$AGG_LOG"
