#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-call5}
mkdir -p "$OUT"
timeout 600 python -m pytest "tests/test_gpu_gemm.py" -x -q > "$OUT/pytest.log" 2>&1; tail -n 3 "$OUT/pytest.log"
bash tools/gpu_ab.sh ${1:-call5}_ab "VSN_OPTS=gemm_breg=0" "VSN_OPTS=gemm_breg=1" "VSN_OPTS=gemm_breg=1 VSN_LIB=$R/ai2bmd_amd/_ab/libvsn_bregw5.so"
