#!/bin/bash
# lab: the aggressor (split3 GEMM only) built with MFMA in VGPR form (no AccVGPRs); victim = stock library, fp32 batch
python -c "import torch; torch.zeros(1).cuda()"
(VSN_LIB=$PWD/ai2bmd_amd/_ab/libvsn_novgprform.so GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- aggressor WITHOUT AccVGPRs"; timeout 300 python tools/lab/determinism_probe.py 10 fp32 2>&1 | grep pid | cut -c1-200
kill %1 2>/dev/null; wait 2>/dev/null
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- aggressor stock (AccVGPRs)"; timeout 300 python tools/lab/determinism_probe.py 10 fp32 2>&1 | grep pid | cut -c1-200
kill %1 2>/dev/null; wait 2>/dev/null
