#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- stock library"; timeout 300 python tools/lab/tap_probe.py 8 400 2>&1 | grep -v amdgpu | cut -c1-160
echo "--- C through a vector load in k_edge_attn"; VSN_LIB=$PWD/ai2bmd_amd/_ab/libvsn_vecC.so timeout 300 python tools/lab/tap_probe.py 8 400 2>&1 | grep -v amdgpu | cut -c1-160
kill %1 2>/dev/null; wait 2>/dev/null
