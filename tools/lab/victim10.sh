#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- HIP_FORCE_DEV_KERNARG=0 (kernel arguments in host memory)"; HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/lab/tap_probe.py 8 400 2>&1 | grep -v amdgpu | cut -c1-130
echo "--- default"; timeout 300 python tools/lab/tap_probe.py 6 400 2>&1 | grep -v amdgpu | cut -c1-130
kill %1 2>/dev/null; wait 2>/dev/null
