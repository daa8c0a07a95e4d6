#!/bin/bash
# lab: the fp32 batch probe next to a split3-GEMM-only neighbour, with the runtime serialising kernels (visibility test)
python -c "import torch; torch.zeros(1).cuda()"
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- default"; timeout 300 python tools/lab/determinism_probe.py 8 fp32 2>&1 | grep pid | cut -c1-160
echo "--- AMD_SERIALIZE_KERNEL=3"; AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/lab/determinism_probe.py 8 fp32 2>&1 | grep pid | cut -c1-160
echo "--- HIP_LAUNCH_BLOCKING=1"; HIP_LAUNCH_BLOCKING=1 timeout 300 python tools/lab/determinism_probe.py 8 fp32 2>&1 | grep pid | cut -c1-160
echo "--- VSN_XCD_REMAP off n/a; victim with 400 fragments (default)"; timeout 300 python tools/lab/tap_probe.py 4 400 2>&1 | grep -v amdgpu | cut -c1-200
kill %1 2>/dev/null; wait 2>/dev/null
