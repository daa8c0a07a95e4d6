#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
(GD_ZERO=1 GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- aggressor multiplies ZERO matrices"; timeout 300 python tools/lab/determinism_probe.py 10 fp32 2>&1 | grep pid | cut -c1-200
kill %1 2>/dev/null; wait 2>/dev/null
