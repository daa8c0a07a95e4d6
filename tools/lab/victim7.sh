#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
echo "--- alone"; timeout 120 tools/lab/victim_micro 60
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- next to a split3-GEMM-only neighbour"; timeout 300 tools/lab/victim_micro 400
kill %1 2>/dev/null; wait 2>/dev/null
