#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
echo "--- alone"; timeout 120 tools/lab/stale_micro 1000 256; timeout 120 tools/lab/stale_micro 3000 16
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- next to a split3-GEMM-only neighbour"; timeout 300 tools/lab/stale_micro 3000 256; timeout 300 tools/lab/stale_micro 10000 16; timeout 300 tools/lab/stale_micro 3000 1024
kill %1 2>/dev/null; wait 2>/dev/null
