#!/bin/bash
# lab: library kernels (torch: matmul, sigmoid / exp / reciprocal / sin chain) and OUR fp32 batch probe as victims next to
# a neighbour that runs nothing but the split3 GEMM (k_gemm3_128) back to back
python -c "import torch; torch.zeros(1).cuda()"
(GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 60000 2>&1 | grep pid | cut -c1-120) &
sleep 10
timeout 300 python tools/lab/torch_victim.py 300 2>&1 | grep victim
timeout 300 python tools/lab/determinism_probe.py 12 fp32 2>&1 | grep pid | cut -c1-260
kill %1 2>/dev/null; wait 2>/dev/null
