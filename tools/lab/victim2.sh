#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
(timeout 900 python tools/lab/determinism_probe.py 1500 split3 2>&1 | grep pid | cut -c1-120) &
sleep 14
timeout 300 python tools/lab/torch_victim.py 300 2>&1 | grep victim
timeout 300 python tools/lab/gemm_determinism.py 200 2>&1 | grep pid
kill %1 2>/dev/null; wait 2>/dev/null
