#!/bin/bash
# lab: fp32 victim A next to an fp32 neighbour B that evaluates OTHER positions (same program, same allocation pattern)
python -c "import torch; torch.zeros(1).cuda()"
(PROBE_SEED=99 timeout 900 python tools/lab/determinism_probe.py 1500 fp32 2>&1 | grep pid | cut -c1-120) &
sleep 14
timeout 300 python tools/lab/determinism_probe.py 20 fp32 2>&1 | grep pid | cut -c1-300
kill %1 2>/dev/null; wait 2>/dev/null
