#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
(timeout 900 python tools/lab/determinism_probe.py 1500 split3 2>&1 | grep pid | cut -c1-120) &
sleep 14
timeout 600 python tools/lab/tap_probe.py 12 400 2>&1 | grep -v amdgpu.ids | cut -c1-300
TAP_DEBUG=1 timeout 600 python tools/lab/tap_probe.py 12 400 2>&1 | grep -v amdgpu.ids | cut -c1-300
kill %1 2>/dev/null; wait 2>/dev/null
