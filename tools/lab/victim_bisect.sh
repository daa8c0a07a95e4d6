#!/bin/bash
# lab: a long-running split3 neighbour (B); the fp32 probe (A) under different options next to it
python -c "import torch; torch.zeros(1).cuda()"
(timeout 900 python tools/lab/determinism_probe.py 1200 split3 2>&1 | grep pid | cut -c1-120) &
sleep 14
for o in "" "fuse_panel=0" "fuse_bwd=0" "fuse_fwd=0" "fuse_head=0" "overlap=0" "split_rev=0"; do
  VSN_OPTS=$o timeout 300 python tools/lab/determinism_probe.py 10 fp32 2>&1 | grep pid | cut -c1-140
done
kill %1 2>/dev/null; wait 2>/dev/null
