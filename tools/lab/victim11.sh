#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
for dt in bf16 fp16; do
  (timeout 200 python tools/lab/burner_dtype.py 60 $dt 2>&1 | grep burner) &
  sleep 12
  echo "--- victim fp32 batch next to library $dt matmuls"; timeout 300 python tools/lab/determinism_probe.py 10 fp32 2>&1 | grep pid | cut -c1-200
  wait
done
