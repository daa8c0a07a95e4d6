#!/bin/bash
python -c "import torch; torch.zeros(1).cuda()"
for v in s3pad s3pad2; do
(VSN_LIB=$PWD/ai2bmd_amd/_ab/libvsn_$v.so GD_ONLY=1 timeout 900 python tools/lab/gemm_determinism.py 90000 2>&1 | grep pid | cut -c1-120) &
sleep 10
echo "--- aggressor $v (64 / 84 KB of LDS per workgroup instead of 48)"; timeout 300 python tools/lab/determinism_probe.py 8 fp32 2>&1 | grep pid | cut -c1-160
kill %1 2>/dev/null; wait 2>/dev/null
done
