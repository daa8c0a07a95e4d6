"""Lab: which kernels does the fp32 GEMM library (torch.mm -> hipBLASLt / rocBLAS) run on the ViSNet product shapes, and
how fast?  Run under `rocprofv3 --kernel-trace` (+ tools/rocpd_stats.py): the Tensile kernel names carry the macro tile,
the MFMA instruction and the pipelining options."""
import time

import torch

dev = "cuda:0"
torch.manual_seed(0)
for M, Nc, K in [(1000000, 768, 256), (1000000, 512, 256), (490000, 1280, 256), (1000000, 256, 512), (1000000, 256, 768),
                 (490000, 256, 1280), (6687, 768, 256), (6687, 512, 256), (3128, 1280, 256), (6687, 256, 512),
                 (6687, 256, 768), (3128, 256, 1280)]:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(Nc, K, device=dev) / K ** 0.5
    C = A @ W.t()
    torch.cuda.synchronize()
    reps = 5 if M > 100000 else 50
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.mm(A, W.t(), out=C)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{M:8d} {Nc:5d} {K:5d} | library fp32 GEMM {us:9.1f} us {2.0 * M * Nc * K / us * 1e-6:7.1f} TFLOP/s", flush=True)
    del A, W, C
    torch.cuda.empty_cache()
