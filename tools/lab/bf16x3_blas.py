"""Lab: what would a TUNED bf16 GEMM make of the six-product form of tools/lab/bf16x3_lab.hip?  The six products are
one bf16 GEMM with the split planes concatenated along K (K' = 6 K):  C = [Ah Ah Am Ah Al Am] . [Wh Wm Wh Wl Wh Wm]^T.
Times torch.mm (hipBLASLt / rocBLAS) on that shape - an upper estimate for a hand-written kernel's MFMA side (the
library reads the 6 K-wide operand from HBM, a fused kernel would split in LDS) - and reports fp32-equivalent TFLOP/s
(2 M Nc K / t) next to the error against an fp64 product."""
import time

import torch

torch.manual_seed(0)
dev = "cuda:0"


def split3(x):
    h = x.to(torch.bfloat16)
    r1 = x - h.float()
    m = r1.to(torch.bfloat16)
    lo = (r1 - m.float()).to(torch.bfloat16)
    return h, m, lo


for M, Nc, K in [(1000000, 768, 256), (1000000, 256, 768), (490000, 1280, 256), (6687, 768, 256), (6687, 256, 768)]:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(Nc, K, device=dev) / K ** 0.5
    ah, am, al = split3(A)
    wh, wm, wl = split3(W)
    A6 = torch.cat([ah, ah, am, ah, al, am], 1).contiguous()
    W6 = torch.cat([wh, wm, wh, wl, wh, wm], 1).contiguous()
    del ah, am, al
    try:
        f = lambda: torch.mm(A6, W6.t(), out_dtype=torch.float32)  # noqa: E731
        C = f()
        kind = "bf16 x bf16 -> f32"
    except Exception as e:  # noqa: BLE001
        f = lambda: torch.mm(A6, W6.t())  # noqa: E731
        C = f()
        kind = f"bf16 out (timing only; out_dtype unsupported: {type(e).__name__})"
    ref = (A[:256].double() @ W.double().t())
    err = (C[:256].double() - ref)
    f32 = (A[:256] @ W.t()).double() - ref
    torch.cuda.synchronize()
    reps = 5 if M > 100000 else 50
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        A @ W.t()
    torch.cuda.synchronize()
    us32 = (time.perf_counter() - t0) / reps * 1e6
    fl = 2.0 * M * Nc * K
    print(f"{M:8d} {Nc:5d} {K:5d} | library {kind}: {us:9.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s fp32-equivalent "
          f"({6 * fl / us * 1e-9:5.2f} PFLOP/s bf16)  rms err {err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.2e} | "
          f"library fp32 GEMM {us32:9.1f} us {fl / us32 * 1e-6:7.1f} TFLOP/s  rms err "
          f"{f32.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.2e}", flush=True)
    del A, W, A6, W6, C
    torch.cuda.empty_cache()
