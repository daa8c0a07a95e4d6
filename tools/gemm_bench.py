#!/usr/bin/env python
"""Micro-benchmark of the fp32 MFMA GEMM tap (vsn_gemm) on the shapes the ViSNet path issues.
    python tools/gemm_bench.py            (on the GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402

SHAPES = [
    # (M, Nc, K, tag)
    (170000, 768, 256, "batch edge dk|dv|f"), (170000, 512, 256, "batch s_proj"), (170000, 256, 512, "batch g_m"),
    (170000, 256, 768, "batch g_f"), (80960, 1280, 256, "batch vec5"), (80960, 256, 1280, "batch g_vh"),
    (10120, 768, 256, "batch qkv/o"), (10120, 256, 768, "batch g_A/g_xh"),
    (6651, 768, 256, "chig edge"), (6651, 512, 256, "chig s_proj"), (6651, 256, 512, "chig g_m"),
    (6651, 256, 768, "chig g_f"), (3128, 1280, 256, "chig vec5"), (3128, 256, 1280, "chig g_vh"),
    (391, 768, 256, "chig qkv/o"), (391, 256, 768, "chig g_A/g_xh"),
]


if os.environ.get("GEMM_SHAPES"):  # "M,Nc,K;M,Nc,K;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) + ("custom",) for t in os.environ["GEMM_SHAPES"].split(";")]


def main():
    hp = default_hparams(embedding_dimension=64, num_layers=1)
    eng = ViSNetEngine(hp, make_state_dict(hp, seed=1), "cuda:0")
    dev = "cuda:0"
    print(f"{'M':>7} {'Nc':>5} {'K':>5}  {'us':>8} {'TFLOP/s':>8}  tag")
    for M, Nc, K, tag in SHAPES:
        A = torch.randn(M, K, device=dev)
        Bt = torch.randn(Nc, K, device=dev) / K ** 0.5
        C = torch.empty(M, Nc, device=dev)
        bias = torch.randn(Nc, device=dev)
        for _ in range(3):
            eng.gemm(A, Bt, C, bias=bias)
        torch.cuda.synchronize()
        n = 20
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            eng.gemm(A, Bt, C, bias=bias)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / n
        print(f"{M:7d} {Nc:5d} {K:5d}  {us:8.1f} {2.0 * M * Nc * K / us / 1e6:8.1f}  {tag}")


if __name__ == "__main__":
    main()
