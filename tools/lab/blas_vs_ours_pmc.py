"""Lab: the library's fp32 GEMM (torch.mm -> Tensile) and the production kernel of gemm.hip (C ABI vsn_gemm) on the same
fragment-batch shapes in one process - run under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`
(+ tools/pmc_summary.py) to see whether the library's lead is MFMA-pipe occupancy or clock."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402

dev = "cuda:0"
hp = default_hparams(num_layers=1)
eng = ViSNetEngine(hp, make_state_dict(hp, seed=1), dev)
torch.manual_seed(0)
for M, Nc, K in [(1000000, 768, 256), (1000000, 256, 768), (490000, 1280, 256)]:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(Nc, K, device=dev) / K ** 0.5
    C1 = torch.empty(M, Nc, device=dev)
    C2 = torch.empty(M, Nc, device=dev)
    for _ in range(4):
        torch.mm(A, W.t(), out=C1)
        eng.gemm(A, W, C2)
    torch.cuda.synchronize()
    print(M, Nc, K, "max |library - ours|", (C1 - C2).abs().max().item(), flush=True)
    del A, W, C1, C2
    torch.cuda.empty_cache()
