"""GPU: the fp32 MFMA GEMM tap (C ABI vsn_gemm) against torch fp64 on the host."""
import numpy as np
import pytest
import torch

from oracle.weights import default_hparams, make_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(lib_built):
    from ai2bmd_amd.visnet_calculator import ViSNetEngine

    hp = default_hparams(embedding_dimension=64, num_layers=1)
    return ViSNetEngine(hp, make_state_dict(hp, seed=1), "cuda:0")


@pytest.mark.parametrize("M,Nc,K", [(1, 32, 32), (37, 64, 96), (300, 96, 64), (1000, 256, 256), (4099, 768, 256),
                                    (2500, 256, 768), (5000, 1280, 256), (129, 32, 512), (6000, 128, 32),
                                    (391, 256, 768), (3128, 256, 1280), (6651, 256, 512), (70000, 256, 768)])
@pytest.mark.parametrize("flags", [0, 1, 2])
def test_gemm_matches_fp64(eng, M, Nc, K, flags):
    g = torch.Generator().manual_seed(M * 7 + Nc + K + flags)
    A = torch.randn(M, K, generator=g)
    Bt = torch.randn(Nc, K, generator=g) / K ** 0.5  # asymmetric, non-square: catches transposes
    bias = torch.randn(Nc, generator=g)
    C0 = torch.randn(M, Nc, generator=g)
    dev = "cuda:0"
    Ad, Bd, bd, Cd = A.to(dev), Bt.to(dev), bias.to(dev), C0.clone().to(dev)
    eng.gemm(Ad, Bd, Cd, bias=bd, flags=flags)
    torch.cuda.synchronize()
    A64 = A.double()
    if flags & 2:
        A64 = A64 * torch.sigmoid(A64)
    ref = A64 @ Bt.double().T + bias.double()
    if flags & 1:
        ref = ref + C0.double()
    got = Cd.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_gemm_strided_views(eng):
    """lda/ldb/ldc larger than the logical widths (column slices of wider buffers)."""
    g = torch.Generator().manual_seed(3)
    big_a = torch.randn(500, 3 * 64, generator=g).to("cuda:0")
    big_b = torch.randn(128, 2 * 64, generator=g).to("cuda:0")
    big_c = torch.zeros(500, 5 * 64, device="cuda:0")
    A = big_a[:, 64:128]
    Bt = big_b[:, 64:]
    C = big_c[:, 128:256]
    eng.gemm(A, Bt, C)
    torch.cuda.synchronize()
    ref = A.cpu().double() @ Bt.cpu().double().T
    assert (C.cpu().double() - ref).abs().max().item() < 1e-4
    assert big_c[:, :128].abs().max().item() == 0 and big_c[:, 256:].abs().max().item() == 0


def test_silu_prologue_extreme_inputs(eng):
    """The activation on v_exp_f32 / v_rcp_f32 (csrc/common.h sigmoid_f) over the whole fp32 range: one-hot weights
    turn the silu(A) product into silu(A) itself.  Saturation (|x| >> 88, where exp overflows / flushes), signed
    zeros, denormals and +-inf limits must come out as torch's silu does, 2 ulp-ish everywhere else."""
    vals = [0.0, -0.0, 1e-42, -1e-42, 1e-30, -1e-30, 1e-3, -1e-3, 0.5, -0.5, 1.0, -1.0, 5.0, -5.0, 20.0, -20.0,
            80.0, -80.0, 87.0, -87.0, 88.5, -88.5, 89.0, -89.0, 100.0, -100.0, 1e4, -1e4, 3e38, -3e38]
    K = 32
    A = torch.zeros(64, K)
    for i, v in enumerate(vals):
        A[i, i % K] = v
    A[40:, :] = torch.linspace(-30, 30, 24 * K).reshape(24, K)
    Bt = torch.eye(K)  # C = silu(A) . I
    Cd = torch.zeros(64, K, device="cuda:0")
    eng.gemm(A.to("cuda:0"), Bt.to("cuda:0"), Cd, flags=2)
    torch.cuda.synchronize()
    got = Cd.cpu().double()
    ref = torch.nn.functional.silu(A.double())
    assert torch.isfinite(got).all(), "silu must saturate, not overflow"
    err = (got - ref).abs()
    tol = 4e-7 * ref.abs().clamp(min=1.0) + 1e-37
    assert (err <= tol).all(), (err / tol).max().item()
