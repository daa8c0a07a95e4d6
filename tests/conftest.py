import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, f"visnet_{name}.npz"), allow_pickle=False)
    out = {k: d[k] for k in d.files}
    out["hparams"] = json.loads(str(out["hparams"]))
    out["weight_seed"] = int(out["weight_seed"])
    return out


GOLDEN_CASES = ["h64_l2", "h128_l3_lmax1", "h64_l2_trunc", "h256_l9_default", "h64_l3_rms", "h64_l3_maxmin",
                "h64_l2_whole", "h64_l2_gauss", "h128_l2_gauss_lmax1", "h64_l2_ssp", "h64_l2_tanh_sig",
                "h128_l2_sig_swish", "h192_l2", "h320_l2_rms", "h384_l2_lmax1", "h448_l2", "h512_l3", "h64_l2_mean",
                "h192_l2_heads3", "h320_l2_heads5", "h192_l3_heads12", "h192_l2_heads6_lmax1_mean_rms"]
HIP_CASES = list(GOLDEN_CASES)


@pytest.fixture(scope="session")
def lib_built():
    from ai2bmd_amd import build

    return build.build()
