"""TEST INFRASTRUCTURE ONLY - the seeded weight generator lives in the product package
(ai2bmd_amd/synthetic.py, bench.py needs it too); re-exported here for the oracle and the tests."""
from ai2bmd_amd.synthetic import *  # noqa: F401,F403
from ai2bmd_amd.synthetic import BIAS_SCALE, default_hparams, make_state_dict, rbf_params, write_lightning_ckpt  # noqa: F401
