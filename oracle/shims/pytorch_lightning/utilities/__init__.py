"""TEST INFRASTRUCTURE ONLY (oracle shim): rank_zero_warn -> warnings.warn
(used at ViSNet/model/visnet.py:6,81,114)."""
import warnings


def rank_zero_warn(msg, *a, **k):
    warnings.warn(msg)
