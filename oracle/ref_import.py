"""TEST INFRASTRUCTURE ONLY - imports the REFERENCE's own ViSNet model source.

Only usable where /root/reference exists (this build container; NOT the GPU
box).  Puts oracle/shims (stand-ins for the un-vendored torch_scatter /
torch_cluster / torch_sparse / torch_geometric / pytorch_lightning wheels)
ahead of /root/reference/src on sys.path and returns
`ViSNet.model.visnet.create_model` (reference: src/ViSNet/model/visnet.py:14-70).
Used by oracle/make_golden.py to generate tests/golden/*.npz and by the CPU
tests that pin oracle/visnet_oracle.py to the reference.
"""
import os
import sys

REFERENCE_SRC = "/root/reference/src"
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "ViSNet", "model"))


def import_reference_create_model():
    if not reference_available():
        raise RuntimeError("reference tree not present")
    for p in (REFERENCE_SRC, _SHIMS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_SRC)
    sys.path.insert(0, _SHIMS)
    from ViSNet.model.visnet import create_model  # type: ignore
    return create_model
