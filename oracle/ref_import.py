"""TEST INFRASTRUCTURE ONLY - imports the REFERENCE's own ViSNet model.

Two places it can come from:
* `/root/reference/src` (this build container; NOT the GPU box): the source tree itself;
* `oracle/_ref/` (everywhere the snapshot travels, the GPU box included): the same modules byte-compiled from that
  tree by `oracle/make_ref.py` (a git-ignored build output, like the .so).

Either way oracle/shims (stand-ins for the un-vendored torch_scatter / torch_cluster / torch_sparse /
torch_geometric / pytorch_lightning wheels) go ahead of it on sys.path and
`ViSNet.model.visnet.create_model` (reference: src/ViSNet/model/visnet.py:14-70) is returned.
Used by oracle/make_golden.py to generate tests/golden/*.npz, by the CPU tests that pin
oracle/visnet_oracle.py to the reference and by bench.py's `cpu_baseline` leg.
"""
import importlib.util
import json
import os
import sys

REFERENCE_SRC = "/root/reference/src"
_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIMS = os.path.join(_HERE, "shims")
COMPILED_REF = os.path.join(_HERE, "_ref")


def reference_available() -> bool:
    """the reference SOURCE tree is here (live-reference tests need more of it than the model package)"""
    return os.path.isdir(os.path.join(REFERENCE_SRC, "ViSNet", "model"))


def compiled_reference_available() -> bool:
    """oracle/_ref holds the reference's model package compiled by THIS interpreter version"""
    man = os.path.join(COMPILED_REF, "MANIFEST.json")
    try:
        with open(man) as fh:
            m = json.load(fh)
    except Exception:
        return False
    return m.get("magic") == importlib.util.MAGIC_NUMBER.hex() and os.path.exists(
        os.path.join(COMPILED_REF, "ViSNet", "model", "visnet.pyc"))


def reference_model_source() -> str | None:
    """where `import_reference_create_model` would import from: "source" | "compiled" | None"""
    if reference_available():
        return "source"
    if compiled_reference_available():
        return "compiled"
    return None


def import_reference_create_model(prefer: str | None = None):
    """prefer="compiled" imports oracle/_ref even where the source tree exists (used to test the recipe)"""
    kind = prefer or reference_model_source()
    if kind == "source" and reference_available():
        root = REFERENCE_SRC
    elif kind == "compiled" and compiled_reference_available():
        root = COMPILED_REF
    else:
        raise RuntimeError("reference model not present (neither /root/reference/src nor oracle/_ref)")
    for p in (REFERENCE_SRC, COMPILED_REF, _SHIMS):
        while p in sys.path:
            sys.path.remove(p)
    loaded = sys.modules.get("ViSNet")
    if loaded is not None and not any(os.path.abspath(p).startswith(root) for p in getattr(loaded, "__path__", [])):
        for name in [n for n in sys.modules if n == "ViSNet" or n.startswith("ViSNet.")]:
            del sys.modules[name]
    sys.path.insert(0, root)
    sys.path.insert(0, _SHIMS)
    from ViSNet.model.visnet import create_model  # type: ignore
    return create_model
