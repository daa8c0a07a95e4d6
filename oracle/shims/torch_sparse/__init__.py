"""TEST INFRASTRUCTURE ONLY (oracle shim). `SparseTensor` is imported but never
used by the reference (ViSNet/model/utils.py:8)."""


class SparseTensor:  # pragma: no cover
    pass
