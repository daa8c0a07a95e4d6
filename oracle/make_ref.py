"""TEST INFRASTRUCTURE ONLY - builds oracle/_ref/: the REFERENCE's own ViSNet model and the caller of its seam, compiled.

Recipe (run by `__graft_entry__.build()` wherever /root/reference exists, i.e. in the build container):
every module of the reference's model package is byte-compiled FROM THE SOURCE WHERE IT LIES
(`/root/reference/src/ViSNet/__init__.py`, `ViSNet/model/{__init__,visnet,visnet_block,utils,output_modules,
priors}.py`, and the five caller-side files `AIMD/fragment.py`, `Calculators/{device_strategy,combiner,bonded}.py`,
`utils/utils.py`) with CPython's own compiler into sourceless `oracle/_ref/ViSNet/**/<module>.pyc`.  No reference source
enters the repository: `oracle/_ref/` is a build output like `libvsn_hip.so` - git-ignored, NOT gpurun-ignored, so it
travels to the GPU box, where `/root/reference` does not exist.  There it is what `bench.py`'s `cpu_baseline` leg
times (kind "reference": the reference's CPU path, not the oracle port) and what the smoke test may check against.

`oracle/_ref/MANIFEST.json` records, per module, the sha256 of the source it was compiled from and the interpreter's
bytecode magic; `oracle.ref_import` refuses a `_ref` built by another interpreter.

    python -m oracle.make_ref        # (re)build
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
OUT = os.path.join(HERE, "_ref")
PACKAGE = "ViSNet"


# the caller side of the seam: loaded by oracle/ref_caller.py so that the
# reference's OWN DLBondedCalculator drives the HIP seam on the GPU box (tests/test_gpu_reference_caller.py)
CALLER_FILES = ("AIMD/fragment.py", "Calculators/device_strategy.py", "Calculators/combiner.py",
                "Calculators/bonded.py", "utils/utils.py")


def _modules():
    root = os.path.join(REF_SRC, PACKAGE)
    for d, _, files in os.walk(root):
        if "__pycache__" in d:
            continue
        for f in sorted(files):
            if f.endswith(".py"):
                yield os.path.relpath(os.path.join(d, f), REF_SRC)
    for f in CALLER_FILES:
        if os.path.exists(os.path.join(REF_SRC, f)):
            yield f


def build(force: bool = False) -> str | None:
    """-> oracle/_ref (built or already current), or None where the reference tree is absent (GPU box: the prebuilt
    directory that travelled with the snapshot is used as it is)."""
    if not os.path.isdir(os.path.join(REF_SRC, PACKAGE, "model")):
        return OUT if os.path.exists(os.path.join(OUT, "MANIFEST.json")) else None
    mods = list(_modules())
    want = dict(magic=importlib.util.MAGIC_NUMBER.hex(), python=sys.version.split()[0], modules={})
    for rel in mods:
        with open(os.path.join(REF_SRC, rel), "rb") as fh:
            want["modules"][rel] = hashlib.sha256(fh.read()).hexdigest()
    man = os.path.join(OUT, "MANIFEST.json")
    if not force and os.path.exists(man):
        try:
            if json.load(open(man)) == want and all(
                    os.path.exists(os.path.join(OUT, r[:-3] + ".pyc")) for r in mods):
                return OUT
        except Exception:
            pass
    shutil.rmtree(OUT, ignore_errors=True)
    for rel in mods:
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: tracebacks name the reference file (reference:src/ViSNet/model/utils.py:NNN), not a repo path
        py_compile.compile(os.path.join(REF_SRC, rel), cfile=dst, dfile=f"reference:src/{rel}", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    with open(man, "w") as fh:
        json.dump(want, fh, indent=1, sort_keys=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
