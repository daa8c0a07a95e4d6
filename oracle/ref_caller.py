"""TEST INFRASTRUCTURE ONLY - imports the REFERENCE's own CALLER of the ViSNet seam.

The caller side of the hot path in the reference is five files:

    AIMD/fragment.py               FragmentData / FragmentInfo                  (:7-70)
    Calculators/device_strategy.py DeviceStrategy (devices, work partitions)    (:84-127)
    Calculators/combiner.py        DipeptideBondedCombiner                      (:12-41)
    Calculators/bonded.py          DLBondedCalculator (__init__/calculate/__call__, :25-123)
    utils/utils.py                 numpy_to_torch (:229-230), RNGPool (:28-49)

`load_reference_caller(get_visnet_model, DistanceFragment)` executes those files - from `/root/reference/src` where
the tree exists, else from `oracle/_ref/` (the same files byte-compiled by oracle/make_ref.py; the GPU box) - with
exactly the two names the reference's `bonded.py` pulls from outside this list injected:

    Calculators.visnet_calculator.get_visnet_model   <- the seam under test (HIP: ai2bmd_amd.visnet_calculator's)
    Fragmentation.DistanceFragment                   <- the fragment producer (HIP: ai2bmd_amd.distancefrag's)

and the absent wheels stubbed (ase: only the class names these files mention at import time; torch_scatter through
oracle/shims).  Nothing of the reference is modified: `DLBondedCalculator.__init__`, `.calculate` (ThreadPoolExecutor
over DeviceStrategy.get_bonded_devices(), chunked work partitions, np.concatenate, scalar/vector split) and
`.__call__` (combiner) run as written.
"""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import sys
import types

from .ref_import import COMPILED_REF, REFERENCE_SRC, _SHIMS, compiled_reference_available, reference_available

CALLER_FILES = ("AIMD/fragment.py", "Calculators/device_strategy.py", "Calculators/combiner.py",
                "Calculators/bonded.py", "utils/utils.py")


def compiled_caller_available() -> bool:
    return compiled_reference_available() and all(
        os.path.exists(os.path.join(COMPILED_REF, f[:-3] + ".pyc")) for f in CALLER_FILES)


def caller_source(prefer: str | None = None) -> str | None:
    if prefer in (None, "source") and reference_available():
        return "source"
    if prefer in (None, "compiled") and compiled_caller_available():
        return "compiled"
    return None


def _exec(name: str, rel: str, kind: str):
    """execute one reference file as module `name` (registered in sys.modules under that name)"""
    if kind == "source":
        spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_SRC, rel))
    else:
        path = os.path.join(COMPILED_REF, rel[:-3] + ".pyc")
        spec = importlib.util.spec_from_loader(name, importlib.machinery.SourcelessFileLoader(name, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_TOUCHED = ("ase", "ase.io", "ase.io.trajectory", "ase.md", "ase.md.md", "AIMD", "AIMD.arguments", "AIMD.fragment",
            "AIMD.protein", "Calculators", "Calculators.combiner", "Calculators.device_strategy",
            "Calculators.visnet_calculator", "Calculators.bonded", "Fragmentation", "utils", "utils.utils",
            "utils.system")


def load_reference_caller(get_visnet_model, DistanceFragment, physical_cores: int = 8, prefer: str | None = None):
    """-> namespace(bonded, DLBondedCalculator, DeviceStrategy, FragmentData, DipeptideBondedCombiner, utils, origin).
    The module names the reference's files import from each other are registered only while they are executed and
    restored afterwards (the returned module objects keep working: they hold their own references)."""
    kind = caller_source(prefer)
    if kind is None:
        raise RuntimeError("reference caller not present (neither /root/reference/src nor a complete oracle/_ref)")
    if _SHIMS not in sys.path:
        sys.path.insert(0, _SHIMS)  # torch_scatter (Calculators/combiner.py:3)
    saved = {k: sys.modules.get(k) for k in _TOUCHED}
    try:
        try:
            import ase  # noqa: F401  (absent in this image; the real package is used where it exists)
            have_ase = hasattr(ase, "__path__")  # (a stub module left by oracle/ref_fragmenter.py is not the package)
        except ImportError:
            have_ase = False
        if not have_ase:
            _stub("ase", Atoms=type("Atoms", (), {}))
            _stub("ase.io")
            _stub("ase.io.trajectory", TrajectoryWriter=type("TrajectoryWriter", (), {}))
            _stub("ase.md")
            _stub("ase.md.md", MolecularDynamics=type("MolecularDynamics", (), {}))
        aimd = _stub("AIMD")
        aimd.arguments = _stub("AIMD.arguments", get=lambda: types.SimpleNamespace())
        frag = _exec("AIMD.fragment", "AIMD/fragment.py", kind)
        aimd.fragment = frag
        _stub("AIMD.protein", Protein=object)
        _stub("utils")
        _stub("utils.system", get_physical_core_count=lambda: physical_cores)
        utils = _exec("utils.utils", "utils/utils.py", kind)
        _stub("Calculators")
        ds = _exec("Calculators.device_strategy", "Calculators/device_strategy.py", kind)
        comb = _exec("Calculators.combiner", "Calculators/combiner.py", kind)
        _stub("Calculators.visnet_calculator", ViSNetModelLike=object, get_visnet_model=get_visnet_model)
        _stub("Fragmentation", DistanceFragment=DistanceFragment)
        bonded = _exec("Calculators.bonded", "Calculators/bonded.py", kind)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return types.SimpleNamespace(bonded=bonded, DLBondedCalculator=bonded.DLBondedCalculator,
                                 DeviceStrategy=ds.DeviceStrategy, FragmentData=frag.FragmentData,
                                 DipeptideBondedCombiner=comb.DipeptideBondedCombiner, utils=utils, origin=kind)
