"""TEST INFRASTRUCTURE ONLY - loads the REFERENCE's own fragmenter
(/root/reference/src/Fragmentation/{basefrag,distancefrag}.py) in the build container with its absent imports
stubbed, so that our fragment plan (ai2bmd_amd/fragmentation.py) can be pinned on what the reference computes.

Stubbed (not restated arithmetic, just import plumbing):
  ase.Atoms (type annotation), AIMD.arguments.get() -> verbose=0, AIMD.preprocess.Preprocess.get_seq_dict_path()
  -> the reference's own utils/seq_dict.pkl, AIMD.protein.Protein (annotation), Calculators.device_strategy
  .DeviceStrategy.get_*_device() -> "cpu", utils.utils.numpy_to_torch / numpy_list_to_torch (two one-liners,
  utils/utils.py:229-233; the real module needs ase), Fragmentation.hydrogen (needs torch_geometric: the graph for
  the hydrogen optimiser is not built here - that part is pinned separately, oracle/make_hydrogen_golden.py).
Real reference code that runs: get_fragments_index, get_hydrogen_indices, calculate_permutation_indices (with
seq_dict.pkl), calculate_select_indices, the index algebra of `fragment`, get_dipeptide_positions,
utils/reference.py, AIMD/fragment.py.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

SRC = "/root/reference/src"


def available():
    return os.path.exists(os.path.join(SRC, "Fragmentation", "distancefrag.py"))


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(SRC, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_distance_fragment():
    def mk(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mk("ase", Atoms=object)
    args = types.SimpleNamespace(verbose=0)
    mk("AIMD", arguments=mk("AIMD.arguments", get=lambda: args))
    _load("AIMD.fragment", "AIMD/fragment.py")
    mk("AIMD.preprocess", Preprocess=type("Preprocess", (), {
        "get_seq_dict_path": staticmethod(lambda: os.path.join(SRC, "utils", "seq_dict.pkl"))}))
    mk("AIMD.protein", Protein=object)
    ds = type("DeviceStrategy", (), {"get_optimiser_device": staticmethod(lambda: "cpu"),
                                     "get_default_device": staticmethod(lambda: "cpu")})
    mk("Calculators")
    mk("Calculators.device_strategy", DeviceStrategy=ds)
    mk("utils")
    _load("utils.reference", "utils/reference.py")
    mk("utils.utils", numpy_to_torch=lambda x, device: torch.from_numpy(x).to(device),
       numpy_list_to_torch=lambda x, device: torch.from_numpy(np.concatenate(x)).to(device))

    class _Batch:
        @staticmethod
        def from_data_list(lst):
            return types.SimpleNamespace(to=lambda device: None)

    mk("Fragmentation")
    mk("Fragmentation.hydrogen", CTable=object, HydrogenOptimizer=lambda max_iter: None, ProteinData=object,
       ProteinDataBatch=_Batch)
    _load("Fragmentation.basefrag", "Fragmentation/basefrag.py")
    mod = _load("Fragmentation.distancefrag", "Fragmentation/distancefrag.py")
    DF = mod.DistanceFragment
    DF.create_protein_graph = staticmethod(lambda resi_info, lengths, constrain: [])  # needs torch_geometric
    return DF


class DuckProtein:
    """what DistanceFragment reads from the reference's `Protein` (an ase.Atoms subclass)"""

    def __init__(self, p):
        self.arrays = {"residuenumbers": np.asarray(p.resnums), "residuenames": np.asarray(p.resnames),
                       "atomtypes": np.asarray(p.names), "positions": np.asarray(p.positions, dtype=np.float64),
                       "numbers": np.asarray(p.numbers)}

    def __len__(self):
        return len(self.arrays["numbers"])

    def get_positions(self):
        return self.arrays["positions"]

    def get_chemical_symbols(self):
        sym = {1: "H", 6: "C", 7: "N", 8: "O", 16: "S"}
        return [sym[int(z)] for z in self.arrays["numbers"]]


def run_reference_fragmenter(p):
    """-> dict with the reference's fragment batch for protein `p` (ai2bmd_amd.fragmentation.ProteinAtoms)"""
    DF = load_distance_fragment()
    frag = DF()
    prot = DuckProtein(p)
    frag.fragment(prot)
    dip_pos = DF.get_dipeptide_positions(prot, "cpu").numpy()
    return dict(
        z=np.asarray(prot.fragments_z), start=np.asarray(prot.fragments_start), end=np.asarray(prot.fragments_end),
        batch=np.asarray(prot.fragments_batch), pos=dip_pos[prot.fragments_index],
        select_index=prot.select_index.numpy(), origin_index=prot.origin_index.numpy(),
        n_dip_rows=int(prot.dipeptides_len),
    )
