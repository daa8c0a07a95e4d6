"""GPU: the REFERENCE's own caller drives the HIP seam.

`Calculators/bonded.py` (DLBondedCalculator.__init__ / calculate / __call__, :25-123), `Calculators/combiner.py`,
`Calculators/device_strategy.py`, `AIMD/fragment.py` and `utils/utils.py` of the reference are executed UNCHANGED - on
the GPU box from `oracle/_ref` (the files byte-compiled by oracle/make_ref.py; `/root/reference` does not exist
there), through oracle/ref_caller.py - with the two names north_star swaps:

    Calculators.visnet_calculator.get_visnet_model  = ai2bmd_amd.visnet_calculator.get_visnet_model   (HIP seam)
    Fragmentation.DistanceFragment                  = ai2bmd_amd.distancefrag.DistanceFragment        (HIP cap-H)

Two `cuda:0` entries in `DeviceStrategy._bonded_devices` make the reference's ThreadPoolExecutor (bonded.py:75-77) run
two model handles from two Python threads, and chunk size 120 atoms makes every handle see several FragmentData
slices (`device_strategy.py:84-127`).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


@pytest.fixture(scope="module")
def ckpt_dir(lib_built, tmp_path_factory):
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict, write_lightning_ckpt

    td = tmp_path_factory.mktemp("ckpt")
    hp = default_hparams()
    write_lightning_ckpt(str(td / "visnet-uni-bench.ckpt"), hp, make_state_dict(hp, seed=2024))
    return str(td)


def _reference_caller(relax=True, handles="shared"):
    """handles: "shared" = the factory as it is - like the reference's `_local_calc` it hands ONE model to both
    'cuda:0' entries, so the reference's two executor threads call the same handle concurrently (serialised inside the
    seam); "separate" = one handle per entry (what two different devices give)."""
    import threading

    from ai2bmd_amd.distancefrag import DistanceFragment
    from ai2bmd_amd.visnet_calculator import ViSNetModel, get_visnet_model
    from oracle.ref_caller import caller_source, load_reference_caller

    assert caller_source() is not None, ("oracle/_ref (built by __graft_entry__.build() -> oracle/make_ref.py) must "
                                         "travel with the snapshot: the reference's caller is not importable")
    calls = []

    class Traced:  # forwards everything; records who called the seam with what
        def __init__(self, inner):
            self.inner, self.device = inner, inner.device

        def dl_potential_loader(self, frag_data):
            calls.append((id(self), threading.get_ident(), len(frag_data), type(frag_data).__module__))
            return self.inner.dl_potential_loader(frag_data)

    def traced_get_visnet_model(model_path, device):
        m = get_visnet_model(model_path, device) if handles == "shared" else ViSNetModel.from_file(
            model_path=model_path, device=device)
        return Traced(m)

    class Fragmenter(DistanceFragment):  # (same class; only the constructor default differs for the "placed" run)
        def __init__(self):
            super().__init__(relax=relax)

    ref = load_reference_caller(traced_get_visnet_model, Fragmenter)
    DS = ref.DeviceStrategy
    DS._gpu_count, DS._bonded_devices, DS._default_device, DS._chunk_size = 1, ["cuda:0", "cuda:0"], "cuda:0", 120
    DS._optimiser_device = "cuda:0"
    return ref, DS, calls


@pytest.mark.parametrize("name,handles", [("chig", "shared"), ("chig", "separate"), ("ww", "separate")])
def test_reference_dl_bonded_calculator_on_the_hip_seam(ckpt_dir, name, handles):
    ref, DS, calls = _reference_caller(handles=handles)
    assert ref.DLBondedCalculator.__module__ == "Calculators.bonded"
    calc = ref.DLBondedCalculator(ckpt_dir, "bench")                  # the reference's own constructor
    assert len(calc.models) == 2 and all(m.device == "cuda:0" for m in calc.models)
    assert (calc.models[0].inner is calc.models[1].inner) == (handles == "shared")
    prot = _protein(name)
    calc.fragment_method.fragment(prot)                               # simulator.py:53-57 initialize_fragcalc
    DS.set_work_partitions(prot.fragments_start.tolist(), prot.fragments_end.tolist())
    work = DS.get_work_partitions()
    assert {w[0] for w in work} == {0, 1} and len(work) >= 4          # both handles, several chunks each
    E, F = calc(prot)                                                 # the reference's own __call__
    # the reference's executor ran the two entries on two worker threads, one call per chunk; every chunk is a slice
    # (`fragments[start:end]`, bonded.py:73) of the FragmentData our fragment producer handed out
    assert len(calls) == len(work) and len({c[0] for c in calls}) == 2 and len({c[1] for c in calls}) == 2
    assert {c[3] for c in calls} == {"ai2bmd_amd.fragment"}
    assert sorted(c[2] for c in calls) == sorted(w[2] - w[1] for w in work)
    g = np.load(os.path.join(GOLDEN, f"visnet_prot_{name}.npz"))
    Fg, Eg = g["Fprot64_relaxed"], float(g["Eprot64_relaxed"])
    assert isinstance(F, np.ndarray) and F.shape == Fg.shape and F.dtype == np.float32
    assert np.abs(F - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max()), np.abs(F - Fg).max()
    assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))
    # ... and equals the mirror class (ai2bmd_amd.bonded.DLBondedCalculator) on the same handles and partitions
    from ai2bmd_amd.bonded import DLBondedCalculator as Mirror

    mirror = Mirror.from_models([m.inner for m in calc.models], chunk_atoms=120, fragment_method=calc.fragment_method)
    Em, Fm = mirror(prot)
    assert mirror._work == [tuple(w) for w in work]
    np.testing.assert_allclose(F, Fm, rtol=0, atol=2e-6)
    assert abs(float(E) - float(Em)) <= 2e-5 * max(1.0, abs(Eg))


def test_reference_caller_against_the_all_reference_chain_golden(ckpt_dir):
    """tests/golden/refchain_chig.npz (oracle/make_refchain_golden.py): fragmenter, model, split and combiner ALL the
    reference's own code, cap hydrogens at their first-guess positions.  Here: the reference's DLBondedCalculator on
    the HIP seam with the HIP fragment producer's relaxation switched off = the same geometry."""
    ref, DS, calls = _reference_caller(relax=False)
    calc = ref.DLBondedCalculator(ckpt_dir, "bench")
    prot = _protein("chig")
    calc.fragment_method.fragment(prot)
    DS.set_work_partitions(prot.fragments_start.tolist(), prot.fragments_end.tolist())
    E, F = calc(prot)
    g = np.load(os.path.join(GOLDEN, "refchain_chig.npz"))
    Fg, Eg = g["Fprot64"], float(g["Eprot64"])
    assert np.abs(F - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max()), np.abs(F - Fg).max()
    assert np.abs(F - Fg).mean() <= 1e-5 * max(1.0, np.abs(Fg).mean())
    assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))
    # no worse than 4x the reference's own fp32 error on this chain
    ref_err = max(np.abs(g["Fprot32"] - Fg).max(), 2e-6)
    assert np.abs(F - Fg).max() <= 4 * ref_err + 1e-6 * np.abs(Fg).max()
    # the recombination indices our fragment producer leaves are what the reference's leaves: torch, default device,
    # the same values in the same order; and the fragment batch is the reference's row for row
    assert torch.is_tensor(prot.select_index) and prot.select_index.device.type == "cuda"
    assert np.array_equal(prot.select_index.cpu().numpy(), g["select_index"])
    assert np.array_equal(prot.origin_index.cpu().numpy(), g["origin_index"])
    assert np.array_equal(prot.fragments_z, g["z"]) and np.array_equal(prot.fragments_start, g["start"])


@pytest.mark.parametrize("case,name", [("chig_nb20", "chig"), ("abd_nb31", "abd"), ("abd_nb24", "abd")])
def test_reference_caller_under_neighbour_truncation(lib_built, tmp_path, case, name):
    """tests/golden/refchain_<case>.npz: the all-reference chain with `max_num_neighbors` lowered until targets
    truncate (Chignolin at 20: 30 % of the targets; ABD at 31 and 24).  `radius_graph` keeps the lowest-index sources,
    so the result depends on the row order of the FragmentData: the reference's DLBondedCalculator on the HIP seam, fed
    by the HIP fragment producer (rows in the reference's AMBER order), reproduces it."""
    from ai2bmd_amd.fragmentation import preprocessed_order
    from ai2bmd_amd.synthetic import default_hparams, make_state_dict, write_lightning_ckpt

    g = np.load(os.path.join(GOLDEN, f"refchain_{case}.npz"))
    hp = default_hparams(max_num_neighbors=int(g["max_num_neighbors"]))
    write_lightning_ckpt(str(tmp_path / "visnet-uni-bench.ckpt"), hp, make_state_dict(hp, seed=int(g["weight_seed"])))
    ref, DS, calls = _reference_caller(relax=False, handles="separate")
    calc = ref.DLBondedCalculator(str(tmp_path), "bench")
    prot = _protein(name)
    if name != "chig":
        prot = preprocessed_order(prot)  # the atom numbering of the golden's protein forces
    calc.fragment_method.fragment(prot)
    assert np.array_equal(prot.fragments_z, g["z"]) and np.array_equal(prot.fragments_end, g["end"])
    DS.set_work_partitions(prot.fragments_start.tolist(), prot.fragments_end.tolist())
    E, F = calc(prot)
    assert len(calls) == len(DS.get_work_partitions()) and int(g["n_truncated"]) >= 1
    Fg, Eg = g["Fprot64"], float(g["Eprot64"])
    assert np.abs(F - Fg).max() <= 1e-4 * max(1.0, np.abs(Fg).max()), np.abs(F - Fg).max()
    assert np.abs(F - Fg).mean() <= 1e-5 * max(1.0, np.abs(Fg).mean())
    assert abs(float(E) - Eg) <= 1e-4 * max(1.0, abs(Eg))
    ref_err = max(np.abs(g["Fprot32"] - Fg).max(), 2e-6)
    assert np.abs(F - Fg).max() <= 4 * ref_err + 1e-6 * np.abs(Fg).max()
