"""CPU, build container only: re-runs the reference's own ViSNet source (through
oracle/shims) against the oracle restatement on fresh seeds.  Skipped where
/root/reference does not exist (the GPU box)."""
import numpy as np
import pytest
import torch

from oracle.inputs import random_fragments
from oracle.ref_import import reference_available
from oracle.visnet_oracle import ViSNetOracle
from oracle.weights import default_hparams, make_state_dict

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("vn,lmax,mnb", [("none", 2, 32), ("rms", 2, 32), ("max_min", 1, 32), ("none", 2, 12)])
def test_live_reference(vn, lmax, mnb):
    from oracle.make_golden import run_reference

    hp = default_hparams(embedding_dimension=64, num_layers=2, vecnorm_type=vn, lmax=lmax, max_num_neighbors=mnb)
    sd = make_state_dict(hp, seed=5)
    z, pos, start, end = random_fragments(55, [19, 0, 12, 36])
    E_ref, F_ref = run_reference(hp, sd, z, pos, start, end, torch.float64)
    E, F, c = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    assert np.diff(c["graph"]["rowptr"]).max() <= mnb
    np.testing.assert_allclose(E, E_ref, atol=1e-10)
    np.testing.assert_allclose(F, F_ref, atol=1e-10)


def _load_reference_module(name, relpath, stubs=()):
    """import one file of the reference tree by path, with the named absent imports stubbed"""
    import importlib.util
    import os
    import sys
    import types

    shim_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims")
    if shim_dir not in sys.path:
        sys.path.insert(0, shim_dir)
    for modname, attrs in stubs:
        m = types.ModuleType(modname)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(modname, m)
    spec = importlib.util.spec_from_file_location(name, os.path.join("/root/reference/src", relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_live_reference_combiner():
    """ai2bmd_amd.bonded.combine_numpy / fragmentation.combine_host against the reference's own
    DipeptideBondedCombiner (Calculators/combiner.py) on a real fragment plan."""
    import os

    from ai2bmd_amd.bonded import combine_numpy
    from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, combine_host
    from conftest import GOLDEN

    ref = _load_reference_module("ref_combiner", "Calculators/combiner.py").DipeptideBondedCombiner
    d = np.load(os.path.join(GOLDEN, "protein_trpcage.npz"))
    plan = build_plan(ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"]))
    rng = np.random.default_rng(3)
    f_all = rng.standard_normal((len(plan.z), 3)).astype(np.float32)
    e_all = rng.standard_normal(len(plan.start)).astype(np.float32)
    rows_dip = plan.row_of_cat[:plan.n_dip_rows]
    rows_ace = plan.row_of_cat[plan.n_dip_rows:]
    f_dip, f_ace = f_all[rows_dip], f_all[rows_ace]
    e_dip, e_ace = e_all[plan.is_dipeptide], e_all[~plan.is_dipeptide]
    F_ref = ref.forces_combine(plan.n_prot, torch.as_tensor(f_dip), torch.as_tensor(f_ace),
                               torch.as_tensor(plan.select_index), torch.as_tensor(plan.origin_index))
    E_ref = float(ref.energy_combine(torch.as_tensor(e_dip), torch.as_tensor(e_ace)))
    E1, F1 = combine_numpy(plan.n_prot, e_dip, f_dip, e_ace, f_ace, plan.select_index, plan.origin_index)
    E2, F2 = combine_host(plan, e_all[:, None], f_all)
    assert abs(float(E1) - E_ref) < 1e-5 and abs(float(E2) - E_ref) < 1e-5
    np.testing.assert_allclose(F1, F_ref, atol=1e-6)
    np.testing.assert_allclose(F2, F_ref, atol=1e-6)


@pytest.mark.parametrize("ndev,chunk", [(1, 9999), (2, 9999), (4, 120), (8, 9999), (3, 70)])
def test_live_reference_work_partitions(lib_built, ndev, chunk):
    """vsn_partition (C) against the reference's own DeviceStrategy._set_combined_work_partitions."""
    from ai2bmd_amd.device_strategy import work_partitions

    mod = _load_reference_module(
        "ref_device_strategy", "Calculators/device_strategy.py",
        stubs=(("AIMD", {}), ("AIMD.fragment", {"FragmentInfo": object}), ("utils", {}),
               ("utils.system", {"get_physical_core_count": lambda: 8})))
    DS = mod.DeviceStrategy
    rng = np.random.default_rng(ndev * 100 + chunk)
    sizes = []
    for _ in range(35):
        sizes += [int(rng.integers(19, 37)), 12]
    sizes.append(int(rng.integers(19, 37)))
    end = np.cumsum(sizes)
    start = end - np.asarray(sizes)
    DS._chunk_size = chunk
    DS._set_combined_work_partitions(list(range(ndev)), start.tolist(), end.tolist())
    assert work_partitions(start, end, ndev, chunk) == [tuple(int(v) for v in t) for t in DS._work_partitions]


def test_live_reference_fragment_data():
    """ai2bmd_amd.fragment.FragmentData against the reference's own class (AIMD/fragment.py; `ase.Atoms` stubbed):
    slicing re-bases start/end/batch the same way, scalar/vector splits agree, including an empty (CYZ) fragment."""
    from ai2bmd_amd.fragment import FragmentData, make_batch_index

    Ref = _load_reference_module("ref_fragment", "AIMD/fragment.py", stubs=(("ase", {"Atoms": object}),)).FragmentData
    sizes = [22, 12, 0, 12, 30, 12, 25]
    end = np.cumsum(sizes)
    start = end - np.asarray(sizes)
    n = int(end[-1])
    args = (np.arange(n), np.arange(3 * n, dtype=np.float32).reshape(n, 3), start, end, make_batch_index(start, end))
    mine, ref = FragmentData(*args), Ref(*args)
    for a, b in zip(mine.scalar_split(), ref.scalar_split()):
        assert np.array_equal(a, b)
    for a, b in zip(mine.vector_split(), ref.vector_split()):
        assert np.array_equal(a, b)
    for sl in (slice(0, 7), slice(1, 4), slice(3, 6), 4, 0):
        m, r = mine[sl], ref[sl]
        for k in ("z", "pos", "start", "end", "batch"):
            assert np.array_equal(getattr(m, k), getattr(r, k)), (sl, k)
        assert len(m) == len(r)


@pytest.mark.parametrize("name", ["chig", "trpcage", "ww", "abd"])
def test_live_reference_fragmenter_matches_golden(name):
    """the committed golden of the reference's fragmenter is what the reference tree produces now (Chignolin: the
    reference's pre-processed example; the C3 / C4 proteins: its examples put into that atom order)"""
    import os

    from ai2bmd_amd.fragmentation import ProteinAtoms, preprocessed_order
    from conftest import GOLDEN
    from oracle.ref_fragmenter import run_reference_fragmenter

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    p = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"])
    r = run_reference_fragmenter(p if name == "chig" else preprocessed_order(p))
    g = np.load(os.path.join(GOLDEN, f"fragref_{name}.npz"))
    assert np.array_equal(r["z"], g["z"]) and np.array_equal(r["select_index"], g["select_index"])
    assert np.array_equal(r["origin_index"], g["origin_index"]) and np.allclose(r["pos"], g["pos"], atol=1e-6)


def test_live_reference_load_model_reads_our_checkpoint(tmp_path):
    """the Lightning-shaped file written by ai2bmd_amd.synthetic.write_lightning_ckpt (what ViSNetModel.from_file
    reads) goes through the reference's own `load_model` (ViSNet/model/visnet.py:73-93) and evaluates to the oracle."""
    from ai2bmd_amd.synthetic import write_lightning_ckpt
    from oracle.ref_import import import_reference_create_model

    hp = default_hparams(embedding_dimension=64, num_layers=2)
    sd = make_state_dict(hp, seed=9)
    path = str(tmp_path / "visnet.ckpt")
    write_lightning_ckpt(path, hp, sd)
    import_reference_create_model()
    import ViSNet.model.visnet as visnet  # the reference's module (through oracle/shims)

    real_script = torch.jit.script
    torch.jit.script = lambda m, *a, **k: m  # the PyG stand-in is plain Python; TorchScript is not part of the format
    try:
        model = visnet.load_model(path, device="cpu")
    finally:
        torch.jit.script = real_script
    z, pos, start, end = random_fragments(77, [22, 12, 30])
    batch = np.repeat(np.arange(len(start)), end - start)
    E_ref, F_ref = model(dict(z=torch.as_tensor(z), pos=torch.as_tensor(pos, dtype=torch.float32),
                              batch=torch.as_tensor(batch)))
    E, F, _ = ViSNetOracle(hp, sd, torch.float64).energy_forces(z, pos, start, end)
    np.testing.assert_allclose(E_ref.detach().numpy(), E, atol=2e-4)
    np.testing.assert_allclose(F_ref.detach().numpy(), F, atol=2e-4)


@pytest.mark.parametrize("origin", ["source", "compiled"])
def test_reference_bonded_calculator_drives_our_seam_unchanged(lib_built, origin):
    """The reference's OWN Calculators/bonded.py (DLBondedCalculator.__init__/calculate/__call__, :25-123), loaded
    through oracle/ref_caller.py - from /root/reference ("source") and from the byte-compiled oracle/_ref the GPU box
    uses ("compiled"; tests/test_gpu_reference_caller.py runs the same caller on the HIP seam) - is pointed at a `get_visnet_model` that returns an object with our
    seam's shape (`dl_potential_loader(FragmentData) -> (e[B,1], f[N,3])` numpy; here backed by the CPU oracle, the HIP
    model needs a GPU) and at a fragment producer handing out OUR FragmentData.  It runs unchanged and its result
    equals the mirror class's (ai2bmd_amd.bonded.DLBondedCalculator) on the same inputs."""
    import os
    import sys
    import types

    from ai2bmd_amd.amber import load_tables
    from ai2bmd_amd.bonded import DLBondedCalculator as Mirror
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, combine_host, fragment_positions
    from ai2bmd_amd.hydrogen import build_hydrogen_plan
    from conftest import GOLDEN
    from oracle.hydrogen_oracle import HydrogenOracle

    hp = default_hparams(embedding_dimension=64, num_layers=2)
    sd = make_state_dict(hp, seed=33)
    oracle = ViSNetOracle(hp, sd, torch.float32)
    calls = []

    class OracleSeam:  # the shape of ai2bmd_amd.visnet_calculator.ViSNetModel
        implemented_properties = ["energy", "forces"]

        def __init__(self, device):
            self.device = device

        def dl_potential_loader(self, frag_data):
            calls.append((self.device, len(frag_data)))
            E, F, _ = oracle.energy_forces(frag_data.z, frag_data.pos, frag_data.start, frag_data.end)
            return E.astype(np.float32).reshape(-1, 1), F.astype(np.float32)

    d = np.load(os.path.join(GOLDEN, "protein_chig.npz"))
    prot = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))
    plan = build_plan(prot)
    hplan = build_hydrogen_plan(prot, plan, load_tables(os.path.join(GOLDEN, "amber_tables.npz")))

    class Fragmenter:  # DistanceFragment.get_fragments on the host (the HIP one needs a GPU)
        def get_fragments(self, p):
            pos = HydrogenOracle(hplan).relax(fragment_positions(plan, p.positions).astype(np.float32))
            ace = hplan.alias >= 0
            pos[ace] = pos[hplan.alias[ace]]
            return FragmentData(plan.z, pos, plan.start, plan.end, make_batch_index(plan.start, plan.end))

    from oracle.ref_caller import load_reference_caller

    ref = load_reference_caller(lambda model_path, device: OracleSeam(device), Fragmenter, prefer=origin)
    assert ref.origin == origin
    ref_bonded = ref.bonded
    DS = ref.DeviceStrategy
    DS._gpu_count, DS._bonded_devices, DS._default_device, DS._chunk_size = 0, ["cpu", "cpu"], "cpu", 120
    DS.set_work_partitions(plan.start.tolist(), plan.end.tolist())
    calc = ref_bonded.DLBondedCalculator("/ckpts", "test")            # the reference's own constructor
    assert [m.device for m in calc.models] == ["cpu", "cpu"]
    prot.select_index = torch.as_tensor(plan.select_index)           # what DistanceFragment.fragment leaves on prot
    prot.origin_index = torch.as_tensor(plan.origin_index)
    E_ref, F_ref = calc(prot)                                         # the reference's own __call__
    assert len(calls) == len(DS.get_work_partitions()) >= 4 and {c[0] for c in calls} == {"cpu"}
    # the mirror class on the same seam objects, fragmenter and partitions
    mirror = Mirror.from_models([OracleSeam("a"), OracleSeam("b")], chunk_atoms=120, fragment_method=Fragmenter())
    prot.select_index, prot.origin_index = plan.select_index, plan.origin_index
    E_m, F_m = mirror(prot)
    assert mirror._work == [tuple(t) for t in DS.get_work_partitions()]
    np.testing.assert_allclose(np.asarray(F_ref), F_m, rtol=0, atol=1e-6)
    assert abs(float(E_ref) - float(E_m)) < 1e-4
    # and both equal the straight evaluation of the whole batch + host recombination
    fd = Fragmenter().get_fragments(prot)
    E_all, F_all, _ = oracle.energy_forces(fd.z, fd.pos, fd.start, fd.end)
    E_h, F_h = combine_host(plan, E_all.reshape(-1, 1), F_all)
    np.testing.assert_allclose(F_m, F_h, rtol=0, atol=2e-5)
    assert abs(float(E_m) - E_h) < 2e-4 * max(1.0, abs(E_h))


def test_hparams_are_recovered_from_a_reference_module():
    """`ViSNetModel(model, device)` accepts a torch module built by the reference's own create_model: the
    hyper-parameters are read off the attributes the reference's ViSNetBlock keeps (visnet_block.py:40-55) and the
    module's state_dict carries the same keys our engine loads."""
    from ai2bmd_amd.visnet_calculator import hparams_of_module
    from oracle.ref_import import import_reference_create_model

    create_model = import_reference_create_model()
    for over in (dict(embedding_dimension=64, num_layers=2),
                 dict(embedding_dimension=128, num_layers=1, lmax=1, rbf_type="gauss", num_rbf=20, activation="ssp",
                      attn_activation="tanh", vecnorm_type="rms", num_heads=4, cutoff=4.5, max_num_neighbors=20)):
        hp = default_hparams(**over)
        model = create_model(hp)
        got = hparams_of_module(model)
        for k in ("embedding_dimension", "num_layers", "num_rbf", "num_heads", "lmax", "max_z", "cutoff",
                  "max_num_neighbors", "vecnorm_type", "rbf_type", "prior_model"):
            assert got[k] == hp[k], (k, got[k], hp[k])
        same = {"silu": ("silu", "swish"), "swish": ("silu", "swish")}
        assert got["activation"] in same.get(hp["activation"], (hp["activation"],))
        assert got["attn_activation"] in same.get(hp["attn_activation"], (hp["attn_activation"],))
        ours = set(make_state_dict(hp, seed=1).keys())
        theirs = set(model.state_dict().keys())
        assert ours == theirs, ours ^ theirs


def test_compiled_reference_recipe_gives_the_source_results_bit_for_bit():
    """oracle/make_ref.py (run by __graft_entry__.build()): the reference's model package byte-compiled into the
    git-ignored oracle/_ref/ is what bench.py's cpu_baseline leg imports on the GPU box, where /root/reference does not
    exist.  Same interpreter, same code objects: energies and forces equal the source tree's to the last bit, and no
    .py file of the reference lands in the repository."""
    import os
    import subprocess
    import sys

    from oracle import make_ref
    from oracle.ref_import import COMPILED_REF, compiled_reference_available

    out = make_ref.build()
    assert out == COMPILED_REF and compiled_reference_available()
    for d, _, files in os.walk(COMPILED_REF):
        assert not [f for f in files if f.endswith(".py")], "reference sources must not be copied"
    # a fresh interpreter per origin: the two packages share the module name `ViSNet`
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle.ref_import import import_reference_create_model
from oracle.weights import default_hparams, make_state_dict
from oracle.inputs import random_fragments
create_model = import_reference_create_model(prefer=sys.argv[1])
import ViSNet
assert (sys.argv[1] == "compiled") == ("_ref" in ViSNet.__path__[0]), ViSNet.__path__
hp = default_hparams(embedding_dimension=64, num_layers=2)
m = create_model(hp)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in make_state_dict(hp, seed=3).items()})
m = m.float().eval()
z, pos, start, end = random_fragments(8, [22, 12, 30])
batch = np.repeat(np.arange(3), [22, 12, 30])
E, F = m(dict(z=torch.as_tensor(z), pos=torch.as_tensor(pos), batch=torch.as_tensor(batch)))
np.save(sys.argv[2], np.concatenate([E.detach().numpy().ravel(), F.detach().numpy().ravel()]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    res = {}
    with tempfile.TemporaryDirectory() as td:
        for origin in ("source", "compiled"):
            f = os.path.join(td, origin + ".npy")
            subprocess.run([sys.executable, "-c", code, origin, f], check=True, timeout=300)
            res[origin] = np.load(f)
    assert np.array_equal(res["source"], res["compiled"])
