"""CPU: fragmentation plan, combine bookkeeping, and the N>1 sharded path over a
world_size-2 gloo group (stub force function: no GPU needed)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def load_protein(name):
    from ai2bmd_amd.fragmentation import ProteinAtoms

    d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
    return ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))


@pytest.mark.parametrize("name,B,N", [("chig", 19, 391), ("trpcage", 39, 737), ("ww", 69, 1387), ("abd", 93, 1850)])
def test_plan_sizes_match_survey(name, B, N):
    from ai2bmd_amd.fragmentation import build_plan, fragment_positions

    prot = load_protein(name)
    plan = build_plan(prot)
    assert len(plan.start) == B and len(plan.z) == N
    sizes = plan.end - plan.start
    assert (sizes[1::2] == 12).all() and sizes[0::2].min() >= 19 and sizes[0::2].max() <= 36
    assert plan.is_dipeptide[0::2].all() and not plan.is_dipeptide[1::2].any()
    pos = fragment_positions(plan, prot.positions)
    cap = plan.src < 0
    d = np.linalg.norm(pos[cap] - prot.positions[plan.acceptor[cap]], axis=1)
    np.testing.assert_allclose(d, plan.length[cap], atol=1e-6)
    assert set(np.round(plan.length[cap].astype(np.float64), 2).tolist()) <= {1.07, 1.02}
    # every protein atom is covered: dipeptide copies minus ACE-NME copies == 1
    cover = np.zeros(plan.n_prot)
    sign = np.where(np.arange(len(plan.row_of_cat)) < plan.n_dip_rows, 1.0, -1.0)
    np.add.at(cover, plan.origin_index, sign[plan.select_index])
    assert (cover == 1).all()


def test_combine_is_linear_and_matches_reference_convention():
    from ai2bmd_amd.fragmentation import build_plan, combine_host

    prot = load_protein("chig")
    plan = build_plan(prot)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((len(plan.z), 3))
    e = rng.standard_normal(len(plan.start))
    E, F = combine_host(plan, e, f)
    # reference convention: cat[F_dip, -F_ace][select] scattered to origin
    vd = np.zeros(len(plan.z), bool)
    for b in range(0, len(plan.start), 2):
        vd[plan.start[b]:plan.end[b]] = True
    cat = np.concatenate([f[vd], -f[~vd]])[plan.select_index]
    F2 = np.zeros_like(F)
    np.add.at(F2, plan.origin_index, cat)
    np.testing.assert_allclose(F, F2, atol=1e-12)
    assert abs(E - (e[0::2].sum() - e[1::2].sum())) < 1e-12
    # a uniform force field on every fragment copy telescopes to the same field on the protein
    _, Fc = combine_host(plan, e, np.ones((len(plan.z), 3)))
    np.testing.assert_allclose(Fc, 1.0)


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, fragment_positions, combine_host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
which = sys.argv[2] if len(sys.argv) > 2 else "ww"
if which == "mini5":  # ACE-TYR-TYR-ASP-NME cut out of Chignolin: 5 fragments, fewer than an 8-rank job has ranks
    d = np.load(os.path.join(sys.argv[1], "tests", "golden", "protein_chig.npz"))
    m = (d["resnums"] <= 4) | (d["resnums"] == 12)
    rn = d["resnums"][m].copy()
    rn[rn == 12] = 5
    prot = ProteinAtoms(d["names"][m], d["resnames"][m], rn, d["numbers"][m], d["positions"][m].astype(np.float64))
else:
    d = np.load(os.path.join(sys.argv[1], "tests", "golden", f"protein_{which}.npz"))
    prot = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))
plan = build_plan(prot)
ff = ShardedFragmentForces(plan, rank, world, "cpu")
lo, hi = ff.atom_lo[rank], ff.atom_hi[rank]
def stub_force(pos):  # deterministic per-row function standing in for vsn_forces
    return np.stack([np.sin(pos[:, 0]) + pos[:, 1], pos[:, 2] ** 2, pos[:, 0] * pos[:, 1]], 1)
def local_fn(prot_pos):
    full = fragment_positions(plan, prot_pos.numpy().astype(np.float64))
    f = stub_force(full[lo:hi])
    e = np.array([full[s:e_].sum() for s, e_ in zip(plan.start[ff.f0:ff.f1], plan.end[ff.f0:ff.f1])])
    return torch.as_tensor(e, dtype=torch.float32), torch.as_tensor(f, dtype=torch.float32)
offs = ff.gathered_force_rows()
def combine_fn(buf):
    f_all = buf.numpy()[(offs[:, None] + np.arange(3)[None, :])]
    return torch.as_tensor(combine_host(plan, np.zeros(len(plan.start)), f_all)[1])
ff.local_fn, ff.combine_fn = local_fn, combine_fn
x = torch.as_tensor(prot.positions, dtype=torch.float32)
E, F = ff.step(x)
full = fragment_positions(plan, prot.positions.astype(np.float32).astype(np.float64))
e_ref = np.array([full[s:e_].sum() for s, e_ in zip(plan.start, plan.end)])
E_ref, F_ref = combine_host(plan, e_ref, stub_force(full))
assert abs(float(E) - E_ref) < 1e-2 * max(1, abs(E_ref)), (float(E), E_ref)
assert np.abs(F.numpy() - F_ref).max() < 1e-3, np.abs(F.numpy() - F_ref).max()
cover = sum(ff.rows)
assert cover == len(plan.z) and ff.ranges[0][0] == 0 and ff.ranges[-1][1] == len(plan.start)
print(f"rank {rank} ok rows={ff.local_rows} frags={ff.f1 - ff.f0}")
dist.destroy_process_group()
'''


WORKER_FOR_ENGINE = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan, fragment_positions, combine_host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
which = sys.argv[2] if len(sys.argv) > 2 else "ww"
if which == "mini5":  # ACE-TYR-TYR-ASP-NME cut out of Chignolin: 5 fragments, fewer than an 8-rank job has ranks
    d = np.load(os.path.join(sys.argv[1], "tests", "golden", "protein_chig.npz"))
    m = (d["resnums"] <= 4) | (d["resnums"] == 12)
    rn = d["resnums"][m].copy()
    rn[rn == 12] = 5
    prot = ProteinAtoms(d["names"][m], d["resnames"][m], rn, d["numbers"][m], d["positions"][m].astype(np.float64))
else:
    d = np.load(os.path.join(sys.argv[1], "tests", "golden", f"protein_{which}.npz"))
    prot = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))
plan = build_plan(prot)

def stub_force(pos):  # deterministic per-row function standing in for the network
    return torch.stack([torch.sin(pos[:, 0]) + pos[:, 1], pos[:, 2] ** 2, pos[:, 0] * pos[:, 1]], 1)

class FakeEngine:
    """stands in for ViSNetEngine: same call, writes INTO the views it is handed (that aliasing is the test)"""
    device, index = "cpu", 0
    calls = 0
    def forces_device(self, z, pos, start, end, e_out, f_out, stream=None):
        assert len(z) == len(pos) == len(f_out) and len(start) == len(end) == len(e_out)
        assert int(start[0]) == 0 and int(end[-1]) == len(z)          # rebased to the shard
        f_out.copy_(stub_force(pos))
        for b, (s, e_) in enumerate(zip(start, end)):
            e_out[b] = pos[int(s):int(e_)].sum()
        self.calls += 1

class TorchTail:
    """stands in for _HipTail (vsn_build_fragments / vsn_combine_with_energy) with the same plan arguments"""
    def stream(self):
        return None
    def fragplan(self, src, acc, tow, ln):
        return tuple(torch.as_tensor(np.asarray(a)) for a in (src, acc, tow, ln))
    def build(self, fp, prot_pos, pos_geo, st):
        src, acc, tow, ln = fp
        m = src >= 0
        out = torch.zeros(len(src), 3)
        out[m] = prot_pos[src[m]]
        a = prot_pos[acc[~m]]
        v = prot_pos[tow[~m]] - a
        out[~m] = a + v / v.norm(dim=1, keepdim=True) * ln[~m][:, None]
        pos_geo[: len(src)] = out
    def combine_plan(self, n_prot, row_of_cat, n_dip_rows, select, origin, e_idx, e_sgn):
        sel = np.asarray(select)
        return dict(rows=torch.as_tensor(np.asarray(row_of_cat)[sel]), origin=torch.as_tensor(np.asarray(origin)),
                    sign=torch.as_tensor(np.where(sel < n_dip_rows, 1.0, -1.0).astype(np.float32)),
                    e_idx=torch.as_tensor(e_idx), e_sgn=torch.as_tensor(e_sgn))
    def combine(self, cp, buf, F_prot, st):
        F_prot.zero_()
        F_prot.index_add_(0, cp["origin"], buf.view(-1, 3)[cp["rows"]] * cp["sign"][:, None])
    def combine_with_energy(self, cp, buf, F_prot, E_tot, st):
        self.combine(cp, buf, F_prot, st)
        E_tot[0] = (buf[cp["e_idx"]] * cp["e_sgn"]).sum()

eng = FakeEngine()
ff = ShardedFragmentForces.for_engine(eng, plan, rank=rank, world=world, tail=TorchTail())
assert ff.direct and ff.fused_tail is None
x = torch.as_tensor(prot.positions, dtype=torch.float32)
for it in range(2):  # second call: the exchange buffers are reused
    xs = x + 0.01 * it
    E, F = ff.step(xs)
    full = fragment_positions(plan, xs.numpy().astype(np.float64))
    e_ref = np.array([full[s:e_].sum() for s, e_ in zip(plan.start, plan.end)])
    E_ref, F_ref = combine_host(plan, e_ref[(plan.end - plan.start) > 0], stub_force(torch.as_tensor(full)).numpy())
    assert abs(float(E) - E_ref) < 1e-2 * max(1, abs(E_ref)), (float(E), E_ref)
    assert np.abs(F.numpy() - F_ref).max() < 1e-3, np.abs(F.numpy() - F_ref).max()
lo, hi = ff.atom_lo[rank], ff.atom_hi[rank]
# a rank that owns no fragment (fewer fragments than ranks) never calls the engine but DID enter both collectives
assert eng.calls == (2 if hi > lo else 0) and (ff.f1 - ff.f0 > 0) == (hi > lo)
# this rank's slot of the gathered buffer IS what the engine wrote through the views; every other rank's slot arrived
mine = ff.recv[rank * ff.slot: rank * ff.slot + (hi - lo) * 3].view(-1, 3)
assert torch.equal(mine, stub_force(ff.frag_pos[: hi - lo]))
for other in range(world):
    if other != rank and ff.rows[other]:
        assert ff.recv[other * ff.slot: other * ff.slot + ff.rows[other] * 3].abs().sum() > 0
assert sum(ff.rows) == len(plan.z) and sum(ff.nfrag) == len(plan.start) and ff.slot % 3 == 0
assert ff.slot >= 3 * max(ff.rows) + max(ff.nfrag)
# without the relaxation a rank gathers only ITS rows of the fragment geometry
assert ff.frag_pos.shape[0] == max(hi - lo, 1)
print(f"rank {rank} for_engine ok rows={ff.local_rows} frags={ff.f1 - ff.f0} slot={ff.slot}")
dist.destroy_process_group()
'''


def test_for_engine_wiring_world2_gloo(tmp_path):
    """The PRODUCT wiring of the N > 1 path - `ShardedFragmentForces.for_engine`: shard-rebased fragment offsets, the
    engine writing through views into the exchange buffer, one all-gather, the combine plan remapped onto the padded
    gathered buffer - run by two gloo ranks with stand-ins for the engine and the two HIP ends (same arguments, torch
    arithmetic).  RCCL differs from this only in the backend string."""
    script = tmp_path / "worker_fe.py"
    script.write_text(WORKER_FOR_ENGINE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 for_engine ok" in r.stdout and "rank 1 for_engine ok" in r.stdout


@pytest.mark.parametrize("which,port", [("ww", 29547), ("mini5", 29549)])
def test_for_engine_wiring_world8_gloo(tmp_path, which, port):
    """The same product wiring at the size the driver's scaling run uses: EIGHT ranks - the WW domain (69 fragments:
    8 or 9 per rank) and a 5-fragment peptide, where three ranks own NOTHING (`nloc == 0`: no engine call, zero-row
    views into the exchange buffer, `max_rows` padding from the largest rank) and still enter the all-gather and
    produce the same recombined forces."""
    script = tmp_path / "worker_fe8.py"
    script.write_text(WORKER_FOR_ENGINE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), ROOT, which]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(8):
        assert f"rank {k} for_engine ok" in r.stdout
    if which == "mini5":
        assert r.stdout.count("rows=0 frags=0") == 3


def test_single_rank_never_enters_a_collective(monkeypatch):
    """SCALE's N = 1 point must be BENCH's code path: with world == 1 the evaluator writes into the buffer the combine
    reads and `torch.distributed` is never touched (it is not even initialised in bench.py's Ctx)."""
    import torch
    import torch.distributed as dist

    from ai2bmd_amd.bonded import ShardedFragmentForces
    from ai2bmd_amd.fragmentation import build_plan

    def boom(*a, **k):
        raise AssertionError("collective entered at world == 1")

    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "barrier", "broadcast"):
        monkeypatch.setattr(dist, name, boom)
    plan = build_plan(load_protein("chig"))
    ff = ShardedFragmentForces(plan, rank=0, world=1, device="cpu")
    assert ff.ranges == [(0, len(plan.start))] and ff.local_rows == len(plan.z)
    n = len(plan.z)
    ff.local_fn = lambda pos: (torch.arange(len(plan.start), dtype=torch.float32), torch.ones(n, 3))
    ff.combine_fn = lambda buf: buf[: n * 3].view(-1, 3).clone()
    E, F = ff.step(torch.zeros(plan.n_prot, 3))
    assert F.shape == (n, 3) and float(F.sum()) == 3 * n and torch.isfinite(E)
    # bench.py: a single process builds no process group at all
    import bench

    a = bench.parse_args(["--stub"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    ctx = bench.Ctx(a)
    assert ctx.world == 1 and ctx.backend is None and not dist.is_initialized()
    ctx.barrier()
    assert ctx.max_over_ranks(1.5) == 1.5


def test_sharded_path_world2_gloo(tmp_path, lib_built):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def _row_permutation(plan, prot_pos, ref):
    """row of OUR batch for every row of the reference's batch (AMBER order there, residue order here):
    matched inside each fragment by (atomic number, position)."""
    from ai2bmd_amd.fragmentation import fragment_positions

    mine = fragment_positions(plan, prot_pos).astype(np.float32)
    perm = -np.ones(len(ref["z"]), dtype=np.int64)
    for b in range(len(plan.start)):
        a0, a1 = int(plan.start[b]), int(plan.end[b])
        assert (a0, a1) == (int(ref["start"][b]), int(ref["end"][b]))
        used = set()
        for r in range(a0, a1):
            d = np.abs(mine[a0:a1] - ref["pos"][r]).max(axis=1)
            d[plan.z[a0:a1] != ref["z"][r]] = np.inf
            d[list(used)] = np.inf  # an atom shared by the two halves of a CYX pair appears twice
            k = int(np.argmin(d))
            assert d[k] < 2e-4, (b, r, d[k])
            used.add(k)
            perm[r] = a0 + k
    return perm


def load_for_reference(name):
    """the protein in the atom order the reference's fragmenter was run on (oracle/make_fragmenter_golden.py): the
    pre-processed Chignolin example as it is, the other examples through preprocessed_order"""
    from ai2bmd_amd.fragmentation import preprocessed_order

    prot = load_protein(name)
    return prot if name.startswith("chig") else preprocessed_order(prot)


def test_preprocessed_order_reproduces_the_reference_example():
    from ai2bmd_amd.fragmentation import preprocessed_order

    prot = load_protein("chig")  # = examples/chig_preprocessed/chig-preeq-nowat.pdb
    again = preprocessed_order(prot)
    assert np.array_equal(again.names, prot.names) and np.array_equal(again.positions, prot.positions)
    shuffled = load_protein("trpcage")  # tleap order: a different one
    assert not np.array_equal(preprocessed_order(shuffled).names, shuffled.names)


@pytest.mark.parametrize("name", ["chig", "chigcyx", "trpcage", "ww", "abd"])
def test_plan_matches_reference_fragmenter(name):
    """golden = the reference's own DistanceFragment.fragment + get_dipeptide_positions (oracle/ref_fragmenter.py).
    The DEFAULT plan is the reference's fragment batch row for row, no permutation in between: atomic numbers,
    fragment ranges, cap-hydrogen first-guess positions, select / origin indices - hence the same FragmentData and,
    when `max_num_neighbors` truncates, the same lowest-index sources kept per target.
    "chigcyx" = Chignolin with a fabricated CYX-CYX bridge (oracle/make_fragmenter_golden.py): the two dipeptides are
    merged into one 44-atom fragment and the second slot stays empty."""
    from ai2bmd_amd.fragment import FragmentData, make_batch_index
    from ai2bmd_amd.fragmentation import build_plan, combine_host, fragment_positions

    prot = load_for_reference(name)
    plan = build_plan(prot)
    ref = np.load(os.path.join(GOLDEN, f"fragref_{name}.npz"))
    fd = FragmentData(plan.z, fragment_positions(plan, prot.positions).astype(np.float32), plan.start, plan.end,
                      make_batch_index(plan.start, plan.end))
    assert np.array_equal(fd.z, ref["z"])
    assert np.array_equal(fd.start, ref["start"]) and np.array_equal(fd.end, ref["end"])
    np.testing.assert_allclose(fd.pos, ref["pos"], atol=2e-5)
    assert plan.n_dip_rows == int(ref["n_dip_rows"])
    assert np.array_equal(plan.select_index, ref["select_index"])
    assert np.array_equal(plan.origin_index, ref["origin_index"])
    if name == "chigcyx":
        sizes = plan.end - plan.start
        assert sizes[2] == 44 and sizes[10] == 0 and plan.cyx_partner[1] == 5 and plan.cyx_partner[5] == -2
    # the protein may come in ANY atom order (atoms are matched to template slots by name): same batch
    shuffled = build_plan(load_protein(name))
    assert np.array_equal(shuffled.z, plan.z) and np.array_equal(shuffled.start, plan.start)
    # forces: random per-row forces recombined by the reference's rule (combiner.py:24-41 with the reference's
    # select / origin indices) == combine_host on the plan
    rng = np.random.default_rng(0)
    f_ref = rng.standard_normal((len(ref["z"]), 3))
    is_dip = np.zeros(len(ref["z"]), bool)
    for b in range(0, len(ref["start"]), 2):
        is_dip[ref["start"][b]:ref["end"][b]] = True
    cat = np.concatenate([f_ref[is_dip], -f_ref[~is_dip]])[ref["select_index"]]
    F_ref = np.zeros((plan.n_prot, 3))
    np.add.at(F_ref, ref["origin_index"], cat)
    n_nonempty = int(((plan.end - plan.start) > 0).sum())  # energies exist for non-empty fragments only
    _, F_mine = combine_host(plan, np.zeros((n_nonempty, 1), np.float32), f_ref.astype(np.float32))
    np.testing.assert_allclose(F_mine, F_ref, atol=1e-5)


@pytest.mark.parametrize("name", ["chig", "chigcyx", "trpcage", "ww", "abd"])
def test_grouped_layout_is_a_row_permutation_of_the_default(name):
    """`order="grouped"` ([previous part | residue | next part], the layout the AMBER order is derived from) holds the
    same atoms at the same places in every fragment and recombines to the same protein forces; building the hydrogen
    plan from either gives the same terms."""
    from ai2bmd_amd.amber import default_tables
    from ai2bmd_amd.fragmentation import build_plan, combine_host, fragment_positions
    from ai2bmd_amd.hydrogen import amber_ordered, build_hydrogen_plan

    prot = load_protein(name)
    plan, grouped = build_plan(prot), build_plan(prot, order="grouped")
    assert grouped.tmpl_slot is None and plan.tmpl_slot is not None
    ref = dict(z=plan.z, pos=fragment_positions(plan, prot.positions).astype(np.float32), start=plan.start, end=plan.end)
    perm = _row_permutation(grouped, prot.positions, ref)   # grouped row of every default row
    assert sorted(perm.tolist()) == list(range(len(plan.z)))
    rng = np.random.default_rng(1)
    f = rng.standard_normal((len(plan.z), 3)).astype(np.float32)
    f_g = np.zeros_like(f)
    f_g[perm] = f
    n_nonempty = int(((plan.end - plan.start) > 0).sum())
    e = np.zeros((n_nonempty, 1), np.float32)
    np.testing.assert_allclose(combine_host(plan, e, f)[1], combine_host(grouped, e, f_g)[1], atol=1e-6)
    again = amber_ordered(prot, plan, default_tables())
    assert again is plan  # already ordered
    h1, h2 = build_hydrogen_plan(prot, plan, default_tables()), build_hydrogen_plan(prot, grouped, default_tables())
    assert len(h1.cap_rows) == len(h2.cap_rows) and len(h1.pair["i"]) == len(h2.pair["i"])
    assert all(np.array_equal(t, np.arange(len(t))) for t in h1.tmpl_index)
    assert np.array_equal(np.sort(perm[h1.cap_rows]), np.sort(h2.cap_rows))
    for tab in ("bond", "angle", "dihedral"):
        a, b = getattr(h1, tab), getattr(h2, tab)
        assert np.array_equal(perm[a["i"]], b["i"]) and np.allclose(a["kf"], b["kf"])


def test_edge_balanced_device_ranges_cover_and_balance():
    """device_ranges(balance="cost"): the reference's cutting rule on cumulative edge counts - contiguous, covering,
    and never worse balanced (max edge cost over ranks) than the atom rule on the example proteins."""
    from ai2bmd_amd.device_strategy import device_ranges, fragment_cost
    from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan

    for name in ("chig", "trpcage", "ww", "abd"):
        d = np.load(os.path.join(GOLDEN, f"protein_{name}.npz"))
        plan = build_plan(ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"],
                                       d["positions"].astype(np.float64)))
        cost = fragment_cost(plan.start, plan.end)
        for w in (1, 2, 4, 8):
            worst = {}
            for rule in ("atoms", "cost"):
                r = device_ranges(plan.start, plan.end, w, balance=rule)
                assert r[0][0] == 0 and r[-1][1] == len(plan.start)
                assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
                worst[rule] = max(int(cost[a:b].sum()) for a, b in r)
            assert worst["cost"] <= worst["atoms"] * 1.02, (name, w, worst)
    with pytest.raises(ValueError):
        device_ranges(plan.start, plan.end, 2, balance="nope")
