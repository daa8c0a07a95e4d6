"""CPU: the C-ABI library loads here (no GPU) and exports every symbol
include/vsn.h declares; host-side logic (FragmentData mirror, work partitions)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(lib_built):
    from ai2bmd_amd import capi

    header = open(os.path.join(ROOT, "include", "vsn.h")).read()
    declared = sorted(set(re.findall(r"\b(vsn_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/vsn.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_shipped_code_objects_hold_no_packed_fp32_valu_instructions(lib_built, tmp_path):
    """`v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32` must not appear in the gfx950 code of the library: on MI355X they
    return wrong halves now and then while a second process uses the GPU (tools/lab/pk_micro.hip, LAB_NOTES section 15;
    ai2bmd_amd/build.py NO_PACKED_FP32).  The GPU-side regression test is in tests/test_gpu_multirank.py; this one
    catches a build whose flags lost the switch, without a GPU."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    lib = shutil.copy(lib_built if isinstance(lib_built, str) else os.path.join(ROOT, "ai2bmd_amd", "libvsn_hip.so"),
                      tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, os.listdir(tmp_path)
    n_inst = 0
    for f in objs:
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        n_inst += dis.count("v_mul_f32") + dis.count("v_fma_f32") + dis.count("v_fmac_f32")
        hits = re.findall(r"v_pk_(?:mul|fma|add)_f32|v_pk_mov_b32", dis)  # (the four instructions of the feature)
        assert not hits, (f, sorted(set(hits)), len(hits))
    assert n_inst > 1000  # (the disassembly really was device code)


def test_create_fails_loudly_without_gpu_or_bad_hparams(lib_built):
    import torch

    from ai2bmd_amd import capi

    L = capi.lib()
    h = C.c_void_p()
    hp = capi.VsnHParams(hidden=100, num_layers=2, num_rbf=32, num_heads=8, lmax=2, max_z=100,
                         max_num_neighbors=32, vecnorm_type=0, has_atomref=0, cutoff=5.0)
    rc = L.vsn_create(C.byref(h), C.byref(hp), 0)
    assert rc != 0 and b"hidden" in L.vsn_last_error(h)
    L.vsn_destroy(h)
    if not torch.cuda.is_available():
        hp.hidden = 64
        rc = L.vsn_create(C.byref(h), C.byref(hp), 0)
        assert rc != 0  # no device: must not pretend to work
        L.vsn_destroy(h)


def ref_partition(devices, start, end, chunk):
    """Python restatement of Calculators/device_strategy.py:84-127 (test oracle)."""
    import bisect

    start, end = list(start), list(end)
    parts = []
    n_blocks, a_end = devices, len(start)
    b_prev = 0
    for i in range(n_blocks):
        block = (end[-1] - start[b_prev]) // (n_blocks - i)
        b_end = bisect.bisect(start, block + start[b_prev])
        b_idx = b_end - 1
        block_end = block + start[b_prev]
        if (block_end - start[b_idx]) < (end[b_idx] - block_end):
            b_end -= 1
        b_end = min(b_end, a_end)
        c_prev = b_prev
        while c_prev != b_end:
            c_end = bisect.bisect(start, chunk + start[c_prev])
            c_idx = c_end - 1
            chunk_end = chunk + start[c_prev]
            if (chunk_end - start[c_idx]) < (end[c_idx] - chunk_end):
                c_end -= 1
            c_end = min(c_end, b_end)
            parts.append((i, c_prev, c_end))
            c_prev = c_end
        b_prev = b_end
    return parts


@pytest.mark.parametrize("ndev,chunk", [(1, 9999), (2, 9999), (4, 120), (8, 9999), (3, 70)])
def test_partition_matches_reference_algorithm(lib_built, ndev, chunk):
    from ai2bmd_amd.device_strategy import work_partitions

    rng = np.random.default_rng(ndev * 100 + chunk)
    sizes = []
    for _ in range(35):  # interleaved dipeptide / ACE-NME like a 37-residue protein (WW domain)
        sizes += [int(rng.integers(19, 37)), 12]
    sizes.append(int(rng.integers(19, 37)))
    end = np.cumsum(sizes)
    start = end - np.asarray(sizes)
    got = work_partitions(start, end, ndev, chunk)
    assert got == ref_partition(ndev, start, end, chunk)
    # cover: every fragment exactly once, in order
    flat = [f for (_, a, b) in got for f in range(a, b)]
    assert flat == list(range(len(sizes)))


def test_fragment_data_mirror():
    from ai2bmd_amd.fragment import FragmentData, make_batch_index

    sizes = [22, 12, 0, 12, 30]  # dip, ace, empty dip (CYZ), ace, dip
    end = np.cumsum(sizes)
    start = end - np.asarray(sizes)
    n = int(end[-1])
    z = np.arange(n)
    pos = np.arange(3 * n, dtype=np.float32).reshape(n, 3)
    fd = FragmentData(z, pos, start, end, make_batch_index(start, end))
    assert len(fd) == 5
    assert fd.batch.max() == 3 and (np.diff(fd.batch) >= 0).all()
    dip, ace = fd.scalar_split()
    assert dip.tolist() == [True, False, False, True] and ace.tolist() == [False, True, True, False]
    vd, va = fd.vector_split()
    assert vd.sum() == 52 and va.sum() == 24 and vd[:22].all() and va[22:46].all() and vd[46:].all()
    sub = fd[1:4]
    assert sub.start.tolist() == [0, 12, 12] and sub.end.tolist() == [12, 12, 24]
    assert sub.z.tolist() == list(range(22, 46)) and sub.batch.min() == 0
    one = fd[4]
    assert len(one) == 1 and one.pos.shape == (30, 3)


def test_plan_entry_points_reject_bad_arguments(lib_built):
    """errno-style codes, no crash: invalid plans are refused before any device work (-22 = EINVAL); on a host
    without a GPU valid plans stop at device selection (-19 = ENODEV) instead of pretending to work."""
    import torch

    from ai2bmd_amd import capi

    L = capi.lib()
    h = C.c_void_p()
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    # combine plan: select / origin out of range
    bad = L.vsn_combine_plan_create(C.byref(h), 0, 4, 6, 4, capi.i64_ptr(i64(range(6))), capi.i64_ptr(i64([0, 9])),
                                    capi.i64_ptr(i64([0, 1])), 2)
    assert bad == -22
    bad = L.vsn_combine_plan_create(C.byref(h), 0, 4, 6, 4, capi.i64_ptr(i64(range(6))), capi.i64_ptr(i64([0, 1])),
                                    capi.i64_ptr(i64([0, 7])), 2)
    assert bad == -22
    # fragment geometry plan: a cap hydrogen whose acceptor equals its direction atom
    src, acc, tow = i64([0, -1]), i64([-1, 3]), i64([-1, 3])
    ln = np.ascontiguousarray([0.0, 1.07], dtype=np.float32)
    assert L.vsn_fragplan_create(C.byref(h), 0, 2, capi.i64_ptr(src), capi.i64_ptr(acc), capi.i64_ptr(tow),
                                 ln.ctypes.data_as(C.POINTER(C.c_float))) == -22
    # hydrogen optimiser: no caps / too many iterations / an occurrence that does not name the cap atom as an end atom
    t = capi.VsnHoptTerms()
    assert L.vsn_hopt_create(C.byref(h), 0, C.byref(t)) == -22
    cap = i64([1])
    occ_ptr = np.ascontiguousarray([0, 1], dtype=np.int32)
    z32 = np.zeros(1, dtype=np.int32)
    one = np.ones(1, dtype=np.float32)
    bi, bj = np.ascontiguousarray([0], np.int32), np.ascontiguousarray([2], np.int32)  # bond 0-2 does not touch row 1
    t.n_rows, t.n_cap, t.cap_rows = 3, 1, cap.ctypes.data_as(C.POINTER(C.c_int64))
    t.n_bond = 1
    t.bond_i, t.bond_j = bi.ctypes.data_as(C.POINTER(C.c_int32)), bj.ctypes.data_as(C.POINTER(C.c_int32))
    t.bond_k, t.bond_r0 = one.ctypes.data_as(C.POINTER(C.c_float)), one.ctypes.data_as(C.POINTER(C.c_float))
    t.occ_ptr = occ_ptr.ctypes.data_as(C.POINTER(C.c_int32))
    t.occ_type = t.occ_term = t.occ_end = z32.ctypes.data_as(C.POINTER(C.c_int32))
    t.occ_w = one.ctypes.data_as(C.POINTER(C.c_float))
    t.max_iter, t.lr, t.tolerance_grad, t.tolerance_change, t.scnb, t.scee = 99, 0.1, 0.1, 0.01, 1.2, 2.0
    assert L.vsn_hopt_create(C.byref(h), 0, C.byref(t)) == -22          # max_iter > 16
    t.max_iter = 10
    rc = L.vsn_hopt_create(C.byref(h), 0, C.byref(t))
    assert rc == (-19 if not torch.cuda.is_available() else -22)          # ENODEV first on a GPU-less host
    # MD / MM plans: non-positive sizes
    assert L.vsn_md_create(C.byref(h), 0, 0, None, C.c_float(1.0), C.c_float(1.0), C.c_float(1.0), 0, C.c_float(0.0),
                           None) == -22
    assert L.vsn_combine_with_energy(None, None, None, None, None) == -22


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launches_its_ranks_and_reports_one_line(world):
    """`python bench.py --gpus N` (no external launcher) must start N ranks itself, time with barrier +
    max-over-ranks and print exactly one JSON line; --stub swaps the GPU step for a sleep on CPU over gloo.
    N = 8 is the driver's scaling run (one rank per GPU of a node)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "5",
                        "--warmup", "1", "--stub"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"] == "gloo"
    assert out["steps_requested"] == 5 and out["steps"] >= 5 and out["data"] == "STUB"
    assert len(lines[0]) < 6000 and r.stdout.strip().splitlines()[-1] == lines[0]   # the result is the LAST stdout line
    # rank k sleeps 2 (k + 1) ms per step: the reported time is the MAX over the ranks
    assert out["ms_per_step"] >= 2.0 * world - 0.1


def test_hparams_of_module_reads_scripted_activation_names():
    """load_model returns torch.jit.script(model) in the reference (visnet.py:92): a scripted sub-module reports its
    class through `original_name`, not through type(...).__name__"""
    from ai2bmd_amd.visnet_calculator import _act_name

    class Scripted:  # what a RecursiveScriptModule exposes
        original_name = "ShiftedSoftplus"

    import torch

    assert _act_name(Scripted()) == "ssp" and _act_name(torch.nn.SiLU()) == "silu" and _act_name(torch.nn.Tanh()) == "tanh"
    with pytest.raises(TypeError):
        _act_name(torch.nn.ReLU())


def test_bench_cpu_baseline_helpers():
    """bench.py's CPU-baseline plumbing without a GPU: the physical core count is read like the reference reads it
    (utils/system.py:28-45: lscpu, sockets x cores per socket), the median protocol keeps >= 3 timed evaluations, and
    the evaluator it times is the REFERENCE's model (kind "reference": the source tree here, oracle/_ref on the GPU
    box) whenever either is present."""
    import subprocess

    import bench
    from oracle.ref_import import reference_model_source
    from oracle.weights import default_hparams, make_state_dict

    n = bench.physical_core_count()
    out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
    if "Core(s) per socket:" in out:
        assert isinstance(n, int) and 1 <= n <= (os.cpu_count() or 1)
    calls = []
    rate, k = bench._median_rate(lambda: calls.append(1), warm=2, timed=5)
    assert k == 5 and len(calls) == 7 and rate > 0
    hp = default_hparams(embedding_dimension=64, num_layers=1)
    kind, origin, fn = bench._cpu_evaluator(hp, make_state_dict(hp, seed=1))
    assert (kind == "reference") == (reference_model_source() is not None)
    from oracle.inputs import random_fragments

    z, pos, start, end = random_fragments(1, [12, 19])
    E, F = fn(z, pos, start, end)
    assert np.asarray(E).reshape(-1).shape == (2,) and np.asarray(F).shape == (len(z), 3)


def test_bench_step_policy_and_defaults():
    """bench.py argument policy: chig_md is BASELINE configs[1] = the 1000-step loop (default steps 1000; a smaller
    --steps is remembered as steps_requested and timed first by run_md), the other workloads time exactly what was
    asked; --min-seconds is off unless asked for."""
    import bench

    a = bench.parse_args([])
    assert a.workload == "chig_md" and a.steps == bench.C2_STEPS == 1000 and a.steps_requested is None
    assert a.min_seconds == 0.0 and a.gpus == 1
    a = bench.parse_args(["--steps", "20", "--warmup", "5"])
    assert a.steps == 20 and a.steps_requested == 20 and a.warmup == 5
    assert bench.parse_args(["--workload", "frag_batch"]).steps == 2
    assert bench.parse_args(["--workload", "trpcage_md"]).steps == 200


def test_compiled_reference_is_refused_for_another_interpreter(tmp_path, monkeypatch):
    """oracle/_ref is bytecode: a manifest written by another interpreter version (different magic) is not used"""
    import json

    from oracle import ref_import

    d = tmp_path / "_ref"
    (d / "ViSNet" / "model").mkdir(parents=True)
    (d / "ViSNet" / "model" / "visnet.pyc").write_bytes(b"x")
    (d / "MANIFEST.json").write_text(json.dumps(dict(magic="00000000", python="0.0", modules={})))
    monkeypatch.setattr(ref_import, "COMPILED_REF", str(d))
    assert not ref_import.compiled_reference_available()
    import importlib.util

    (d / "MANIFEST.json").write_text(json.dumps(dict(magic=importlib.util.MAGIC_NUMBER.hex(), python="x", modules={})))
    assert ref_import.compiled_reference_available()


def test_preprocessed_order_rejects_unknown_atoms():
    from ai2bmd_amd.fragmentation import ProteinAtoms, preprocessed_order

    p = ProteinAtoms(np.array(["CH3", "C", "O", "H1", "H2", "HX"]), np.array(["ACE"] * 6), np.array([1] * 6),
                     np.array([6, 6, 8, 1, 1, 1]), np.zeros((6, 3)))
    with pytest.raises(ValueError):
        preprocessed_order(p)


def _check_result_line(d):
    """what the measurement contract asks of the ONE line (compact form, bench.compact_line)"""
    assert d["metric"] == "MD steps/sec on Chignolin" and d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-4 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    if "c2_loop" in d["config"]:  # `value` on exactly the requested K, the 1000-step loop of configs[1] beside it
        assert d["steps"] == d["steps_requested"] and d["config"]["c2_loop"]["steps"] == 1000
        assert abs(d["config"]["c2_loop"]["value"] - d["value"]) < 0.05 * d["value"]
    else:                         # (records before that: `value` on the 1000-step loop, the requested K beside it)
        assert d["steps"] >= 1000 and d["config"]["requested_run"]["steps"] == d["steps_requested"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["peak"] == 157.3 and 0.3 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > 0)
    h = r["hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0
    assert abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-4
    if "rocprof" in h:  # live dispatch timestamps agree with the committed kernel trace of the same build
        assert 0.9 < h["rocprof"]["live_over_trace"] < 1.1
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["unit"] == "force evaluations/s" and c["physical_cores"] >= 1
    assert c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["reference_layout_evals_per_s"] > 0
    summ = d["config"]["secondary_summary"]
    modes = [k for k in summ if "split3" in k]
    assert modes and all(k.endswith("_optin") for k in modes) and "split3" not in d["metric"]


def test_bench_line_is_compact_and_parseable():
    """the driver could not parse a 23 KB line.  bench.compact_line turns the full record (here: the
    round-4 record, 23 KB, seven nested secondaries) into a single line < 6 KB that json round-trips and still carries
    metric / value / roofline (+ hbm) / cpu_baseline / parity / the secondary summary; the emergency trimming never
    lets it exceed the limit."""
    import json

    import bench

    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full, "gpurun_out/bench_full_chig_md_n1.json")
    line = json.dumps(c, separators=(",", ":"))
    assert len(line) < 6000 and "\n" not in line and json.loads(line) == c
    _check_result_line(c)
    assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["roofline"]["frac"] == pytest.approx(
        full["roofline"]["frac"], rel=1e-5)
    assert set(c["config"]["secondary_summary"]) >= {"trpcage_md", "frag_batch", "frag_stream_pcie", "chig_md_mm",
                                                     "chig_md_h128l6", "chig_md_split3_optin"}
    assert "secondary" not in c and "layouts" not in c["cpu_baseline"] and "all_scatter_kernels" not in c["roofline"]["hbm"]
    # a record blown up far past the limit still yields a line under it
    fat = json.loads(json.dumps(full))
    fat["secondary"] = fat["secondary"] + [dict(s_, metric=f"extra metric number {i} " + "x" * 60) for i, s_ in
                                           enumerate(fat["secondary"] * 6)]
    assert len(json.dumps(bench.compact_line(fat), separators=(",", ":"))) <= bench.LINE_LIMIT
    # non-finite values never reach the line
    bad = json.loads(json.dumps(full))
    bad["roofline"]["traffic"] = float("nan")
    assert bench.compact_line(bad)["roofline"]["traffic"] is None


def test_committed_bench_line_keeps_the_contract():
    """The bench line committed with the round's profiles (profiles/r*_bench_line.json = the LAST stdout line of
    bench.py on the GPU box, exactly what the driver parses) is compact and carries what the measurement contract asks
    for: BASELINE's metric and unit, the 1000-step Chignolin loop, a roofline block for the dominant GEMM with the
    scatter path nested as roofline.hbm, a CPU baseline that timed the REFERENCE's model with the node's physical core
    count, no model keys in config, the split mode only as a labelled opt-in secondary."""
    import glob
    import json

    import bench

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    assert files
    raw = open(files[-1]).read().strip().splitlines()[-1]
    d = json.loads(raw)
    if "secondary" in d:  # a full record of rounds <= 4
        d = bench.compact_line(d)
    else:
        assert len(raw) <= bench.LINE_LIMIT
    _check_result_line(d)


def test_parity_failure_is_agreed_over_the_ranks_before_anybody_raises(monkeypatch):
    """bench.parity_check: a mismatch raises ParityFailure (a SystemExit: fatal for the primary workload, caught per
    secondary in main()).  With a process group the verdict is all-reduced first, so every rank leaves the workload at the
    same point - a rank that passed raises too when another one failed (a lone raise would leave the others waiting in
    their next collective)."""
    import numpy as np

    import bench

    E, F = np.ones(3), np.ones((5, 3))
    ok = bench.parity_check("same", E, F, E, F)
    assert ok["max_dF"] == 0.0
    with pytest.raises(bench.ParityFailure) as ei:
        bench.parity_check("off", E, F + 1e-2, E, F)
    assert isinstance(ei.value, SystemExit) and "PARITY FAILURE" in str(ei.value)

    class TwoRanks:  # the other rank failed: max over ranks of the failure flag is 1
        world = 2

        @staticmethod
        def max_over_ranks(v):
            return max(v, 1.0)

    monkeypatch.setattr(bench, "_CTX", TwoRanks)
    with pytest.raises(bench.ParityFailure, match="another rank"):
        bench.parity_check("mine is fine", E, F, E, F)
    # the in-stream check settles its verdict after the loop: no collective inside the clock
    assert bench.parity_check("local", E, F, E, F, collective=False)["max_dE"] == 0.0
