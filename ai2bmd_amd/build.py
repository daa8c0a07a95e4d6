"""Builds libvsn_hip.so (gfx950 only) in-tree with hipcc.

`python -m ai2bmd_amd.build` or `ai2bmd_amd.build.build()`.  hipcc cross-compiles
without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libvsn_hip.so")
SOURCES = ["gemm.hip", "graph.hip", "layer_fwd.hip", "layer_bwd.hip", "fused.hip", "vecnorm.hip", "head.hip", "head_fused.hip", "md.hip", "mm.hip", "hydrogen.hip", "p2p.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", "pgemm.h", "gemm_s3.h", "tail.h", "md_body.h", os.path.join("..", "..", "include", "vsn.h")]
# md.hip holds every kernel whose per-atom arithmetic is shared between two launches (the step ends stand-alone and
# fused with the integrator halves, csrc/tail.h): contraction is off for the whole file, so bit-identity between those
# launches does not hang on per-function pragmas
# hydrogen.hip launches the first Langevin half + fragment gather in front of its own optimiser (md_body.h, tail.h): the
# same flag as md.hip, so the shared bodies compile to the same arithmetic in both files (measured: with contraction on
# in this file - plain `fast` disregards the pragmas, `fast-honor-pragmas` still left 14 velocities one ulp off - the
# fused start of a step was not bitwise the two launches it replaces)
PER_FILE_FLAGS = {"md.hip": ["-ffp-contract=off"], "hydrogen.hip": ["-ffp-contract=off"]}
# NO packed fp32 VALU instructions in the device code (target feature "packed-fp32-ops" off; the host pass prints an
# "ignoring feature" note).  Measured on MI355X (tools/lab/pk_micro.hip, LAB_NOTES section 15): while a SECOND PROCESS
# runs kernels on the same GPU, `v_pk_mul_f32 d, x, a op_sel:[0,1]` (both halves times a.hi - what the compiler emits
# for "row times one factor") returns a wrong LOW half in lanes 48..63 about 3e-6 of the time - the attention walk's `m`
# rows came out with 16 zeroed channels and a fragment batch differed by 1e-2 eV/A from call to call.  One process per GPU
# never shows it; a library must not depend on that.  Scalar fp32 VALU is the same IEEE arithmetic (the compiler's
# contraction choices move the last ulp: smoke |dF| against fp64 1.18e-6 -> 1.22e-6) and costs nothing here: the walks
# wait for memory and the products run on MFMA (Chignolin 450.4 -> 454.9 steps/s, alternating runs on one box).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", *NO_PACKED_FP32, "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()

    def unit_digest(src):  # one translation unit: its source, every header, its flags
        h = hashlib.sha256()
        for f in [src] + HEADERS:
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(FLAGS + PER_FILE_FLAGS.get(src, [])).encode())
        return h.hexdigest()

    def cc(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        ostamp, od = obj + ".stamp", unit_digest(src)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == od:
            return obj  # unchanged since it was compiled: only the edited units are rebuilt
        cmd = [hipcc, *FLAGS, *PER_FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(ostamp, "w") as fh:
            fh.write(od)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
