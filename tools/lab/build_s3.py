"""Lab: a variant library whose GROUPED GEMM launches (the per-layer products of a single-protein step) run as 3 x bf16
split products (tools/lab/gemm_s3_body.h) when VSN_SPLIT3=1 - NOT a product mode: the product sources are untouched, a
patched COPY of csrc/gemm.hip is compiled into ai2bmd_amd/_ab/libvsn_s3.so (git-ignored) and selected with VSN_LIB.

    python tools/lab/build_s3.py [--single-stage | --b-in-registers]     (-> libvsn_s3.so / _s3sb.so / _s3br.so)
    VSN_LIB=ai2bmd_amd/_ab/libvsn_s3.so VSN_SPLIT3=1 python bench.py --no-secondary --no-cpu-baseline --steps 600
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ai2bmd_amd import build as B  # noqa: E402


def patched_source(single):
    s = open(os.path.join(B.CSRC, "gemm.hip")).read()

    def once(old, new):
        nonlocal s
        assert s.count(old) == 1, old
        s = s.replace(old, new)

    once("namespace vsn {\n", '#include <map>\nnamespace vsn {\n#include "%s"\n' % os.path.join(ROOT, "tools", "lab", "gemm_s3_body.h"))
    if single:  # one 24-KiB stage of six bf16 planes fits in the product's 32 KiB
        pass
    else:
        once("  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 64) * 32];  // 32 KiB\n",
             "  __shared__ __attribute__((aligned(16))) float smem[12288];  // 48 KiB: two stages of six bf16 planes\n")
    once("""  gemm_body<64, 64, 2, 2, true, 0, 32>(d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K,
                                           d.flags, d.ksplit, d.part, b, smem);""",
         """  if (d.flags & VSN_S3_FLAG)
    gemm_body3(d.A, d.lda, reinterpret_cast<const unsigned short*>(d.Bt), d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc,
               d.K, d.flags, d.ksplit, d.part, b, smem);
  else
    gemm_body<64, 64, 2, 2, true, 0, 32>(d.A, d.lda, d.Bt, d.ldb, d.C, d.ldc, d.bias, d.M, d.Mptr, d.Nc, d.K,
                                         d.flags, d.ksplit, d.part, b, smem);""")
    once("    GemmDesc d = descs[i];\n    if (d.M <= 0) continue;\n",
         "    GemmDesc d = descs[i];\n    if (d.M <= 0) continue;\n    s3_patch(d, st);\n")
    return s


def main():
    B.build()  # stock objects up to date
    out = os.path.join(B.HERE, "_ab")
    os.makedirs(out, exist_ok=True)
    single = "--single-stage" in sys.argv
    breg = "--b-in-registers" in sys.argv  # weights packed in fragment order, straight from L2 (24 KiB of LDS: fits)
    tag = "s3br" if breg else "s3sb" if single else "s3"
    single = single or breg
    src = os.path.join(out, f"gemm_{tag}_gen.hip")
    open(src, "w").write(patched_source(single))
    obj = os.path.join(out, f"gemm_{tag}.o")
    subprocess.run([B._hipcc(), *B.FLAGS, *(["-DS3_BREG=1"] if breg else ["-DS3_DB=0"] if single else []), "-I", B.CSRC, "-c", src, "-o", obj], check=True)
    objs = [obj if s_ == "gemm.hip" else os.path.join(B.OBJ, s_.replace(".hip", ".o")) for s_ in B.SOURCES]
    lib = os.path.join(out, f"libvsn_{tag}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
