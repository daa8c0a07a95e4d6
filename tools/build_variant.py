"""A/B builds: recompile some translation units with extra -D flags and link them with the stock objects into
ai2bmd_amd/_ab/libvsn_<tag>.so (git-ignored; travels to the GPU box).  Select at run time with VSN_LIB=<path>.

    python tools/build_variant.py <tag> <file.hip>[,<file.hip>...] -DNAME=VALUE [...]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ai2bmd_amd import build as B  # noqa: E402


def main():
    tag, files, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build()  # stock objects up to date
    out = os.path.join(B.HERE, "_ab")
    os.makedirs(out, exist_ok=True)

    def cc(src):
        obj = os.path.join(out, f"{tag}_{src.replace('.hip', '.o')}")
        subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(src, []), *defs, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(4) as ex:
        new = dict(zip(files, ex.map(cc, files)))
    objs = [new.get(s, os.path.join(B.OBJ, s.replace(".hip", ".o"))) for s in B.SOURCES]
    lib = os.path.join(out, f"libvsn_{tag}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    for o in new.values():
        os.remove(o)
    print(lib)


if __name__ == "__main__":
    main()
