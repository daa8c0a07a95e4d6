"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc --offload-arch=gfx950 -O3 ... -c x.hip -o x.o -Rpass-analysis=kernel-resource-usage 2> x.ru.txt
    python tools/kernel_resources.py x.ru.txt [substring ...]
"""
import re
import sys


def table(path, filters):
    txt = open(path).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        if filters and not any(f in name for f in filters):
            continue

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        occ, lds = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{name[:78]:78s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} spill={g('VGPRs Spill'):3d} "
              f"sgpr={g('SGPRs'):3d} occ={occ} lds={lds}")


if __name__ == "__main__":
    table(sys.argv[1], sys.argv[2:])
