#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.
    python tools/pmc_summary.py <dir-with-b_counter_collection.csv> [...]
FETCH_SIZE / WRITE_SIZE are in KB (64-byte requests / 1024); on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) - the caller applies that correction."""
import collections
import csv
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


if __name__ == "__main__":
    out = collections.defaultdict(dict)
    for d in sys.argv[1:]:
        acc, dur = load(f"{d}/b_counter_collection.csv")
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
            out[k]["calls"] = len(dur[k])
            out[k]["avg_us"] = sum(dur[k]) / len(dur[k]) / 1e3
    # derived: HBM bytes (2*FETCH_SIZE + WRITE_SIZE, KB; gfx950 reports half of wide coalesced reads) and GB/s,
    # MFMA utilisation = MFMA busy cycles / (kernel cycles * 1024 SIMDs), kernel cycles = SQ_BUSY_CYCLES / 32 SEs
    for v in out.values():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_MB"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6
            v["hbm_GBps"] = v["hbm_MB"] / 1e3 / (v["avg_us"] * 1e-6)
        if v.get("SQ_BUSY_CYCLES"):
            v["clk_GHz"] = v["SQ_BUSY_CYCLES"] / 32 / (v["avg_us"] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                v["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["SQ_BUSY_CYCLES"] / 32 * 1024)
    names = sorted({c for v in out.values() for c in v})
    print("kernel," + ",".join(names))
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
        print(k.replace(",", ";") + "," + ",".join(f"{v.get(c, float('nan')):.4g}" for c in names))
