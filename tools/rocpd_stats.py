#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into a per-kernel stats table
(the `--stats` view): calls, total/avg/min/max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         f"max(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} "
         f"order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for n, c, t, a, mn, mx in rows:
        n = n.replace(",", ";")
        print(f"\"{n}\",{c},{t},{a:.0f},{mn},{mx},{100.0 * t / tot:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
