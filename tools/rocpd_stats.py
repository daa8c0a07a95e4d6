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


def by_grid(path, pattern):
    """per (kernel, grid) breakdown for kernels whose name contains `pattern`"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    dcols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    gx = "grid_size_x" if "grid_size_x" in dcols else "grid_x"
    q = (f"select s.{name_col}, d.{gx}, count(*), avg(d.end - d.start), min(d.end - d.start) from {disp} d "
         f"join {sym} s on d.kernel_id = s.id where s.{name_col} like ? group by s.{name_col}, d.{gx} order by 1, 2")
    print("kernel,grid_x,calls,avg_ns,min_ns")
    for n, g, c, a, mn in cur.execute(q, (f"%{pattern}%",)):
        print(f"\"{n[:60]}\",{g},{c},{a:.0f},{mn}")


if __name__ == "__main__":
    if len(sys.argv) > 2:
        by_grid(sys.argv[1], sys.argv[2])
    else:
        main(sys.argv[1])
