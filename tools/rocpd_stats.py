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


def busy(path, lo=0.5, hi=1.0):
    """GPU busy fraction over the [lo, hi] fraction of the trace: union of kernel intervals / wall span"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    iv = sorted(cur.execute(f"select start, end from {disp}"))
    t0, t1 = iv[0][0], max(e for _, e in iv)
    a, b = t0 + int((t1 - t0) * lo), t0 + int((t1 - t0) * hi)
    iv = [(s, e) for s, e in iv if a <= s <= b]
    union, summ, cur_s, cur_e = 0, 0, iv[0][0], iv[0][1]
    for s, e in iv:
        summ += e - s
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = max(e for _, e in iv) - iv[0][0]
    print(f"dispatches {len(iv)} span_ms {span / 1e6:.3f} busy_union_ms {union / 1e6:.3f} "
          f"sum_kernel_ms {summ / 1e6:.3f} busy_frac {union / span:.3f}")


def gaps(path, lo=0.3, hi=0.6):
    """idle gaps between kernels (start - latest end so far) inside the [lo, hi] fraction of the trace"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    iv = sorted(cur.execute(f"select start, end from {disp}"))
    t0, t1 = iv[0][0], max(e for _, e in iv)
    a, b = t0 + int((t1 - t0) * lo), t0 + int((t1 - t0) * hi)
    iv = [(s, e) for s, e in iv if a <= s <= b]
    g, cur_e = [], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            g.append(s - cur_e)
        cur_e = max(cur_e, e)
    g.sort()
    n = len(g)
    print(f"kernels {len(iv)} gaps {n} total_gap_ms {sum(g) / 1e6:.3f} span_ms {(iv[-1][1] - iv[0][0]) / 1e6:.3f}")
    for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99):
        print(f"  p{int(q * 100):02d} {g[int(q * (n - 1))] / 1e3:.2f} us")


def timeline(path, anchor="k_build_fragments", which=-12):
    """one MD step as a kernel sequence: offset of every dispatch from the step's first kernel (`anchor`),
    duration, queue (stream) and grid - the `which`-th occurrence of the anchor (negative = from the end; the
    default skips bench.py's closing instrumented pass, whose five steps carry a HIP event pair per GEMM launch)"""
    which = int(which)
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    dcols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    gx = "grid_size_x" if "grid_size_x" in dcols else "grid_x"
    qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else None)
    q = (f"select d.start, d.end, s.{name_col}, d.{gx}, {('d.' + qcol) if qcol else '0'} from {disp} d "
         f"join {sym} s on d.kernel_id = s.id order by d.start")
    rows = list(cur.execute(q))
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    a, b = idx[which], idx[which + 1]
    t0 = rows[a][0]
    print("offset_us,dur_us,gap_before_us,queue,grid_x,kernel")
    prev_end = t0
    for st, en, n, g, qd in rows[a:b]:
        short = n.split("(")[0].replace("_ZN3vsn", "").replace(".kd", "")[:48]
        print(f"{(st - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{(st - prev_end) / 1e3:.1f},{qd},{g},{short}")
        prev_end = max(prev_end, en)
    print(f"# step span {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} dispatches")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--timeline":
        timeline(sys.argv[1], *sys.argv[3:5])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1], *[float(v) for v in sys.argv[3:5]])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--busy":
        busy(sys.argv[1], *[float(v) for v in sys.argv[3:5]])
        sys.exit(0)
    if len(sys.argv) > 2:
        by_grid(sys.argv[1], sys.argv[2])
    else:
        main(sys.argv[1])
