#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd SQLite) outputs into the text tables committed under profiles/.
usage: rocpd_summary.py <kernel_trace.db> [<pmc.db> ...]"""
import sqlite3
import sys


def kernel_stats(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = [f"# kernel trace: {path}", f"{'kernel':60s} {'calls':>7s} {'total_ms':>12s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s}"]
    for name, n, t, a, mn, mx in rows:
        out.append(f"{name[:60]:60s} {n:7d} {t / 1e6:12.3f} {a / 1e6:10.4f} {mn / 1e6:10.4f} {mx / 1e6:10.4f} {100.0 * t / tot:6.2f}")
    extra = [c for c in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_size", "grid_size") if c in cols]
    if extra:
        out.append("")
        out.append("# per-kernel resources (first dispatch): " + ", ".join(extra))
        for name, *_ in rows:
            r = cur.execute(f"select {', '.join(extra)} from kernels where name = ? limit 1", (name,)).fetchone()
            out.append(f"{name[:60]:60s} " + " ".join(f"{c}={v}" for c, v in zip(extra, r)))
    return "\n".join(out)


def pmc_stats(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    out = [f"# pmc: {path}  (columns: {cols})"]
    try:
        rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                           "group by kernel_name, counter_name order by 4 desc").fetchall()
        out.append(f"{'kernel':50s} {'counter':14s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s}")
        for k, c, n, s, a in rows:
            out.append(f"{k[:50]:50s} {c:14s} {n:10d} {s:16.1f} {a:16.1f}")
    except sqlite3.Error as e:
        out.append(f"(could not aggregate: {e})")
    return "\n".join(out)


if __name__ == "__main__":
    print(kernel_stats(sys.argv[1]))
    for p in sys.argv[2:]:
        print()
        print(pmc_stats(p))
