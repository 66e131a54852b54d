#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` sqlite result (rocpd .db) as a kernel-stats table.

usage: python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt
Same columns as rocprofv3's kernel_stats.csv: name, calls, total ns, average ns, %, min, max.
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            f"from kernels group by {name_col} order by sum(end-start) desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for n, c, tot, avg, mn, mx in rows:
        print(f"  {c:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * tot / total:6.2f}  {n}")
    print(f"# total kernel time {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
