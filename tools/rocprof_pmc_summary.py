#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 `--pmc ...` sqlite result (rocpd .db).

usage: python tools/rocprof_pmc_summary.py <results.db> [name-filter]
Counter instances of one dispatch (per XCD / SE) are summed; the table shows the mean per launch.
"""
import collections
import sqlite3
import sys


def main(path, flt=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (name, dispatch) -> counter -> sum
    dur = {}
    for name, disp, cname, val, d in cur.execute(
            "select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
        if flt and flt not in name:
            continue
        per[(name, disp)][cname] += val
        dur[(name, disp)] = d
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (name, disp), cs in per.items():
        for c, v in cs.items():
            agg[name][c].append(v)
        agg[name]["__dur_ns"].append(dur[(name, disp)])
    print(f"# rocprofv3 --pmc summary of {path} (mean per launch; instances summed per dispatch)")
    for name in sorted(agg, key=lambda n: -sum(agg[n]["__dur_ns"])):
        cs = agg[name]
        n = len(cs["__dur_ns"])
        print(f"{name[:150]}")
        print(f"    launches {n}   avg duration (profiled) {sum(cs['__dur_ns']) / n / 1e3:.1f} us")
        for c in sorted(k for k in cs if not k.startswith("__")):
            v = cs[c]
            print(f"    {c:34s} mean {sum(v) / len(v):16.1f}   min {min(v):16.1f}   max {max(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
