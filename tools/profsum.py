#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs under gpurun_out/prof into a text file for profiles/."""
import glob, os, sqlite3, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out = []
tr = os.path.join(root, "trace", "trace_results.db")
if os.path.exists(tr):
    c = sqlite3.connect(tr)
    out.append("== kernel trace (rocprofv3 --kernel-trace --stats): per-kernel durations ==")
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out.append(f"{'kernel':40s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'min_ms':>12s} {'max_ms':>12s} {'pct':>7s}")
    for n, k, s, a, mn, mx in rows:
        out.append(f"{n[:40]:40s} {k:6d} {s/1e6:12.3f} {a/1e6:12.3f} {mn/1e6:12.3f} {mx/1e6:12.3f} {100*s/tot:7.2f}")
for db in sorted(glob.glob(os.path.join(root, "pmc_*", "pmc_results.db"))):
    c = sqlite3.connect(db)
    out.append(f"\n== PMC pass {os.path.basename(os.path.dirname(db))} (per-kernel sum over dispatches) ==")
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        for kn, cn, v, n in rows:
            out.append(f"{kn[:32]:32s} {cn:28s} {v:20.0f}  (n={n})")
    except Exception as e:
        out.append(f"  (query failed: {e}; columns {cols if 'cols' in dir() else '?'})")
print("\n".join(out))
