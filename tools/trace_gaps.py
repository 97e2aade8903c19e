#!/usr/bin/env python3
"""Timeline of one traced bench run (rocprofv3 --kernel-trace, sqlite output): per batch the compute kernels' span, the idle gap to the
next batch's first kernel, and the copy kernels that ran meanwhile.  usage: trace_gaps.py <dir with *_results.db>"""
import glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels") or t == "kernels"]
print("tables:", [t for t in tabs if "kernel" in t][:8])
rows = list(c.execute("select name, start, end, queue_id from kernels order by start")) if "kernels" in tabs else []
if not rows:
    v = [t for t in tabs if "kernel" in t.lower()]
    print("no 'kernels' view; candidates", v); sys.exit(0)
t0 = rows[0][1]
comp = [(n.split("(")[0], (s - t0) / 1e6, (e - t0) / 1e6, q) for n, s, e, q in rows]
setups = [i for i, r in enumerate(comp) if r[0] == "k_setup"]
for a, b in zip(setups, setups[1:] + [len(comp)]):
    ks = [r for r in comp[a:b] if r[0].startswith("k_")]
    cp = [r for r in comp[a:b] if "copy" in r[0].lower()]
    busy = sum(r[2] - r[1] for r in ks)
    print("batch: first %.1f ms last end %.1f span %.1f kernel-busy %.1f idle-in-span %.1f | copy kernels %d, %.1f ms, queues %s" % (
        ks[0][1], ks[-1][2], ks[-1][2] - ks[0][1], busy, ks[-1][2] - ks[0][1] - busy, len(cp), sum(r[2] - r[1] for r in cp), sorted(set(r[3] for r in cp))))
ends = [max(r[2] for r in comp[a:b] if r[0].startswith("k_")) for a, b in zip(setups, setups[1:] + [len(comp)])]
starts = [comp[a][1] for a in setups]
print("gaps between batches (ms):", ["%.1f" % (starts[i + 1] - ends[i]) for i in range(len(starts) - 1)])
