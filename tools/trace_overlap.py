#!/usr/bin/env python3
"""Overlap of the draft stage with the polish stage in a traced bench run (rocprofv3 --kernel-trace, sqlite output).
For every k_poa / k_align16 / k_align dispatch: the part of its interval during which some k_polish dispatch was running, plus
the union-busy time of the two stages and the wall span.  usage: trace_overlap.py <dir with *_results.db>"""
import glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True)[0]
c = sqlite3.connect(db)
def kname(n):                                   # 'void k_polish_t<512, 2, 8>(KParams, ...)' -> 'k_polish'
    n = n.split("(")[0].split("<")[0].strip()
    n = n[5:] if n.startswith("void ") else n
    return "k_polish" if n == "k_polish_t" else n
rows = [(kname(n), s, e, q) for n, s, e, q in c.execute("select name, start, end, queue_id from kernels order by start")]
t0 = min(r[1] for r in rows)
pol = [(s, e) for n, s, e, q in rows if n in ("k_polish", "k_stitch", "k_kinetics")]
dra = [(n, s, e) for n, s, e, q in rows if n in ("k_poa_init", "k_poa_dp", "k_poa_thread", "k_poa_finish", "k_align16", "k_align", "k_rescue", "k_post", "k_setup")]
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    a, b = union(a), union(b); i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
ms = lambda x: x / 1e6
queues = {}
for n, s, e, q in rows: queues.setdefault(n, set()).add(q)
print("queues per kernel:", {k: sorted(v) for k, v in queues.items() if k.startswith("k_")})
ub_p, ub_d = sum(e - s for s, e in union(pol)), sum(e - s for s, e in union([(s, e) for _, s, e in dra]))
both = inter(pol, [(s, e) for _, s, e in dra])
span = max(r[2] for r in rows if r[0].startswith("k_")) - min(r[1] for r in rows if r[0].startswith("k_"))
print("polish-stage busy %.1f ms, draft-stage busy %.1f ms, both at once %.1f ms (%.0f %% of the draft stage), wall span of all kernels %.1f ms"
      % (ms(ub_p), ms(ub_d), ms(both), 100.0 * both / max(1, ub_d), ms(span)))
for name in ("k_poa", "k_align16"):
    ks = [(s, e) for n, s, e in dra if n == name]
    if not ks: continue
    tot = sum(e - s for s, e in ks); ov = sum(inter([k], pol) for k in ks)
    big = sorted(ks, key=lambda k: k[0] - k[1])[:8]
    print("%s: %d dispatches, %.1f ms in total, %.1f ms (%.0f %%) under k_polish; longest dispatches (start ms, duration ms, under polish ms): %s"
          % (name, len(ks), ms(tot), ms(ov), 100.0 * ov / max(1, tot), [(round(ms(s - t0), 1), round(ms(e - s), 1), round(ms(inter([(s, e)], pol)), 1)) for s, e in sorted(big)]))
pp = sorted((s, e) for n, s, e, q in rows if n == "k_polish")
print("k_polish dispatches (start ms, duration ms):", [(round(ms(s - t0), 1), round(ms(e - s), 1)) for s, e in pp])
