"""POA cost per threaded read: draft_ms for maxPoaCoverage 1..6 at full occupancy (8192 ZMWs, 10 x 10 kb)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from ccs_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
b = api.synth(n, 10, 10000, seed=5)
prev = 0.0
for cov in (1, 2, 3, 4, 5, 6):
    o = api.default_opts(); o.max_poa_cov = cov
    h = api.Handle(0, opts=o)
    h.upload(b); h.run(); h.sync(); h.run(); h.sync()
    t = h.timings()
    print(f"cov {cov}: draft {t.draft_ms:7.1f} ms  (+{t.draft_ms - prev:6.1f} for read {cov})", flush=True)
    prev = t.draft_ms
    h.close()
