import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from ccs_amd import api
n = 4096
b = api.synth(n, 10, 10000, seed=5)
for cov in (4, 5, 6, 7, 8):
    o = api.default_opts(); o.max_poa_cov = cov
    h = api.Handle(0, opts=o)
    h.upload(b); h.run(); h.sync(); h.run(); h.sync()
    t = h.timings(); res = h.download()
    ok = res.status == 0
    # exactness vs truth: length + 24-mer containment
    exact = 0
    for z in range(0, n, 16):
        tpl = b.tpl[b.tpl_off[z]:b.tpl_off[z+1]]; s = res.sequence(z)
        exact += int(len(s) == len(tpl) and np.array_equal(s, tpl))
    print(f"cov {cov:2d}: poa {t.draft_ms:7.1f} align {t.align_ms:6.1f} polish {t.polish_ms:7.1f} total {t.total_ms:7.1f} ms | ok {ok.mean():.4f} rq {res.rq[ok].mean():.6f} iters/win {res.iters.sum()/res.n_windows.sum():.3f} exact {exact}/{n//16}", flush=True)
    h.close()
