import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ccs_amd import api
b = api.synth(4096, 10, 10000, seed=0xC0FFEE)
for cov in (3, 4, 5, 6, 7):
    o = api.default_opts(); o.max_poa_cov = cov
    h = api.Handle(0, opts=o)
    h.upload(b); h.run(); h.sync(); h.run(); h.sync()
    t = h.timings(); r = h.download() if hasattr(h, "download") else None
    res = h.consensus(b)
    err = 0
    ok = (res.status == 0)
    print("cov", cov, "ms draft %.1f align %.1f polish %.1f total %.1f" % (t.draft_ms, t.align_ms, t.polish_ms, t.draft_ms + t.align_ms + t.polish_ms), "success", int(ok.sum()), "mean rq %.6f" % res.rq[ok].mean(), "iters", int(res.iters.sum()))
    h.close()
