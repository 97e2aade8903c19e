import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ccs_amd import api
b = api.synth(4096, 10, 10000, seed=0xC0FFEE)
h = api.Handle(0)
h.upload(b); h.run(); h.sync(); h.run(); h.sync()
t = h.timings(); print("ms", t.draft_ms, t.align_ms, t.polish_ms)
