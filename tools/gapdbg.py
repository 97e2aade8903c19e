import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ccs_amd import api
Z = int(sys.argv[1]); D = int(sys.argv[2])
bs = [api.synth(Z, 10, 10000, seed=0xC0FFEE, first_zmw_id=i * Z).pinned() for i in range(D)]
o = api.default_opts(); o.serial_stages = int(sys.argv[3]) if len(sys.argv) > 3 else 1
h = api.Handle(0, opts=o)
def cap(b):
    cb = b.c_struct()
    return int(api.lib().ccsx_result_layout(api.C.byref(cb), api._ptr(np.zeros(b.n_zmw + 1, np.int64), api.C.c_int64)))
big = max(bs, key=cap)
res = [api.Results.allocate(big, pinned=True, raw=False) for _ in range(3)]
tick = []; sub = []
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
prev_end = None
for k in range(N + 3):
    if k >= 3:
        t = tick[k - 3]; h.wait(t); tt = h.ticket_timings(t); h.release(t)
        gap = 0.0 if prev_end is None else tt.start_ms - prev_end
        prev_end = tt.end_ms
        print("gap %6.1f " % gap, end="")
        print("ticket %d start %.1f end %.1f dur %.1f | draft %.1f align %.1f queue %.1f polish %.1f | host submit %.1f ms" % (t, tt.start_ms, tt.end_ms, tt.end_ms - tt.start_ms, tt.draft_ms, tt.align_ms, tt.queue_ms, tt.polish_ms, sub[k - 3] * 1e3))
    if k < N:
        t0 = time.perf_counter(); tick.append(h.submit(bs[k % D], res[k % 3])); sub.append(time.perf_counter() - t0)
