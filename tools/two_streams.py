import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from ccs_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
b = api.synth(n, 10, 10000, seed=5)
parts = [b.slice(i * n // k, (i + 1) * n // k) for i in range(k)]
hs = [api.Handle(0) for _ in range(k)]
for h, p in zip(hs, parts): h.upload(p)
for h in hs: h.sync()
for rep in range(3):
    t = time.perf_counter()
    for h in hs: h.run()
    for h in hs: h.sync()
    dt = time.perf_counter() - t
    print(f"{k} handles x {n//k}: {dt*1e3:.1f} ms  {n/dt:.0f} ZMW/s", flush=True)
