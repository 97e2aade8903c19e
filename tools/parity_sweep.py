"""One-off wide parity sweep on the GPU box: HIP path vs CPU restatement over many seeds and shapes (bit-exact or fail).

python tools/parity_sweep.py [n_batches]   — each batch: a few hundred ZMWs of one shape family, with and without kinetics.
"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from ccs_amd import api
import oracle_lib as O

SHAPES = [  # (n_zmw, passes, length)
    (384, (3, 50), (300, 6000)),       # configs[4]-like mix, scaled down so the CPU side stays in seconds
    (256, 10, (9000, 11000)),          # headline shape
    (96, 30, (15000, 21000)),          # deep + long
    (1024, 3, (400, 1500)),            # configs[0]-like
    (200, (1, 70), (50, 900)),         # degenerate sizes: single pass, > 64 passes, tiny inserts
    (64, (5, 12), (30000, 45000)),     # long inserts near --max-length
]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else len(SHAPES)
bad = 0
for k in range(nb):
    n, p, l = SHAPES[k % len(SHAPES)]
    seed = 1000 + 17 * k
    b = api.synth(n, p, l, seed=seed)
    rng = np.random.default_rng(seed)
    b.ipd = rng.integers(0, 256, len(b.bases)).astype(np.uint8)
    for kin in (0, 1):
        o = api.default_opts(); o.hifi_kinetics = kin
        if k % 3 == 2: o.max_poa_cov = 3 + (k % 5)
        h = api.Handle(0, opts=o)
        t0 = time.time(); res = h.consensus(b); tg = time.time() - t0
        ref = api.Results.allocate(b, kinetics=bool(kin))
        t0 = time.time(); O.consensus_batch(h.model, o, b, ref, nthreads=16); tc = time.time() - t0
        ok = (np.array_equal(res.status, ref.status) and np.array_equal(res.seq_len, ref.seq_len) and np.array_equal(res.np_, ref.np_)
              and np.array_equal(res.iters, ref.iters) and np.array_equal(res.fn, ref.fn) and np.array_equal(res.rn, ref.rn)
              and np.array_equal(res.rq, ref.rq) and np.array_equal(res.ec, ref.ec))
        dq = 0.0
        for z in range(n):
            if not ok: break
            ok = np.array_equal(res.sequence(z), ref.sequence(z)) and np.array_equal(res.quals(z), ref.quals(z))
            if ok and len(res.raw(z)): dq = max(dq, float(np.abs(res.raw(z) - ref.raw(z)).max()))
            if ok and kin: ok = np.array_equal(res.kinetics(z), ref.kinetics(z))
        st = np.bincount(res.status, minlength=9)
        print(f"batch {k} shape {(n, p, l)} kin {kin} cov {o.max_poa_cov}: {'OK ' if ok else 'MISMATCH'} max|dQV| {dq:.2e} status {st.tolist()} gpu {tg:.2f}s cpu {tc:.1f}s", flush=True)
        bad += 0 if ok else 1
        h.close()
print("FAILED" if bad else "ALL BIT-EXACT")
sys.exit(1 if bad else 0)
