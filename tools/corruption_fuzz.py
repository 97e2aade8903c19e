"""Fuzz of the robustness paths on the GPU box (round 3: + partial passes, low-complexity templates, off-model channels): HIP path vs CPU restatement (bit-exact or fail) on batches whose passes are
corrupted the way real subreads are — foreign blocks of 8..600 bases (one or several per pass, anywhere incl. the first / last
bases), missing stretches, junk passes, truncated passes — under random option sets (max_insertion_size, fallback draft,
candidate filter, kinetics).  Exercises k_rescue (split alignment), the large-insertion trim, the fallback draft and every gate.

python tools/corruption_fuzz.py [n_batches] [seed0]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
from ccs_amd import api
import oracle_lib as O


def make_batch(k, seed0):
    """batch k of the fuzz with seed seed0: (batch, opts, ZMWs, lmax, corrupted passes, passes) - also used by tests/test_gpu_parity.py"""
    rng = np.random.default_rng(seed0 + k)
    n = int(rng.integers(8, 40))
    lmax = int(rng.choice([300, 1500, 4000, 12000]))
    if rng.random() < 0.25:                                 # round 3: low-complexity templates / off-model channels (tools/lowcx.py)
        import lowcx
        base = lowcx.make(n, (int(rng.integers(3, 6)), int(rng.integers(6, 24))), (max(60, lmax // 4), lmax), seed0 + 31 * k,
                          channel=float(rng.choice([0.6, 1.0, 1.5])), tpl=("lowcx" if rng.random() < 0.6 else None), hp_boost=float(rng.choice([1.0, 2.5])))
    elif rng.random() < 0.12:                               # round 4: many passes (SPEC v5: up to 255 are used, k_polish takes them in groups of 32)
        n, lmax = int(rng.integers(4, 12)), int(rng.choice([300, 1500]))
        base = api.synth(n, (int(rng.integers(20, 60)), int(rng.integers(60, 300))), (max(40, lmax // 4), lmax), seed=seed0 + 31 * k)
    else:
        base = api.synth(n, (int(rng.integers(1, 6)), int(rng.integers(6, 24))), (max(40, lmax // 4), lmax), seed=seed0 + 31 * k)
    base.ipd = rng.integers(0, 256, len(base.bases)).astype(np.uint8)
    bases, pw, ipd, off = [], [], [], [0]
    ncorr = 0
    flags = base.flags.copy()
    # round 3: partial passes — the last one or two passes of some ZMWs are truncated at one end and flagged (bit 1, bit 2 = adapter at the end)
    part = {}
    for z in range(n):
        nr = int(base.read_off[z + 1] - base.read_off[z])
        if nr >= 4 and rng.random() < 0.35:
            for q in range(nr - int(rng.integers(1, 3)), nr):
                part[int(base.read_off[z]) + q] = int(rng.integers(0, 2))
    for r in range(int(base.read_off[-1])):
        a, b = int(base.base_off[r]), int(base.base_off[r + 1])
        bb, pp, ii = base.bases[a:b].copy(), base.pw[a:b].copy(), base.ipd[a:b].copy()
        u = rng.random()
        if u < 0.30:                                        # 1..3 foreign blocks
            for _ in range(int(rng.integers(1, 4))):
                size = int(rng.choice([8, 20, 33, 45, 70, 150, 600]))
                at = int(rng.integers(0, len(bb) + 1))
                blk = rng.integers(0, 4, size, dtype=np.uint8)
                bb = np.concatenate([bb[:at], blk, bb[at:]]); pp = np.concatenate([pp[:at], rng.integers(1, 4, size).astype(np.uint8), pp[at:]])
                ii = np.concatenate([ii[:at], rng.integers(0, 256, size).astype(np.uint8), ii[at:]])
            ncorr += 1
        elif u < 0.38 and len(bb) > 80:                     # a missing stretch
            at, size = int(rng.integers(0, len(bb) - 40)), int(rng.integers(10, 200))
            bb = np.concatenate([bb[:at], bb[at + size:]]); pp = np.concatenate([pp[:at], pp[at + size:]]); ii = np.concatenate([ii[:at], ii[at + size:]])
            ncorr += 1
        elif u < 0.43:                                      # junk
            bb = rng.integers(0, 4, len(bb), dtype=np.uint8); ncorr += 1
        elif u < 0.47 and len(bb) > 60:                     # truncated
            cut = int(rng.integers(20, len(bb)))
            bb, pp, ii = bb[:cut], pp[:cut], ii[:cut]; ncorr += 1
        if r in part and len(bb) > 120:
            keep = int(len(bb) * rng.uniform(0.2, 0.9))
            if part[r]: bb, pp, ii = bb[len(bb) - keep:], pp[len(pp) - keep:], ii[len(ii) - keep:]; flags[r] |= 2 | 4
            else: bb, pp, ii = bb[:keep], pp[:keep], ii[:keep]; flags[r] |= 2
        elif r in part: flags[r] |= 2 | (4 if part[r] else 0)
        bases.append(bb); pw.append(pp); ipd.append(ii); off.append(off[-1] + len(bb))
    batch = api.Batch(base.zmw_id, base.snr, base.read_off, np.array(off, np.int64), np.concatenate(bases), np.concatenate(pw),
                      np.concatenate(ipd), flags, base.tpl_off, base.tpl)
    o = api.default_opts()
    o.max_insertion_size = int(rng.choice([0, 30, 10, 5, -1])); o.no_fallback_draft = int(rng.random() < 0.25)
    o.disable_heuristics = int(rng.random() < 0.2); o.hifi_kinetics = int(rng.random() < 0.4); o.max_poa_cov = int(rng.choice([3, 5, 7]))
    o.min_rq = float(rng.choice([0.99, 0.9, 0.0]))
    o.top_passes = int(rng.choice([0, 60, 60, 100, 33]))
    return batch, o, n, lmax, ncorr, int(base.read_off[-1])


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # first batch (to reproduce one batch of a longer run)
    verbose = len(sys.argv) > 4
    bad = 0
    for k in range(k0, k0 + nb):
        batch, o, n, lmax, ncorr, npass = make_batch(k, seed0)
        h = api.Handle(0, opts=o)
        t0 = time.time(); res = h.consensus(batch); tg = time.time() - t0
        ref = api.Results.allocate(batch, kinetics=bool(o.hifi_kinetics))
        t0 = time.time(); O.consensus_batch(h.model, o, batch, ref, nthreads=16); tc = time.time() - t0
        ok = all(np.array_equal(getattr(res, f), getattr(ref, f)) for f in ("status", "seq_len", "np_", "iters", "fn", "rn", "rq", "ec"))
        for z in range(n):
            if not ok: break
            ok = np.array_equal(res.sequence(z), ref.sequence(z)) and np.array_equal(res.quals(z), ref.quals(z)) and np.array_equal(res.raw(z), ref.raw(z))
            if ok and o.hifi_kinetics: ok = np.array_equal(res.kinetics(z), ref.kinetics(z))
        if verbose and not ok:
            for f in ("status", "seq_len", "np_", "iters", "fn", "rn", "rq", "ec"):
                a, b = getattr(res, f), getattr(ref, f)
                if not np.array_equal(a, b): print("  ", f, "gpu", a[a != b][:8], "cpu", b[a != b][:8], "zmws", np.nonzero(a != b)[0][:8])
            for z in range(n):
                if not np.array_equal(res.sequence(z), ref.sequence(z)):
                    nr = int(batch.read_off[z + 1] - batch.read_off[z]); r0 = int(batch.read_off[z])
                    print("   zmw", z, "seq differs; passes", nr, "lens", np.diff(batch.base_off[r0:r0 + nr + 1]).tolist(), "flags", batch.flags[r0:r0 + nr].tolist(), "status", res.status[z], ref.status[z])
        st = np.bincount(res.status, minlength=10)
        print(f"batch {k} n {n} lmax {lmax} corrupted passes {ncorr}/{npass} maxins {o.max_insertion_size} nofb {o.no_fallback_draft} "
              f"noheur {o.disable_heuristics} kin {o.hifi_kinetics} cov {o.max_poa_cov}: {'OK ' if ok else 'MISMATCH'} status {st.tolist()} gpu {tg:.2f}s cpu {tc:.1f}s", flush=True)
        bad += 0 if ok else 1
        h.close()
    print("FAILED" if bad else "ALL BIT-EXACT")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
