#!/usr/bin/env python3
"""Generates tests/golden/golden_v7.npz: inputs + expected outputs of the hot path at SPEC version 7.  Round 5 made two steps: v6 = the banded alpha /
beta fill — every array of golden_v6.npz equalled its counterpart in round 4's golden_v5.npz (the band is exactly free, profiles/r05_band_study.txt; `--compare
OLD.npz` checks such a claim) — and v7 = honest QVs (skip-probability floor Q50, repeat-count floor): against v6 only qual / raw_qv / rq (and the statuses that
follow from rq) change, sequences, np, ec, iterations and windows do not (`--compare OLD.npz --qv-only`).

The reference mount is documentation-only (no source, binary or test vectors: SURVEY.md §0/§8c), so these
vectors come from this repository's own CPU restatement (oracle/ccs_oracle.c, "parity unpinned") at the
specification version in DESIGN.md §SPEC.  They freeze the specification: any change to the oracle or the
kernels that alters results must regenerate this file deliberately (and bump CCSX_SPEC_VERSION / ORC_SPEC_VERSION:
the file carries the version and tests/test_golden.py refuses a library or oracle of another one).

Every SPEC path that is not taken by plain synthetic data has a case here that PROVABLY takes it: the generator asserts
through the oracle's path counters that the large-insertion trim, the split alignment (interior, s = 0 and s = Ld), the
fallback draft, the last-resort draft, the 16 -> 64-row alignment retry, a z-score drop, a NON_CONVERGENT window and the
partial-pass alignment fired in its case.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))

from ccs_amd import api  # noqa: E402
import oracle_lib as O  # noqa: E402


def with_blocks(base, where, rng):
    """copy of `base` with foreign blocks inserted: where = {(zmw, pass): (position as a fraction of the pass or 'start' / 'end', length)}"""
    bases, pw, ipd, off = [], [], [], [0]
    for r in range(int(base.read_off[-1])):
        a, b = int(base.base_off[r]), int(base.base_off[r + 1])
        bb, pp, ii = base.bases[a:b], base.pw[a:b], base.ipd[a:b]
        z = int(np.searchsorted(base.read_off, r, side="right") - 1)
        key = (z, r - int(base.read_off[z]))
        if key in where:
            pos, ln = where[key]
            at = 0 if pos == "start" else (len(bb) if pos == "end" else int(pos * len(bb)))
            blk = rng.integers(0, 4, ln, dtype=np.uint8)
            bb = np.concatenate([bb[:at], blk, bb[at:]]); pp = np.concatenate([pp[:at], np.full(ln, 2, np.uint8), pp[at:]])
            ii = np.concatenate([ii[:at], np.full(ln, 5, np.uint8), ii[at:]])
        bases.append(bb); pw.append(pp); ipd.append(ii); off.append(off[-1] + len(bb))
    return api.Batch(base.zmw_id, base.snr, base.read_off, np.array(off, np.int64), np.concatenate(bases), np.concatenate(pw),
                     np.concatenate(ipd), base.flags.copy(), base.tpl_off, base.tpl)


def junk(batch, reads, rng):
    for z, q in reads:
        r = int(batch.read_off[z]) + q
        a, b = int(batch.base_off[r]), int(batch.base_off[r + 1])
        batch.bases[a:b] = rng.integers(0, 4, b - a, dtype=np.uint8)
    return batch


def cases():
    import lowcx
    import test_oracle_draft as T
    rng = np.random.default_rng(2026)
    yield "p3_l300", api.synth(3, 3, 300, seed=101), {}
    yield "p5_l700", api.synth(2, 5, 700, seed=102), {}
    yield "p10_l2000", api.synth(2, 10, 2000, seed=103), {"zdrop": 1}
    yield "mix", api.synth(3, (3, 9), (150, 900), seed=104), {}
    yield "c2_one", api.synth(1, 10, 10000, seed=105), {"zdrop": 1}                       # one ZMW at the headline size
    yield "trim", with_blocks(api.synth(2, 8, 1500, seed=90), {(1, 5): (0.5, 40), (1, 6): (0.4, 45)}, rng), {"trim": 1, "split": 1}
    yield "split", with_blocks(api.synth(3, 8, 2000, seed=95), {(0, 2): (0.5, 150), (1, 2): ("start", 120), (2, 4): ("end", 90),
                                                                (0, 6): (0.3, 400)}, rng), {"split": 3, "split_s0": 1, "split_sLd": 1, "trim": 1}
    yield "fallback", junk(api.synth(2, 7, (500, 1200), seed=97), [(1, 0)], rng), {"fallback": 1}
    yield "lastresort", T._junk_backbones_batch().slice(1, 3), {"fallback": 1, "third_draft": 1}
    yield "retry64", with_blocks(api.synth(2, 6, 1500, seed=98), {(0, 5): (0.5, 12), (1, 3): (0.6, 14)}, rng), {"retry64": 1}
    yield "lowcx", lowcx.make(3, 10, 5000, 7, tpl="lowcx").slice(2, 3), {"nonconv_win": 1}           # the one ZMW of 144 that still ends NON_CONVERGENT at SPEC v5
    # SPEC v5 "band saturation": low-complexity templates on which the 16-row band locks onto a wrong repeat phase although it passes the gate;
    # the saturated passes are re-aligned with 64 rows and every ZMW succeeds (tools/acc_eval.py tpl=lowcx: 33 -> 48 of 48)
    yield "lowcx_rescue", lowcx.make(3, 10, (1500, 3000), 402, tpl="lowcx"), {"saturated": 10}
    yield "partial", T.partial_pass_batch(n=2, seed=58, nfull=5, length=(800, 1500)), {"partial_used": 4}
    import test_oracle_filter as TF
    # SPEC v8: a tandem tract that begins and ends inside the visible window template at more than ten passes: its repeat-count floor is scaled down by the coverage
    yield "closed_tract", lowcx.make(2, 16, 1200, 811, tpl="lowcx"), {"closed_tract": 1}
    yield "split2", TF._two_block_batch(sizes=(250, 250, 250), fr=(0.2, 0.5, 0.8))[1], {"split2": 2}     # SPEC v4: three blocks per pass


PATH_KEYS = O.COUNT_NAMES[:11] + ["split2", "saturated", "closed_tract"]          # = tests/golden_util.py PATHS


def main():
    out = {}
    m, o = api.default_model(), api.default_opts()
    names = []
    for name, b, must in cases():
        r = api.Results.allocate(b)
        O.counts_reset()
        O.consensus_batch(m, o, b, r)
        c = O.counts()
        for k, v in must.items():
            assert c[k] >= v, f"case {name}: path {k} fired {c[k]} times, {v} wanted"
        if name == "lowcx":
            assert (r.status == 4).any()
        if name == "lowcx_rescue":
            assert (r.status == 0).all()
        for k in ("zmw_id", "snr", "read_off", "base_off", "bases", "pw", "ipd", "flags", "tpl_off", "tpl"):
            out[f"{name}/in/{k}"] = getattr(b, k)
        for k in ("seq_off", "status", "seq_len", "seq", "qual", "raw_qv", "rq", "np_", "ec", "iters", "n_windows", "fn", "rn"):
            out[f"{name}/out/{k}"] = getattr(r, k)
        out[f"{name}/draft0"] = O.poa_draft(b, 0, o.max_poa_cov)
        out[f"{name}/paths"] = np.array([c[k] for k in PATH_KEYS], np.int64)
        names.append(name)
        print(f"{name:12s} zmws {b.n_zmw} status {r.status.tolist()} paths {{{', '.join(f'{k} {c[k]}' for k in PATH_KEYS if c[k])}}}")
    out["model_bytes"] = np.frombuffer(bytes(m), np.uint8)
    out["spec_version"] = np.array([O.spec_version()], np.int32)
    out["cases"] = np.array(names)
    name = "golden_v%d.npz" % O.spec_version()
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "with", len(out), "arrays, SPEC version", O.spec_version())
    if "--compare" in sys.argv:                       # e.g. `git show HEAD~:tests/golden/golden_v5.npz > /tmp/v5.npz`: a SPEC change that claims to be result-free
        old = np.load(sys.argv[sys.argv.index("--compare") + 1])
        diff = [k for k in old.files if k != "spec_version" and not (k in out and np.array_equal(old[k], out[k]))]
        if "--qv-only" in sys.argv:                   # a SPEC change that claims to touch the QVs only
            diff = [k for k in diff if not k.endswith(("/out/qual", "/out/raw_qv", "/out/rq", "/out/status"))]
        print("arrays of the old file that differ:", diff or "none", "| arrays only in the new file:", [k for k in out if k not in old.files] or "none")
        assert not diff


if __name__ == "__main__":
    main()
