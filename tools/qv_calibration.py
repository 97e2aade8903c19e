#!/usr/bin/env python3
"""Predicted vs empirical accuracy of the consensus (docs/how-does-ccs-work.md:103-106 and docs/img/ccs-acc.png are the
only accuracy statements the reference makes: "the predicted accuracy is the mean of the per-base QVs").  For every data set — on-model,
error channel x 1.5, indels x 2.5 inside homopolymers, low-complexity templates — the CPU restatement (oracle) makes the consensus; the
reads are binned by their predicted quality (rq) and the bases by their phred QV, and each bin's predicted error count is set against the
errors found by aligning the consensus to the true template (oracle/ccs_oracle.c orc_error_positions).

Round 6: the consensus comes from the HIP library when a GPU is there (bit-identical to the restatement — tests/test_gpu_parity.py — and fast
enough for >= 1000 ZMWs per set), the sets include the headline 10 kb shape and a 3-50-pass x 1-10 kb mix, and opts.max_qv can be set (MAX_QV=93).

    python tools/qv_calibration.py [N_ZMW=1024] > profiles/r06_qv_calibration.txt      (also writes profiles/r06_qv_calibration.json)
"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import ctypes as C
import numpy as np
from ccs_amd import api
import oracle_lib as O
import lowcx

N = int(sys.argv[1]) if (__name__ == "__main__" and len(sys.argv) > 1) else 1024
MAX_QV = int(os.environ.get("MAX_QV", "0"))
TAG = os.environ.get("QVCAL_TAG", "r06_qv_calibration" + (f"_maxqv{MAX_QV}" if MAX_QV else ""))
# (name, generator arguments, passes, length, share of N)
DATASETS = [("on-model", dict(), 10, 5000, 1.0), ("channel x1.5", dict(channel=1.5), 10, 5000, 1.0), ("hp_boost 2.5", dict(hp_boost=2.5), 10, 5000, 1.0),
            ("lowcx", dict(tpl="lowcx"), 10, 5000, 1.0), ("on-model 10 x 10 kb", dict(), 10, 10000, 0.5), ("on-model 3-50 passes x 1-10 kb", dict(), (3, 50), (1000, 10000), 1.0),
            ("lowcx 30 passes x 3 kb", dict(tpl="lowcx"), 30, 3000, 0.5)]
RQ_BINS = [(0, 20), (20, 25), (25, 30), (30, 35), (35, 40), (40, 99)]
QV_BINS = [(0, 10), (10, 20), (20, 30), (30, 40), (40, 50), (50, 60), (60, 94)]
q = lambda p: 99.0 if p <= 0 else -10.0 * np.log10(p)


def error_positions(cons, truth):
    cons = np.ascontiguousarray(cons, np.uint8); truth = np.ascontiguousarray(truth, np.uint8)
    err = np.zeros(len(cons), np.uint8)
    f = O.lib().orc_error_positions
    f.restype = C.c_int
    d = f(cons.ctypes.data_as(C.POINTER(C.c_uint8)), len(cons), truth.ctypes.data_as(C.POINTER(C.c_uint8)), len(truth), 400, err.ctypes.data_as(C.POINTER(C.c_uint8)))
    return d, err


def main():
    import torch                                               # (BEFORE the first call into libccsx.so: torch brings its own HIP runtime, and whichever runtime opens the
    gpu = torch.cuda.is_available()                            # device second finds none — bench.py and the tests import torch first as well)
    m, o = api.default_model(), api.default_opts()
    o.min_rq = 0.0                                             # every consensus is kept: calibration needs the low-rq reads too
    o.max_qv = MAX_QV
    h = None
    for attempt in range(4):                                   # (a device can be transiently unavailable right after another process released it)
        if not gpu: break
        try:
            h = api.Handle(0, opts=o); break
        except RuntimeError as e:
            print(f"# Handle: {e}; retrying", file=sys.stderr); import time; time.sleep(3)
    if gpu and h is None: raise SystemExit("no device")
    out = {"spec_version": O.spec_version(), "zmws_per_dataset": N, "max_qv": MAX_QV or 50, "engine": "HIP library" if gpu else "CPU restatement", "headline": {}, "datasets": {}}
    print(f"# predicted vs empirical accuracy, SPEC v{O.spec_version()}, max_qv {MAX_QV or 50}, {N} ZMWs per data set (half for the 10 kb / 30-pass sets), {'HIP library on one MI355X' if gpu else 'CPU restatement'} (tools/qv_calibration.py)")
    print("# a read's predicted error count = (1 - rq) x length; a base's = 10^(-QV/10); empirical = errors of the consensus against the true template")
    for name, kw, P_, L_, share in DATASETS:
        n_ = max(8, int(N * share))
        b = lowcx.make(n_, P_, L_, 60, **kw)
        if gpu:
            r = h.consensus(b)
        else:
            r = api.Results.allocate(b)
            O.consensus_batch(m, o, b, r, nthreads=8)
        reads, bases = [], []
        st = {api.STATUS_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(r.status, return_counts=True))}
        for z in range(b.n_zmw):
            if r.status[z] not in (0, 7):
                continue
            cons, qual = r.sequence(z), r.quals(z)
            d, err = error_positions(cons, b.tpl[b.tpl_off[z]:b.tpl_off[z + 1]])
            if d < 0:
                continue
            reads.append((float(r.rq[z]), len(cons), d))
            bases.append(np.stack([qual.astype(np.int32), err.astype(np.int32)], 1))
        bases = np.concatenate(bases)
        nb = sum(x[1] for x in reads); pe = sum((1 - x[0]) * x[1] for x in reads); ee = sum(x[2] for x in reads)
        print(f"\n## {name}: status {st}; {len(reads)} reads, {nb} bases; predicted {1e6 * pe / nb:.0f} ppm (Q{q(pe / nb):.1f}), empirical {1e6 * ee / nb:.0f} ppm (Q{q(ee / nb):.1f}), ratio {ee / max(pe, 1e-9):.2f}")
        out["headline"][name] = {"reads": len(reads), "passes": P_, "length": L_, "predicted_ppm": round(1e6 * pe / nb, 1), "empirical_ppm": round(1e6 * ee / nb, 1),
                                 "empirical_over_predicted": round(ee / max(pe, 1e-9), 2), "status": st}
        rows_r, rows_b = [], []
        print("   reads by predicted quality (rq):    bin      reads      bases   predicted Q   empirical Q   errors pred / found")
        for lo, hi in RQ_BINS:
            sel = [x for x in reads if lo <= q(1 - x[0]) < hi]
            if not sel: continue
            n_ = sum(x[1] for x in sel); p_ = sum((1 - x[0]) * x[1] for x in sel); e_ = sum(x[2] for x in sel)
            print(f"                                   Q{lo:2d}-{hi:2d}   {len(sel):7d} {n_:10d}   {q(p_ / n_):11.1f}   {q(e_ / n_):11.1f}   {p_:10.1f} / {e_}")
            rows_r.append({"bin": [lo, hi], "reads": len(sel), "bases": n_, "predicted_q": round(q(p_ / n_), 2), "empirical_q": round(q(e_ / n_), 2), "errors": e_, "errors_predicted": round(p_, 2)})
        print("   bases by phred QV:                   bin                 bases   predicted Q   empirical Q   errors pred / found")
        for lo, hi in QV_BINS:
            sel = bases[(bases[:, 0] >= lo) & (bases[:, 0] < hi)]
            if not len(sel): continue
            p_ = float(np.sum(10.0 ** (-sel[:, 0] / 10.0))); e_ = int(sel[:, 1].sum())
            print(f"                                   Q{lo:2d}-{hi:2d}           {len(sel):10d}   {q(p_ / len(sel)):11.1f}   {q(e_ / len(sel)):11.1f}   {p_:10.1f} / {e_}")
            rows_b.append({"bin": [lo, hi], "bases": int(len(sel)), "predicted_q": round(q(p_ / len(sel)), 2), "empirical_q": round(q(e_ / len(sel)), 2), "errors": e_, "errors_predicted": round(p_, 2)})
        out["datasets"][name] = {"by_read_rq": rows_r, "by_base_qv": rows_b}
    if h is not None: h.close()
    json.dump(out, open(os.path.join(R, "profiles" if not os.environ.get("GRAFT_REPO_ROOT") else "gpurun_out", TAG + ".json"), "w"), indent=1)


if __name__ == "__main__":
    main()
