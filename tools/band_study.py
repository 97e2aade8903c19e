#!/usr/bin/env python3
"""SPEC v6 "banded fill" against the full matrices (SPEC v5), on the CPU restatement: the whole pipeline per data set with the full alpha / beta
matrices and with the band [min(0,J-I) - (Wr + lo), max(0,J-I) + (Wr + hi)], compared per ZMW (sequence, phred QVs, rq, polish rounds).
  usage: python tools/band_study.py [N_ZMW] > profiles/r05_band_study.txt"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
from ccs_amd import api
import oracle_lib as O
import lowcx

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SETS = [("on-model 10x5kb", dict(P=10, L=5000)), ("on-model 5x5kb", dict(P=5, L=5000)), ("channel 1.5", dict(P=10, L=5000, channel=1.5)),
        ("hp_boost 2.5", dict(P=10, L=5000, hp_boost=2.5)), ("lowcx", dict(P=10, L=5000, tpl="lowcx")), ("30 passes x 3kb", dict(P=30, L=3000)),
        ("on-model 10x5kb, no filter", dict(P=10, L=5000, disable_heuristics=1))]
MARGINS = [(2, 2), (1, 2), (0, 2), (1, 1), (0, 0)]


def run(batch, opts):
    res = api.Results.allocate(batch)
    O.consensus_batch(api.default_model(), opts, batch, res, nthreads=8)
    return res


print("# SPEC v6 banded alpha / beta fill against the full matrices, CPU restatement, %d ZMWs per data set (tools/lowcx.py generator, seed 77)." % N)
print("# Band = diagonals j - i in [min(0, J - I) - (Wr + lo), max(0, J - I) + (Wr + hi)], Wr = SCORE_BAND + max(0, |I - J| - 2); cells outside are exact zeros.")
for name, kw in SETS:
    opts = api.default_opts()
    if kw.get("disable_heuristics"): opts.disable_heuristics = 1
    b = lowcx.make(N, kw["P"], kw["L"], 77, channel=kw.get("channel", 1.0), tpl=kw.get("tpl"), hp_boost=kw.get("hp_boost", 1.0))
    O.lib().orc_set_fill_band(-1)
    full = run(b, opts)
    for lo, hi in MARGINS:
        O.lib().orc_set_fill_margins(lo, hi)
        r = run(b, opts)
        same_seq = sum(int(full.status[z] == r.status[z] and np.array_equal(full.sequence(z), r.sequence(z))) for z in range(b.n_zmw))
        same_qv = sum(int(full.status[z] == r.status[z] and np.array_equal(full.sequence(z), r.sequence(z)) and np.array_equal(full.quals(z), r.quals(z))) for z in range(b.n_zmw))
        same_raw = sum(int(full.status[z] == r.status[z] and np.array_equal(full.sequence(z), r.sequence(z)) and np.array_equal(full.raw(z), r.raw(z))) for z in range(b.n_zmw))
        drq = float(np.max(np.abs(full.rq.astype(np.float64) - r.rq.astype(np.float64))))
        print("%-28s margins (%d, %d): identical sequences %d/%d, identical phred QVs %d/%d, identical raw QVs %d/%d, max |d rq| %.2e, rounds %d vs %d" %
              (name, lo, hi, same_seq, b.n_zmw, same_qv, b.n_zmw, same_raw, b.n_zmw, drq, int(full.iters.sum()), int(r.iters.sum())), flush=True)
O.lib().orc_set_fill_margins(2, 2)
