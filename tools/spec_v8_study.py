#!/usr/bin/env python3
"""SPEC v8 candidates against SPEC v7 on the CPU restatement, in the format of tools/band_study.py:
  fma  — every multiply-add of the alpha / beta fill, the mutation extension and the link is one fused multiply-add (orc_set_fma);
  clip — a scoring row is evaluated iff the cells it reads lie on the fill band (orc_set_score_clip): one compare per row on the device, no band tests.
Per data set and rule: sequences / phred QVs / raw QVs changed against v7, max |d rq|, consensus errors against the TRUTH, yield (SUCCESS), polish rounds per
window, predicted / empirical error ratio.  A rule ships only if the errors against the truth and the calibration are no worse.
  usage: python tools/spec_v8_study.py [N_ZMW] > profiles/r06_spec_v8_study.txt"""
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
RULES = [("v7", 0, 0), ("fma", 1, 0), ("clip", 0, 1), ("fma+clip", 1, 1)]


def run(batch, opts):
    res = api.Results.allocate(batch)
    O.consensus_batch(api.default_model(), opts, batch, res, nthreads=8)
    return res


def truth_stats(b, r):
    ok = np.nonzero(r.status == 0)[0]
    err = sum(O.edit_distance(r.sequence(z), b.tpl[b.tpl_off[z]:b.tpl_off[z + 1]]) for z in ok)
    nb = int(sum(len(r.sequence(z)) for z in ok))
    pred = float(sum((1.0 - float(r.rq[z])) * len(r.sequence(z)) for z in ok))
    return len(ok), err, nb, pred


print("# SPEC v8 candidates against SPEC v7, CPU restatement, %d ZMWs per data set (tools/lowcx.py generator, seed 77; the sets of profiles/r05_band_study.txt)." % N)
print("# fma = fused multiply-add recurrences (fill, extension, link); clip = scoring rows restricted to the fill band.  'errors' = edit distance of the SUCCESS reads to the true templates;")
print("# 'emp/pred' = those errors / sum over reads of (1 - rq) * length (1.0 = calibrated, > 1 = over-confident).")
tot = {k: [0, 0, 0.0] for k, _, _ in RULES}
for name, kw in SETS:
    opts = api.default_opts()
    if kw.get("disable_heuristics"): opts.disable_heuristics = 1
    b = lowcx.make(N, kw["P"], kw["L"], 77, channel=kw.get("channel", 1.0), tpl=kw.get("tpl"), hp_boost=kw.get("hp_boost", 1.0))
    base = None
    for rule, fma, clip in RULES:
        O.lib().orc_set_fma(fma); O.lib().orc_set_score_clip(clip)
        r = run(b, opts)
        if base is None: base = r
        same = [bool(base.status[z] == r.status[z] and np.array_equal(base.sequence(z), r.sequence(z))) for z in range(b.n_zmw)]
        dseq = b.n_zmw - sum(same)
        dqv = sum(int(not (same[z] and np.array_equal(base.quals(z), r.quals(z)))) for z in range(b.n_zmw))
        draw = sum(int(not (same[z] and np.array_equal(base.raw(z), r.raw(z)))) for z in range(b.n_zmw))
        mraw = max([float(np.max(np.abs(base.raw(z) - r.raw(z)))) if same[z] and len(r.raw(z)) else 0.0 for z in range(b.n_zmw)])
        drq = float(np.max(np.abs(base.rq.astype(np.float64) - r.rq.astype(np.float64))))
        nok, err, nb, pred = truth_stats(b, r)
        tot[rule][0] += err; tot[rule][1] += nb; tot[rule][2] += pred
        print("%-28s %-9s sequences changed %2d/%d, phred QVs changed %2d, raw QVs changed %2d (max |d raw QV| %.2e), max |d rq| %.2e | errors %5d / %d b, SUCCESS %d/%d, rounds/window %.4f, emp/pred %.3f" %
              (name, rule, dseq, b.n_zmw, dqv, draw, mraw, drq, err, nb, nok, b.n_zmw, r.iters.sum() / max(1, r.n_windows.sum()), err / max(pred, 1e-9)), flush=True)
print("# totals over the seven sets:")
for rule, _, _ in RULES:
    e, nb, pred = tot[rule]
    print("#   %-9s errors %d / %d b (%.1f ppm), emp/pred %.3f" % (rule, e, nb, 1e6 * e / max(1, nb), e / max(pred, 1e-9)))
