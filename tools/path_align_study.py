#!/usr/bin/env python3
"""VERDICT r05 item 3 (skipped twice as "a SPEC change, not studied"): take the step-3 alignment of the passes that were threaded into the POA (at most
max_poa_cov = 5 of a ZMW's passes) from their path through the graph instead of a second banded DP against the draft.  CPU restatement, study hook
orc_set_path_align; per data set with the hook off / on: consensus errors against the truth, yield, polish rounds per window, z-score drops, passes used.
    python tools/path_align_study.py [N_ZMW=48] > profiles/r06_path_align_study.txt"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
from ccs_amd import api
import oracle_lib as O
import lowcx

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SETS = [("on-model 10x5kb", dict(P=10, L=5000)), ("on-model 5x5kb", dict(P=5, L=5000)), ("channel 1.5", dict(P=10, L=5000, channel=1.5)),
        ("hp_boost 2.5", dict(P=10, L=5000, hp_boost=2.5)), ("lowcx", dict(P=10, L=5000, tpl="lowcx"))]
print("# step-3 alignments of the threaded passes from their POA path (study hook) against the SPEC's second DP, CPU restatement, %d ZMWs per set" % N)
for name, kw in SETS:
    b = lowcx.make(N, kw["P"], kw["L"], 77, channel=kw.get("channel", 1.0), tpl=kw.get("tpl"), hp_boost=kw.get("hp_boost", 1.0))
    base = None
    for on in (0, 1):
        O.lib().orc_set_path_align(on)
        r = api.Results.allocate(b)
        O.counts_reset()
        O.consensus_batch(api.default_model(), api.default_opts(), b, r, nthreads=8)
        c = O.counts()
        ok = np.nonzero(r.status == 0)[0]
        err = sum(O.edit_distance(r.sequence(z), b.tpl[b.tpl_off[z]:b.tpl_off[z + 1]]) for z in ok)
        nb = int(sum(len(r.sequence(z)) for z in ok))
        if base is None: base = r
        same = sum(int(base.status[z] == r.status[z] and np.array_equal(base.sequence(z), r.sequence(z))) for z in range(b.n_zmw))
        print("%-18s path-align %d: errors %5d / %d b, SUCCESS %d/%d, identical sequences %d/%d, rounds/window %.4f, z-drops %d, retry64 %d, mean np %.2f, mean ec %.2f, cells align %d per ZMW" %
              (name, on, err, nb, len(ok), b.n_zmw, same, b.n_zmw, r.iters.sum() / max(1, r.n_windows.sum()), c["zdrop"], c["retry64"], r.np_[ok].mean() if len(ok) else 0, r.ec[ok].mean() if len(ok) else 0,
               c["cells_align"] // max(1, c["zmws"])), flush=True)
O.lib().orc_set_path_align(0)
