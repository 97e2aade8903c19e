#!/usr/bin/env python3
"""Accuracy of the SPEC on synthetic ZMWs with the CPU restatement (oracle): consensus errors against the true templates, rq, polish
rounds, which paths fired, counted cell updates.  Used for every SPEC decision of DESIGN.md §2 and for the off-model sweeps
(profiles/r03_offmodel.txt).   usage: acc_eval.py N PASSES LENGTH SEED [key=value ...]
  keys: any ccsx_opts field; poa_band / align_band1 / score_band / skip_margin = the SPEC's approximations (oracle test hooks); channel=<x>: error-channel multiplier applied to the reads (extra substitutions / indels on top of the
  generator's); tpl=lowcx: low-complexity templates (tools/lowcx.py); env:NAME=VALUE sets an oracle experiment variable."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from ccs_amd import api
import oracle_lib as O


def evaluate(batch, opts, model=None, nthreads=8, label=""):
    model = model or api.default_model()
    res = api.Results.allocate(batch)
    O.counts_reset()
    t0 = time.time()
    O.consensus_batch(model, opts, batch, res, nthreads=nthreads)
    dt = time.time() - t0
    ok = np.nonzero((res.status == 0) | (res.status == 7))[0]
    err = sum(O.edit_distance(res.sequence(z), batch.tpl[batch.tpl_off[z]:batch.tpl_off[z + 1]]) for z in ok)
    nb = int(sum(batch.tpl_off[z + 1] - batch.tpl_off[z] for z in ok))
    c = O.counts()
    st = {api.STATUS_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(res.status, return_counts=True))}
    print(f"{label:28s} errors {err:5d} / {nb} b ({1e6 * err / max(1, nb):7.1f} ppm)  rq {res.rq[ok].mean():.6f}  rounds/win {res.iters.sum() / max(1, res.n_windows.sum()):.3f} "
          f" status {st}  paths {{trim {c['trim']} split {c['split']} fallback {c['fallback']} retry64 {c['retry64']} sat {c['saturated']} zdrop {c['zdrop']} poa_wide {c['poa_wide']}}} "
          f" cells/ZMW poa {c['cells_poa'] // max(1, c['zmws'])} align {c['cells_align'] // max(1, c['zmws'])} fill {c['cells_fill'] // max(1, c['zmws'])} score {c['cells_score'] // max(1, c['zmws'])}  {dt:.1f}s", flush=True)
    return res, err, c


if __name__ == "__main__":
    n, P, Ln, seed = (int(x) for x in sys.argv[1:5])
    opts = api.default_opts()
    channel, tpl, hp_boost = 1.0, None, 1.0
    for kv in sys.argv[5:]:
        k, v = kv.split("=", 1)
        if k.startswith("env:"): os.environ[k[4:]] = v
        elif k in ("poa_band", "align_band1", "score_band", "skip_margin", "sat_rows", "sat_gain", "fill_band"): getattr(O.lib(), "orc_set_" + k)(int(v))   # SPEC approximation knobs
        elif k == "channel": channel = float(v)
        elif k == "tpl": tpl = v
        elif k == "hp_boost": hp_boost = float(v)
        else: setattr(opts, k, type(getattr(opts, k))(float(v)))
    sys.path.insert(0, os.path.join(R, "tools"))
    import lowcx
    b = lowcx.make(n, P, Ln, seed, channel=channel, tpl=tpl, hp_boost=hp_boost)
    evaluate(b, opts, label=" ".join(sys.argv[5:]) or "default")
