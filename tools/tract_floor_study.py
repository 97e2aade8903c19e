#!/usr/bin/env python3
"""How many consensus errors does a tandem tract carry, as a function of the number of passes?  (the SPEC v7 repeat-count floor p_err >= 0.2 / L was
calibrated at 10 passes only.)  Low-complexity templates (tools/lowcx.py) and on-model random templates at several pass counts; the consensus comes from the HIP
library when a GPU is there (bit-identical to the CPU restatement, and ~ 1000 x faster), else from the restatement.  Per pass count: the period-1..4 tandem tracts
of the consensus that are long enough for the floor (>= 8 / 10 / 12 / 16 bases), the consensus errors that fall inside them (alignment to the true template),
errors per tract, and predicted / empirical errors inside and outside the tracts.
    python tools/tract_floor_study.py [N_ZMW=256] [LENGTH=3000] > profiles/r06_tract_floor_by_passes.txt"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
from ccs_amd import api
import oracle_lib as O
import lowcx
from qv_calibration import error_positions

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
MINLEN = (8, 10, 12, 16)


def tract_mask(s):
    """bases of s inside a maximal period-p tandem tract (p = 1..4) of at least MINLEN[p-1] bases; also the number of such tracts"""
    m = np.zeros(len(s), bool); nt = 0
    for p in range(1, 5):
        eq = np.concatenate([s[:-p] == s[p:], [False]]) if len(s) > p else np.zeros(1, bool)
        i = 0
        while i < len(eq):
            if eq[i]:
                j = i
                while j < len(eq) and eq[j]: j += 1
                if j - i + p >= MINLEN[p - 1]:
                    if not m[i:j + p].all(): nt += 1
                    m[i:j + p] = True
                i = j
            else:
                i += 1
    return m, nt


def consensus(batch, opts, handle):
    if handle is not None:
        return handle.consensus(batch)
    r = api.Results.allocate(batch)
    O.consensus_batch(api.default_model(), opts, batch, r, nthreads=8)
    return r


def main():
    import torch
    gpu = torch.cuda.is_available()
    opts = api.default_opts(); opts.min_rq = 0.0
    h = api.Handle(0, opts=opts) if gpu else None
    print(f"# consensus errors inside tandem tracts by pass count, SPEC v{api.lib().ccsx_spec_version()}, {N} ZMWs x {L} bp per row, engine: {'HIP library' if gpu else 'CPU restatement'}")
    print("# tract = maximal period-p run (p = 1..4) of >= 8 / 10 / 12 / 16 bases in the consensus; 'pred' = sum of 10^(-QV/10) over the bases, 'found' = errors against the true template")
    for tpl in ("lowcx", None):
        for P in (3, 5, 8, 10, 15, 20, 30):
            b = lowcx.make(N, P, L, 4100 + P, tpl=tpl)
            r = consensus(b, opts, h)
            nt = nin = nout = ein = eout = 0; pin = pout = 0.0; nreads = 0
            for z in range(b.n_zmw):
                if r.status[z] not in (0, 7): continue
                cons, qual = r.sequence(z), r.quals(z).astype(np.float64)
                d, err = error_positions(cons, b.tpl[b.tpl_off[z]:b.tpl_off[z + 1]])
                if d < 0: continue
                m, k = tract_mask(cons)
                pe = 10.0 ** (-qual / 10.0)
                nt += k; nin += int(m.sum()); nout += int((~m).sum()); ein += int(err[m].sum()); eout += int(err[~m].sum())
                pin += float(pe[m].sum()); pout += float(pe[~m].sum()); nreads += 1
            print(f"{'lowcx ' if tpl else 'random'} {P:2d} passes: reads {nreads:4d}  tracts {nt:6d}  tract bases {nin:8d}  errors in tracts {ein:6d} = {ein / max(nt, 1):.3f} per tract (pred {pin:8.1f}, found/pred {ein / max(pin, 1e-9):.2f})"
                  f" | elsewhere: bases {nout:8d} errors {eout:6d} (pred {pout:8.1f}, found/pred {eout / max(pout, 1e-9):.2f})", flush=True)
    if h is not None: h.close()


if __name__ == "__main__":
    main()
