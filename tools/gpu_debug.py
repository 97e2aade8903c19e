import sys, os, json, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from ccs_amd import api
import oracle_lib as O
h = api.Handle(0)
for (n, P, L, seed) in [(2, 3, 200, 1), (4, 6, 900, 11), (3, 10, 2000, 2)]:
    batch = api.synth(n, P, L, seed=seed)
    h.upload(batch); h.run(); h.sync()
    t = h.timings()
    print('cfg', n, P, L, 'ms', t.setup_ms, t.draft_ms, t.align_ms, t.polish_ms, t.stitch_ms, flush=True)
    res = h.download()
    ref = api.Results.allocate(batch); O.consensus_batch(h.model, h.opts, batch, ref)
    for z in range(n):
        d = h.stage_draft(z); dr = O.poa_draft(batch, z, h.opts.max_poa_cov)
        print(' z', z, 'draft eq', np.array_equal(d, dr), len(d), len(dr), 'status', res.status[z], ref.status[z], 'np', res.np_[z], ref.np_[z],
              'nw', res.n_windows[z], ref.n_windows[z], 'iters', res.iters[z], ref.iters[z], 'len', res.seq_len[z], ref.seq_len[z], flush=True)
        if not np.array_equal(d, dr):
            k = next((i for i in range(min(len(d), len(dr))) if d[i] != dr[i]), None); print('   first draft diff at', k)
            continue
        wb = h.stage_windows(z); print('   windows eq', np.array_equal(wb, O.windows(dr)))
        r0 = int(batch.read_off[z])
        need = sorted({0, len(d)} | {int(b) - 2 for b in wb[1:-1]} | {int(b) + 2 for b in wb[1:-1]})
        for r in range(r0, int(batch.read_off[z+1])):
            bases, _ = batch.read(r); rev = (batch.flags[r] & 1) != (batch.flags[r0] & 1)
            rs_ref, v_ref, sc_ref = O.align(O.orient(bases, rev), dr)
            rs, v, sc = h.stage_align(r, len(d))
            ok = (v, sc) == (v_ref, sc_ref) and (not v or [int(rs[c]) for c in need] == [int(rs_ref[c]) for c in need])
            if not ok:
                bad = [c for c in need if rs[c] != rs_ref[c]][:5]
                print('   read', r, 'align mismatch', v, v_ref, sc, sc_ref, 'cols', bad, [int(rs[c]) for c in bad], [int(rs_ref[c]) for c in bad])
        s, sr = res.sequence(z), ref.sequence(z)
        print('   seq eq', np.array_equal(s, sr), 'qv maxdiff', float(np.max(np.abs(res.raw(z)[:min(len(s),len(sr))] - ref.raw(z)[:min(len(s),len(sr))]))) if len(s) and len(sr) else None, 'rq', res.rq[z], ref.rq[z], 'ec', res.ec[z], ref.ec[z])
        if not np.array_equal(s, sr):
            k = next((i for i in range(min(len(s), len(sr))) if s[i] != sr[i]), None); print('   first seq diff at', k)
