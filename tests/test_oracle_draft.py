"""POA draft / alignment / windowing known answers and recover-the-truth fuzz (item 4)."""
import numpy as np
import pytest

from ccs_amd import api
import oracle_lib as O


def _batch_from_reads(reads, flags=None):
    n = len(reads)
    flags = np.zeros(n, np.uint8) if flags is None else np.asarray(flags, np.uint8)
    bo = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    bases = np.concatenate(reads).astype(np.uint8)
    return api.Batch(np.zeros(1, np.int32), np.array([[9, 16, 8, 13]], np.float32), np.array([0, n], np.int32), bo, bases,
                     np.full(len(bases), 2, np.uint8), np.ones(len(bases), np.uint8), flags)


def test_poa_known_answers(built):
    rng = np.random.default_rng(0)
    t = rng.integers(0, 4, 300).astype(np.uint8)
    # identical reads -> the read
    assert np.array_equal(O.poa_draft(_batch_from_reads([t, t, t]), 0), t)
    # majority vote on a SNP
    s = t.copy(); s[100] = (s[100] + 1) & 3
    assert np.array_equal(O.poa_draft(_batch_from_reads([s, t, t]), 0), t)
    assert np.array_equal(O.poa_draft(_batch_from_reads([t, s, s]), 0), s)
    # one read with an insertion / one with a deletion are out-voted
    ins = np.insert(t, 150, (t[150] + 2) & 3)
    dele = np.delete(t, 200)
    assert np.array_equal(O.poa_draft(_batch_from_reads([ins, t, t]), 0), t)
    assert np.array_equal(O.poa_draft(_batch_from_reads([t, dele, t, t]), 0), t)
    # reverse-strand passes are oriented by their flag
    rc = (3 - t[::-1]).astype(np.uint8)
    assert np.array_equal(O.poa_draft(_batch_from_reads([t, rc, t], [0, 1, 0]), 0), t)
    assert np.array_equal(O.poa_draft(_batch_from_reads([rc, t, rc], [1, 0, 1]), 0), rc)
    # max_poa_cov=1 -> the first read verbatim
    assert np.array_equal(O.poa_draft(_batch_from_reads([s, t, t]), 0, max_poa_cov=1), s)


def test_align_entry_rows(built):
    rng = np.random.default_rng(1)
    d = rng.integers(0, 4, 500).astype(np.uint8)
    rs, v, sc = O.align(d, d)
    assert v == 1 and sc == 3 * 500 and np.array_equal(rs, np.arange(501))
    # 3 extra read bases before draft position 250: they are consumed while waiting AT state 250,
    # so entry rows are unchanged up to 250 and shifted by 3 after it
    x = (d[250] + 2) & 3
    r = np.concatenate([d[:250], [x, x, x], d[250:]]).astype(np.uint8)
    rs, v, _ = O.align(r, d)
    assert v == 1 and rs[250] == 250 and rs[251] == 254 and rs[500] == 503 and rs[0] == 0
    # a deleted draft base: entry rows stall
    r = np.delete(d, 100)
    rs, v, _ = O.align(r, d)
    assert v == 1 and rs[500] == 499 and (np.diff(rs) >= 0).all() and (np.diff(rs) <= 1).all()
    # junk does not align
    junk = rng.integers(0, 4, 500).astype(np.uint8)
    assert O.align(junk, d)[1] == 0
    # a read far longer than the band can absorb fails cleanly
    assert O.align(np.concatenate([d, d]), d)[1] == 0


def test_windows_properties(built):
    rng = np.random.default_rng(2)
    for L in [5, 28, 29, 30, 51, 52, 100, 1000, 10007]:
        d = rng.integers(0, 4, L).astype(np.uint8)
        d[40:60] = 1 if L > 60 else d[40:60]      # a long homopolymer
        b = O.windows(d)
        assert b[0] == 0 and b[-1] == L and (np.diff(b) > 0).all()
        core = np.diff(b)
        assert core[:-1].min(initial=22) >= 19 and core[:-1].max(initial=22) <= 25 and core[-1] <= 28
        assert (core + 4 <= 31).all() or len(core) == 1

        def bad(x):                                # SPEC: a break inside a tandem repeat of period 1..4
            return any(x - p >= 0 and x + p <= L and np.array_equal(d[x - p:x], d[x:x + p]) for p in range(1, 5))
        for k, x in enumerate(b[1:-1]):            # a bad break is only allowed when all seven candidates around cur+22 are bad
            if bad(x):
                cur = b[k]
                assert x == cur + 22 and all(bad(cur + 22 + o) for o in (0, 1, -1, 2, -2, 3, -3))


def test_windows_avoid_simple_repeats(built):
    """docs/how-does-ccs-work.md:58-60: windows do not break homopolymers ... 4-mer repeats when a +-3 shift avoids it"""
    rng = np.random.default_rng(5)
    for unit in ("A", "TG", "CAG", "ACGT"):
        d = rng.integers(0, 4, 200).astype(np.uint8)
        rep = np.array(["ACGT".index(c) for c in unit * 3], np.uint8)[:5]      # a short tandem repeat across the target break
        d[20:20 + len(rep)] = rep
        d[19] = (rep[0] + 1) & 3 if unit != "A" else 1
        b = O.windows(d)
        x = int(b[1])
        assert 19 <= x <= 25
        assert not any(np.array_equal(d[x - p:x], d[x:x + p]) for p in range(1, 5))


@pytest.mark.parametrize("passes,length,max_err", [(8, 600, 2), (12, 1500, 2)])
def test_recover_the_truth(built, passes, length, max_err):
    batch = api.synth(6, passes, length, seed=123)
    m, o = api.default_model(), api.default_opts()
    res = api.Results.allocate(batch)
    O.consensus_batch(m, o, batch, res, nthreads=4)
    import difflib
    total = 0
    for z in range(batch.n_zmw):
        tpl = batch.tpl[batch.tpl_off[z]:batch.tpl_off[z + 1]]
        s = res.sequence(z)
        assert res.status[z] in (0, 7)
        ops = [op for op in difflib.SequenceMatcher(None, bytes(s + 65), bytes(tpl + 65), autojunk=False).get_opcodes() if op[0] != "equal"]
        assert len(ops) <= max_err
        total += len(ops)
        # rq is 1 - mean(p_err) of the emitted bases (docs/how-does-ccs-work.md:104-106)
        p = 10.0 ** (-res.raw(z).astype(np.float64) / 10.0)
        assert abs((1 - p.mean()) - res.rq[z]) < 2e-4
        assert res.np_[z] == passes and abs(res.ec[z] - passes) < 0.5
    assert total <= 4


def test_status_taxonomy(built):
    m, o = api.default_model(), api.default_opts()
    two = api.synth(2, 2, 200, seed=1)
    r = api.Results.allocate(two); O.consensus_batch(m, o, two, r)
    assert list(r.status) == [1, 1] and list(r.seq_len) == [0, 0]          # TOO_FEW_PASSES
    b = api.synth(1, 4, 300, seed=2)
    o2 = api.default_opts(); o2.max_length = 100
    r = api.Results.allocate(b); O.consensus_batch(m, o2, b, r)
    assert r.status[0] == 6                                                # TOO_LONG
    o3 = api.default_opts(); o3.min_length = 1000
    r = api.Results.allocate(b); O.consensus_batch(m, o3, b, r)
    assert r.status[0] == 5                                                # TOO_SHORT
    # 3 of 4 reads replaced by junk -> TOO_MANY_UNUSABLE
    rng = np.random.default_rng(3)
    for k in (1, 2, 3):
        a, e = int(b.base_off[k]), int(b.base_off[k + 1])
        b.bases[a:e] = rng.integers(0, 4, e - a, dtype=np.uint8)
    r = api.Results.allocate(b); O.consensus_batch(m, o, b, r)
    assert r.status[0] == 3 and r.np_[0] <= 2


def test_accuracy_rises_with_passes(built):
    """[DOC] docs/img/ccs-acc.png via docs/faq/accuracy-vs-passes.md:13 — about Q20 at 4-5 passes, about Q30 at 10,
    higher beyond; subreads ~90 % accurate (docs/how-does-ccs-work.md:46).  Statistical acceptance on synthetic
    data: empirical quality is monotone in the pass count and lands in broad bands around those figures, and the
    predicted accuracy (rq) tracks the empirical one."""
    import difflib
    m, o = api.default_model(), api.default_opts()
    o.min_rq = 0.0
    emp = {}
    for passes in (4, 10, 20):
        batch = api.synth(16, passes, 1000, seed=500 + passes)
        res = api.Results.allocate(batch)
        O.consensus_batch(m, o, batch, res, nthreads=4)
        errs = bases = 0
        for z in range(batch.n_zmw):
            tpl = batch.tpl[batch.tpl_off[z]:batch.tpl_off[z + 1]]
            s = res.sequence(z)
            sm = difflib.SequenceMatcher(None, bytes(s + 65), bytes(tpl + 65), autojunk=False)
            errs += sum(max(i2 - i1, j2 - j1) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != "equal")
            bases += len(tpl)
        emp[passes] = (errs / bases, float(1.0 - res.rq.mean()))
    q = {p: -10 * np.log10(max(e[0], 1e-6)) for p, e in emp.items()}
    assert q[4] < q[10] <= q[20] + 1e-9
    assert 12 <= q[4] <= 28 and 25 <= q[10] <= 45 and q[20] >= 30
    for p, (e_emp, e_pred) in emp.items():                     # rq is calibrated within a factor ~4 of the truth
        assert e_pred < 4 * max(e_emp, 2e-4) + 1e-4 and e_emp < 4 * e_pred + 2e-3, (p, e_emp, e_pred)


def _junk_backbones_batch():
    """7 passes per ZMW; pass 0 and the pass whose length is the median are junk, the other five are good: the first draft and the
    fallback draft both start from junk, the last resort (SPEC "draft cascade") takes a good pass as the draft itself"""
    from ccs_amd import api
    rng = np.random.default_rng(11)
    base = api.synth(6, 7, (900, 1400), seed=77)
    for z in (1, 2, 4):
        r0 = int(base.read_off[z])
        lens = np.diff(base.base_off[r0:r0 + 8])
        med = sorted((int(l), q) for q, l in enumerate(lens))[7 // 2][1]        # element n/2 of the sorted lengths
        for q in {0, med}:
            a, b = int(base.base_off[r0 + q]), int(base.base_off[r0 + q + 1])
            base.bases[a:b] = rng.integers(0, 4, b - a, dtype=np.uint8)
    return base


def test_last_resort_draft_takes_a_pass_as_the_draft(built):
    """docs/faq/accuracy-vs-passes.md:41-46 (a cascade of draft generators, from fast and unstable to slow and robust)"""
    from ccs_amd import api
    import oracle_lib as O
    batch = _junk_backbones_batch()
    o = api.default_opts()
    res = api.Results.allocate(batch)
    O.counts_reset()
    O.consensus_batch(api.default_model(), o, batch, res)
    c = O.counts()
    assert c["third_draft"] >= 2 and c["fallback"] >= c["third_draft"]
    ok = (res.status == 0) | (res.status == 7)
    assert ok.all(), res.status                                   # with the cascade every ZMW gets a consensus
    for z in (1, 2, 4):                                           # ... that is close to the truth, in the orientation of a good pass
        t = batch.tpl[batch.tpl_off[z]:batch.tpl_off[z + 1]]
        s = res.sequence(z)
        d = min(O.edit_distance(s, t), O.edit_distance(s, (3 - t[::-1]).astype(np.uint8)))
        assert d <= 12, (z, d)
        assert res.np_[z] == 5
    o.no_fallback_draft = 1
    res2 = api.Results.allocate(batch)
    O.consensus_batch(api.default_model(), o, batch, res2)
    assert set(res2.status[[1, 2, 4]].tolist()) <= {2, 3}        # DRAFT_FAILURE / TOO_MANY_UNUSABLE without it


def partial_pass_batch(n=6, seed=55, nfull=6, length=(1200, 2500)):
    """nfull full-length passes + two partial ones per ZMW (made by truncating two more passes): pass nfull keeps its first 60 % (it starts
    at an adapter: the LAST subread of a polymerase read), pass nfull+1 its last 50 % (it ends at an adapter: the FIRST subread;
    flag bit 2).  flags: bit 0 strand, bit 1 partial, bit 2 the adapter is at the pass's end."""
    from ccs_amd import api
    base = api.synth(n, nfull + 2, length, seed=seed)
    bases, pw, ipd, off, flags = [], [], [], [0], base.flags.copy()
    for r in range(int(base.read_off[-1])):
        a, b = int(base.base_off[r]), int(base.base_off[r + 1])
        z = int(np.searchsorted(base.read_off, r, side="right") - 1)
        q = r - int(base.read_off[z])
        if q == nfull: b = a + (6 * (b - a)) // 10; flags[r] |= 2
        elif q == nfull + 1: a = b - (b - a) // 2; flags[r] |= 2 | 4
        bases.append(base.bases[a:b]); pw.append(base.pw[a:b]); ipd.append(base.ipd[a:b]); off.append(off[-1] + (b - a))
    return api.Batch(base.zmw_id, base.snr, base.read_off, np.array(off, np.int64), np.concatenate(bases), np.concatenate(pw),
                     np.concatenate(ipd), flags, base.tpl_off, base.tpl)


def test_partial_passes_serve_the_polish_but_are_not_passes(built):
    """docs/faq/accuracy-vs-passes.md:26-29: np = full-length passes, ec ~ np + 1 because the polish also uses the partial ones"""
    from ccs_amd import api
    import oracle_lib as O
    batch = partial_pass_batch()
    full_only = api.synth(6, 8, (1200, 2500), seed=55)
    keep = [r for z in range(6) for r in range(int(full_only.read_off[z]), int(full_only.read_off[z]) + 6)]
    fo = api.Batch(full_only.zmw_id, full_only.snr, np.arange(0, 37, 6, dtype=np.int32),
                   np.concatenate([[0], np.cumsum([int(full_only.base_off[r + 1] - full_only.base_off[r]) for r in keep])]).astype(np.int64),
                   np.concatenate([full_only.bases[int(full_only.base_off[r]):int(full_only.base_off[r + 1])] for r in keep]),
                   np.concatenate([full_only.pw[int(full_only.base_off[r]):int(full_only.base_off[r + 1])] for r in keep]),
                   np.concatenate([full_only.ipd[int(full_only.base_off[r]):int(full_only.base_off[r + 1])] for r in keep]),
                   full_only.flags[keep].copy(), full_only.tpl_off, full_only.tpl)
    o, m = api.default_opts(), api.default_model()
    res, ref = api.Results.allocate(batch), api.Results.allocate(fo)
    O.counts_reset()
    O.consensus_batch(m, o, batch, res)
    assert O.counts()["partial_used"] == 12                       # both partial passes of every ZMW found their place
    O.consensus_batch(m, o, fo, ref)
    assert np.array_equal(res.np_, ref.np_) and (res.np_ == 6).all()          # they are not passes ...
    assert np.array_equal(res.fn, ref.fn) and np.array_equal(res.rn, ref.rn)
    assert (res.ec > ref.ec + 0.8).all() and (res.ec < ref.ec + 1.3).all()   # ... but 0.6 + 0.5 of a pass more coverage per window
    assert ((res.status == 0) | (res.status == 7)).all()
    for z in range(6):                                            # the draft does not see them
        assert np.array_equal(O.poa_draft(batch, z), O.poa_draft(fo, z))
    assert res.rq.mean() > ref.rq.mean()                          # more evidence, higher predicted accuracy
    err = lambda r, b: sum(O.edit_distance(r.sequence(z), b.tpl[b.tpl_off[z]:b.tpl_off[z + 1]]) for z in range(6))
    assert err(res, batch) <= err(ref, fo) + 1


def test_more_than_64_passes_are_used(built):
    """SPEC v5 (docs/faq/accuracy-vs-passes.md:49-52, `--top-passes 0` = unlimited): up to 255 passes of a ZMW are used; np is the mode over
    windows of the passes used, so it reports more than 64; accuracy does not get worse with the extra passes; 300 passes are capped at 255"""
    o = api.default_opts(); o.top_passes = 0
    m = api.default_model()
    b = api.synth(1, 100, 600, seed=71)
    r = api.Results.allocate(b)
    O.consensus_batch(m, o, b, r)
    assert r.status[0] == 0 and 90 <= r.np_[0] <= 100
    o64 = api.default_opts(); o64.top_passes = 64
    r64 = api.Results.allocate(b)
    O.consensus_batch(m, o64, b, r64)
    assert r64.np_[0] <= 64 and r.rq[0] >= r64.rq[0] - 1e-6
    big = api.synth(1, 300, 200, seed=72)
    rb = api.Results.allocate(big)
    O.consensus_batch(m, o, big, rb)
    assert rb.np_[0] == 255


def test_polish_seam_of_the_oracle_equals_its_fused_path(built):
    """orc_polish_zmw (the checker's restatement of ccsx_polish_batch) on the first draft reproduces the fused oracle wherever the first draft is final;
    CCSX_QV_ONLY returns the sequence as given after exactly one round per window"""
    from ccs_amd import api
    import oracle_lib as O
    b = api.synth(4, 8, 900, seed=77)
    m, o = api.default_model(), api.default_opts()
    fused = O.consensus_batch(m, o, b, api.Results.allocate(b))
    d = api.Drafts.allocate(b)
    for z in range(b.n_zmw):
        d.set_draft(z, O.poa_draft(b, z, o.max_poa_cov), backbone=0)
    split = O.polish_batch(m, o, b, d, api.Results.allocate(b))
    assert np.array_equal(split.status, fused.status) and np.array_equal(split.iters, fused.iters)
    for z in range(b.n_zmw):
        assert np.array_equal(split.sequence(z), fused.sequence(z)) and np.array_equal(split.raw(z), fused.raw(z))
    d2 = api.Drafts.allocate(b)
    for z in range(b.n_zmw):
        d2.set_draft(z, fused.sequence(z), backbone=0)
    qv = O.polish_batch(m, o, b, d2, api.Results.allocate(b), flags=api.QV_ONLY)
    assert (qv.iters == qv.n_windows).all()
    full = O.polish_batch(m, o, b, d2, api.Results.allocate(b))
    for z in range(b.n_zmw):
        assert np.array_equal(qv.sequence(z), fused.sequence(z))
        # the windows of this run are cut from the consensus, not from the draft the fused run polished, so its QVs are those of a polish that STARTS from the
        # consensus (equal wherever that polish changes nothing), and close to — not identical with — the fused run's: same phred at most bases, same rq to 1e-3
        if np.array_equal(full.sequence(z), fused.sequence(z)): assert np.array_equal(qv.quals(z), full.quals(z))
        assert (qv.quals(z) == fused.quals(z)).mean() > 0.85 and abs(float(qv.rq[z]) - float(fused.rq[z])) < 1e-3
