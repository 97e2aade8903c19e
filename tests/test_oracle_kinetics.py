"""First-principles checks of the HiFi-kinetics SPEC on the CPU oracle.

Reference behaviour: docs/faq/kinetics.md:8-18 (averaged kinetics per strand, forward = orientation of SEQ) and the
tag table docs/faq/bam-output.md:13-23 (fi fp ri rp are CodecV1 byte arrays, fn / rn pass counts).
"""
import numpy as np
import pytest

import oracle_lib as O
from ccs_amd import api


def test_codec_v1_roundtrip_and_rounding():
    dec = [O.codec_decode(c) for c in range(256)]
    assert dec[:64] == list(range(64))
    assert dec[64] == 64 and dec[127] == 190 and dec[128] == 192 and dec[191] == 444 and dec[192] == 448 and dec[255] == 952
    assert all(b > a for a, b in zip(dec, dec[1:]))                     # strictly monotone
    for c in range(256):
        assert O.codec_encode(dec[c]) == c                              # representable values are fixed points
    for f in range(0, 1200):
        c = O.codec_encode(f)
        err = abs(dec[c] - f) if f <= 952 else 0
        best = min(abs(d - f) for d in dec) if f <= 952 else 0
        assert err == best, (f, c)                                      # nearest representable value
        if f <= 952 and c > 0 and abs(dec[c - 1] - f) == err:
            assert dec[c] > f                                           # ties go up
    assert O.codec_encode(5000) == 255


def _rc(t):
    return (3 - np.asarray(t)[::-1]).astype(np.uint8)


def test_kinetics_read_perfect_and_indels():
    rng = np.random.default_rng(5)
    t = rng.integers(0, 4, 26).astype(np.uint8)
    ipd = rng.integers(0, 256, 26).astype(np.uint8)
    pw = rng.integers(0, 256, 26).astype(np.uint8)
    si, sp, cn = O.kinetics_read(t, t, ipd, pw, 0)
    assert (cn == 1).all()
    assert (si == [O.codec_decode(c) for c in ipd]).all() and (sp == [O.codec_decode(c) for c in pw]).all()
    # reverse-strand read: template handed over in read orientation, sums come back in forward columns
    tr = _rc(t)
    si, sp, cn = O.kinetics_read(tr, tr, ipd, pw, 1)
    assert (cn == 1).all()
    assert (si == [O.codec_decode(c) for c in ipd[::-1]]).all()
    # a deleted template column gets no observation, every other column keeps its own base
    t2 = np.array([0, 1, 2, 3] * 6, np.uint8)
    keep = np.ones(len(t2), bool); keep[9] = False
    rb = t2[keep]
    ip2 = np.arange(len(rb), dtype=np.uint8) + 1
    si, _, cn = O.kinetics_read(t2, rb, ip2, ip2, 0)
    assert cn[9] == 0 and cn.sum() == len(rb)
    assert (si[keep] == ip2).all()
    # an inserted read base is attributed to no column
    rb3 = np.insert(t2, 12, (t2[12] + 2) & 3)
    ip3 = np.arange(len(rb3), dtype=np.uint8) + 1
    si, _, cn = O.kinetics_read(t2, rb3, ip3, ip3, 0)
    assert (cn == 1).all() and 13 not in si.tolist() and si.sum() == ip3.sum() - 13
    # a mismatching base is not attributed (matches only)
    rb4 = t2.copy(); rb4[7] = (rb4[7] + 1) & 3
    si, _, cn = O.kinetics_read(t2, rb4, np.full(len(rb4), 9, np.uint8), np.full(len(rb4), 2, np.uint8), 0)
    assert cn[7] == 0 and cn.sum() == len(t2) - 1


def _base_coded(batch):
    """Kinetics that are a pure function of the sequenced base: IPD code 10+10*base, PW code 1+(base&1)*2."""
    b = batch
    b.ipd = (10 + 10 * b.bases).astype(np.uint8)
    b.pw = (1 + (b.bases & 1) * 2).astype(np.uint8)
    return b


def test_whole_path_kinetics_follow_the_strand():
    """Only bases that agree with the consensus are averaged, so base-coded kinetics come back exactly: the forward
    planes carry the code of SEQ's base, the reverse planes the code of its complement (kinetics.md:11-13)."""
    batch = _base_coded(api.synth(6, 9, 700, seed=31))
    model, opts = api.default_model(), api.default_opts()
    opts.hifi_kinetics = 1
    res = O.consensus_batch(model, opts, batch, api.Results.allocate(batch, kinetics=True))
    plain = O.consensus_batch(model, opts, batch, api.Results.allocate(batch))
    assert (res.seq == plain.seq).all() and (res.raw_qv == plain.raw_qv).all()     # kinetics never touch the consensus
    covered = 0
    for z in range(batch.n_zmw):
        assert res.status[z] == 0
        assert res.fn[z] + res.rn[z] == res.np_[z] and res.fn[z] >= 4 and res.rn[z] >= 4
        s = res.sequence(z).astype(int)
        fi, fp, ri, rp = res.kinetics(z).astype(int)
        ok_f, ok_r = fi > 0, ri > 0
        assert (fi[ok_f] == 10 + 10 * s[ok_f]).all() and (fp[ok_f] == 1 + (s[ok_f] & 1) * 2).all()
        assert (ri[ok_r] == 10 + 10 * (3 - s[ok_r])).all() and (rp[ok_r] == 1 + ((3 - s[ok_r]) & 1) * 2).all()
        covered += ok_f.mean() + ok_r.mean()
    assert covered / (2 * batch.n_zmw) > 0.999                                     # ~every position has an observation


def test_kinetics_average_is_rounded_mean():
    """Random kinetics: each plane value is the CodecV1 code of the rounded integer mean of some subset of the pass
    values, so it must lie between the min and max decoded frame seen at that strand (coarse but model-free)."""
    batch = api.synth(3, 8, 400, seed=77)
    rng = np.random.default_rng(1)
    batch.ipd = rng.integers(1, 200, len(batch.bases)).astype(np.uint8)
    model, opts = api.default_model(), api.default_opts()
    res = O.consensus_batch(model, opts, batch, api.Results.allocate(batch, kinetics=True))
    for z in range(batch.n_zmw):
        fi = res.kinetics(z)[0]
        dec = np.array([O.codec_decode(c) for c in fi])
        assert dec.max() <= O.codec_decode(199) and (fi > 0).mean() > 0.99
        # averaging 4 passes shrinks the spread: the std of the means is well below the std of single values
        single = np.array([O.codec_decode(c) for c in batch.ipd[:5000]])
        assert dec.std() < 0.7 * single.std()
