"""First-principles tests of the CPU restatement (the reference ships no tests or golden vectors:
SURVEY.md §4, §8c).  These are the checks the public unanimity unit tests made, re-created:
brute-force HMM, alpha/beta agreement, mutation equivalence, normalisation."""
import numpy as np
import pytest

from ccs_amd import api
import oracle_lib as O


@pytest.fixture(scope="module")
def tabs(built):
    m = api.default_model()
    return O.tables(m, np.array([9.0, 16.0, 8.0, 13.0], np.float32))


def _rand_case(rng, J=None, noisy=True):
    J = J or int(rng.integers(3, 31))
    t = rng.integers(0, 4, J).astype(np.uint8)
    r = []
    for b in t:
        while noisy and rng.random() < 0.08:
            r.append(int(rng.integers(0, 4)))
        if noisy and rng.random() < 0.06:
            continue
        r.append(int(b) if (not noisy or rng.random() > 0.03) else int(rng.integers(0, 4)))
    r = r[:63]
    obs = (np.array(r, np.int64) * 3 + rng.integers(0, 3, len(r))).astype(np.uint8)
    lf = int(rng.integers(0, 5))
    return t, obs, lf


def test_det_log2_exp2_accuracy(built):
    L = O.lib()
    xs = np.concatenate([np.logspace(-37, 37, 4001), [1.0, 2.0, 0.5, 1.4142135, 1.4142137]]).astype(np.float32)
    got = np.array([L.orc_log2f(float(x)) for x in xs])
    assert np.max(np.abs(got - np.log2(xs.astype(np.float64)))) < 2e-5
    assert L.orc_log2f(0.0) == -127.0 and L.orc_log2f(1e-42) == -127.0
    ys = np.linspace(-120, 55, 3001).astype(np.float32)
    e = np.array([L.orc_exp2f(float(y)) for y in ys], np.float64)
    assert np.max(np.abs(e / np.exp2(ys.astype(np.float64)) - 1)) < 2e-6


def test_tables_are_a_probability_model(tabs):
    ME, INS, DL = tabs
    # every state's outgoing mass (match+branch+stick emissions, deletion) sums to 1 (x4 per emitted base)
    tot = ME.sum(1) / 4 + INS.sum(1) / 4 + DL
    assert np.allclose(tot, 1.0, atol=2e-6)
    assert (ME > 0).all() and (INS > 0).all() and (DL > 0).all()
    assert len({tuple(np.round(r, 9)) for r in ME}) == 16   # all 16 dinucleotide contexts distinct


def test_fill_matches_bruteforce_and_alpha_equals_beta(tabs):
    ME, INS, DL = tabs
    rng = np.random.default_rng(1)
    for _ in range(200):
        t, obs, lf = _rand_case(rng)
        a, b = O.window_likelihood(ME, INS, DL, t, lf, obs)
        bf = O.bruteforce_likelihood(ME, INS, DL, t, lf, obs)
        assert bf > 0
        assert abs(a / bf - 1) < 2e-5
        assert abs(b / bf - 1) < 2e-5


def _apply(t, ty, c, x):
    t = list(t)
    if ty == 0:
        t[c] = x
    elif ty == 2:
        t.insert(c, x)
    else:
        del t[c]
    return np.array(t, np.uint8)


def test_mutation_extend_link_equals_full_refill(tabs):
    """LL(mutation) via <=2 extended alpha columns + beta link == LL of the explicitly mutated template,
    for every mutation lane incl. the first / last columns."""
    ME, INS, DL = tabs
    rng = np.random.default_rng(2)
    n = 0
    for _ in range(40):
        t, obs, lf = _rand_case(rng, J=int(rng.integers(3, 30)))
        for m in range(256):
            res, valid, ty, c, x = O.mutation_likelihood(ME, INS, DL, t, lf, obs, m)
            if not valid:
                continue
            t2 = _apply(t, ty, c, x)
            if len(t2) == 0:
                continue
            full, _ = O.window_likelihood(ME, INS, DL, t2, lf, obs)
            assert full > 0
            assert abs(res / full - 1) < 3e-5, (m, ty, c, x, res, full)
            n += 1
    assert n > 4000


def test_mutation_lane_pruning(tabs):
    """deletions only at the first base of a homopolymer run, insertions never duplicate the previous base"""
    ME, INS, DL = tabs
    t = np.array([0, 0, 0, 1, 2, 2, 3], np.uint8)
    obs = (t.astype(np.int64) * 3).astype(np.uint8)
    valid = {}
    for m in range(256):
        _, v, ty, c, x = O.mutation_likelihood(ME, INS, DL, t, 4, obs, m)
        if v:
            valid.setdefault(ty, set()).add((c, x))
    assert {c for c, _ in valid[1]} == {0, 3, 4, 6}
    assert (1, 0) not in valid[2] and (1, 1) in valid[2] and (7, 3) not in valid[2] and (7, 0) in valid[2]
    assert all(x != t[c] for c, x in valid[0]) and len(valid[0]) == 21


def test_polish_fixes_a_draft_error_and_reports_low_qv_without_evidence(tabs):
    ME, INS, DL = tabs
    rng = np.random.default_rng(5)
    truth = rng.integers(0, 4, 26).astype(np.uint8)
    for kind in ("sub", "ins", "del"):
        d = list(truth)
        if kind == "sub":
            d[12] = (d[12] + 1) & 3
        elif kind == "ins":
            d.insert(12, (truth[12] + 2) & 3)
        else:
            del d[12]
        d = np.array(d, np.uint8)
        fwd = (truth.astype(np.int64) * 3 + 1).astype(np.uint8)
        rc = ((3 - truth[::-1]).astype(np.int64) * 3 + 1).astype(np.uint8)
        obs = [fwd, rc, fwd, rc, fwd, rc]
        out = O.polish_window(ME, INS, DL, d, 2, len(d) - 2, 4, 4, obs, [0, 1, 0, 1, 0, 1])
        core = truth[2:-2]
        assert np.array_equal(out["seq"], core), kind
        assert out["nonconv"] == 0 and out["iters"] == 2 and out["nvalid"] == 6
        assert out["qv"].min() > 30
    none = O.polish_window(ME, INS, DL, truth, 2, 24, 4, 4, [None, None], [0, 1])
    assert none["nvalid"] == 0 and none["qv"].max() < 3 and np.array_equal(none["seq"], truth[2:24])


def test_diagonal_band_schedule():
    """Groundwork for the diagonal-band fill of DESIGN.md 8.8 (tools/diag_fill_model.py, a schedule model, not a kernel): filling the band of diagonals in the
    proposed lane / step order — a lane owns two adjacent diagonals and walks a staircase, one value shifted in from a neighbouring lane per step — gives gamma, alpha
    and beta bit for bit as the oracle's column order does, alpha(I, J) equals beta(0, 0), and a typical 26 x 26 window needs 8 lanes per read."""
    import diag_fill_model as M
    rng = np.random.default_rng(5)
    F = np.float32
    for trial in range(12):
        J = int(rng.integers(6, 32)); I = max(1, J + int(rng.integers(-7, 8)))
        ME = rng.random((16, 12)).astype(F) * F(0.3); INS = rng.random((16, 12)).astype(F) * F(0.1); DL = rng.random(16).astype(F) * F(0.1)
        k = rng.integers(0, 16, J + 1); o = rng.integers(0, 12, I + 1)
        dlo, dhi = M.band(I, J, int(rng.integers(0, 4)))
        a = M.fill_by_columns(ME, INS, DL, k, o, I, J, dlo, dhi)
        b = M.fill_by_staircase(ME, INS, DL, k, o, I, J, dlo, dhi)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (trial, I, J, dlo, dhi)
        assert a[1][I, J] > 0 and abs(a[1][I, J] - a[2][0, 0]) <= 1e-5 * a[1][I, J]
    dlo, dhi = M.band(26, 26, 2)
    assert (dhi - dlo + 2) // 2 == 8
