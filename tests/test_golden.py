"""Committed golden vectors (tests/golden/golden_v7.npz, made by tests/golden/make_golden.py from the
oracle at SPEC version 7): the oracle must keep reproducing them on CPU, the HIP path must reproduce them on the GPU; a library or
oracle of another SPEC version is refused (SPEC drift fails loudly: VERDICT r02 item 3e)."""
import numpy as np
import pytest

from ccs_amd import api
import oracle_lib as O
import golden_util as G


def test_spec_versions_agree(built):
    """the golden vectors, the library (ccsx_spec_version) and the oracle (ORC_SPEC_VERSION) carry the same SPEC version"""
    assert G.spec_version() == G.kin_spec_version() == api.lib().ccsx_spec_version() == O.spec_version()
    assert sorted(np.load(G.GOLDEN)["cases"].tolist()) == sorted(G.CASES)


@pytest.mark.parametrize("case", G.CASES)
def test_oracle_reproduces_golden(built, case):
    batch, exp, draft0, model_bytes = G.load(case)
    m = api.default_model()
    assert bytes(m) == model_bytes.tobytes(), "SYN-1 parameter set changed: regenerate the golden vectors deliberately"
    res = api.Results.allocate(batch)
    O.counts_reset()
    O.consensus_batch(m, api.default_opts(), batch, res, nthreads=2)
    c = O.counts()
    G.check(res, exp, qv_tol=0.0)
    assert {k: c[k] for k in G.PATHS} == exp["paths"], "the case no longer takes the SPEC paths it was built for"
    assert np.array_equal(O.poa_draft(batch, 0), draft0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.CASES)
def test_gpu_reproduces_golden(built, case):
    batch, exp, draft0, _ = G.load(case)
    assert G.spec_version() == api.lib().ccsx_spec_version()
    h = api.Handle(0)
    res = h.consensus(batch)
    G.check(res, exp)
    if case not in ("fallback", "lastresort") or exp["paths"]["fallback"] == 0:
        assert np.array_equal(h.stage_draft(0), draft0)      # (draft0 = the FIRST draft of ZMW 0; after a fallback the stage holds the later one)
    h.close()


# ---- HiFi kinetics (tests/golden/golden_kin_v7.npz, made by tests/golden/make_golden_kinetics.py) --------------------
KIN_CASES = ["p5_l700", "mix", "partial"]


def _kin_case(case):
    import os
    batch, exp, _, _ = G.load(case)
    g = np.load(G.GOLDEN_KIN)
    batch.ipd = np.ascontiguousarray(g[f"{case}/ipd"])
    return batch, exp, g[f"{case}/kin"], g[f"{case}/fn"], g[f"{case}/rn"]


def _check_kin(res, exp, kin, fn, rn):
    G.check(res, exp)
    assert np.array_equal(res.fn, fn) and np.array_equal(res.rn, rn)
    for z in range(len(fn)):
        o, n = int(exp["seq_off"][z]), int(exp["seq_len"][z])
        assert np.array_equal(res.kinetics(z), kin[:, o:o + n]), f"zmw {z} kinetics"


@pytest.mark.parametrize("case", KIN_CASES)
def test_oracle_reproduces_golden_kinetics(built, case):
    batch, exp, kin, fn, rn = _kin_case(case)
    opts = api.default_opts(); opts.hifi_kinetics = 1
    res = api.Results.allocate(batch, kinetics=True)
    O.consensus_batch(api.default_model(), opts, batch, res, nthreads=2)
    _check_kin(res, exp, kin, fn, rn)


@pytest.mark.gpu
@pytest.mark.parametrize("case", KIN_CASES)
def test_gpu_reproduces_golden_kinetics(built, case):
    batch, exp, kin, fn, rn = _kin_case(case)
    opts = api.default_opts(); opts.hifi_kinetics = 1
    h = api.Handle(0, opts=opts)
    try:
        _check_kin(h.consensus(batch), exp, kin, fn, rn)
    finally:
        h.close()
