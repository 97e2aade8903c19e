import os

import numpy as np

from ccs_amd import api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v8.npz")
GOLDEN_KIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_kin_v8.npz")
# every case of the file (tests/golden/make_golden.py): four plain shapes, one ZMW at the headline size, and one case per SPEC path that
# plain synthetic data does not take (the generator asserts that the path fired)
CASES = ["p3_l300", "p5_l700", "p10_l2000", "mix", "c2_one", "trim", "split", "fallback", "lastresort", "retry64", "lowcx", "lowcx_rescue", "partial", "split2", "closed_tract"]
PATHS = ["trim", "split", "split_s0", "split_sLd", "fallback", "retry64", "zdrop", "nonconv_win", "poa_wide", "third_draft", "partial_used", "split2", "saturated", "closed_tract"]
OUT_KEYS = ("seq_off", "status", "seq_len", "seq", "qual", "raw_qv", "rq", "np_", "ec", "iters", "n_windows", "fn", "rn")


def spec_version():
    return int(np.load(GOLDEN)["spec_version"][0])


def kin_spec_version():
    return int(np.load(GOLDEN_KIN)["spec_version"][0])


def load(case):
    g = np.load(GOLDEN)
    kin = {k: np.ascontiguousarray(g[f"{case}/in/{k}"]) for k in
           ("zmw_id", "snr", "read_off", "base_off", "bases", "pw", "ipd", "flags", "tpl_off", "tpl")}
    batch = api.Batch(**kin)
    exp = {k: g[f"{case}/out/{k}"] for k in OUT_KEYS}
    exp["paths"] = dict(zip(PATHS, g[f"{case}/paths"].tolist()))
    return batch, exp, g[f"{case}/draft0"], g["model_bytes"]


def check(res, exp, qv_tol=1e-4):
    assert np.array_equal(res.status, exp["status"])
    assert np.array_equal(res.seq_len, exp["seq_len"])
    assert np.array_equal(res.np_, exp["np_"]) and np.array_equal(res.iters, exp["iters"]) and np.array_equal(res.n_windows, exp["n_windows"])
    assert np.array_equal(res.fn, exp["fn"]) and np.array_equal(res.rn, exp["rn"])
    for z in range(len(exp["status"])):
        o, n = int(exp["seq_off"][z]), int(exp["seq_len"][z])
        assert np.array_equal(res.sequence(z), exp["seq"][o:o + n]), f"zmw {z} sequence"
        assert np.array_equal(res.quals(z), exp["qual"][o:o + n]), f"zmw {z} qual"
        assert np.allclose(res.raw(z), exp["raw_qv"][o:o + n], atol=qv_tol, rtol=0), f"zmw {z} raw qv"
    assert np.allclose(res.rq, exp["rq"], atol=1e-6, rtol=0) and np.allclose(res.ec, exp["ec"], atol=1e-6, rtol=0)
