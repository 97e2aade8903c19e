"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them on CPU, the HIP path must reproduce them on the GPU."""
import numpy as np
import pytest

from ccs_amd import api
import oracle_lib as O
import golden_util as G


@pytest.mark.parametrize("case", G.CASES)
def test_oracle_reproduces_golden(built, case):
    batch, exp, draft0, model_bytes = G.load(case)
    m = api.default_model()
    assert bytes(m) == model_bytes.tobytes(), "SYN-1 parameter set changed: regenerate the golden vectors deliberately"
    res = api.Results.allocate(batch)
    O.consensus_batch(m, api.default_opts(), batch, res, nthreads=2)
    G.check(res, exp, qv_tol=0.0)
    assert np.array_equal(O.poa_draft(batch, 0), draft0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.CASES)
def test_gpu_reproduces_golden(built, case):
    batch, exp, draft0, _ = G.load(case)
    h = api.Handle(0)
    res = h.consensus(batch)
    G.check(res, exp)
    assert np.array_equal(h.stage_draft(0), draft0)
    h.close()
