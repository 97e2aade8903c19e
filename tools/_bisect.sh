cd $GRAFT_REPO_ROOT
for v in 3 4 8; do
  CCSX_EXTRA_FLAGS="-DPW_CHUNK_READS=$v" python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
  echo "== PW_CHUNK_READS=$v"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_path or headline" 2>&1 | tail -4
done
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from ccs_amd import api
import oracle_lib as O
b = api.synth(6, 10, 2000, seed=2)
h = api.Handle(0); res = h.consensus(b)
ref = api.Results.allocate(b); O.consensus_batch(h.model, h.opts, b, ref, nthreads=8)
for z in range(b.n_zmw):
    d = np.nonzero(res.quals(z) != ref.quals(z))[0]
    if len(d): print("zmw", z, "len", res.seq_len[z], "ndiff", len(d), "first", d[:12].tolist(), "gpu", res.raw(z)[d[:6]].tolist(), "ref", ref.raw(z)[d[:6]].tolist())
w = h.stage_windows(0); print("windows zmw0:", w[:12].tolist())
PY
