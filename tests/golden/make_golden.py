#!/usr/bin/env python3
"""Generates tests/golden/golden_v2.npz: inputs + expected outputs of the hot path.

The reference mount is documentation-only (no source, binary or test vectors: SURVEY.md §0/§8c), so these
vectors come from this repository's own CPU restatement (oracle/ccs_oracle.c, "parity unpinned") at the
specification version in DESIGN.md §SPEC.  They freeze the specification: any change to the oracle or the
kernels that alters results must regenerate this file deliberately.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from ccs_amd import api  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = [("p3_l300", 3, 3, 300, 101), ("p5_l700", 2, 5, 700, 102), ("p10_l2000", 2, 10, 2000, 103),
         ("mix", 3, (3, 9), (150, 900), 104)]


def main():
    out = {}
    m, o = api.default_model(), api.default_opts()
    for name, n, passes, length, seed in CASES:
        b = api.synth(n, passes, length, seed=seed)
        r = api.Results.allocate(b)
        O.consensus_batch(m, o, b, r)
        for k in ("zmw_id", "snr", "read_off", "base_off", "bases", "pw", "ipd", "flags", "tpl_off", "tpl"):
            out[f"{name}/in/{k}"] = getattr(b, k)
        for k in ("seq_off", "status", "seq_len", "seq", "qual", "raw_qv", "rq", "np_", "ec", "iters", "n_windows"):
            out[f"{name}/out/{k}"] = getattr(r, k)
        out[f"{name}/draft0"] = O.poa_draft(b, 0, o.max_poa_cov)
    out["model_bytes"] = np.frombuffer(bytes(m), np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
