#!/usr/bin/env python3
"""Generates tests/golden/golden_kin_v8.npz: HiFi-kinetics expectations (SPEC DESIGN.md §2 "HiFi kinetics") on top of two
cases of golden_v8.npz (same subreads, pw untouched, ipd replaced by seeded CodecV1 codes over the whole 0..255 range).

Like golden_v8 these vectors come from the repository's own CPU restatement (docs-only reference, "parity unpinned");
they freeze the kinetics specification.   python tests/golden/make_golden_kinetics.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from ccs_amd import api  # noqa: E402
import oracle_lib as O  # noqa: E402
import golden_util as G  # noqa: E402

CASES = [("p5_l700", 201), ("mix", 202), ("partial", 203)]


def main():
    out = {}
    m, o = api.default_model(), api.default_opts()
    o.hifi_kinetics = 1
    for case, seed in CASES:
        b, exp, _, _ = G.load(case)
        b.ipd = np.random.default_rng(seed).integers(0, 256, len(b.bases)).astype(np.uint8)
        r = api.Results.allocate(b, kinetics=True)
        O.consensus_batch(m, o, b, r)
        G.check(r, exp, qv_tol=0.0)                 # kinetics never change the consensus
        out[f"{case}/ipd"] = b.ipd
        out[f"{case}/kin"] = r.kin
        out[f"{case}/fn"] = r.fn
        out[f"{case}/rn"] = r.rn
    out["spec_version"] = np.array([O.spec_version()], np.int32)      # (the kinetics vectors carry the SPEC version like golden_v8.npz)
    assert O.spec_version() == G.spec_version()
    np.savez_compressed(os.path.join(HERE, "golden_kin_v8.npz"), **out)
    print("wrote golden_kin_v8.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
