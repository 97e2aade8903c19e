#!/usr/bin/env python3
"""Synthetic ZMWs the C generator (ccsx_synth_generate) does not make: LOW-COMPLEXITY templates (homopolymer runs of 5-60 bases, 2-4-mer
tandem repeats of 20-500 bp: docs/faq/low-complexity.md:11-18) and OFF-MODEL error channels (rates scaled by `channel`, indels boosted
inside homopolymers by `hp_boost`) — the inputs of the low-complexity parity fuzz (tests/test_gpu_parity.py) and of the off-model
sweeps of every SPEC approximation (tools/acc_eval.py, profiles/r03_offmodel.txt).  Pure numpy, independent of the library's generator:
the SPEC approximations were tuned on the C generator's channel, these inputs are the non-circular evidence available offline."""
from __future__ import annotations

import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from ccs_amd import api  # noqa: E402


def lowcx_template(rng, L):
    """random sequence interleaved with homopolymer runs (5-60) and tandem repeats (unit 2-4, 20-500 bp): about half of the bases"""
    parts, n = [], 0
    while n < L:
        kind = rng.integers(0, 4)
        if kind == 0:
            seg = rng.integers(0, 4, rng.integers(20, 200), dtype=np.uint8)
        elif kind == 1:
            seg = np.full(rng.integers(5, 61), rng.integers(0, 4), np.uint8)
        else:
            u = rng.integers(2, 5)
            unit = rng.integers(0, 4, u, dtype=np.uint8)
            while len(set(unit.tolist())) == 1:
                unit = rng.integers(0, 4, u, dtype=np.uint8)
            total = int(rng.integers(20, 501)) if kind == 2 else int(rng.integers(20, 80))
            seg = np.tile(unit, total // u + 1)[:total]
        parts.append(seg); n += len(seg)
    return np.concatenate(parts)[:L].astype(np.uint8)


def sequence_read(rng, t, channel=1.0, hp_boost=1.0):
    """one pass over template t through the SURVEY.md §8d channel x `channel`: P(del) .04, P(sub) .01, insertion before a base .06 (half a
    copy of that base = branch, half random); pulse widths {.3 .3 .4} (inserted bases {.6 .25 .15}); indels x hp_boost inside homopolymers"""
    L = len(t)
    hp = np.zeros(L, bool)
    hp[1:] |= t[1:] == t[:-1]; hp[:-1] |= t[:-1] == t[1:]
    boost = np.where(hp, hp_boost, 1.0)
    dele = rng.random(L) < 0.04 * channel * boost
    sub = rng.random(L) < 0.01 * channel
    ins = rng.random(L) < 0.06 * channel * boost
    main = np.where(sub, (t + rng.integers(1, 4, L)) & 3, t).astype(np.uint8)
    insb = np.where(rng.random(L) < 0.5, t, rng.integers(0, 4, L)).astype(np.uint8)
    pwm = np.searchsorted([0.3, 0.6], rng.random(L)) + 1
    pwi = np.searchsorted([0.6, 0.85], rng.random(L)) + 1
    bases = np.stack([insb, main], 1).ravel()
    pw = np.stack([pwi, pwm], 1).ravel().astype(np.uint8)
    keep = np.stack([ins, ~dele], 1).ravel()
    return bases[keep], pw[keep]


def make(n, passes, length, seed, channel=1.0, tpl=None, hp_boost=1.0):
    """an api.Batch of n ZMWs.  passes / length: int or (lo, hi).  tpl None + channel 1 + hp_boost 1 = the library's own generator."""
    if tpl is None and channel == 1.0 and hp_boost == 1.0:
        return api.synth(n, passes, length, seed=seed)
    rng = np.random.default_rng(seed)
    plo, phi = (passes, passes) if isinstance(passes, int) else passes
    llo, lhi = (length, length) if isinstance(length, int) else length
    zmw_id, snr, read_off, base_off, flags, tpls, tpl_off = [], [], [0], [0], [], [], [0]
    bases, pws = [], []
    for z in range(n):
        P = int(rng.integers(plo, phi + 1))
        L = int(round(np.exp(rng.uniform(np.log(llo), np.log(lhi))))) if lhi > llo else llo
        t = lowcx_template(rng, L) if tpl == "lowcx" else rng.integers(0, 4, L, dtype=np.uint8)
        tpls.append(t); tpl_off.append(tpl_off[-1] + L)
        zmw_id.append(z)
        snr.append(np.maximum(4.0, np.array([9.0, 16.0, 8.0, 13.0]) * (1 + 0.1 * rng.standard_normal(4))))
        for k in range(P):
            b, p = sequence_read(rng, t, channel, hp_boost)
            if k & 1:
                b, p = (3 - b[::-1]).astype(np.uint8), p[::-1]
            bases.append(b); pws.append(p); flags.append(k & 1)
            base_off.append(base_off[-1] + len(b))
        read_off.append(read_off[-1] + P)
    nb = base_off[-1]
    return api.Batch(np.array(zmw_id, np.int32), np.ascontiguousarray(np.array(snr, np.float32)), np.array(read_off, np.int32),
                     np.array(base_off, np.int64), np.ascontiguousarray(np.concatenate(bases), np.uint8),
                     np.ascontiguousarray(np.concatenate(pws), np.uint8), rng.integers(1, 61, nb).astype(np.uint8),
                     np.array(flags, np.uint8), tpl_off=np.array(tpl_off, np.int64), tpl=np.concatenate(tpls).astype(np.uint8))
