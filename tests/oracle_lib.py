"""ctypes access to oracle/libccs_oracle.so — the CPU restatement (test infrastructure, parity unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libccs_oracle.so")

NCTX, NOBS, JMAX, IMAX = 16, 12, 31, 63

_lib = None


def build():
    src = os.path.join(ORACLE_DIR, "ccs_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(ORACLE_SO)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.orc_log2f.restype = C.c_float
        L.orc_log2f.argtypes = [C.c_float]
        L.orc_exp2f.restype = C.c_float
        L.orc_exp2f.argtypes = [C.c_float]
        L.orc_bruteforce_likelihood.restype = C.c_double
        L.orc_window_mutation_likelihood.restype = C.c_float
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def tables(model, snr):
    """A0: (ME[16,12], INS[16,12], DL[16]) for one ZMW.  `model` is a ccs_amd.api.Model (same byte layout)."""
    ME = np.zeros((NCTX, NOBS), np.float32)
    INS = np.zeros((NCTX, NOBS), np.float32)
    DL = np.zeros(NCTX, np.float32)
    snr = np.ascontiguousarray(snr, np.float32)
    lib().orc_tables(C.byref(model), _p(snr, C.c_float), _p(ME, C.c_float), _p(INS, C.c_float), _p(DL, C.c_float))
    return ME, INS, DL


def poa_draft(batch, z, max_poa_cov=5):
    r0, r1 = int(batch.read_off[z]), int(batch.read_off[z + 1])
    while r1 > r0 and (batch.flags[r1 - 1] & 2):           # partial passes (a suffix of the ZMW's reads) never enter the draft
        r1 -= 1
    b0 = int(batch.base_off[r0])
    rel = (batch.base_off[r0:r1 + 1] - b0).astype(np.int64)
    bases = np.ascontiguousarray(batch.bases[b0:int(batch.base_off[r1])])
    flags = np.ascontiguousarray(batch.flags[r0:r1])
    maxL = int(np.max(np.diff(rel)))
    dcap = maxL + maxL // 4 + 64
    vcap = (5 * maxL) // 2 + 256
    draft = np.zeros(dcap, np.uint8)
    n = lib().orc_poa_draft(r1 - r0, _p(rel, C.c_int64), _p(bases, C.c_uint8), _p(flags, C.c_uint8), max_poa_cov,
                            vcap, _p(draft, C.c_uint8), dcap)
    return draft[:n].copy()


def orient(bases, rev):
    return (3 - bases[::-1]).astype(np.uint8) if rev else bases.copy()


def align(read_oriented, draft):
    r = np.ascontiguousarray(read_oriented, np.uint8)
    d = np.ascontiguousarray(draft, np.uint8)
    rs = np.zeros(len(d) + 1, np.int32)
    sc = C.c_int32()
    v = lib().orc_align(_p(r, C.c_uint8), len(r), _p(d, C.c_uint8), len(d), _p(rs, C.c_int32), C.byref(sc))
    return rs, v, sc.value


def windows(draft):
    d = np.ascontiguousarray(draft, np.uint8)
    cap = len(d) // 19 + 4
    b = np.zeros(cap, np.int32)
    n = lib().orc_windows(_p(d, C.c_uint8), len(d), _p(b, C.c_int32), cap)
    return b[: n + 1].copy()


def polish_window(ME, INS, DL, tpl, cs, ce, lf, rf, obs_list, strand):
    """obs_list[r]: uint8 array of native-orientation obs codes, or None if the read is unusable here."""
    n = len(obs_list)
    tpl = np.ascontiguousarray(tpl, np.uint8)
    bufs = [np.ascontiguousarray(o if o is not None else np.zeros(1, np.uint8), np.uint8) for o in obs_list]
    ptrs = (C.POINTER(C.c_uint8) * n)(*[_p(b, C.c_uint8) for b in bufs])
    I = np.array([len(o) if o is not None else -1 for o in obs_list], np.int32)
    st = np.ascontiguousarray(strand, np.uint8)
    seq = np.zeros(JMAX + 1, np.uint8)
    perr = np.zeros(JMAX + 1, np.float32)
    qv = np.zeros(JMAX + 1, np.float32)
    ln, nv, nc = C.c_int32(), C.c_int32(), C.c_int32()
    delta = np.zeros(256, np.float32)
    it = lib().orc_polish_window(_p(ME, C.c_float), _p(INS, C.c_float), _p(DL, C.c_float), _p(tpl, C.c_uint8), len(tpl),
                                 int(cs), int(ce), int(lf), int(rf), n, ptrs, _p(I, C.c_int32), _p(st, C.c_uint8), _p(seq, C.c_uint8),
                                 _p(perr, C.c_float), _p(qv, C.c_float), C.byref(ln), C.byref(nv), C.byref(nc),
                                 _p(delta, C.c_float))
    k = ln.value
    return dict(seq=seq[:k].copy(), perr=perr[:k].copy(), qv=qv[:k].copy(), nvalid=nv.value, nonconv=nc.value,
                iters=it, delta=delta)


def window_likelihood(ME, INS, DL, tpl, lf, obs):
    t = np.ascontiguousarray(tpl, np.uint8)
    o = np.ascontiguousarray(obs, np.uint8)
    a, b = C.c_float(), C.c_float()
    lib().orc_window_likelihood(_p(ME, C.c_float), _p(INS, C.c_float), _p(DL, C.c_float), _p(t, C.c_uint8), len(t), lf,
                                _p(o, C.c_uint8), len(o), C.byref(a), C.byref(b))
    return a.value, b.value


def mutation_likelihood(ME, INS, DL, tpl, lf, obs, m):
    t = np.ascontiguousarray(tpl, np.uint8)
    o = np.ascontiguousarray(obs, np.uint8)
    v, ty, c, x = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    res = lib().orc_window_mutation_likelihood(_p(ME, C.c_float), _p(INS, C.c_float), _p(DL, C.c_float), _p(t, C.c_uint8),
                                               len(t), lf, _p(o, C.c_uint8), len(o), m, C.byref(v), C.byref(ty),
                                               C.byref(c), C.byref(x))
    return res, v.value, ty.value, c.value, x.value


def bruteforce_likelihood(ME, INS, DL, tpl, lf, obs):
    t = np.ascontiguousarray(tpl, np.uint8)
    o = np.ascontiguousarray(obs, np.uint8)
    return lib().orc_bruteforce_likelihood(_p(ME, C.c_float), _p(INS, C.c_float), _p(DL, C.c_float), _p(t, C.c_uint8),
                                           len(t), lf, _p(o, C.c_uint8), len(o))


def consensus_batch(model, opts, batch, results, nthreads=1):
    """Whole-path oracle over a ccs_amd.api.Batch into a ccs_amd.api.Results (same layout as the product).
    HiFi kinetics (fi/fp/ri/rp planes) are computed iff `results` was allocated with kinetics=True."""
    L = lib()
    kin = results.kin is not None
    nullp = C.POINTER(C.c_uint8)()
    planes = [_p(results.kin[k], C.c_uint8) for k in range(4)] if kin else [nullp] * 4
    L.orc_consensus_batch_kin(C.byref(model), C.byref(opts), batch.n_zmw, _p(batch.snr, C.c_float),
                              _p(batch.read_off, C.c_int32), _p(batch.base_off, C.c_int64), _p(batch.bases, C.c_uint8),
                              _p(batch.pw, C.c_uint8), _p(batch.flags, C.c_uint8), _p(results.seq_off, C.c_int64),
                              _p(results.status, C.c_int32), _p(results.seq_len, C.c_int32), _p(results.seq, C.c_uint8),
                              _p(results.qual, C.c_uint8), _p(results.raw_qv, C.c_float), _p(results.rq, C.c_float),
                              _p(results.np_, C.c_int32), _p(results.ec, C.c_float), _p(results.iters, C.c_int32),
                              _p(results.n_windows, C.c_int32), int(nthreads),
                              _p(batch.ipd, C.c_uint8) if kin else nullp, *planes,
                              _p(results.fn, C.c_int32), _p(results.rn, C.c_int32))
    return results


class _ZmwOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("seq_len", C.c_int32), ("np", C.c_int32), ("iters", C.c_int32), ("n_windows", C.c_int32),
                ("rq", C.c_float), ("ec", C.c_float), ("fn", C.c_int32), ("rn", C.c_int32)]


def polish_batch(model, opts, batch, drafts, results, flags=0):
    """The polish seam on the CPU restatement (orc_polish_zmw per ZMW): alignment cascade + windows + polish + QVs on the drafts of a
    ccs_amd.api.Drafts; flags bit 0 = CCSX_QV_ONLY.  Mirrors ccsx_polish_batch (include/ccsx.h)."""
    L = lib()
    for z in range(batch.n_zmw):
        r0, r1 = int(batch.read_off[z]), int(batch.read_off[z + 1])
        b0 = int(batch.base_off[r0])
        rel = np.ascontiguousarray(batch.base_off[r0:r1 + 1] - b0)
        o, cap = int(results.seq_off[z]), int(results.seq_off[z + 1] - results.seq_off[z])
        d = np.ascontiguousarray(drafts.draft(z))
        out = _ZmwOut()
        seq, qual, raw = results.seq[o:o + cap], results.qual[o:o + cap], results.raw_qv[o:o + cap]
        L.orc_polish_zmw(C.byref(model), C.byref(opts), _p(np.ascontiguousarray(batch.snr[z]), C.c_float), r1 - r0, _p(rel, C.c_int64),
                         _p(batch.bases[b0:], C.c_uint8), _p(batch.pw[b0:], C.c_uint8), _p(batch.flags[r0:], C.c_uint8),
                         _p(d, C.c_uint8) if len(d) else C.POINTER(C.c_uint8)(), len(d), int(drafts.backbone[z]), int(flags),
                         _p(seq, C.c_uint8), _p(qual, C.c_uint8), _p(raw, C.c_float), C.c_int64(cap), C.byref(out))
        results.status[z], results.seq_len[z], results.np_[z], results.iters[z], results.n_windows[z] = out.status, out.seq_len, out.np, out.iters, out.n_windows
        results.rq[z], results.ec[z], results.fn[z], results.rn[z] = out.rq, out.ec, out.fn, out.rn
    return results


def codec_decode(c):
    return lib().orc_codec_v1_decode(int(c))


def codec_encode(f):
    return lib().orc_codec_v1_encode(int(f))


def kinetics_read(tpl_read_orient, read_bases, ipd, pw, strand):
    """orc_kinetics_read on one read: returns (sum_ipd, sum_pw, cnt) indexed by FORWARD window column."""
    t = np.ascontiguousarray(tpl_read_orient, np.uint8)
    rb = np.ascontiguousarray(read_bases, np.uint8)
    ip = np.ascontiguousarray(ipd, np.uint8)
    pwc = np.ascontiguousarray(pw, np.uint8)
    si, sp, cn = (np.zeros(JMAX + 1, np.uint32) for _ in range(3))
    lib().orc_kinetics_read(_p(t, C.c_uint8), len(t), _p(rb, C.c_uint8), _p(ip, C.c_uint8), _p(pwc, C.c_uint8), len(rb),
                            int(strand), _p(si, C.c_uint32), _p(sp, C.c_uint32), _p(cn, C.c_uint32))
    return si[: len(t)], sp[: len(t)], cn[: len(t)]


COUNT_NAMES = ["trim", "split", "split_s0", "split_sLd", "fallback", "retry64", "zdrop", "nonconv_win", "poa_wide", "third_draft",
               "partial_used", "cells_poa", "cells_align", "cells_fill", "cells_score", "zmws", "split2", "saturated", "closed_tract"]


def counts_reset():
    lib().orc_counts_reset()


def counts():
    """path / work counters since the last counts_reset(): which SPEC paths fired (trim, split alignment, fallback draft, 64-row
    retry, z-score drop, ...) and the DP cell updates per stage (SURVEY.md §8d secondary figure)"""
    n = lib().orc_counts_n()
    a = np.zeros(n, np.int64)
    lib().orc_counts_get(_p(a, C.c_int64))
    return dict(zip(COUNT_NAMES, a.tolist()))


def edit_distance(a, b, band=400):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_edit_distance(_p(a, C.c_uint8), len(a), _p(b, C.c_uint8), len(b), int(band)))


def spec_version():
    return int(lib().orc_spec_version())
