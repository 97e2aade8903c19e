"""ctypes mirror of include/ccsx.h — the host side used by tests, bench.py and the smoke test.

The product is libccsx.so (hand-written HIP kernels for gfx950 + a C++ host behind a C ABI).  This
module only marshals numpy arrays into the C structs; it contains no algorithmic code and there is
NO CPU fallback: if the library is missing, loading raises, and if no GPU is usable, `Handle()` raises.

Reference interface this mirrors: the "Draft Stage"/"Polish Stage" GPU consumers of
docs/img/ccs-impl.png and the per-ZMW outputs of docs/faq/bam-output.md:9-30 (rq, np, ec, sn, zm).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCSX_LIB", os.path.join(_HERE, "libccsx.so"))

BAND, MAXPRED, WIN_CORE, WIN_OVERHANG, JMAX, IMAX, MAX_ITER, NCTX, NOBS = 64, 7, 22, 2, 31, 63, 8, 16, 12   # include/ccsx.h (tests/test_abi.py compares)

STATUS_NAMES = {
    0: "SUCCESS", 1: "TOO_FEW_PASSES", 2: "DRAFT_FAILURE", 3: "TOO_MANY_UNUSABLE", 4: "NON_CONVERGENT",
    5: "TOO_SHORT", 6: "TOO_LONG", 7: "LOW_RQ", 8: "EMPTY_WINDOW", 9: "CAPACITY",
}


class Model(C.Structure):
    _fields_ = [
        ("name", C.c_char * 32),
        ("snr_lo", C.c_float), ("snr_hi", C.c_float),
        ("trans_poly", C.c_float * 4 * 3 * NCTX),
        ("em_match", C.c_float * NOBS * NCTX),
        ("em_branch", C.c_float * 3 * NCTX),
        ("em_stick", C.c_float * 3 * NCTX),
    ]


class Opts(C.Structure):
    _fields_ = [
        ("max_poa_cov", C.c_int32), ("min_passes", C.c_int32), ("top_passes", C.c_int32),
        ("min_length", C.c_int32), ("max_length", C.c_int32), ("min_rq", C.c_float),
        ("poa_slots", C.c_int32), ("hifi_kinetics", C.c_int32), ("disable_heuristics", C.c_int32), ("min_zscore", C.c_float), ("handles_per_device", C.c_int32), ("no_fallback_draft", C.c_int32), ("max_insertion_size", C.c_int32), ("serial_stages", C.c_int32), ("max_qv", C.c_int32),
    ]


class CBatch(C.Structure):
    _fields_ = [
        ("n_zmw", C.c_int32), ("n_reads", C.c_int32), ("n_bases", C.c_int64),
        ("zmw_id", C.POINTER(C.c_int32)), ("snr", C.POINTER(C.c_float)),
        ("read_off", C.POINTER(C.c_int32)), ("base_off", C.POINTER(C.c_int64)),
        ("bases", C.POINTER(C.c_uint8)), ("pw", C.POINTER(C.c_uint8)), ("ipd", C.POINTER(C.c_uint8)),
        ("flags", C.POINTER(C.c_uint8)),
    ]


class CResults(C.Structure):
    _fields_ = [
        ("n_zmw", C.c_int32), ("seq_capacity", C.c_int64),
        ("seq_off", C.POINTER(C.c_int64)), ("status", C.POINTER(C.c_int32)), ("seq_len", C.POINTER(C.c_int32)),
        ("seq", C.POINTER(C.c_uint8)), ("qual", C.POINTER(C.c_uint8)), ("raw_qv", C.POINTER(C.c_float)),
        ("rq", C.POINTER(C.c_float)), ("np", C.POINTER(C.c_int32)), ("ec", C.POINTER(C.c_float)),
        ("iters", C.POINTER(C.c_int32)), ("n_windows", C.POINTER(C.c_int32)),
        ("fi", C.POINTER(C.c_uint8)), ("fp", C.POINTER(C.c_uint8)), ("ri", C.POINTER(C.c_uint8)), ("rp", C.POINTER(C.c_uint8)),
        ("fn", C.POINTER(C.c_int32)), ("rn", C.POINTER(C.c_int32)),
    ]


class CDrafts(C.Structure):
    _fields_ = [
        ("n_zmw", C.c_int32), ("seq_capacity", C.c_int64), ("win_capacity", C.c_int64),
        ("seq_off", C.POINTER(C.c_int64)), ("win_off", C.POINTER(C.c_int64)), ("status", C.POINTER(C.c_int32)), ("len", C.POINTER(C.c_int32)),
        ("seq", C.POINTER(C.c_uint8)), ("backbone", C.POINTER(C.c_int32)), ("n_windows", C.POINTER(C.c_int32)), ("win_bounds", C.POINTER(C.c_int32)),
    ]


QV_ONLY = 1     # ccsx_polish_batch flag (CCSX_QV_ONLY)


class Timings(C.Structure):
    _fields_ = [
        ("setup_ms", C.c_float), ("draft_ms", C.c_float), ("align_ms", C.c_float), ("polish_ms", C.c_float),
        ("stitch_ms", C.c_float), ("total_ms", C.c_float), ("polish_workgroups", C.c_int64),
        ("queue_ms", C.c_float), ("reserved_", C.c_float), ("start_ms", C.c_double), ("end_ms", C.c_double),
    ]


class CSynth(C.Structure):
    _fields_ = [("batch", CBatch), ("tpl_off", C.POINTER(C.c_int64)), ("tpl", C.POINTER(C.c_uint8))]


# every symbol include/ccsx.h declares (tests/test_abi.py checks the library exports all of them)
EXPORTS = [
    "ccsx_abi_version", "ccsx_spec_version", "ccsx_last_error", "ccsx_device_count", "ccsx_model_default", "ccsx_opts_default",
    "ccsx_create", "ccsx_destroy", "ccsx_result_layout", "ccsx_consensus_batch", "ccsx_upload", "ccsx_run",
    "ccsx_sync", "ccsx_download", "ccsx_get_timings", "ccsx_stage_draft", "ccsx_stage_align",
    "ccsx_stage_windows", "ccsx_synth_generate", "ccsx_synth_free", "ccsx_alloc_pinned", "ccsx_free_pinned",
    "ccsx_submit", "ccsx_wait", "ccsx_poll", "ccsx_ticket_timings",
    "ccsx_model_from_json", "ccsx_model_load", "ccsx_model_to_json", "ccsx_model_for_chemistry",
    "ccsx_build_flags", "ccsx_runtime_switches", "ccsx_pci_numa_node", "ccsx_device_numa_node", "ccsx_bind_thread_to_node", "ccsx_bind_thread_to_device", "ccsx_draft_layout", "ccsx_draft_batch", "ccsx_polish_batch", "ccsx_submit_draft", "ccsx_submit_polish",
]

_lib = None


def lib() -> C.CDLL:
    """Load libccsx.so (fails loudly when it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.ccsx_last_error.restype = C.c_char_p
        L.ccsx_result_layout.restype = C.c_int64
        L.ccsx_result_layout.argtypes = [C.POINTER(CBatch), C.POINTER(C.c_int64)]
        L.ccsx_create.argtypes = [C.c_int, C.POINTER(Model), C.POINTER(Opts), C.POINTER(C.c_void_p)]
        for f in ("ccsx_destroy", "ccsx_run", "ccsx_sync"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.ccsx_upload.argtypes = [C.c_void_p, C.POINTER(CBatch)]
        L.ccsx_download.argtypes = [C.c_void_p, C.POINTER(CResults)]
        L.ccsx_consensus_batch.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CResults)]
        L.ccsx_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
        L.ccsx_submit.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CResults), C.POINTER(C.c_int64)]
        L.ccsx_wait.argtypes = [C.c_void_p, C.c_int64]
        L.ccsx_poll.argtypes = [C.c_void_p, C.c_int64]
        L.ccsx_ticket_timings.argtypes = [C.c_void_p, C.c_int64, C.POINTER(Timings)]
        L.ccsx_stage_draft.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_int32)]
        L.ccsx_stage_align.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ccsx_stage_windows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
        L.ccsx_synth_generate.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_uint64, C.POINTER(C.POINTER(CSynth))]
        L.ccsx_synth_free.argtypes = [C.POINTER(CSynth)]
        L.ccsx_alloc_pinned.restype = C.c_void_p
        L.ccsx_alloc_pinned.argtypes = [C.c_size_t]
        L.ccsx_free_pinned.argtypes = [C.c_void_p]
        L.ccsx_model_default.argtypes = [C.POINTER(Model)]
        L.ccsx_model_from_json.argtypes = [C.c_char_p, C.POINTER(Model)]
        L.ccsx_model_load.argtypes = [C.c_char_p, C.POINTER(Model)]
        L.ccsx_model_to_json.restype = C.c_int64
        L.ccsx_model_to_json.argtypes = [C.POINTER(Model), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64]
        L.ccsx_model_for_chemistry.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Model)]
        L.ccsx_opts_default.argtypes = [C.POINTER(Opts)]
        L.ccsx_build_flags.restype = C.c_char_p
        L.ccsx_runtime_switches.restype = C.c_char_p
        L.ccsx_draft_layout.restype = None
        L.ccsx_draft_layout.argtypes = [C.POINTER(CBatch), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.ccsx_draft_batch.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CDrafts)]
        L.ccsx_polish_batch.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CDrafts), C.POINTER(CResults), C.c_uint32]
        L.ccsx_submit_draft.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CDrafts), C.POINTER(C.c_int64)]
        L.ccsx_submit_polish.argtypes = [C.c_void_p, C.POINTER(CBatch), C.POINTER(CDrafts), C.POINTER(CResults), C.c_uint32, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class _Pinned:
    """Owner of one ccsx_alloc_pinned block."""

    def __init__(self, p):
        self.p = p

    def __del__(self):
        try:
            if self.p:
                lib().ccsx_free_pinned(self.p)
                self.p = None
        except Exception:
            pass


@dataclass
class Batch:
    """Input batch, SoA + CSR (include/ccsx.h ccsx_batch).  All arrays are C-contiguous numpy."""
    zmw_id: np.ndarray
    snr: np.ndarray
    read_off: np.ndarray
    base_off: np.ndarray
    bases: np.ndarray
    pw: np.ndarray
    ipd: np.ndarray
    flags: np.ndarray
    tpl_off: np.ndarray | None = None
    tpl: np.ndarray | None = None

    @property
    def n_zmw(self) -> int:
        return len(self.zmw_id)

    def c_struct(self) -> CBatch:
        b = CBatch()
        b.n_zmw = self.n_zmw
        b.n_reads = int(self.read_off[-1])
        b.n_bases = int(self.base_off[-1])
        b.zmw_id = _ptr(self.zmw_id, C.c_int32)
        b.snr = _ptr(self.snr, C.c_float)
        b.read_off = _ptr(self.read_off, C.c_int32)
        b.base_off = _ptr(self.base_off, C.c_int64)
        b.bases = _ptr(self.bases, C.c_uint8)
        b.pw = _ptr(self.pw, C.c_uint8)
        b.ipd = _ptr(self.ipd, C.c_uint8)
        b.flags = _ptr(self.flags, C.c_uint8)
        return b

    def read(self, r: int):
        a, b = int(self.base_off[r]), int(self.base_off[r + 1])
        return self.bases[a:b], self.pw[a:b]

    def algorithmic_bytes(self) -> int:
        """3*sum(len) + per ZMW (16 + 2*L_out + 32); L_out ~ template length."""
        n = self.n_zmw
        if self.tpl_off is not None:
            lout = int(self.tpl_off[-1])
        else:
            lout = int(self.base_off[-1]) // max(1, int(self.read_off[-1])) * n
        return 3 * int(self.base_off[-1]) + 48 * n + 2 * lout

    def pinned(self) -> "Batch":
        """Copy of this batch whose per-base arrays live in page-locked host memory (ccsx_alloc_pinned): uploads run at PCIe
        rate instead of through the runtime's pageable staging.  The buffers are released with the returned object."""
        L = lib()
        out = Batch(self.zmw_id, self.snr, self.read_off, self.base_off, None, None, None, self.flags, self.tpl_off, self.tpl)
        keep = []
        for name in ("bases", "pw", "ipd"):
            src = getattr(self, name)
            n = max(1, src.nbytes)
            p = L.ccsx_alloc_pinned(n)
            if not p:
                raise RuntimeError("ccsx_alloc_pinned failed: " + L.ccsx_last_error().decode())
            keep.append(_Pinned(p))
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))[: len(src)]
            arr[:] = src
            setattr(out, name, arr)
        out._pinned = keep
        return out

    def slice(self, z0: int, z1: int) -> "Batch":
        r0, r1 = int(self.read_off[z0]), int(self.read_off[z1])
        b0, b1 = int(self.base_off[r0]), int(self.base_off[r1])
        kw = {}
        if self.tpl_off is not None:
            t0, t1 = int(self.tpl_off[z0]), int(self.tpl_off[z1])
            kw = dict(tpl_off=(self.tpl_off[z0:z1 + 1] - t0).copy(), tpl=self.tpl[t0:t1].copy())
        return Batch(self.zmw_id[z0:z1].copy(), self.snr[z0:z1].copy(), (self.read_off[z0:z1 + 1] - r0).copy(),
                     (self.base_off[r0:r1 + 1] - b0).copy(), self.bases[b0:b1].copy(), self.pw[b0:b1].copy(),
                     self.ipd[b0:b1].copy(), self.flags[r0:r1].copy(), **kw)


def concat(parts) -> "Batch":
    """the ZMWs of several batches as one batch, in order (ZMW ids are renumbered 0..n-1)"""
    read_off, base_off, tpl_off = [np.zeros(1, np.int32)], [np.zeros(1, np.int64)], [np.zeros(1, np.int64)]
    for b in parts:
        read_off.append(b.read_off[1:] + read_off[-1][-1]); base_off.append(b.base_off[1:] + base_off[-1][-1])
        tpl_off.append(b.tpl_off[1:] + tpl_off[-1][-1])
    cat = lambda k: np.ascontiguousarray(np.concatenate([getattr(b, k) for b in parts]))
    n = sum(b.n_zmw for b in parts)
    return Batch(np.arange(n, dtype=np.int32), cat("snr"), np.concatenate(read_off).astype(np.int32), np.concatenate(base_off).astype(np.int64),
                 cat("bases"), cat("pw"), cat("ipd"), cat("flags"), np.concatenate(tpl_off).astype(np.int64), cat("tpl"))


def synth(n_zmw: int, passes, length, seed: int = 1, first_zmw_id: int = 0) -> Batch:
    """Deterministic synthetic subreads (ccsx_synth_generate).  passes/length: int or (lo, hi)."""
    plo, phi = (passes, passes) if isinstance(passes, int) else passes
    llo, lhi = (length, length) if isinstance(length, int) else length
    L = lib()
    p = C.POINTER(CSynth)()
    if L.ccsx_synth_generate(n_zmw, first_zmw_id, plo, phi, llo, lhi, seed, C.byref(p)) != 0:
        raise RuntimeError(L.ccsx_last_error().decode())
    s = p.contents
    b = s.batch
    R, NB = b.n_reads, b.n_bases

    def cp(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)

    out = Batch(cp(b.zmw_id, n_zmw, np.int32), cp(b.snr, 4 * n_zmw, np.float32).reshape(n_zmw, 4),
                cp(b.read_off, n_zmw + 1, np.int32), cp(b.base_off, R + 1, np.int64), cp(b.bases, NB, np.uint8),
                cp(b.pw, NB, np.uint8), cp(b.ipd, NB, np.uint8), cp(b.flags, R, np.uint8),
                tpl_off=cp(s.tpl_off, n_zmw + 1, np.int64))
    out.tpl = cp(s.tpl, int(out.tpl_off[-1]), np.uint8)
    L.ccsx_synth_free(p)
    return out


def default_model() -> Model:
    m = Model()
    lib().ccsx_model_default(C.byref(m))
    return m


def set_model_name(m: Model, name: str) -> Model:
    """name of a parameter set, zero padded (the blob is compared / hashed as bytes; ctypes assignment stops at the first NUL)"""
    raw = name.encode()[:31].ljust(32, b"\0")
    C.memmove(C.addressof(m), raw, 32)
    return m


def model_to_json(m: Model, chemistry=None) -> str:
    """json text of a parameter set (ccsx_model_to_json); chemistry = (binding kit, sequencing kit, basecaller version) or None"""
    L = lib()
    tri = [c.encode() for c in chemistry] if chemistry else [None, None, None]
    n = L.ccsx_model_to_json(C.byref(m), *tri, None, 0)
    buf = C.create_string_buffer(n + 1)
    L.ccsx_model_to_json(C.byref(m), *tri, buf, n + 1)
    return buf.value.decode()


def model_from_json(text: str) -> Model:
    m = Model()
    if lib().ccsx_model_from_json(text.encode(), C.byref(m)) != 0:
        raise RuntimeError(lib().ccsx_last_error().decode())
    return m


def model_load(path: str) -> Model:
    m = Model()
    if lib().ccsx_model_load(path.encode(), C.byref(m)) != 0:
        raise RuntimeError(lib().ccsx_last_error().decode())
    return m


def model_for_chemistry(binding_kit: str, sequencing_kit: str, basecaller_version: str) -> Model:
    m = Model()
    if lib().ccsx_model_for_chemistry(binding_kit.encode(), sequencing_kit.encode(), basecaller_version.encode(), C.byref(m)) != 0:
        raise RuntimeError(lib().ccsx_last_error().decode())
    return m


def default_opts() -> Opts:
    o = Opts()
    lib().ccsx_opts_default(C.byref(o))
    return o


@dataclass
class Results:
    seq_off: np.ndarray
    status: np.ndarray
    seq_len: np.ndarray
    seq: np.ndarray
    qual: np.ndarray
    raw_qv: np.ndarray
    rq: np.ndarray
    np_: np.ndarray
    ec: np.ndarray
    iters: np.ndarray
    n_windows: np.ndarray
    fn: np.ndarray = None
    rn: np.ndarray = None
    kin: np.ndarray | None = None        # [4, capacity] planes fi, fp, ri, rp (CodecV1 codes); None without kinetics

    @staticmethod
    def allocate(batch: Batch, kinetics: bool = False, pinned: bool = False, raw: bool = True) -> "Results":
        """pinned=True puts every array in page-locked memory (ccsx_alloc_pinned): asynchronous downloads (Handle.submit)
        then run by DMA at PCIe rate."""
        n = batch.n_zmw
        cb = batch.c_struct()
        off = np.zeros(n + 1, np.int64)
        cap = lib().ccsx_result_layout(C.byref(cb), _ptr(off, C.c_int64))
        keep = []

        def z(shape, dt):
            if not pinned:
                return np.zeros(shape, dt)
            nb = max(1, int(np.prod(shape)) * np.dtype(dt).itemsize)
            p = lib().ccsx_alloc_pinned(nb)
            if not p:
                raise RuntimeError("ccsx_alloc_pinned failed: " + lib().ccsx_last_error().decode())
            keep.append(_Pinned(p))
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb,))[: int(np.prod(shape)) * np.dtype(dt).itemsize]
            a = a.view(dt).reshape(shape)
            a[...] = 0
            return a
        r = Results(off, z(n, np.int32), z(n, np.int32), z(cap, np.uint8), z(cap, np.uint8),
                    z(cap, np.float32) if raw else None, z(n, np.float32), z(n, np.int32), z(n, np.float32),
                    z(n, np.int32), z(n, np.int32), z(n, np.int32), z(n, np.int32),
                    z((4, cap), np.uint8) if kinetics else None)
        r._pinned = keep
        return r

    def c_struct(self) -> CResults:
        r = CResults()
        r.n_zmw = len(self.status)
        r.seq_capacity = len(self.seq)
        r.seq_off = _ptr(self.seq_off, C.c_int64)
        r.status = _ptr(self.status, C.c_int32)
        r.seq_len = _ptr(self.seq_len, C.c_int32)
        r.seq = _ptr(self.seq, C.c_uint8)
        r.qual = _ptr(self.qual, C.c_uint8)
        r.raw_qv = _ptr(self.raw_qv, C.c_float) if self.raw_qv is not None else None   # optional output: NULL = not downloaded
        r.rq = _ptr(self.rq, C.c_float)
        r.np = _ptr(self.np_, C.c_int32)
        r.ec = _ptr(self.ec, C.c_float)
        r.iters = _ptr(self.iters, C.c_int32)
        r.n_windows = _ptr(self.n_windows, C.c_int32)
        r.fn = _ptr(self.fn, C.c_int32)
        r.rn = _ptr(self.rn, C.c_int32)
        if self.kin is not None:
            r.fi, r.fp, r.ri, r.rp = (_ptr(self.kin[k], C.c_uint8) for k in range(4))
        return r

    def sequence(self, z: int) -> np.ndarray:
        o = int(self.seq_off[z])
        return self.seq[o:o + int(self.seq_len[z])]

    def quals(self, z: int) -> np.ndarray:
        o = int(self.seq_off[z])
        return self.qual[o:o + int(self.seq_len[z])]

    def raw(self, z: int) -> np.ndarray:
        o = int(self.seq_off[z])
        return self.raw_qv[o:o + int(self.seq_len[z])]

    def kinetics(self, z: int) -> np.ndarray:
        """[4, seq_len] CodecV1 codes (fi, fp, ri, rp) of ZMW z, orientation of SEQ."""
        o = int(self.seq_off[z])
        return self.kin[:, o:o + int(self.seq_len[z])]


@dataclass
class Drafts:
    """ccsx_drafts: what the draft seam hands to the polish seam (include/ccsx.h)."""
    seq_off: np.ndarray
    win_off: np.ndarray
    status: np.ndarray
    len: np.ndarray
    seq: np.ndarray
    backbone: np.ndarray
    n_windows: np.ndarray
    win_bounds: np.ndarray

    @staticmethod
    def allocate(batch: Batch, pinned: bool | None = None) -> "Drafts":
        """pinned (default wherever there is a device): the arrays the ticketed seams copy asynchronously live in page-locked memory, as include/ccsx.h asks
        (with pageable arrays the copies block the submitting thread until the stage has finished)"""
        if pinned is None:
            pinned = lib().ccsx_device_count() > 0
        n = batch.n_zmw
        cb = batch.c_struct()
        so, wo = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        sc, wc = C.c_int64(), C.c_int64()
        lib().ccsx_draft_layout(C.byref(cb), _ptr(so, C.c_int64), _ptr(wo, C.c_int64), C.byref(sc), C.byref(wc))
        keep = []

        def z(count, dt):
            if not pinned:
                return np.zeros(count, dt)
            nb = max(1, int(count) * np.dtype(dt).itemsize)
            p = lib().ccsx_alloc_pinned(nb)
            if not p:
                raise RuntimeError("ccsx_alloc_pinned failed: " + lib().ccsx_last_error().decode())
            keep.append(_Pinned(p))
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb,))[: int(count) * np.dtype(dt).itemsize].view(dt)
            a[...] = 0
            return a
        d = Drafts(so, wo, z(n, np.int32), z(n, np.int32), z(sc.value, np.uint8), z(n, np.int32), z(n, np.int32), z(wc.value, np.int32))
        d._pinned = keep
        return d

    def c_struct(self) -> CDrafts:
        d = CDrafts()
        d.n_zmw = len(self.len); d.seq_capacity = len(self.seq); d.win_capacity = len(self.win_bounds)
        d.seq_off = _ptr(self.seq_off, C.c_int64); d.win_off = _ptr(self.win_off, C.c_int64)
        d.status = _ptr(self.status, C.c_int32); d.len = _ptr(self.len, C.c_int32); d.seq = _ptr(self.seq, C.c_uint8)
        d.backbone = _ptr(self.backbone, C.c_int32); d.n_windows = _ptr(self.n_windows, C.c_int32); d.win_bounds = _ptr(self.win_bounds, C.c_int32)
        return d

    def draft(self, z: int) -> np.ndarray:
        o = int(self.seq_off[z])
        return self.seq[o:o + int(self.len[z])]

    def set_draft(self, z: int, bases: np.ndarray, backbone: int = 0):
        o = int(self.seq_off[z])
        assert len(bases) <= int(self.seq_off[z + 1]) - o
        self.seq[o:o + len(bases)] = bases
        self.len[z] = len(bases); self.backbone[z] = backbone

    def windows(self, z: int) -> np.ndarray:
        o = int(self.win_off[z])
        return self.win_bounds[o:o + int(self.n_windows[z]) + 1]


class Handle:
    """One consensus engine bound to one GPU (ccsx_create).  Not thread-safe: one per worker per GPU."""

    def __init__(self, device: int = 0, model: Model | None = None, opts: Opts | None = None):
        self._L = lib()
        self.model = model or default_model()
        self.opts = opts or default_opts()
        self._h = C.c_void_p()
        rc = self._L.ccsx_create(device, C.byref(self.model), C.byref(self.opts), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("ccsx_create failed: " + self._L.ccsx_last_error().decode())
        self._keep = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: " + self._L.ccsx_last_error().decode())

    def consensus(self, batch: Batch) -> Results:
        res = Results.allocate(batch, kinetics=bool(self.opts.hifi_kinetics))
        cb, cr = batch.c_struct(), res.c_struct()
        self._check(self._L.ccsx_consensus_batch(self._h, C.byref(cb), C.byref(cr)), "ccsx_consensus_batch")
        return res

    # ---- the two seams (docs/img/ccs-impl.png): draft stage alone, polish stage on caller-supplied drafts
    def draft(self, batch: Batch) -> "Drafts":
        d = Drafts.allocate(batch)
        cb, cd = batch.c_struct(), d.c_struct()
        self._check(self._L.ccsx_draft_batch(self._h, C.byref(cb), C.byref(cd)), "ccsx_draft_batch")
        return d

    def polish(self, batch: Batch, drafts: "Drafts", flags: int = 0) -> Results:
        res = Results.allocate(batch, kinetics=bool(self.opts.hifi_kinetics))
        cb, cd, cr = batch.c_struct(), drafts.c_struct(), res.c_struct()
        self._check(self._L.ccsx_polish_batch(self._h, C.byref(cb), C.byref(cd), C.byref(cr), flags), "ccsx_polish_batch")
        return res

    def upload(self, batch: Batch):
        self._keep = batch
        cb = batch.c_struct()
        self._check(self._L.ccsx_upload(self._h, C.byref(cb)), "ccsx_upload")

    def run(self):
        self._check(self._L.ccsx_run(self._h), "ccsx_run")

    def sync(self):
        self._check(self._L.ccsx_sync(self._h), "ccsx_sync")

    def download(self) -> Results:
        res = Results.allocate(self._keep, kinetics=bool(self.opts.hifi_kinetics))
        cr = res.c_struct()
        self._check(self._L.ccsx_download(self._h, C.byref(cr)), "ccsx_download")
        return res

    def timings(self) -> Timings:
        t = Timings()
        self._check(self._L.ccsx_get_timings(self._h, C.byref(t)), "ccsx_get_timings")
        return t

    # ---- asynchronous pipeline (ccsx_submit / ccsx_wait): up to three batches in flight, copies under compute
    def submit(self, batch: Batch, res: "Results") -> int:
        cb, cr = batch.c_struct(), res.c_struct()
        t = C.c_int64()
        self._check(self._L.ccsx_submit(self._h, C.byref(cb), C.byref(cr), C.byref(t)), "ccsx_submit")
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[t.value] = (batch, res, cb, cr)      # the C structs and arrays must outlive the ticket
        return t.value

    def wait(self, ticket: int) -> "Results":
        self._check(self._L.ccsx_wait(self._h, ticket), "ccsx_wait")
        return self._inflight[ticket][1]

    def release(self, ticket: int):
        self._inflight.pop(ticket, None)

    def poll(self, ticket: int) -> bool:
        rc = self._L.ccsx_poll(self._h, ticket)
        if rc < 0:
            self._check(rc, "ccsx_poll")
        return rc == 1

    def ticket_timings(self, ticket: int) -> Timings:
        t = Timings()
        self._check(self._L.ccsx_ticket_timings(self._h, ticket, C.byref(t)), "ccsx_ticket_timings")
        return t

    def stage_draft(self, z: int) -> np.ndarray:
        cap = 1 << 17
        buf = np.zeros(cap, np.uint8)
        n = C.c_int32()
        self._check(self._L.ccsx_stage_draft(self._h, z, _ptr(buf, C.c_uint8), cap, C.byref(n)), "ccsx_stage_draft")
        return buf[: n.value].copy()

    def stage_align(self, r: int, ld: int):
        buf = np.zeros(ld + 1, np.int32)
        v, s = C.c_int32(), C.c_int32()
        self._check(self._L.ccsx_stage_align(self._h, r, _ptr(buf, C.c_int32), ld + 1, C.byref(v), C.byref(s)), "ccsx_stage_align")
        return buf, v.value, s.value

    def stage_windows(self, z: int) -> np.ndarray:
        cap = 1 << 14
        buf = np.zeros(cap, np.int32)
        n = C.c_int32()
        self._check(self._L.ccsx_stage_windows(self._h, z, _ptr(buf, C.c_int32), cap, C.byref(n)), "ccsx_stage_windows")
        return buf[: n.value + 1].copy()

    def close(self):
        if self._h:
            self._L.ccsx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count() -> int:
    return lib().ccsx_device_count()
