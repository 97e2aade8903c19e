"""Tiny pure-python BAM reader for the tests (BGZF = concatenated gzip members; no pysam/htslib in the image)."""
import gzip
import struct

import numpy as np

NIB = {1: 0, 2: 1, 4: 2, 8: 3}


def read_bam(path):
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p); p += 4
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p); p += 4 + l + 4
    recs = []
    while p < len(data):
        bs, = struct.unpack_from("<i", data, p); p += 4
        r = data[p:p + bs]; p += bs
        l_name = r[8]; n_cig, flag = struct.unpack_from("<HH", r, 12)
        l_seq, = struct.unpack_from("<i", r, 16)
        name = r[32:32 + l_name - 1].decode()
        q = 32 + l_name + 4 * n_cig
        packed = np.frombuffer(r, np.uint8, (l_seq + 1) // 2, q)
        nib = np.empty(2 * len(packed), np.uint8); nib[0::2] = packed >> 4; nib[1::2] = packed & 15
        seq = np.array([NIB.get(int(x), 9) for x in nib[:l_seq]], np.uint8)
        q += (l_seq + 1) // 2
        qual = np.frombuffer(r, np.uint8, l_seq, q).copy(); q += l_seq
        tags = {}
        while q < len(r):
            tag = r[q:q + 2].decode(); ty = chr(r[q + 2]); q += 3
            if ty == "Z":
                e = r.index(b"\0", q); tags[tag] = r[q:e].decode(); q = e + 1
            elif ty == "B":
                st = chr(r[q]); n, = struct.unpack_from("<i", r, q + 1); q += 5
                dt = {"c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16, "i": np.int32, "I": np.uint32, "f": np.float32}[st]
                tags[tag] = np.frombuffer(r, dt, n, q).copy(); q += n * np.dtype(dt).itemsize
            else:
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f", "A": "<c"}[ty]
                tags[tag], = struct.unpack_from(fmt, r, q); q += struct.calcsize(fmt)
        recs.append(dict(name=name, flag=flag, seq=seq, qual=qual, tags=tags))
    return text, recs


# ---- tiny writer (adversarial inputs for the C++ reader) ------------------------------------------------------------
def _bgzf_block(payload: bytes, extra_first: bytes = b"") -> bytes:
    import zlib
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(payload) + co.flush()
    extra = extra_first + b"BC" + struct.pack("<H", 2) + b"\0\0"
    total = 12 + len(extra) + len(comp) + 8
    extra = extra_first + b"BC" + struct.pack("<H", 2) + struct.pack("<H", total - 1)
    hdr = b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", len(extra))
    return hdr + extra + comp + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def record(name: str, seq: str, tags: list, raw_tail: bytes = b"") -> bytes:
    """tags: list of (tag, type, value); type in 'i','f','Z','BC','BS','Bf'; raw_tail: bytes appended after the tags."""
    nib = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    n = len(seq)
    packed = bytearray((n + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= nib[ch] << (0 if (i & 1) else 4)
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(name) + 1, 255, 4680, 0, 4, n, -1, -1, 0) + name.encode() + b"\0" + bytes(packed) + b"\xff" * n
    for tag, ty, v in tags:
        body += tag.encode()
        if ty == "i": body += b"i" + struct.pack("<i", v)
        elif ty == "f": body += b"f" + struct.pack("<f", v)
        elif ty == "Z": body += b"Z" + v.encode() + b"\0"
        elif ty == "BC": body += b"BC" + struct.pack("<i", len(v)) + np.asarray(v, np.uint8).tobytes()
        elif ty == "BS": body += b"BS" + struct.pack("<i", len(v)) + np.asarray(v, "<u2").tobytes()
        elif ty == "Bf": body += b"Bf" + struct.pack("<i", len(v)) + np.asarray(v, "<f4").tobytes()
        else: raise ValueError(ty)
    body += raw_tail
    return struct.pack("<i", len(body)) + body


def write_bam(path, text: str, records: list, block: int = 0xff00, extra_first: bytes = b"", empty_blocks: bool = False, eof: bool = True):
    data = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 0) + b"".join(records)
    with open(path, "wb") as f:
        for k, a in enumerate(range(0, len(data), block)):
            f.write(_bgzf_block(data[a:a + block], extra_first))
            if empty_blocks and k % 3 == 1:
                f.write(_bgzf_block(b"", extra_first))
        if eof:
            f.write(BGZF_EOF)


def read_bam_raw_records(path):
    """(header text, [raw record bytes incl. block_size]) — to rewrite a BAM with another header"""
    data = gzip.open(path, "rb").read()
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p); p += 4
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p); p += 4 + l + 4
    recs = []
    while p < len(data):
        bs, = struct.unpack_from("<i", data, p)
        recs.append(data[p:p + 4 + bs]); p += 4 + bs
    return text, recs


# ---- PacBio BAM index (.pbi) and BGZF virtual offsets -----------------------------------------------------------------
def read_pbi(path):
    """columns of a .pbi (layout as ccs_amd/csrc/bam_io.h documents it): dict of numpy arrays"""
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"PBI\x01"
    version, flags, n = struct.unpack_from("<IHI", data, 4)
    p = 32
    out = {"version": version, "flags": flags, "n": n}
    for key, dt in (("rg_id", np.int32), ("q_start", np.int32), ("q_end", np.int32), ("hole", np.int32), ("read_qual", np.float32),
                    ("ctxt", np.uint8), ("file_offset", np.int64)):
        out[key] = np.frombuffer(data, dt, n, p).copy(); p += n * np.dtype(dt).itemsize
    assert p == len(data)
    return out


def record_virtual_offsets(path):
    """BGZF virtual offset (block start << 16 | offset in the inflated block) of every BAM record, by walking the blocks"""
    import zlib
    raw = open(path, "rb").read()
    blocks, p, upos = [], 0, 0                      # (uncompressed start, compressed start, inflated size)
    chunks = []
    while p < len(raw):
        assert raw[p:p + 2] == b"\x1f\x8b"
        xlen, = struct.unpack_from("<H", raw, p + 10)
        q, bsize = p + 12, None
        while q < p + 12 + xlen:
            si, sl = raw[q:q + 2], struct.unpack_from("<H", raw, q + 2)[0]
            if si == b"BC": bsize = struct.unpack_from("<H", raw, q + 4)[0] + 1
            q += 4 + sl
        payload = zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
        blocks.append((upos, p, len(payload))); chunks.append(payload)
        upos += len(payload); p += bsize
    data = b"".join(chunks)
    l_text, = struct.unpack_from("<i", data, 4)
    u = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, u); u += 4
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, u); u += 4 + l + 4
    starts = np.array([b[0] for b in blocks]); sizes = np.array([b[2] for b in blocks])
    offs = []
    while u < len(data):
        k = int(np.searchsorted(starts, u, side="right") - 1)
        while sizes[k] == 0 or u >= starts[k] + sizes[k]: k += 1          # (empty blocks)
        offs.append((blocks[k][1] << 16) | (u - blocks[k][0]))
        bs, = struct.unpack_from("<i", data, u); u += 4 + bs
    return np.array(offs, np.int64)
