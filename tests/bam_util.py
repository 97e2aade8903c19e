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
