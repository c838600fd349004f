"""Tiny BAM reader/writer for the tests (plays the role noodles plays in the reference's integration test)."""
import gzip
import struct
import zlib

_CODE = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path, header_text, refs, records):
    """records: dicts name, flags, seq, qual (Phred+33 string), tags: list of (tag, type, value) with type in Z,i,f,A"""
    out = b"BAM\x01" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    for r in records:
        seq, name = r["seq"], r["name"].encode() + b"\0"
        packed = bytearray()
        for i in range(0, len(seq), 2):
            packed.append(_CODE.index(seq[i]) << 4 | (_CODE.index(seq[i + 1]) if i + 1 < len(seq) else 0))
        aux = b""
        for tag, ty, val in r.get("tags", []):
            aux += tag.encode() + ty.encode()
            aux += {"Z": lambda v: v.encode() + b"\0", "i": lambda v: struct.pack("<i", v), "f": lambda v: struct.pack("<f", v),
                    "A": lambda v: v.encode()}[ty](val)
        body = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, r["flags"], len(seq), -1, -1, 0) + name + bytes(packed) + \
            bytes(ord(c) - 33 for c in r["qual"]) + aux
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for i in range(0, len(out), 60000):
            f.write(_bgzf_block(out[i:i + 60000]))
        f.write(_bgzf_block(b""))


def read_bam(path):
    """-> (header_text, refs, records) with decoded fields and a dict of aux tags"""
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", d, 4)
    text = d[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        ln, = struct.unpack_from("<i", d, o)
        name = d[o + 4:o + 4 + ln - 1].decode()
        lr, = struct.unpack_from("<i", d, o + 4 + ln)
        refs.append((name, lr))
        o += 8 + ln
    recs = []
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        b = d[o + 4:o + 4 + bs]
        o += 4 + bs
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        name = b[p:p + l_name - 1].decode()
        p += l_name
        cigar = ""
        for i in range(n_cig):
            v, = struct.unpack_from("<I", b, p + 4 * i)
            cigar += f"{v >> 4}{'MIDNSHP=X'[v & 15]}"
        p += 4 * n_cig
        seq = "".join(_CODE[(b[p + i // 2] >> (0 if i % 2 else 4)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = "".join(chr(q + 33) for q in b[p:p + l_seq])
        p += l_seq
        tags, order = {}, []
        while p < len(b):
            tag, ty = b[p:p + 2].decode(), chr(b[p + 2])
            p += 3
            if ty == "Z":
                e = b.index(b"\0", p)
                val = b[p:e].decode()
                p = e + 1
            elif ty == "A":
                val = chr(b[p]); p += 1
            elif ty in "cC":
                val = struct.unpack_from("<b" if ty == "c" else "<B", b, p)[0]; p += 1
            elif ty in "sS":
                val = struct.unpack_from("<h" if ty == "s" else "<H", b, p)[0]; p += 2
            elif ty in "iI":
                val = struct.unpack_from("<i" if ty == "i" else "<I", b, p)[0]; p += 4
            elif ty == "f":
                val = struct.unpack_from("<f", b, p)[0]; p += 4
            elif ty == "B":
                sub, n = chr(b[p]), struct.unpack_from("<I", b, p + 1)[0]
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                val = (sub, list(struct.unpack_from(f"<{n}{fmt}", b, p + 5))); p += 5 + n * struct.calcsize(fmt)
            else:
                raise ValueError(ty)
            tags[tag] = (ty, val)
            order.append(tag)
        recs.append({"name": name, "flags": flag, "tid": tid, "pos": pos, "mapq": mapq, "cigar": cigar, "seq": seq, "qual": qual,
                     "tags": tags, "tag_order": order, "bin": _bin})
    return text, refs, recs
