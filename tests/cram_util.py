"""A small CRAM 3.0 writer for the tests of mapad_amd/csrc/cli/cram_io.hpp.

Written from the CRAM format specification 3.0 (samtools/hts-specs), like the reader it exercises: there is no CRAM file, no samtools and
no noodles in this image, so the two can only be checked against each other and against the specification's rules restated here (parity
unpinned).  The writer deliberately spreads the data series over every encoding and block compression the reader knows:
EXTERNAL, HUFFMAN (constant and multi-symbol), BETA, GAMMA, SUBEXP, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP; raw, gzip, rANS 4x8 order 0 and 1.
"""
import gzip
import struct
import zlib


# ---- integers ---------------------------------------------------------------------------------------------------------------------
def itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v):
    for extra in range(9):
        bits = 7 - extra + 8 * extra if extra < 8 else 64
        if v < (1 << bits) or extra == 8:
            body = v.to_bytes(extra + 1, "big") if extra < 8 else b"\x00" + v.to_bytes(8, "big")
            first = ((0xFF << (8 - extra)) & 0xFF) | (body[0] if extra < 7 else 0)
            return bytes([first]) + body[1:]
    raise ValueError(v)


# ---- rANS 4x8 ---------------------------------------------------------------------------------------------------------------------
def _normalise(counts, total=4095):
    n = sum(counts.values())
    f = {s: max(1, c * total // n) for s, c in counts.items()}
    top = max(f, key=lambda s: f[s])
    f[top] += total - sum(f.values())
    assert f[top] >= 1 and sum(f.values()) == total
    return f


def _symbol_list(syms, payload):
    """symbols ascending, runs of consecutive symbols as {first, second, run length}; payload(s) = the bytes that follow symbol s"""
    out = bytearray()
    syms = sorted(syms)
    implied = 0
    for i, s in enumerate(syms):
        if implied:
            implied -= 1
        else:
            out.append(s)
            if i > 0 and syms[i - 1] == s - 1:
                run = 0
                while i + run + 1 < len(syms) and syms[i + run + 1] == s + run + 1:
                    run += 1
                out.append(run)
                implied = run
        out += payload(s)
    out.append(0)
    return bytes(out)


def _freq_bytes(f):
    return bytes([f]) if f < 128 else bytes([0x80 | (f >> 8), f & 0xFF])


class _Rans:
    L = 1 << 23

    def __init__(self):
        self.x = [self.L] * 4
        self.rev = bytearray()

    def put(self, k, f, c):
        x = self.x[k]
        x_max = ((self.L >> 12) << 8) * f
        while x >= x_max:
            self.rev.append(x & 0xFF)
            x >>= 8
        self.x[k] = ((x // f) << 12) + (x % f) + c

    def finish(self):
        for k in (3, 2, 1, 0):
            self.rev += self.x[k].to_bytes(4, "big")
        return bytes(reversed(self.rev))


def rans_encode(data, order):
    n = len(data)
    if order == 0:
        counts = {}
        for b in data:
            counts[b] = counts.get(b, 0) + 1
        F = _normalise(counts) if n else {0: 4095}
        C, acc = {}, 0
        for s in sorted(F):
            C[s] = acc
            acc += F[s]
        table = _symbol_list(F.keys(), lambda s: _freq_bytes(F[s]))
        r = _Rans()
        full = n & ~3
        for k in reversed(range(n & 3)):
            r.put(k, F[data[full + k]], C[data[full + k]])
        for i in range(full - 4, -1, -4):
            for k in (3, 2, 1, 0):
                r.put(k, F[data[i + k]], C[data[i + k]])
        body = table + r.finish()
    else:
        q = n >> 2
        ctx_of = lambda pos, i: data[pos - 1] if i > 0 else 0  # noqa: E731
        counts = {}
        for k in range(4):
            hi = n if k == 3 else (k + 1) * q
            for pos in range(k * q, hi):
                c = ctx_of(pos, pos - k * q)
                counts.setdefault(c, {})
                counts[c][data[pos]] = counts[c].get(data[pos], 0) + 1
        if not counts:
            counts = {0: {0: 1}}
        F = {c: _normalise(v) for c, v in counts.items()}
        C = {}
        for c, f in F.items():
            acc = 0
            C[c] = {}
            for s in sorted(f):
                C[c][s] = acc
                acc += f[s]
        table = _symbol_list(F.keys(), lambda c: _symbol_list(F[c].keys(), lambda s: _freq_bytes(F[c][s])))
        r = _Rans()
        for pos in range(n - 1, 4 * q - 1, -1):
            c = ctx_of(pos, pos - 3 * q)
            r.put(3, F[c][data[pos]], C[c][data[pos]])
        for i in range(q - 1, -1, -1):
            for k in (3, 2, 1, 0):
                pos = k * q + i
                c = ctx_of(pos, i)
                r.put(k, F[c][data[pos]], C[c][data[pos]])
        body = table + r.finish()
    return bytes([order]) + struct.pack("<II", len(body), n) + body


# ---- blocks and containers ------------------------------------------------------------------------------------------------------
RAW, GZIP, RANS0, RANS1 = "raw", "gzip", "rans0", "rans1"
FILE_HEADER, COMPRESSION_HEADER, SLICE_HEADER, EXTERNAL_DATA, CORE_DATA = 0, 1, 2, 4, 5


MANGLE = None  # tests of damaged files: a function applied to a block's compressed bytes before the checksum is taken


def block(method, content_type, content_id, data):
    data = bytes(data)
    if method == RAW:
        m, comp = 0, data
    elif method == GZIP:
        m, comp = 1, gzip.compress(data)
    elif method in (RANS0, RANS1):
        m, comp = 4, rans_encode(data, 0 if method == RANS0 else 1)
    else:  # a method id the reader must refuse (bzip2 = 2, lzma = 3, the CRAM 3.1 codecs 5-8)
        m, comp = int(method), data
    if MANGLE is not None and content_type == EXTERNAL_DATA:
        comp = MANGLE(m, comp)
    b = bytes([m, content_type]) + itf8(content_id) + itf8(len(comp)) + itf8(len(data)) + comp
    return b + struct.pack("<I", zlib.crc32(b))


def container(ref_id, start, span, n_records, record_counter, bases, blocks, landmarks):
    body = b"".join(blocks)
    h = struct.pack("<i", len(body)) + itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_records) + ltf8(record_counter) + ltf8(bases) + itf8(len(blocks))
    h += itf8(len(landmarks)) + b"".join(itf8(x) for x in landmarks)
    return h + struct.pack("<I", zlib.crc32(h)) + body


def eof_container():
    blk = block(RAW, COMPRESSION_HEADER, 0, itf8(1) + itf8(0) + itf8(1) + itf8(0) + itf8(1) + itf8(0))  # three empty maps
    return container(-1, 4542278, 0, 0, 0, 0, [blk], [])


class Bits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for i in reversed(range(n)):
            self.bits.append((v >> i) & 1)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(sum(b[i + k] << (7 - k) for k in range(8)) for i in range(0, len(b), 8))


# ---- encodings: a description (what goes into the compression header) plus how a value is written ------------------------------------
class Enc:
    def params(self):
        raise NotImplementedError

    def header(self):
        p = self.params()
        return itf8(self.kind) + itf8(len(p)) + p


class External(Enc):
    kind = 1

    def __init__(self, cid, as_bytes=False):
        self.cid, self.as_bytes = cid, as_bytes

    def params(self):
        return itf8(self.cid)

    def put(self, st, v):
        st.ext.setdefault(self.cid, bytearray()).extend(bytes([v]) if self.as_bytes else itf8(v))

    def put_raw(self, st, data):
        st.ext.setdefault(self.cid, bytearray()).extend(data)


class Huffman(Enc):
    kind = 3

    def __init__(self, lengths):  # {symbol: code length}
        self.lengths = dict(lengths)
        order = sorted(self.lengths, key=lambda s: (self.lengths[s], s))
        self.codes, code, prev = {}, 0, self.lengths[order[0]]
        for s in order:
            code <<= self.lengths[s] - prev
            prev = self.lengths[s]
            self.codes[s] = code
            code += 1

    def params(self):
        syms = list(self.lengths)
        return itf8(len(syms)) + b"".join(itf8(s) for s in syms) + itf8(len(syms)) + b"".join(itf8(self.lengths[s]) for s in syms)

    def put(self, st, v):
        st.core.put(self.codes[v], self.lengths[v])


class Beta(Enc):
    kind = 6

    def __init__(self, offset, nbits):
        self.offset, self.nbits = offset, nbits

    def params(self):
        return itf8(self.offset) + itf8(self.nbits)

    def put(self, st, v):
        assert 0 <= v + self.offset < (1 << self.nbits)
        st.core.put(v + self.offset, self.nbits)


class Gamma(Enc):
    kind = 9

    def __init__(self, offset):
        self.offset = offset

    def params(self):
        return itf8(self.offset)

    def put(self, st, v):
        x = v + self.offset
        assert x >= 1
        nb = x.bit_length() - 1
        st.core.put(0, nb)
        st.core.put(x, nb + 1)


class Subexp(Enc):
    kind = 7

    def __init__(self, offset, k):
        self.offset, self.k = offset, k

    def params(self):
        return itf8(self.offset) + itf8(self.k)

    def put(self, st, v):
        x = v + self.offset
        assert x >= 0
        if x < (1 << self.k):
            st.core.put(0, 1)
            st.core.put(x, self.k)
        else:
            b = x.bit_length() - 1
            u = b - self.k + 1
            st.core.put((1 << u) - 1, u)
            st.core.put(0, 1)
            st.core.put(x & ((1 << b) - 1), b)


class ByteArrayStop(Enc):
    kind = 5

    def __init__(self, stop, cid):
        self.stop, self.cid = stop, cid

    def params(self):
        return bytes([self.stop]) + itf8(self.cid)

    def put(self, st, data):
        assert self.stop not in data
        st.ext.setdefault(self.cid, bytearray()).extend(bytes(data) + bytes([self.stop]))


class ByteArrayLen(Enc):
    kind = 4

    def __init__(self, len_enc, val_enc):
        self.len_enc, self.val_enc = len_enc, val_enc

    def params(self):
        return self.len_enc.header() + self.val_enc.header()

    def put(self, st, data):
        self.len_enc.put(st, len(data))
        if isinstance(self.val_enc, External):
            self.val_enc.put_raw(st, data)
        else:
            for b in data:
                self.val_enc.put(st, b)


class SliceStreams:
    def __init__(self):
        self.core = Bits()
        self.ext = {}


def compression_header(series, tag_encs, tag_lines, names=True, ap_delta=True, ref_required=False, sub_matrix=None):
    """series: {"BF": Enc, ...}; tag_encs: {(b"XI", "Z"): Enc}; tag_lines: list of lists of (tag, type)"""
    td = b"".join(b"".join(t + ty.encode() for t, ty in line) + b"\0" for line in tag_lines)
    pm = [(b"RN", bytes([int(names)])), (b"AP", bytes([int(ap_delta)])), (b"RR", bytes([int(ref_required)])), (b"TD", itf8(len(td)) + td)]
    if sub_matrix is not None:
        pm.append((b"SM", bytes(sub_matrix)))
    pmap = itf8(len(pm)) + b"".join(k + v for k, v in pm)
    smap = itf8(len(series)) + b"".join(k.encode() + e.header() for k, e in series.items())
    tmap = itf8(len(tag_encs)) + b"".join(itf8(t[0] << 16 | t[1] << 8 | ord(ty)) + e.header() for (t, ty), e in tag_encs.items())
    return itf8(len(pmap)) + pmap + itf8(len(smap)) + smap + itf8(len(tmap)) + tmap


def slice_blocks(ref_id, start, span, n_records, record_counter, streams, methods, embedded_ref=None, embedded_id=99):
    """the slice header block followed by the core block and the external blocks; methods: {content id: compression} (default gzip)"""
    ext = dict(streams.ext)
    if embedded_ref is not None:
        ext[embedded_id] = embedded_ref
    ids = sorted(ext)
    sh = itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_records) + ltf8(record_counter) + itf8(1 + len(ids)) + itf8(len(ids)) + b"".join(itf8(i) for i in ids)
    sh += itf8(embedded_id if embedded_ref is not None else -1) + bytes(16)
    out = [block(RAW, SLICE_HEADER, 0, sh), block(methods.get("core", RAW), CORE_DATA, 0, streams.core.bytes())]
    for i in ids:
        out.append(block(methods.get(i, GZIP), EXTERNAL_DATA, i, ext[i]))
    return out


def file_start(header_text, minor=0):
    text = header_text.encode()
    hdr_block = block(GZIP, FILE_HEADER, 0, struct.pack("<i", len(text)) + text)
    return b"CRAM" + bytes([3, minor]) + b"mapad-amd test file\0"[:20].ljust(20, b"\0") + container(0, 0, 0, 0, 0, 0, [hdr_block], [0])


def aux_value(ty, v):
    """the BAM encoding of one tag value (what CRAM stores for a tag)"""
    if ty == "Z":
        return v.encode() + b"\0"
    if ty == "i":
        return struct.pack("<i", v)
    if ty == "C":
        return bytes([v])
    if ty == "f":
        return struct.pack("<f", v)
    if ty == "A":
        return v.encode()
    raise ValueError(ty)
