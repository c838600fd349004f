"""`mapad-amd worker`: the reference's dispatcher <-> worker protocol (src/distributed/mod.rs:14-44, worker.rs:45-236,
input_chunk_reader.rs:246-253).  A stand-in dispatcher written here from the reference's struct definitions and bincode 1.3's format
(fixed-width little-endian integers, u64 lengths, u32 enum variants) sends TaskSheets and decodes the ResultSheets.

Wire parity is UNPINNED: no reference binary can run in this environment, so encoder and decoder below are checked against the same
reading of the reference's sources as csrc/cli/wire.hpp, not against bytes a real dispatcher produced.
"""
import socket
import struct
import subprocess

import numpy as np
import pytest

import mapad_amd
from mapad_amd import build as mbuild
from mapad_amd import presets, synth

# BamAuxField variants (record.rs:21-42) -> (variant index, payload encoder)
_AUX = {"A": (0, lambda v: struct.pack("<B", v)), "c": (1, lambda v: struct.pack("<b", v)), "C": (2, lambda v: struct.pack("<B", v)),
        "s": (3, lambda v: struct.pack("<h", v)), "S": (4, lambda v: struct.pack("<H", v)), "i": (5, lambda v: struct.pack("<i", v)),
        "I": (6, lambda v: struct.pack("<I", v)), "f": (7, lambda v: struct.pack("<f", v)), "d": (8, lambda v: struct.pack("<d", v)),
        "Z": (9, lambda v: struct.pack("<Q", len(v)) + v), "H": (10, lambda v: struct.pack("<Q", len(v)) + v),
        "Bc": (11, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}b", *v)), "BC": (12, lambda v: struct.pack("<Q", len(v)) + bytes(v)),
        "Bs": (13, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}h", *v)), "BS": (14, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}H", *v)),
        "Bi": (15, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}i", *v)), "BI": (16, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}I", *v)),
        "Bf": (17, lambda v: struct.pack("<Q", len(v)) + struct.pack(f"<{len(v)}f", *v))}


def encode_record(seq, quals, name=None, tags=(), flags=0):
    out = struct.pack("<Q", len(seq)) + bytes(seq) + struct.pack("<Q", len(quals)) + bytes(quals)
    out += b"\x00" if name is None else b"\x01" + struct.pack("<Q", len(name)) + name
    out += struct.pack("<Q", len(tags))
    for tag, kind, value in tags:
        variant, enc = _AUX[kind]
        out += tag + struct.pack("<I", variant) + enc(value)
    return out + struct.pack("<H", flags)


def encode_params(p):
    """AlignmentParameters (map/mod.rs:21-31) from a mapad_params_t; the caches the reference carries along are sent empty."""
    vec0 = struct.pack("<Q", 0)
    if p.model_kind == 0:
        lib = (struct.pack("<Iff", 0, p.five_prime_overhang, p.three_prime_overhang) if p.library_prep == 0 else struct.pack("<If", 1, p.five_prime_overhang))
        use_default = b"\x01" + struct.pack("<f", 1e-3) if p.ignore_base_quality else b"\x00"
        model = struct.pack("<I", 0) + lib + struct.pack("<fff", p.ds_deamination_rate, p.ss_deamination_rate, p.divergence) + use_default
        model += struct.pack("<Q", 3) + struct.pack("<fff", 0.1, 0.2, 0.3) + b"\x01" + struct.pack("<h", 7)  # a cache and a flank offset to be skipped
    elif p.model_kind == 1:  # VindijaPwm { [f32; 7], f32, f32 } (sequence_difference_models.rs:340-345,384-396): a fixed-size array has no length on the wire
        model = struct.pack("<I9f", 1, 0.4, 0.25, 0.1, 0.06, 0.05, 0.04, 0.03, 0.02, 0.0005)
    else:
        model = struct.pack("<Ifff", 2, p.deam_score, p.mm_score, p.match_score)
    if p.bound_kind == 0:    # MAPAD_BOUND_DISCRETE = variant 1 of MismatchBoundDispatch
        bound = struct.pack("<Ifff", 1, p.poisson_threshold, p.base_error_rate, -7.0) + vec0
    elif p.bound_kind == 1:  # Continuous = variant 0
        bound = struct.pack("<Ifff", 0, p.cutoff, p.exponent, -7.0) + struct.pack("<Q", 2) + struct.pack("<ff", 1.0, 2.0)
    else:
        bound = struct.pack("<Iff", 2, p.threshold, p.repr_mm_bound)
    return model + bound + struct.pack("<ffQBBB", p.penalty_gap_open, p.penalty_gap_extend, p.chunk_size, p.gap_dist_ends, p.max_num_gaps_open, 1 if p.stack_limit_abort else 0)


def encode_task(chunk_id, records, reference=None, params=None):
    body = struct.pack("<Q", chunk_id) + struct.pack("<Q", len(records)) + b"".join(records)
    body += b"\x00" if reference is None else b"\x01" + struct.pack("<Q", len(reference)) + reference
    body += b"\x00" if params is None else b"\x01" + encode_params(params)
    return struct.pack("<Q", len(body) + 8) + body


def _recv(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        assert chunk, "worker closed the connection"
        buf += chunk
    return buf


def read_result(conn, records):
    """-> (chunk_id, [per record: list of (lower, lower_rev, size, score_bits, ops)], durations); checks that every record is echoed byte for byte"""
    size, = struct.unpack("<Q", _recv(conn, 8))
    msg = _recv(conn, size - 8)
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, msg, pos)
        pos += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    chunk_id, n = take("<Q"), take("<Q")
    assert n == len(records)
    hits, durations = [], []
    for rec in records:
        assert msg[pos:pos + len(rec)] == rec
        pos += len(rec)
        per_read = []
        for _ in range(take("<Q")):
            lower, lower_rev, sz = take("<QQQ")
            score_bits = take("<I")
            ops = []
            for _ in range(take("<Q")):
                kind = take("<I")
                p = take("<H")
                base = take("<B") if kind in (1, 3) else 0
                ops.append(kind << 24 | base << 16 | p)
            per_read.append((lower, lower_rev, sz, score_bits, ops))
        hits.append(per_read)
        secs, nanos = take("<QI")
        assert nanos < 10 ** 9
        durations.append(secs + nanos * 1e-9)
    assert pos == len(msg)
    return chunk_id, hits, durations


def _serve():
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    srv.settimeout(120)
    return srv, srv.getsockname()[1]


def _cli():
    mapad_amd.lib()
    return mbuild.build_cli()


def test_worker_framing_and_record_echo_without_a_gpu():
    """--dry_run: tasks are decoded (every BamAuxField variant, names present and absent, parameters of both bounds) and answered with
    the records byte for byte and empty hit heaps; the worker leaves when the dispatcher closes the connection (worker.rs:203-206)."""
    srv, port = _serve()
    proc = subprocess.Popen([_cli(), "worker", "--host", "127.0.0.1", "--port", str(port), "--dry_run"], stderr=subprocess.PIPE)
    conn, _ = srv.accept()
    all_tags = [(b"XA", "A", 65), (b"Xc", "c", -5), (b"XC", "C", 200), (b"Xs", "s", -300), (b"XS", "S", 60000), (b"Xi", "i", -70000), (b"XI", "I", 4000000000),
                (b"Xf", "f", 1.5), (b"Xd", "d", 2.25), (b"XZ", "Z", b"hello"), (b"XH", "H", b"1AE3"), (b"Ba", "Bc", [-1, 2]), (b"Bb", "BC", [1, 2, 3]),
                (b"Bd", "Bs", [-300]), (b"Be", "BS", [60000, 1]), (b"Bg", "Bi", [-70000]), (b"Bh", "BI", [4000000000]), (b"Bj", "Bf", [0.5, 0.25])]
    recs = [encode_record(b"ACGTACGT", [30] * 8, b"read1", all_tags, 0x4D), encode_record(b"", [], None, (), 0), encode_record(b"GATTACA", [40] * 7, b"r3")]
    vindija = mapad_amd.make_params(presets.resolve(presets.DAMAGE))
    vindija.model_kind = 1  # round 5: the worker takes VindijaPwm (worker.rs:57-75 takes any model) — its values are VindijaPwm::new()'s constants, checked on arrival
    for chunk, (ref, prm) in enumerate([(b"/nonexistent/ref.fa", mapad_amd.make_params(presets.resolve(presets.DAMAGE))), (None, None),
                                        (b"x", mapad_amd.make_params(presets.resolve(presets.CONTINUOUS))), (b"y", vindija)]):
        conn.sendall(encode_task(100 + chunk, recs, ref, prm))
        chunk_id, hits, durations = read_result(conn, recs)
        assert chunk_id == 100 + chunk and hits == [[], [], []] and len(durations) == 3
    conn.close()
    assert proc.wait(timeout=60) == 0
    assert b"4 task(s), 12 reads" in proc.stderr.read()
    srv.close()


@pytest.mark.parametrize("damage", ["short_body", "bad_option_tag", "trailing_bytes", "unknown_model"])
def test_worker_rejects_malformed_tasks(damage):
    """a task that does not decode ends the worker with an error message and a non-zero status (the reference: InvalidData, worker.rs:229-235) —
    it is never answered and never crashes the process"""
    srv, port = _serve()
    proc = subprocess.Popen([_cli(), "worker", "--host", "127.0.0.1", "--port", str(port), "--dry_run"], stderr=subprocess.PIPE)
    conn, _ = srv.accept()
    good = encode_task(1, [encode_record(b"ACGT", [30] * 4, b"r")], b"ref", mapad_amd.make_params(presets.resolve(presets.DAMAGE)))
    if damage == "short_body":       # the header promises more records than the body holds
        msg = bytearray(good); msg[16:24] = struct.pack("<Q", 5); msg = bytes(msg)
    elif damage == "bad_option_tag":  # Option tag of the record name is neither 0 nor 1
        rec = bytearray(encode_record(b"ACGT", [30] * 4, b"r")); rec[8 + 4 + 8 + 4] = 7
        msg = encode_task(1, [bytes(rec)])
    elif damage == "trailing_bytes":
        body = good[8:] + b"xx"
        msg = struct.pack("<Q", len(body) + 8) + body
    else:                            # VindijaPwm (variant 1) with values other than VindijaPwm::new()'s
        body = struct.pack("<QQ", 1, 0) + b"\x00" + b"\x01" + struct.pack("<I", 1) + b"\x00" * 64
        msg = struct.pack("<Q", len(body) + 8) + body
    conn.sendall(msg)
    assert proc.wait(timeout=60) != 0
    err = proc.stderr.read().decode()
    assert any(k in err for k in ("shorter than its contents", "bad Option tag", "trailing bytes", "VindijaPwm", "bad sequence length", "bad record count"))
    conn.close(); srv.close()


@pytest.mark.gpu
def test_worker_returns_the_hits_of_the_batch_api(tmp_path):
    """Two tasks (the first names the index and the parameters): hits, scores and edit tracks per record equal mapad_map_batch's,
    in BinaryHeap array order; an empty and an over-length read come back without hits."""
    g = synth.genome(150_000, seed=5)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as f:
        f.write(">chr1\n" + g.tobytes().decode() + "\n")
    subprocess.check_call([_cli(), "index", "-g", fa])
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=6, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(30, 90), indel_frac=0.05)
    params = mapad_amd.make_params(presets.resolve(presets.DAMAGE))
    idx = mapad_amd.Index.open(fa)
    ctx = mapad_amd.Context(idx, params, 0)
    want = ctx.map_batch(seqs, quals, offsets)
    ctx.close()

    def rec(i):
        s, e = int(offsets[i]), int(offsets[i + 1])
        return encode_record(seqs[s:e].tobytes(), quals[s:e].tolist(), b"r%d" % i, [(b"XI", "Z", b"ACGTAC")] if i % 3 == 0 else (), 0)

    srv, port = _serve()
    proc = subprocess.Popen([_cli(), "worker", "--host", "127.0.0.1", "--port", str(port)], stderr=subprocess.PIPE)
    conn, _ = srv.accept()
    long_read = g[:33_000].tobytes()  # beyond i16::MAX (record.rs:144-150)
    n_hits = 0
    for chunk, (lo, hi) in enumerate([(0, 900), (900, 1500)]):
        recs = [rec(i) for i in range(lo, hi)] + [encode_record(b"", [], b"empty"), encode_record(long_read, [30] * len(long_read), b"too_long")]
        conn.sendall(encode_task(chunk, recs, fa.encode() if chunk == 0 else None, params if chunk == 0 else None))
        chunk_id, hits, _ = read_result(conn, recs)
        assert chunk_id == chunk and hits[-2:] == [[], []]
        for k, i in enumerate(range(lo, hi)):
            h0, h1 = int(want.hit_begin[i]), int(want.hit_begin[i + 1])
            exp = []
            for h in want.hits_arr[h0:h1]:
                o = int(h["ops_offset"])
                exp.append((int(h["lower"]), int(h["lower_rev"]), int(h["size"]), int(np.float32(h["score"]).view(np.uint32)), want.ops[o:o + int(h["n_ops"])].tolist()))
            assert hits[k] == exp
            n_hits += len(exp)
    conn.close()
    assert proc.wait(timeout=120) == 0
    srv.close()
    assert n_hits > 1000
