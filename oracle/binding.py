"""ctypes binding of the CPU ORACLE (oracle/_build/libmapad_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the product package (mapad_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmapad_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("model_kind", C.c_int32), ("library_prep", C.c_int32),
        ("five_prime_overhang", C.c_float), ("three_prime_overhang", C.c_float),
        ("ds_deamination_rate", C.c_float), ("ss_deamination_rate", C.c_float), ("divergence", C.c_float),
        ("ignore_base_quality", C.c_int32),
        ("deam_score", C.c_float), ("mm_score", C.c_float), ("match_score", C.c_float),
        ("bound_kind", C.c_int32),
        ("poisson_threshold", C.c_float), ("base_error_rate", C.c_float),
        ("cutoff", C.c_float), ("exponent", C.c_float),
        ("threshold", C.c_float), ("repr_mm_bound", C.c_float),
        ("penalty_gap_open", C.c_float), ("penalty_gap_extend", C.c_float),
        ("gap_dist_ends", C.c_int32), ("max_num_gaps_open", C.c_int32), ("stack_limit_abort", C.c_int32),
        ("stack_limit", C.c_uint32), ("edit_tree_limit", C.c_uint32),
        ("heap_variant", C.c_int32),
    ]


MODEL_KINDS = {"simple_adna": 0, "vindija_pwm": 1, "test": 2}
BOUND_KINDS = {"discrete": 0, "continuous": 1, "test": 2}
LIBRARY_PREPS = {"single_stranded": 0, "double_stranded": 1}


def f32(x):
    return float(np.float32(x))


def make_params(d):
    """dict (fixture style) -> Params.  Unknown keys raise."""
    p = Params()
    d = dict(d)
    p.model_kind = MODEL_KINDS[d.pop("model")]
    p.bound_kind = BOUND_KINDS[d.pop("bound")]
    p.library_prep = LIBRARY_PREPS[d.pop("library", "single_stranded")]
    for k, v in d.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH) for f in ("capi.cpp", "mapad_oracle.hpp")
    ):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u64, u32, i32, i64, f = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int64, C.c_float
        PP = C.POINTER(Params)
        sig = {
            "mo_index_build": (vp, [C.c_char_p, u64, C.c_char_p, u32]),
            "mo_index_from_bwt": (vp, [vp, u64, C.c_char_p, u32]),
            "mo_index_free": (None, [vp]),
            "mo_index_len": (u64, [vp]),
            "mo_index_get_bwt": (None, [vp, vp]),
            "mo_index_get_sa": (i32, [vp, vp]),
            "mo_index_get_less": (None, [vp, vp, i32]),
            "mo_index_occ": (u64, [vp, u64, i32]),
            "mo_index_extend": (None, [vp, u64, u64, u64, vp]),
            "mo_index_add_contig": (None, [vp, u64, u64, C.c_char_p]),
            "mo_index_set_original_symbol": (None, [vp, u64, i32]),
            "mo_index_sample_sa": (i32, [vp, u64]),
            "mo_index_set_sampled_sa": (None, [vp, vp, u64, u64, vp, vp, u64]),
            "mo_ssa_get": (i32, [vp, u64, C.POINTER(u64)]),
            "mo_sdm_get": (f, [PP, u64, u64, i32, i32, i32]),
            "mo_sdm_repr_mm": (f, [PP]),
            "mo_sdm_min_penalty": (f, [PP, u64, u64, i32, i32, i32]),
            "mo_sdm_alignment_start": (i32, [PP, u64]),
            "mo_mb_reject": (i32, [PP, f, u64]),
            "mo_mb_reject_iterative": (i32, [PP, f, f]),
            "mo_mb_remaining_frac": (f, [PP, f, u64]),
            "mo_discrete_get": (f, [f, f, u64]),
            "mo_sdm_table": (None, [PP, u64, i32, vp]),
            "mo_d_array": (i32, [vp, PP, C.c_char_p, vp, u64, i64, vp]),
            "mo_d_array_get": (f, [vp, PP, C.c_char_p, vp, u64, i64, i32, i32]),
            "mo_map_batch": (vp, [vp, PP, vp, vp, vp, u64, i32, i32]),
            "mo_result_free": (None, [vp]),
            "mo_result_total_hits": (u64, [vp]),
            "mo_result_total_ops": (u64, [vp]),
            "mo_result_export": (None, [vp, vp, vp, vp, vp, vp]),
            "mo_result_counters": (None, [vp, vp]),
            "mo_result_d_array": (i32, [vp, u64, vp]),
            "mo_hit_bam_fields": (i32, [vp, vp, u64, u64, i32, u64, i32, C.c_char_p, C.c_char_p, C.POINTER(i32)]),
            "mo_records_tsv": (vp, [vp, vp, PP, vp, vp, vp, vp, u64]),
            "mo_free": (None, [vp]),
            "mo_records_from_hits": (vp, [vp, PP, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, i32]),
            "mo_hit_records_text_bytes": (u64, [vp]),
            "mo_hit_records_error": (C.c_char_p, [vp]),
            "mo_hit_records_export": (None, [vp, vp, vp]),
            "mo_hit_records_free": (None, [vp]),
            "mo_prrange": (i64, [u64, u64, u64, vp, u64]),
            "mo_prrange_count": (i32, [u64, u64, u64, u64, C.POINTER(u64), C.POINTER(u64)]),
            "mo_tree_script": (i32, [vp, i32, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


OP_KINDS = ("I", "D", "M", "X")  # Insertion, Deletion, Match, Mismatch


def unpack_op(u):
    u = int(u)
    return (OP_KINDS[u >> 24], u & 0xFFFF, chr((u >> 16) & 0xFF) if (u >> 16) & 0xFF else "")


class OracleIndex:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index build failed")
        self.h = handle

    @classmethod
    def from_text(cls, text: bytes, alphabet="$ACGT", occ_k=3):
        return cls(lib().mo_index_build(text, len(text), alphabet.encode(), occ_k))

    @classmethod
    def from_bwt(cls, bwt: np.ndarray, alphabet="$ACGTX", occ_k=128):
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        return cls(lib().mo_index_from_bwt(_ptr(bwt), bwt.size, alphabet.encode(), occ_k))

    def __del__(self):
        try:
            if self.h:
                lib().mo_index_free(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return int(lib().mo_index_len(self.h))

    def bwt(self):
        out = np.empty(len(self), dtype=np.uint8)
        lib().mo_index_get_bwt(self.h, _ptr(out))
        return out

    def sa(self):
        out = np.empty(len(self), dtype=np.uint64)
        if lib().mo_index_get_sa(self.h, _ptr(out)) != 0:
            raise RuntimeError("no SA")
        return out

    def less(self, n=7):
        out = np.zeros(n, dtype=np.uint64)
        lib().mo_index_get_less(self.h, _ptr(out), n)
        return out

    def occ(self, r, a):
        return int(lib().mo_index_occ(self.h, r, a))

    def extend(self, lower, lower_rev, size):
        out = np.zeros(12, dtype=np.uint64)
        lib().mo_index_extend(self.h, lower, lower_rev, size, _ptr(out))
        return out.reshape(4, 3)

    def add_contig(self, start, end, name):
        lib().mo_index_add_contig(self.h, start, end, name.encode())

    def set_original_symbol(self, pos, sym):
        lib().mo_index_set_original_symbol(self.h, pos, ord(sym))

    def sample_sa(self, rate=32):
        if lib().mo_index_sample_sa(self.h, rate) != 0:
            raise RuntimeError("no SA to sample")

    def set_sampled_sa(self, sample, rate, extra_rows, extra_vals):
        sample = np.ascontiguousarray(sample, dtype=np.uint64)
        er = np.ascontiguousarray(extra_rows, dtype=np.uint64)
        ev = np.ascontiguousarray(extra_vals, dtype=np.uint64)
        lib().mo_index_set_sampled_sa(self.h, _ptr(sample), sample.size, rate, _ptr(er), _ptr(ev), er.size)

    def ssa_get(self, row):
        out = C.c_uint64()
        if lib().mo_ssa_get(self.h, row, C.byref(out)) != 0:
            raise RuntimeError("ssa_get failed")
        return out.value

    def d_array(self, params, seq: bytes, qual, split=-1):
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        out = np.zeros(len(seq), dtype=np.float32)
        lib().mo_d_array(self.h, C.byref(params), seq, _ptr(qual), len(seq), split, _ptr(out))
        return out

    def d_array_get(self, params, seq: bytes, qual, split, k, l):
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        return lib().mo_d_array_get(self.h, C.byref(params), seq, _ptr(qual), len(seq), split, k, l)

    def records_from_hits(self, params, hit_begin, intervals, scores, op_begin, ops, seqs, quals, offsets, flags=None, first_read_index=0, n_threads=1):
        """intervals_to_record (mapping.rs:402-718 restated) over hits found elsewhere, given per read in BinaryHeap array order: hit_begin u64[n + 1], intervals
        u64[n_hits, 3], scores f32[n_hits], op_begin u64[n_hits + 1] into the packed edit operations `ops`.  The index needs its sampled suffix array and contigs
        (set_sampled_sa / sample_sa, add_contig).  Returns (records: MO_RECORD_DTYPE[n], text: uint8 — [CIGAR][MD][XA] of the mapped reads in read order)."""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)  # noqa: E731
        hit_begin, intervals, scores, op_begin, ops = a(hit_begin, np.uint64), a(intervals, np.uint64).reshape(-1, 3), a(scores, np.float32), a(op_begin, np.uint64), a(ops, np.uint32)
        seqs, quals, offsets = a(seqs, np.uint8), a(quals, np.uint8), a(offsets, np.uint64)
        n = offsets.size - 1
        assert hit_begin.size == n + 1 and op_begin.size == int(hit_begin[-1]) + 1 and intervals.shape[0] == int(hit_begin[-1]) == scores.size
        fl = a(flags, np.uint16) if flags is not None else None
        pad = lambda x: x if x.size else np.zeros(1, x.dtype)  # noqa: E731
        h = lib().mo_records_from_hits(self.h, C.byref(params), _ptr(hit_begin), _ptr(pad(intervals.reshape(-1))), _ptr(pad(scores)), _ptr(op_begin), _ptr(pad(ops)),
                                       _ptr(pad(seqs)), _ptr(pad(quals)), _ptr(offsets), _ptr(fl) if fl is not None else None, n, int(first_read_index), int(n_threads))
        try:
            err = lib().mo_hit_records_error(h)
            if err:
                raise RuntimeError("oracle intervals_to_record: " + err.decode())
            recs = np.zeros(max(n, 1), MO_RECORD_DTYPE)
            text = np.zeros(max(int(lib().mo_hit_records_text_bytes(h)), 1), np.uint8)
            lib().mo_hit_records_export(h, _ptr(recs), _ptr(text))
            return recs[:n], text[:int(lib().mo_hit_records_text_bytes(h))]
        finally:
            lib().mo_hit_records_free(h)

    def map_batch(self, params, reads, quals, n_threads=1, keep_d=False):
        """reads: list[bytes]; quals: list[array-like u8] -> OracleResult"""
        seqs, qs, offsets = pack_reads(reads, quals)
        r = lib().mo_map_batch(self.h, C.byref(params), _ptr(seqs), _ptr(qs), _ptr(offsets), len(reads), n_threads, int(keep_d))
        return OracleResult(self, r, params, seqs, qs, offsets)


# capi.cpp: MoRecord (64 bytes)
MO_RECORD_DTYPE = np.dtype({"names": ["pos", "tid", "as_bits", "xs_bits", "nm", "x0", "x1", "cigar_len", "md_len", "xa_len", "flags", "mapq", "mapped", "reverse", "has_xs", "xt", "text_off"],
                            "formats": ["<i8", "<i4", "<u4", "<u4", "<i4", "<i4", "<i4", "<u4", "<u4", "<u4", "<u2", "u1", "u1", "u1", "u1", "u1", "<u8"],
                            "offsets": [0, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 46, 47, 48, 49, 50, 56], "itemsize": 64})


def pack_reads(reads, quals):
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    qs = np.concatenate([np.asarray(q, dtype=np.uint8) for q in quals]) if reads else np.zeros(0, np.uint8)
    if seqs.size == 0:
        seqs = np.zeros(1, np.uint8)
        qs = np.zeros(1, np.uint8)
    return seqs, np.ascontiguousarray(qs), offsets


class OracleResult:
    def __init__(self, index, handle, params, seqs, qs, offsets):
        self.index, self.h, self.params = index, handle, params
        self.seqs, self.qs, self.offsets = seqs, qs, offsets
        self.n = len(offsets) - 1
        L = lib()
        nh, no = int(L.mo_result_total_hits(handle)), int(L.mo_result_total_ops(handle))
        self.hit_offsets = np.zeros(self.n + 1, dtype=np.uint64)
        self.intervals = np.zeros((max(nh, 1), 3), dtype=np.uint64)
        self.scores = np.zeros(max(nh, 1), dtype=np.float32)
        self.op_offsets = np.zeros(nh + 1, dtype=np.uint64)
        self.ops = np.zeros(max(no, 1), dtype=np.uint32)
        L.mo_result_export(handle, _ptr(self.hit_offsets), _ptr(self.intervals), _ptr(self.scores), _ptr(self.op_offsets), _ptr(self.ops))
        self.intervals = self.intervals[:nh]
        self.scores = self.scores[:nh]
        self.ops = self.ops[:no]
        self.counters = np.zeros((self.n, 6), dtype=np.uint64)
        L.mo_result_counters(handle, _ptr(self.counters))

    def __del__(self):
        try:
            if self.h:
                lib().mo_result_free(self.h)
                self.h = None
        except Exception:
            pass

    def hits(self, read):
        """list of dicts in BinaryHeap array order"""
        out = []
        for h in range(int(self.hit_offsets[read]), int(self.hit_offsets[read + 1])):
            ops = self.ops[int(self.op_offsets[h]):int(self.op_offsets[h + 1])]
            out.append({"interval": tuple(int(x) for x in self.intervals[h]), "score": self.scores[h], "ops": [unpack_op(u) for u in ops],
                        "ops_raw": ops.copy()})
        return out

    def d_array(self, read):
        n = int(self.offsets[read + 1] - self.offsets[read])
        out = np.zeros(n, dtype=np.float32)
        if lib().mo_result_d_array(self.h, read, _ptr(out)) != 0:
            raise RuntimeError("d arrays not kept")
        return out

    def bam_fields(self, read, hit, backward=False, abs_pos=0, use_orig=False):
        n_ops = 4 * 70000
        cig, md, nm = C.create_string_buffer(n_ops), C.create_string_buffer(n_ops), C.c_int32()
        lib().mo_hit_bam_fields(self.index.h, self.h, read, hit, int(backward), abs_pos, int(use_orig), cig, md, C.byref(nm))
        return cig.value.decode(), md.value.decode(), nm.value

    def records(self, flags=None):
        fl = np.ascontiguousarray(flags, dtype=np.uint16) if flags is not None else None
        p = lib().mo_records_tsv(self.index.h, self.h, C.byref(self.params), _ptr(self.seqs), _ptr(self.qs), _ptr(self.offsets),
                                 _ptr(fl) if fl is not None else None, self.n)
        s = C.string_at(p).decode()
        lib().mo_free(p)
        keys = ["flags", "tid", "pos", "mapq", "cigar", "seq", "qual", "as_bits", "nm", "md", "xa", "x0", "x1", "xs_bits", "xt"]
        return [dict(zip(keys, line.split("\t"))) for line in s.splitlines()]


def prrange(start, end, seed, max_out=1 << 20):
    out = np.zeros(max_out, dtype=np.uint64)
    n = lib().mo_prrange(start, end, seed, _ptr(out), max_out)
    return None if n < 0 else out[:n]
