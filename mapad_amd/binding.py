"""ctypes mirror of include/mapad_amd.h (the C ABI of libmapad_amd.so).

Host-side plumbing only: this module never computes alignments itself and has no CPU fallback — every mapping call goes
through the HIP library and raises MapadError if the library or a gfx950 device is missing.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))


class MapadError(RuntimeError):
    CODES = {-1: "invalid argument", -2: "I/O error", -3: "index version mismatch", -4: "parse error", -5: "no gfx950 device (no CPU fallback)",
             -6: "HIP call failed", -7: "out of memory", -8: "read too long", -9: "input beyond a documented limit of this entry point"}

    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"{what}: {self.CODES.get(code, 'error')} ({code})")


class Params(C.Structure):
    """mapad_params_t"""
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
        ("chunk_size", C.c_uint64),
    ]


class Hit(C.Structure):
    """mapad_hit_t"""
    _fields_ = [("lower", C.c_uint64), ("lower_rev", C.c_uint64), ("size", C.c_uint64), ("alignment_score", C.c_float),
                ("n_ops", C.c_uint32), ("ops_offset", C.c_uint32), ("reserved", C.c_uint32)]


HIT_DTYPE = np.dtype([("lower", "<u8"), ("lower_rev", "<u8"), ("size", "<u8"), ("score", "<f4"), ("n_ops", "<u4"), ("ops_offset", "<u4"), ("reserved", "<u4")])
COUNTER_DTYPE = np.dtype([("e_search", "<u4"), ("e_darray", "<u4"), ("n_push", "<u4"), ("n_pop", "<u4"), ("n_node", "<u4"), ("n_hits", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 40


class BatchResultC(C.Structure):
    """mapad_batch_result_t"""
    _fields_ = [("n_reads", C.c_uint64), ("n_hits", C.c_uint64), ("n_ops", C.c_uint64), ("hit_begin", C.c_void_p), ("hits", C.c_void_p),
                ("ops", C.c_void_p), ("status", C.c_void_p), ("counters", C.c_void_p), ("d_arrays", C.c_void_p), ("n_second_pass", C.c_uint64), ("n_third_pass", C.c_uint64)]


class RecordC(C.Structure):
    """mapad_record_t"""
    _fields_ = [("flags", C.c_uint16), ("mapq", C.c_uint8), ("mapped", C.c_uint8), ("reverse", C.c_uint8), ("tid", C.c_int32), ("pos", C.c_int64),
                ("as_score", C.c_float), ("xs_score", C.c_float), ("nm", C.c_int32), ("x0", C.c_int32), ("x1", C.c_int32), ("has_xs", C.c_uint8),
                ("xt", C.c_char), ("cigar_off", C.c_uint32), ("cigar_len", C.c_uint32), ("md_off", C.c_uint32), ("md_len", C.c_uint32),
                ("xa_off", C.c_uint32), ("xa_len", C.c_uint32)]


class RecordsC(C.Structure):
    _fields_ = [("n", C.c_uint64), ("recs", C.POINTER(RecordC)), ("text", C.c_void_p), ("text_len", C.c_uint64)]


MODEL_KINDS = {"simple_adna": 0, "vindija_pwm": 1, "test": 2}
BOUND_KINDS = {"discrete": 0, "continuous": 1, "test": 2}
LIBRARY_PREPS = {"single_stranded": 0, "double_stranded": 1}

# every symbol include/mapad_amd.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _i32, _f, _u8 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_uint8
_PP = C.POINTER(Params)
SYMBOLS = {
    "mapad_version": (C.c_char_p, []),
    "mapad_params_from_cli": (_i32, [_PP, _i32, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i32, _i32, _i32, _i32, _u64]),
    "mapad_sdm_get": (_f, [_PP, _u64, _u64, _u8, _u8, _u8]),
    "mapad_sdm_representative_mismatch_penalty": (_f, [_PP]),
    "mapad_sdm_min_penalty": (_f, [_PP, _u64, _u64, _u8, _u8, _i32]),
    "mapad_sdm_alignment_start": (_i32, [_PP, _u64]),
    "mapad_mb_reject": (_i32, [_PP, _f, _u64]),
    "mapad_mb_reject_iterative": (_i32, [_PP, _f, _f]),
    "mapad_mb_remaining_frac_of_repr_mm": (_f, [_PP, _f, _u64]),
    "mapad_index_build": (_i32, [_vp, _vp, _vp, _u32, _u64, C.POINTER(_vp)]),
    "mapad_index_build_gpu": (_i32, [_vp, _vp, _vp, _u32, _u64, _i32, C.POINTER(_vp)]),
    "mapad_index_open": (_i32, [C.c_char_p, C.POINTER(_vp)]),
    "mapad_index_save": (_i32, [_vp, C.c_char_p]),
    "mapad_index_free": (None, [_vp]),
    "mapad_index_text_len": (_u64, [_vp]),
    "mapad_index_copy_bwt": (_i32, [_vp, _vp]),
    "mapad_index_n_contigs": (_u32, [_vp]),
    "mapad_index_contig": (_i32, [_vp, _u32, C.POINTER(C.c_char_p), C.POINTER(_u64), C.POINTER(_u64)]),
    "mapad_index_sa_sample_len": (_u64, [_vp]),
    "mapad_index_sa_extra_len": (_u64, [_vp]),
    "mapad_index_copy_sa": (_i32, [_vp, _vp, _vp, _vp]),
    "mapad_index_device_view": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_u64), _vp, _vp]),
    "mapad_index_sa_get": (_i32, [_vp, _u64, C.POINTER(_u64)]),
    "mapad_index_sa_get_batch": (_i32, [_vp, _vp, _u64, _vp]),
    "mapad_ctx_create": (_i32, [_vp, _PP, _i32, C.POINTER(_vp)]),
    "mapad_ctx_destroy": (None, [_vp]),
    "mapad_ctx_set_stream": (_i32, [_vp, _vp]),
    "mapad_ctx_set_fetch_d_arrays": (_i32, [_vp, _i32]),
    "mapad_ctx_prepare_lengths": (_i32, [_vp, _vp, _u32]),
    "mapad_map_batch": (_i32, [_vp, _vp, _vp, _vp, _u64, C.POINTER(C.POINTER(BatchResultC))]),
    "mapad_batch_result_free": (None, [C.POINTER(BatchResultC)]),
    "mapad_submit_batch": (_i32, [_vp, _vp, _vp, _vp, _u64]),
    "mapad_host_alloc": (_vp, [C.c_size_t]),
    "mapad_host_free": (None, [_vp]),
    "mapad_map_batch_device": (_i32, [_vp, _vp, _vp, _vp, _u64, _u32]),
    "mapad_fetch_result": (_i32, [_vp, C.POINTER(C.POINTER(BatchResultC))]),
    "mapad_ctx_set_pipeline_depth": (_i32, [_vp, _i32]),
    "mapad_ctx_select_batch": (_i32, [_vp, _i32]),
    "mapad_ctx_reserve": (_i32, [_vp, _u64, _u64, _u32, _i32]),
    "mapad_kernel_history": (_i32, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "mapad_compact_result_device": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u64)]),
    "mapad_device_result_ptrs": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "mapad_last_batch_counters": (_i32, [_vp, _vp]),
    "mapad_last_kernel_ms": (_i32, [_vp, _vp]),
    "mapad_last_launch_info": (_i32, [_vp, _vp]),
    "mapad_host_cpus": (C.c_uint32, []),
    "mapad_ctx_set_reserved_cus": (_i32, [_vp, C.c_int32]),
    "mapad_ctx_set_tail_pops": (_i32, [_vp, C.c_uint32]),
    "mapad_tail_set_local_world": (C.c_uint32, [C.c_uint32]),
    "mapad_last_tail_info": (_i32, [_vp, _vp]),
    "mapad_hits_to_records": (_i32, [_vp, _PP, C.POINTER(BatchResultC), _vp, _vp, _vp, _vp, _u64, C.POINTER(C.POINTER(RecordsC))]),
    "mapad_records_free": (None, [C.POINTER(RecordsC)]),
    "mapad_records_seed_at": (_u64, [_u64, _u64]),
    "mapad_sa_locate": (_i32, [_vp, _vp, _u64, _vp]),
    "mapad_last_locate_info": (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(_u64), C.POINTER(_u64)]),
    "mapad_hits_to_records_gpu": (_i32, [_vp, C.POINTER(BatchResultC), _vp, _vp, _vp, _vp, _u64, C.POINTER(C.POINTER(RecordsC))]),
    "mapad_records_device": (_i32, [_vp, _u64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u64)]),
    "mapad_hits_to_coords_gpu": (_i32, [_vp, C.POINTER(BatchResultC), _u64, C.POINTER(_vp)]),
    "mapad_coords_to_records": (_i32, [_vp, _PP, C.POINTER(BatchResultC), _vp, _vp, C.POINTER(C.POINTER(RecordsC))]),
    "mapad_coords_free": (None, [_vp]),
}

_lib = None


def lib():
    """Loads libmapad_amd.so (building it in-tree if the sources are newer).  Raises if it cannot be loaded."""
    global _lib
    if _lib is None:
        path = os.environ.get("MAPAD_AMD_LIB")  # an alternative build of the library (profiling / A-B variants)
        if not path:
            hv = _build.selected_variant()  # MAPAD_HEAP_VARIANT=1..3: the library built with another reading of the heap's tie rules (csrc/heap_core.hpp)
            path = _build.lib_path(hv)
            if _build.needs_build(hv):
                path = _build.build(heap_variant=hv)
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)  # AttributeError = the library does not export what the header declares
            except AttributeError:
                if os.environ.get("MAPAD_AMD_LIB") and os.environ.get("MAPAD_AMD_LIB_OLD_ABI"):  # an A/B run against an older build of the library (profiles/dev/ab3.sh)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise MapadError(rc, what)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_params(d):
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


def params_from_cli(library="single_stranded", five_prime_overhang=0.0, three_prime_overhang=0.0, ds_deamination_rate=0.0, ss_deamination_rate=0.0,
                    divergence=0.02, poisson_prob=0.03, as_cutoff=0.0, as_cutoff_exponent=1.0, indel_rate=0.001, gap_extension_penalty=1.0,
                    gap_dist_ends=5, max_num_gaps_open=2, ignore_base_quality=False, no_search_limit_recovery=False, chunk_size=250000):
    """build_alignment_parameters (src/main.rs:418-499); poisson_prob=None selects the Continuous bound."""
    p = Params()
    _check(lib().mapad_params_from_cli(C.byref(p), LIBRARY_PREPS[library], five_prime_overhang, three_prime_overhang, ds_deamination_rate,
                                       ss_deamination_rate, divergence, -1.0 if poisson_prob is None else poisson_prob, as_cutoff, as_cutoff_exponent,
                                       indel_rate, gap_extension_penalty, gap_dist_ends, max_num_gaps_open, int(ignore_base_quality),
                                       int(no_search_limit_recovery), chunk_size), "mapad_params_from_cli")
    return p


class BatchResult:
    """Owns a mapad_batch_result_t and exposes it as numpy views (copied, so the C object can be freed)."""

    def __init__(self, cptr, free_fn):
        r = cptr.contents
        self.n_reads, self.n_hits, self.n_ops = int(r.n_reads), int(r.n_hits), int(r.n_ops)
        self.n_second_pass, self.n_third_pass = int(r.n_second_pass), int(r.n_third_pass)

        def arr(ptr, dtype, n):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr), dtype=dtype).copy()

        self.hit_begin = arr(r.hit_begin, np.uint64, self.n_reads + 1)
        self.hits_arr = arr(r.hits, HIT_DTYPE, self.n_hits)
        self.ops = arr(r.ops, np.uint32, self.n_ops)
        self.status = arr(r.status, np.uint32, self.n_reads)
        self.counters = arr(r.counters, COUNTER_DTYPE, self.n_reads)
        total = 0
        self._cptr, self._free = cptr, free_fn
        self._d_ptr = r.d_arrays
        self._arr = arr

    def d_arrays(self, offsets):
        total = int(offsets[-1])
        return self._arr(self._d_ptr, np.float32, total)

    def hits(self, read):
        """hits of one read in BinaryHeap array order: list of {"interval", "score", "ops_raw"}"""
        out = []
        for h in self.hits_arr[int(self.hit_begin[read]):int(self.hit_begin[read + 1])]:
            ops = self.ops[int(h["ops_offset"]):int(h["ops_offset"]) + int(h["n_ops"])]
            out.append({"interval": (int(h["lower"]), int(h["lower_rev"]), int(h["size"])), "score": np.float32(h["score"]), "ops_raw": ops.copy()})
        return out

    def close(self):
        if self._cptr is not None:
            self._free(self._cptr)
            self._cptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def build(cls, contigs, seed=1234, device=None):
        """contigs: list of (name: str, seq: bytes / uint8 array).  device=None: suffix sorting on the host (SA-IS);
        device=k: on GPU k (mapad_index_build_gpu) — same index, byte for byte."""
        n = len(contigs)
        names = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        bufs = [np.ascontiguousarray(np.frombuffer(c[1], dtype=np.uint8) if isinstance(c[1], (bytes, bytearray)) else c[1], dtype=np.uint8) for c in contigs]
        seqs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = (C.c_uint64 * n)(*[b.size for b in bufs])
        out = C.c_void_p()
        if device is None:
            _check(lib().mapad_index_build(names, seqs, lens, n, seed, C.byref(out)), "mapad_index_build")
        else:
            _check(lib().mapad_index_build_gpu(names, seqs, lens, n, seed, int(device), C.byref(out)), "mapad_index_build_gpu")
        return cls(out)

    @classmethod
    def open(cls, prefix):
        out = C.c_void_p()
        _check(lib().mapad_index_open(prefix.encode(), C.byref(out)), "mapad_index_open")
        return cls(out)

    def save(self, prefix):
        _check(lib().mapad_index_save(self.h, prefix.encode()), "mapad_index_save")

    def __del__(self):
        try:
            if self.h:
                lib().mapad_index_free(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return int(lib().mapad_index_text_len(self.h))

    def bwt(self):
        out = np.empty(len(self), dtype=np.uint8)
        _check(lib().mapad_index_copy_bwt(self.h, _ptr(out)), "mapad_index_copy_bwt")
        return out

    def contigs(self):
        out = []
        for i in range(lib().mapad_index_n_contigs(self.h)):
            name, s, e = C.c_char_p(), C.c_uint64(), C.c_uint64()
            _check(lib().mapad_index_contig(self.h, i, C.byref(name), C.byref(s), C.byref(e)), "mapad_index_contig")
            out.append((name.value.decode(), s.value, e.value))
        return out

    def sampled_sa(self):
        ns, ne = int(lib().mapad_index_sa_sample_len(self.h)), int(lib().mapad_index_sa_extra_len(self.h))
        sample, er, ev = np.zeros(ns, np.uint64), np.zeros(max(ne, 1), np.uint64), np.zeros(max(ne, 1), np.uint64)
        _check(lib().mapad_index_copy_sa(self.h, _ptr(sample), _ptr(er), _ptr(ev)), "mapad_index_copy_sa")
        return sample, er[:ne], ev[:ne]

    def sa_get(self, row):
        out = C.c_uint64()
        _check(lib().mapad_index_sa_get(self.h, row, C.byref(out)), "mapad_index_sa_get")
        return out.value

    def sa_get_batch(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(rows.size, np.uint64)
        _check(lib().mapad_index_sa_get_batch(self.h, _ptr(rows) if rows.size else None, rows.size, _ptr(out) if rows.size else None), "mapad_index_sa_get_batch")
        return out

    def device_view(self):
        blocks, nb = C.c_void_p(), C.c_uint64()
        less, sent = np.zeros(8, np.uint64), np.zeros(2, np.uint64)
        _check(lib().mapad_index_device_view(self.h, C.byref(blocks), C.byref(nb), _ptr(less), _ptr(sent)), "mapad_index_device_view")
        return blocks.value, nb.value, less, sent

    def blocks(self):
        """the rank blocks (8 x u64 per 96 rows) as a numpy view of the index's host copy"""
        ptr, nb, _, _ = self.device_view()
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(int(nb) * 8,))


class Context:
    """One GPU with the index resident in HBM (mapad_ctx_t)."""

    def __init__(self, index, params, device_id=0):
        self.index, self.params = index, params
        out = C.c_void_p()
        _check(lib().mapad_ctx_create(index.h, C.byref(params), device_id, C.byref(out)), "mapad_ctx_create")
        self.h = out

    def close(self):
        if self.h:
            lib().mapad_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().mapad_ctx_set_stream(self.h, stream_ptr), "mapad_ctx_set_stream")

    def set_fetch_d_arrays(self, on):
        _check(lib().mapad_ctx_set_fetch_d_arrays(self.h, int(on)), "mapad_ctx_set_fetch_d_arrays")

    def prepare_lengths(self, lens):
        a = np.ascontiguousarray(lens, dtype=np.uint32)
        _check(lib().mapad_ctx_prepare_lengths(self.h, _ptr(a), a.size), "mapad_ctx_prepare_lengths")

    def map_batch(self, seqs, quals, offsets):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = C.POINTER(BatchResultC)()
        _check(lib().mapad_map_batch(self.h, _ptr(seqs), _ptr(quals), _ptr(offsets), offsets.size - 1, C.byref(out)), "mapad_map_batch")
        return BatchResult(out, lib().mapad_batch_result_free)

    def submit_batch(self, seqs, quals, offsets):
        """asynchronous map_batch: returns once the reads are staged and the kernels are enqueued (collect with select_batch + fetch)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(lib().mapad_submit_batch(self.h, _ptr(seqs), _ptr(quals), _ptr(offsets), offsets.size - 1), "mapad_submit_batch")

    def map_batch_device(self, d_seqs, d_quals, d_offsets, n_reads, max_read_len):
        _check(lib().mapad_map_batch_device(self.h, d_seqs, d_quals, d_offsets, n_reads, max_read_len), "mapad_map_batch_device")

    def fetch(self):
        out = C.POINTER(BatchResultC)()
        _check(lib().mapad_fetch_result(self.h, C.byref(out)), "mapad_fetch_result")
        return BatchResult(out, lib().mapad_batch_result_free)

    def set_pipeline_depth(self, depth):
        _check(lib().mapad_ctx_set_pipeline_depth(self.h, int(depth)), "mapad_ctx_set_pipeline_depth")

    def reserve(self, n_reads, total_bases, max_read_len, host_inputs=False):
        _check(lib().mapad_ctx_reserve(self.h, int(n_reads), int(total_bases), int(max_read_len), int(host_inputs)), "mapad_ctx_reserve")

    def select_batch(self, age):
        _check(lib().mapad_ctx_select_batch(self.h, int(age)), "mapad_ctx_select_batch")

    def kernel_history(self, cap=4096):
        """(n_launches, 4) float32: ms from the first launch's start to each launch's four event marks; clears the history"""
        out = np.zeros((cap, 4), np.float32)
        n = C.c_uint32()
        _check(lib().mapad_kernel_history(self.h, _ptr(out), cap, C.byref(n)), "mapad_kernel_history")
        return out[:min(cap, int(n.value))].copy()

    def records_device(self, seed=0):
        """record fields of the selected batch on the device: (d_records [n x 88 B], d_text, d_pairs, text_bytes, n_pairs) — what the multi-GPU gather sends"""
        p = [C.c_void_p() for _ in range(3)]
        nt, npairs = _u64(), _u64()
        _check(lib().mapad_records_device(self.h, int(seed), C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(nt), C.byref(npairs)), "mapad_records_device")
        return p[0].value, p[1].value, p[2].value, int(nt.value), int(npairs.value)

    def compact_device(self):
        """device-side order-preserving collect of the last batch: (d_hit_begin, d_hits, d_ops, n_hits, n_ops)"""
        p = [C.c_void_p() for _ in range(3)]
        nh, no = C.c_uint64(), C.c_uint64()
        _check(lib().mapad_compact_result_device(self.h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(nh), C.byref(no)), "mapad_compact_result_device")
        return p[0].value, p[1].value, p[2].value, int(nh.value), int(no.value)

    def device_result_ptrs(self):
        p = [C.c_void_p() for _ in range(5)]
        _check(lib().mapad_device_result_ptrs(self.h, *[C.byref(x) for x in p]), "mapad_device_result_ptrs")
        return [x.value for x in p]

    def last_counters(self):
        out = np.zeros(6, np.uint64)
        _check(lib().mapad_last_batch_counters(self.h, _ptr(out)), "mapad_last_batch_counters")
        return out

    def kernel_ms(self):
        """HIP-event durations (ms) of the last batch: (D arrays + ordering, search over every read + retries, full-limit search)"""
        out = np.zeros(3, np.float32)
        _check(lib().mapad_last_kernel_ms(self.h, _ptr(out)), "mapad_last_kernel_ms")
        return out

    def set_reserved_cus(self, n):
        """Leave the last n CUs free of this context's launches (room for RCCL's transfer kernels beside a search; before the first batch, depth >= 2)."""
        _check(lib().mapad_ctx_set_reserved_cus(self.h, int(n)), "mapad_ctx_set_reserved_cus")

    def set_tail_pops(self, pops):
        """Pop budget of a read on the GPU before the library's host threads take it over (0 = never; csrc/host_tail.hpp)."""
        _check(lib().mapad_ctx_set_tail_pops(self.h, int(pops)), "mapad_ctx_set_tail_pops")

    def tail_info(self):
        """{reads, gpu_pops, host_pops, host_us (wall), threads, budget, ..., host_thread_us (summed over the threads)} of the selected batch's host tail (after its collect / fetch)."""
        out = np.zeros(16, np.uint64)
        _check(lib().mapad_last_tail_info(self.h, _ptr(out)), "mapad_last_tail_info")
        d = dict(zip(("reads", "gpu_pops", "host_pops", "host_us", "threads", "budget", "host_e_search", "host_n_push", "host_n_node", "host_thread_us",
                      "seen_live", "reads_dry_class", "reads_full_limit", "min_class", "continued", "handed_over_with_state"), (int(x) for x in out)))
        d["reads_idle_tier"] = d["min_class"] >> 32  # word 13: dry-class threshold | reads handed over below the budget because a worker was idle << 32
        d["min_class"] &= 0xFFFFFFFF
        return d

    def launch_info(self):
        out = np.zeros(8, np.uint32)
        _check(lib().mapad_last_launch_info(self.h, _ptr(out)), "mapad_last_launch_info")
        return out

    def sa_locate(self, rows):
        """Suffix-array values of BWT rows, located by the device kernel (UINT64_MAX for rows past the text)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(rows.size, np.uint64)
        _check(lib().mapad_sa_locate(self.h, _ptr(rows) if rows.size else None, rows.size, _ptr(out) if rows.size else None), "mapad_sa_locate")
        return out

    def locate_info(self):
        ms, rows, steps = C.c_float(), C.c_uint64(), C.c_uint64()
        _check(lib().mapad_last_locate_info(self.h, C.byref(ms), C.byref(rows), C.byref(steps)), "mapad_last_locate_info")
        return float(ms.value), int(rows.value), int(steps.value)

    def hits_to_records(self, result_cptr_owner, seqs, quals, offsets, in_flags=None, seed=0, as_arrays=False):
        """mapad_hits_to_records_gpu -> list of dicts, same as hits_to_records() with the SA lookups done on the device.  as_arrays: (records as a numpy structured
        array with RecordC's fields, text bytes) instead — for millions of reads."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        fl = None if in_flags is None else np.ascontiguousarray(in_flags, dtype=np.uint16)
        out = C.POINTER(RecordsC)()
        _check(lib().mapad_hits_to_records_gpu(self.h, result_cptr_owner._cptr, _ptr(seqs), _ptr(quals), _ptr(offsets),
                                               _ptr(fl) if fl is not None else None, seed, C.byref(out)), "mapad_hits_to_records_gpu")
        return _records_arrays(out) if as_arrays else _decode_records(out)


def _records_arrays(out):
    """mapad_records_t -> (records as a numpy structured array with RecordC's fields, text bytes as uint8), copied; frees the C object"""
    r = out.contents
    n = int(r.n)
    recs = np.frombuffer(C.string_at(C.addressof(r.recs.contents), n * C.sizeof(RecordC)), np.dtype(RecordC)).copy() if n else np.zeros(0, np.dtype(RecordC))
    text = np.frombuffer(C.string_at(r.text, r.text_len), np.uint8).copy() if r.text_len else np.zeros(0, np.uint8)
    lib().mapad_records_free(out)
    return recs, text


def hits_to_records(index, params, result_cptr_owner, seqs, quals, offsets, in_flags=None, seed=0, as_arrays=False):
    """mapad_hits_to_records -> list of dicts (decoded record fields); as_arrays: (records structured array, text bytes)."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    fl = None if in_flags is None else np.ascontiguousarray(in_flags, dtype=np.uint16)
    out = C.POINTER(RecordsC)()
    _check(lib().mapad_hits_to_records(index.h, C.byref(params), result_cptr_owner._cptr, _ptr(seqs), _ptr(quals), _ptr(offsets),
                                       _ptr(fl) if fl is not None else None, seed, C.byref(out)), "mapad_hits_to_records")
    return _records_arrays(out) if as_arrays else _decode_records(out)


def _decode_records(out):
    r = out.contents
    text = C.string_at(r.text, r.text_len)
    recs = []
    for i in range(r.n):
        c = r.recs[i]
        recs.append({"flags": c.flags, "mapq": c.mapq, "mapped": bool(c.mapped), "reverse": bool(c.reverse), "tid": c.tid, "pos": c.pos,
                     "as": np.float32(c.as_score), "xs": np.float32(c.xs_score) if c.has_xs else None, "nm": c.nm, "x0": c.x0, "x1": c.x1,
                     "xt": c.xt.decode() if c.mapped else None,
                     "cigar": text[c.cigar_off:c.cigar_off + c.cigar_len].decode(), "md": text[c.md_off:c.md_off + c.md_len].decode(),
                     "xa": text[c.xa_off:c.xa_off + c.xa_len].decode()})
    lib().mapad_records_free(out)
    return recs
