"""Shared comparison helpers: product results (GPU or host emulation of the kernels) vs the CPU oracle."""
import numpy as np

from oracle import binding as ob

import mapad_amd

from mapad_amd.presets import CONTINUOUS, DAMAGE, DOUBLE_STRANDED, IGNORE_BQ, NO_DAMAGE, VINDIJA  # noqa: F401  (benchmark parameter presets)


def split_reads(seqs, quals, offsets):
    n = len(offsets) - 1
    reads = [seqs[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n)]
    qs = [quals[int(offsets[i]):int(offsets[i + 1])] for i in range(n)]
    return reads, qs


def assert_same_as_oracle(ores, pres, offsets, check_d=True, check_counters=True):
    """ores: oracle OracleResult (keep_d=True if check_d); pres: mapad_amd BatchResult.  Bit-exact comparison of every read:
    hit count, BinaryHeap array order, intervals, f32 score bits, edit tracks, D arrays, event counters."""
    n = len(offsets) - 1
    assert pres.n_reads == n == ores.n
    assert np.array_equal(pres.hit_begin, ores.hit_offsets), "hit counts differ"
    assert np.array_equal(pres.hits_arr["lower"], ores.intervals[:, 0])
    assert np.array_equal(pres.hits_arr["lower_rev"], ores.intervals[:, 1])
    assert np.array_equal(pres.hits_arr["size"], ores.intervals[:, 2])
    assert np.array_equal(pres.hits_arr["score"].view(np.uint32), ores.scores.view(np.uint32)), "score bits differ"
    assert np.array_equal(pres.hits_arr["n_ops"].astype(np.uint64), np.diff(ores.op_offsets))
    assert np.array_equal(pres.ops, ores.ops), "edit tracks differ"
    if check_d:
        d = pres.d_arrays(offsets)
        for i in range(n):
            assert np.array_equal(d[int(offsets[i]):int(offsets[i + 1])].view(np.uint32), ores.d_array(i).view(np.uint32)), f"D array of read {i}"
    if check_counters:
        c = pres.counters
        got = np.stack([c["e_search"], c["e_darray"], c["n_push"], c["n_pop"], c["n_node"], c["n_hits"]], axis=1).astype(np.uint64)
        assert np.array_equal(got, ores.counters), "event counters differ"


def algorithmic_bytes(counters_sum, total_bases):
    """256*(E_search + E_darray) + 40*(N_push + N_pop) + 8*N_node + 6*L"""
    e_search, e_darray, n_push, n_pop, n_node = [int(x) for x in counters_sum[:5]]
    return 256 * (e_search + e_darray) + 40 * (n_push + n_pop) + 8 * n_node + 6 * int(total_bases)


def oracle_threads():
    """Threads for the oracle: the CPUs this process may really use (os.cpu_count() capped by the cgroup's CPU-time quota)."""
    import os
    n = os.cpu_count() or 8
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, round(int(q) / int(p))))
    except Exception:
        pass
    return n


# ---- record level (POS / strand / CIGAR / MD / NM / MAPQ / X0 / X1 / XA / XS / XT / AS): device-built records vs the oracle's intervals_to_record --------------
def _gather(pool, starts, lens):
    """bytes [starts[i], starts[i] + lens[i]) of `pool` for every i, concatenated in order (vectorised)"""
    lens = lens.astype(np.int64)
    total = int(lens.sum())
    if total == 0:
        return pool[:0]
    first = np.cumsum(lens) - lens
    return pool[np.repeat(starts.astype(np.int64) - first, lens) + np.arange(total, dtype=np.int64)]


_FIXED = np.dtype([("pos", "<i8"), ("tid", "<i4"), ("as_bits", "<u4"), ("xs_bits", "<u4"), ("nm", "<i4"), ("x0", "<i4"), ("x1", "<i4"), ("flags", "<u2"), ("mapq", "u1"),
                   ("mapped", "u1"), ("reverse", "u1"), ("has_xs", "u1"), ("xt", "u1")])


def canonical_records(recs, text, oracle_side):
    """-> (fixed fields per read, (cigar, md, xa) as (lengths, bytes in read order)).  recs: the product's mapad_record_t array (binding.RecordC fields) with its text
    pool, or the oracle's MO_RECORD_DTYPE array with its pool.  Fields that mean nothing on an unmapped read (and XS without has_xs) are zeroed on both sides."""
    n = len(recs)
    f = np.zeros(n, _FIXED)
    m = recs["mapped"] != 0
    for k in ("pos", "tid", "flags", "mapq"):
        f[k] = recs[k]
    f["mapped"] = m
    has_xs = (recs["has_xs"] != 0) & m
    f["has_xs"] = has_xs
    if oracle_side:
        as_bits, xs_bits, xt = recs["as_bits"], recs["xs_bits"], recs["xt"]
        off = recs["text_off"].astype(np.int64)
        starts = (off, off + recs["cigar_len"], off + recs["cigar_len"].astype(np.int64) + recs["md_len"])
    else:
        as_bits, xs_bits, xt = recs["as_score"].view(np.uint32), recs["xs_score"].view(np.uint32), recs["xt"].view(np.uint8)
        starts = (recs["cigar_off"], recs["md_off"], recs["xa_off"])
    f["as_bits"] = np.where(m, as_bits, 0)
    f["xs_bits"] = np.where(has_xs, xs_bits, 0)
    f["xt"] = np.where(m, xt, 0)
    for k in ("nm", "x0", "x1", "reverse"):
        f[k] = np.where(m, recs[k], 0)
    texts = []
    for st, k in zip(starts, ("cigar_len", "md_len", "xa_len")):
        lens = np.where(m, recs[k], 0).astype(np.int64)
        texts.append((lens, _gather(np.asarray(text, np.uint8), np.asarray(st), lens)))
    return f, texts


def compare_records(prod, ora):
    """prod / ora: canonical_records(...) of the same reads.  Returns (number of reads that differ, their first indices, per-field counts)."""
    (pf, pt), (of, ot) = prod, ora
    assert len(pf) == len(of)
    bad = np.zeros(len(pf), bool)
    per_field = {}
    for k in _FIXED.names:
        d = pf[k] != of[k]
        per_field[k] = int(d.sum())
        bad |= d
    for name, (pl, pb), (ol, ob_) in zip(("cigar", "md", "xa"), pt, ot):
        dl = pl != ol
        per_field[name + "_len"] = int(dl.sum())
        bad |= dl
        if not dl.any() and pb.size:  # same lengths everywhere: compare the bytes and attribute mismatches to reads
            neq = (pb != ob_).astype(np.int64)
            if neq.any():
                nz = pl > 0
                first = (np.cumsum(pl) - pl)[nz]
                per_read = np.add.reduceat(neq, first)
                d = np.zeros(len(pf), bool)
                d[np.flatnonzero(nz)] = per_read > 0
                per_field[name] = int(d.sum())
                bad |= d
            else:
                per_field[name] = 0
    return int(bad.sum()), np.flatnonzero(bad)[:10].tolist(), per_field


def records_digest(canon):
    """sha256 over the canonical form: equal digests = every compared field of every read equal"""
    import hashlib
    f, texts = canon
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(f).tobytes())
    for lens, b in texts:
        h.update(np.ascontiguousarray(lens.astype(np.uint32)).tobytes())
        h.update(np.ascontiguousarray(b).tobytes())
    return h.hexdigest()


def oracle_records_from_product_hits(oidx, oparams, res, seqs, quals, offsets, lo=0, hi=None, n_threads=1):
    """The oracle's intervals_to_record over the PRODUCT's hits of reads [lo, hi) of a batch result (proven identical to the oracle's own hits elsewhere), with the
    stand-ins for rand::rng() of reads lo .. hi - 1 (seed 0: oracle/capi.cpp: seed_for == csrc/postproc_core.hpp: seed_for_hd(0, ...))."""
    hi = len(offsets) - 1 if hi is None else hi
    hb = res.hit_begin.astype(np.int64)
    h0, h1 = int(hb[lo]), int(hb[hi])
    hits = res.hits_arr[h0:h1]
    n_ops = hits["n_ops"].astype(np.int64)
    o0 = int(hits["ops_offset"][0]) if h1 > h0 else 0
    op_begin = np.zeros(h1 - h0 + 1, np.uint64)
    op_begin[1:] = np.cumsum(n_ops)
    assert h1 == h0 or np.array_equal(hits["ops_offset"].astype(np.int64) - o0, op_begin[:-1].astype(np.int64)), "edit tracks of a result are laid out in read order"
    intervals = np.stack([hits["lower"], hits["lower_rev"], hits["size"]], axis=1) if h1 > h0 else np.zeros((0, 3), np.uint64)
    s0, s1 = int(offsets[lo]), int(offsets[hi])
    return oidx.records_from_hits(oparams, (hb[lo:hi + 1] - h0).astype(np.uint64), intervals, hits["score"], op_begin, res.ops[o0:o0 + int(n_ops.sum())],
                                  seqs[s0:s1], quals[s0:s1], (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64), first_read_index=lo, n_threads=n_threads)


def check_ungapped_records_against_the_text(genome, recs, text, seqs, offsets, contig_starts=None):
    """Ground truth that needs no index at all: for every mapped read whose CIGAR is `<L>M`, the number of mismatches between the read (reverse-complemented if it
    mapped to the reverse strand) and the reference text at (tid, POS) must be the record's NM.  Returns (reads checked, reads that fail)."""
    comp = np.zeros(256, np.uint8)
    comp[:] = ord("N")
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lens = np.diff(offsets.astype(np.int64))
    m = recs["mapped"] != 0
    checked = failed = 0
    text = np.asarray(text, np.uint8)
    for L in np.unique(lens[m]):
        want = np.frombuffer(f"{int(L)}M".encode(), np.uint8)
        idx = np.flatnonzero(m & (lens == L) & (recs["cigar_len"] == want.size))
        if not idx.size:
            continue
        cig = text[recs["cigar_off"][idx].astype(np.int64)[:, None] + np.arange(want.size)]
        idx = idx[(cig == want).all(axis=1)]
        for c0 in range(0, idx.size, 200_000):
            sel = idx[c0:c0 + 200_000]
            pos = recs["pos"][sel].astype(np.int64)
            if contig_starts is not None:
                pos = pos + np.asarray(contig_starts, np.int64)[recs["tid"][sel]]
            ref = genome[pos[:, None] + np.arange(int(L))]
            rd = seqs[offsets[sel].astype(np.int64)[:, None] + np.arange(int(L))]
            rev = recs["reverse"][sel] != 0
            rd = np.where(rev[:, None], comp[rd[:, ::-1]], rd)
            mism = (ref != rd).sum(axis=1)
            failed += int((mism != recs["nm"][sel]).sum())
            checked += sel.size
    return checked, failed
