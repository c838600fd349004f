"""GPU tests of the on-device SA locate: the kernel's LF walk vs the host's SampledSuffixArray::get restatement,
and the record fields built from device-located positions vs the host path."""
import numpy as np
import pytest

import mapad_amd
from mapad_amd import synth

from kat_util import load, resolve_params
from parity_util import DAMAGE, NO_DAMAGE
from test_oracle_kats import integration_reads

pytestmark = pytest.mark.gpu


def _genome_with_x_runs(n, seed):
    g = np.array(synth.genome(n, seed=seed), dtype=np.uint8).copy()
    g[1000:1040] = ord("N")     # a run of >= 20 ambiguous bases stays 'X' in the index (indexing.rs:98-107)
    g[5000] = ord("R")          # single ambiguity codes are replaced by a random compatible base
    g[n // 2:n // 2 + 25] = ord("N")
    return g.tobytes()


@pytest.mark.parametrize("with_x", [False, True], ids=["acgt", "with_X_runs"])
def test_sa_locate_matches_host_walk(with_x):
    g = _genome_with_x_runs(60_000, 3) if with_x else synth.genome(60_000, seed=3)
    idx = mapad_amd.Index.build([("chr1", g), ("chr2", synth.genome(3_000, seed=4))])
    n = len(idx)
    ctx = mapad_amd.Context(idx, mapad_amd.make_params(resolve_params(NO_DAMAGE)), 0)
    try:
        rows = np.arange(n, dtype=np.uint64)
        got = ctx.sa_locate(rows)
        want = np.array([idx.sa_get(int(r)) for r in rows], dtype=np.uint64)
        assert np.array_equal(got, want)
        assert np.array_equal(np.sort(got), rows)  # the suffix array is a permutation of the text positions
        ms, n_rows, steps = ctx.locate_info()
        assert n_rows == n and 16 * n < steps < 48 * n  # rows are sampled by row number (every 32nd): a walk takes 31 LF steps on average, not at most
        # ragged / out-of-range / empty requests
        odd = np.array([0, n - 1, n, n + 5, 31, 32, 33], dtype=np.uint64)
        out = ctx.sa_locate(odd)
        assert out[2] == np.uint64(0xFFFFFFFFFFFFFFFF) and out[3] == np.uint64(0xFFFFFFFFFFFFFFFF)
        assert [int(x) for x in out[[0, 1, 4, 5, 6]]] == [idx.sa_get(int(r)) for r in (0, n - 1, 31, 32, 33)]
        assert ctx.sa_locate(np.zeros(0, np.uint64)).size == 0
    finally:
        ctx.close()


def test_records_from_device_located_positions_equal_host_path():
    g = synth.genome(400_000, seed=17)
    seqs, quals, offsets = synth.reads(g, 3000, 50, seed=9, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    idx = mapad_amd.Index.build([("chr1", g)])
    params = mapad_amd.make_params(resolve_params(DAMAGE))
    ctx = mapad_amd.Context(idx, params, 0)
    try:
        res = ctx.map_batch(seqs, quals, offsets)
        host = mapad_amd.hits_to_records(idx, params, res, seqs, quals, offsets, seed=77)
        dev = ctx.hits_to_records(res, seqs, quals, offsets, seed=77)
        assert dev == host
        assert sum(r["mapped"] for r in dev) > 2000
    finally:
        ctx.close()


def test_caller_built_result_goes_through_the_device_post_search():
    """mapad_hits_to_records_gpu takes any mapad_batch_result_t, not only one this library handed out (shards merged on another rank, a result kept
    on disk): such a struct has no private half, its hits go to the device first, and the records are the same."""
    import ctypes as C
    from mapad_amd import binding
    g = synth.genome(200_000, seed=31)
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=12, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    idx = mapad_amd.Index.build([("chr1", g)])
    params = mapad_amd.make_params(resolve_params(DAMAGE))
    ctx = mapad_amd.Context(idx, params, 0)
    try:
        res = ctx.map_batch(seqs, quals, offsets)
        want = ctx.hits_to_records(res, seqs, quals, offsets, seed=5)
        hb, hits, ops = res.hit_begin.copy(), res.hits_arr.copy(), res.ops.copy()
        status, counters = res.status.copy(), res.counters.copy()
        own = binding.BatchResultC(n_reads=res.n_reads, n_hits=res.n_hits, n_ops=res.n_ops, hit_begin=hb.ctypes.data, hits=hits.ctypes.data if hits.size else None,
                                   ops=ops.ctypes.data if ops.size else None, status=status.ctypes.data, counters=counters.ctypes.data, d_arrays=None,
                                   n_second_pass=0, n_third_pass=0)

        class Mine:
            _cptr = C.pointer(own)
        assert ctx.hits_to_records(Mine, seqs, quals, offsets, seed=5) == want
        binding.lib().mapad_batch_result_free(Mine._cptr)  # not the library's to free: left alone
        assert int(own.n_reads) == res.n_reads and ctx.hits_to_records(Mine, seqs, quals, offsets, seed=5) == want
        assert sum(r["mapped"] for r in want) > 1000
    finally:
        ctx.close()


def test_result_of_a_destroyed_context_is_not_taken_for_the_new_contexts_batch():
    """The device-resident shortcut of the post-search recognises "this batch slot still holds that result" by a process-wide launch number.  A result that
    outlives its context must go the long way (its hits are uploaded) when it is handed to a NEW context — which may sit at the same address, in the same
    slot, after the same number of launches, holding a different batch."""
    g = synth.genome(200_000, seed=41)
    idx = mapad_amd.Index.build([("chr1", g)])
    params = mapad_amd.make_params(resolve_params(DAMAGE))
    a_reads = synth.reads(g, 1200, 50, seed=1, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    b_reads = synth.reads(g, 1200, 50, seed=2, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    ctx_a = mapad_amd.Context(idx, params, 0)
    res_a = ctx_a.map_batch(*a_reads)
    want = ctx_a.hits_to_records(res_a, *a_reads, seed=3)
    ctx_a.close()
    for _ in range(3):  # the allocator usually hands the same block out again at once; a few tries make that likely, none of them may go wrong
        ctx_b = mapad_amd.Context(idx, params, 0)
        try:
            res_b = ctx_b.map_batch(*b_reads)  # slot 0, first launch of this context: the state the old scheme could not tell from ctx_a's
            assert ctx_b.hits_to_records(res_a, *a_reads, seed=3) == want
            assert ctx_b.hits_to_records(res_b, *b_reads, seed=3) == mapad_amd.hits_to_records(idx, params, res_b, *b_reads, seed=3)
        finally:
            ctx_b.close()
    assert sum(r["mapped"] for r in want) > 800


@pytest.mark.parametrize("text", ["device", "host"])
def test_device_coordinates_on_contigs_x_runs_and_multi_row_hits(text, monkeypatch):
    """The records kernel (postproc_core.hpp: strand, contig, X0 / X1, XA candidates, PrRange order) against the host restatement on a
    multi-contig text with X runs, repeats (hit intervals of many rows: the permutation matters) and reads that straddle contig ends."""
    monkeypatch.setenv("MAPAD_RECORDS_TEXT", text)  # device: CIGAR / MD / XA text by text_kernel (text_core.hpp), MAPQ from its pairs; host: strings on host threads
    rng = np.random.default_rng(5)
    unit = synth.genome(3_000, seed=21)
    parts = [np.array(synth.genome(50_000, seed=20), dtype=np.uint8).copy(), np.concatenate([unit, unit, unit, unit, unit]), np.array(synth.genome(700, seed=22)),
             np.array(synth.genome(40_000, seed=23), dtype=np.uint8).copy()]
    parts[0][2000:2030] = ord("N"); parts[3][100] = ord("N")
    g = np.concatenate(parts)
    contigs = [("c0", parts[0].tobytes()), ("rep", parts[1].tobytes()), ("tiny", parts[2].tobytes()), ("c3", parts[3].tobytes())]
    idx = mapad_amd.Index.build(contigs, seed=3, device=0)
    seqs, quals, offsets = synth.reads(g, 4000, 50, seed=10, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(30, 70), indel_frac=0.05)
    params = mapad_amd.make_params(resolve_params(DAMAGE))
    ctx = mapad_amd.Context(idx, params, 0)
    try:
        res = ctx.map_batch(seqs, quals, offsets)
        assert (res.hits_arr["size"] >= 5).sum() > 100  # reads from the repeat
        for seed in (1, 99):
            host = mapad_amd.hits_to_records(idx, params, res, seqs, quals, offsets, seed=seed)
            dev = ctx.hits_to_records(res, seqs, quals, offsets, seed=seed)
            assert dev == host
        assert sum(1 for r in host if r["mapped"] and r["xa"]) > 100 and len({r["tid"] for r in host if r["mapped"]}) == 4
    finally:
        ctx.close()


@pytest.mark.parametrize("text", ["device", "host"])
def test_integration_records_through_device_locate(monkeypatch, text):
    """tests/integration_tests.rs expectation, with the suffix-array lookups (and, text = device, the CIGAR / MD / XA strings) done by kernels."""
    monkeypatch.setenv("MAPAD_RECORDS_TEXT", text)
    from test_host_logic import check_integration_records
    k = load("integration")
    monkeypatch.delenv("MAPAD_INDEX_FIXED_REPLACEMENT", raising=False)  # StdRng(1234) itself must draw the base the reference's expectation implies
    idx = mapad_amd.Index.build([(c["name"], c["seq"].encode()) for c in k["contigs"]], seed=1234)
    params = mapad_amd.make_params(resolve_params(k["params"]))
    reads, quals = integration_reads(k)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    qs = np.concatenate(quals)
    ctx = mapad_amd.Context(idx, params, 0)
    try:
        res = ctx.map_batch(seqs, qs, offsets)
        flags = np.array([r["flags"] for r in k["reads"]], dtype=np.uint16)
        recs = ctx.hits_to_records(res, seqs, qs, offsets, in_flags=flags)
        check_integration_records(k, recs)
    finally:
        ctx.close()
