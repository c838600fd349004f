"""GPU suffix sorting (mapad_index_build_gpu, csrc/index_gpu.hip) vs the host SA-IS path: every product of `mapad index`
(src/index/indexing.rs:163-195) byte for byte, on degenerate, repetitive, ambiguous and large texts; and the n > 2^32 index
(the north-star's 3 Gbp class: 64-bit intervals, src/map/fmd_index.rs:185-189) mapped on the GPU against the oracle."""
import os
import time

import numpy as np
import pytest

import mapad_amd
from mapad_amd import synth
from oracle import binding as ob

from kat_util import resolve_params
from parity_util import (DAMAGE, NO_DAMAGE, assert_same_as_oracle, canonical_records, check_ungapped_records_against_the_text, compare_records,
                         oracle_records_from_product_hits, oracle_threads, split_reads)

pytestmark = pytest.mark.gpu


def assert_same_index(a, b):
    assert len(a) == len(b)
    assert np.array_equal(a.bwt(), b.bwt()), "BWT"
    for x, y, what in zip(a.sampled_sa(), b.sampled_sa(), ("SA sample", "extra rows", "extra values")):
        assert np.array_equal(x, y), what
    (_, nba, la, sa_), (_, nbb, lb, sb) = a.device_view(), b.device_view()
    assert nba == nbb and np.array_equal(la, lb) and np.array_equal(sa_, sb), "Less / sentinel rows"
    assert np.array_equal(a.blocks(), b.blocks()), "rank blocks"
    assert a.contigs() == b.contigs()


def _texts():
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for n in (3, 4, 5, 7, 8, 9, 31, 32, 33, 255, 256, 257, 1000, 4097):
        out.append((f"uniform{n}", [("c", acgt[rng.integers(0, 4, n)].tobytes())]))
    out.append(("polyA", [("c", b"A" * 70_000)]))                                   # one group of n/2 suffixes: 16 doubling rounds
    out.append(("tandem", [("c", b"ACGTTGCA" * 20_000)]))                           # period-8 repeat + its own reverse complement
    out.append(("two_copies", [("c", (synth.genome(150_000, seed=5).tobytes()) * 2)]))  # LCP 150 000
    out.append(("palindrome", [("c", synth.genome(60_000, seed=6).tobytes() + synth.revcomp(synth.genome(60_000, seed=6)).tobytes())]))  # text == its revcomp
    g = synth.genome(300_000, seed=7)
    g[1000:1040] = ord("N"); g[5000:5019] = ord("N"); g[77] = ord("R"); g[200_000:200_500] = ord("N"); g[299_990:] = ord("N")
    out.append(("ambiguous_multi_contig", [("chrA", g[:120_000].tobytes()), ("chrB", g[120_000:120_001].tobytes()), ("chrC", g[120_001:].tobytes().lower())]))
    return out


@pytest.mark.parametrize("name,contigs", _texts(), ids=[t[0] for t in _texts()])
def test_gpu_index_equals_host_index(name, contigs):
    host = mapad_amd.Index.build(contigs, seed=42)
    dev = mapad_amd.Index.build(contigs, seed=42, device=0)
    assert_same_index(host, dev)


def test_gpu_index_equals_oracle_index_directly():
    """The GPU-built index against the ORACLE's own construction (naive suffix sort of its own text, oracle/mapad_oracle.hpp), not via the product's
    host SA-IS: multi-contig texts with long N runs (-> X, deterministic), lower case, and a 1 Mbp genome; BWT, sampled SA and SA values of random rows."""
    rng = np.random.default_rng(5)
    g = synth.genome(1_000_000, seed=11)
    g2 = g[:200_000].copy()
    g2[3000:3100] = ord("N"); g2[150_000:150_020] = ord("N"); g2[199_900:] = ord("N")
    cases = [[("chr1", g.tobytes())],
             [("a", g2[:70_000].tobytes()), ("b", g2[70_000:70_003].tobytes()), ("c", g2[70_003:].tobytes().lower())]]
    for contigs in cases:
        dev = mapad_amd.Index.build(contigs, seed=1234, device=0)
        text = b"".join(s for _, s in contigs).upper().replace(b"N", b"X")
        o = ob.OracleIndex.from_text(text, "$ACGTX", 128)
        assert np.array_equal(dev.bwt(), o.bwt())
        sa = o.sa()
        rows = np.concatenate([rng.integers(0, len(sa), 5000), np.array([0, 1, len(sa) - 1])]).astype(np.uint64)
        assert [int(x) for x in dev.sa_get_batch(rows)] == [int(sa[int(r)]) for r in rows]


def test_c1_workload_matches_oracle():
    """BASELINE.json configs[0] as specified: the 5 386 bp genome, 1 000 synthetic 50 bp reads, -p 0.03, no-damage model — every read against the oracle."""
    g = synth.genome(5_386, seed=1234)
    seqs, quals, offsets = synth.reads(g, 1000, 50, seed=4321 + 1, qual=40)
    rp = resolve_params(NO_DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)], seed=1234, device=0)
    oidx = ob.OracleIndex.from_text(g.tobytes(), "$ACGTX", 128)
    assert np.array_equal(pidx.bwt(), oidx.bwt())
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    res = ctx.map_batch(seqs, quals, offsets)
    ctx.close()
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)


def test_gpu_index_equals_host_index_100mbp():
    g = synth.genome(100_000_000, seed=1234)
    t0 = time.time()
    dev = mapad_amd.Index.build([("chr1", g)], device=0)
    t1 = time.time()
    host = mapad_amd.Index.build([("chr1", g)])
    t2 = time.time()
    print(f"\n100 Mbp (n = {len(dev)}): GPU build {t1 - t0:.1f} s, host SA-IS build {t2 - t1:.1f} s")
    assert_same_index(host, dev)


def test_index_beyond_2_pow_32_rows_maps_like_the_oracle(monkeypatch):
    """3 Gbp text (the genome of BASELINE.json's C4/C5) -> n = 6e9 > 2^32 rows: the index is built on the GPU, 100 K reads are mapped on the GPU and compared bit for bit
    (hits, scores, edit tracks, D arrays, the six event counters) with the oracle running on its own structures over the same BWT;
    SA positions located on the device are checked against the genome itself."""
    monkeypatch.setenv("MAPAD_INDEX_VERBOSE", "1")
    G = int(os.environ.get("MAPAD_TEST_BIG_GENOME", 3_000_000_000))
    t0 = time.time()
    g = synth.genome(G, seed=1234)
    # a random text has no repeats, and without them no read maps to several rows (X0 > 1, XA entries, low MAPQ, rows drawn by PrRange): 64 stretches of 300 bp are
    # copied 20 Mbp further on, and their reverse complements 10 Mbp further on, spread over the whole text (reads from them are added to the first batch below)
    rep_src = [1_000 + k * (G // 70) for k in range(64)] if G >= 100_000_000 else []
    for a in rep_src:
        g[a + 20_000_000:a + 20_000_300] = g[a:a + 300]
        g[a + 10_000_000:a + 10_000_300] = synth.revcomp(g[a:a + 300])
    t1 = time.time()
    pidx = mapad_amd.Index.build([("chr1", g)], device=0)
    t2 = time.time()
    n = len(pidx)
    assert n == 2 * G + 2 and (G < 2_147_483_648 or n > 2 ** 32)
    print(f"\ngenome {t1 - t0:.1f} s, GPU index of n = {n} rows {t2 - t1:.1f} s")
    # the index against the text: unique 40-mers (exact matches) must be located where they were cut, on both strands
    rng = np.random.default_rng(3)
    pos = np.concatenate([rng.integers(0, G - 40, 2000), np.array([0, 1, G - 41, G - 40])])
    rp = resolve_params(NO_DAMAGE)
    params = mapad_amd.make_params(rp)
    ctx = mapad_amd.Context(pidx, params, 0)
    fwd = np.concatenate([g[p:p + 40] for p in pos])
    rc = np.concatenate([synth.revcomp(g[p:p + 40]) for p in pos])
    offs = (np.arange(len(pos) + 1) * 40).astype(np.uint64)
    for seqs, strand in ((fwd, 0), (rc, 1)):
        res = ctx.map_batch(seqs, np.full(len(seqs), 40, np.uint8), offs)
        best = res.hit_begin[:-1].astype(np.int64)
        assert (np.diff(res.hit_begin.astype(np.int64)) >= 1).all()
        h = res.hits_arr[best]
        assert (h["size"] == 1).all() and (h["score"] == 0.0).all()
        located = ctx.sa_locate(h["lower"].astype(np.uint64)).astype(np.int64)
        if strand == 0:
            assert np.array_equal(located, pos)
        else:  # the occurrence lies in the reverse-complement half of the text (mapping.rs:617-628)
            assert np.array_equal(n - located - 40 - 1, pos)
    # reads vs the oracle
    n_reads = int(os.environ.get("MAPAD_TEST_BIG_READS", 100_000))
    half = n_reads // 2
    first = synth.reads(g, half, 50, seed=4325, qual=40)
    for k, a in enumerate(rep_src):  # 30 reads from every repeated stretch
        r = synth.reads(g[a:a + 300], 30, 50, seed=5000 + k, qual=40, exo_frac=0.0)
        first = (np.concatenate([first[0], r[0]]), np.concatenate([first[1], r[1]]), np.concatenate([first[2], r[2][1:] + first[2][-1]]))
    batches = [(NO_DAMAGE, first),
               (DAMAGE, synth.reads(g, n_reads - half, 50, seed=4326, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)))]
    # C5's read mix (BASELINE.json configs[4]: 35-100 bp, 5 % of the reads with a 1-2 bp indel, damage model, Phred 20-40) on the 3 Gbp index.  The
    # reference's limits are scaled down 10x (STACK_LIMIT / EDIT_TREE_LIMIT, mapping.rs:52-54: the recovery code is the same and the heaviest
    # reads then take seconds instead of a minute each), and the size classes beyond 128 Ki nodes get no arenas (the base arenas of a 3 Gbp index hold 64 Ki), so that the reads which
    # need them are re-run by the full-limit stage (wavefront-per-read kernel).
    n_c5 = int(os.environ.get("MAPAD_TEST_C5_READS", 20_000))
    c5_limits = {"stack_limit": int(os.environ.get("MAPAD_TEST_C5_STACK_LIMIT", 200_000)), "edit_tree_limit": int(os.environ.get("MAPAD_TEST_C5_TREE_LIMIT", 1_000_000))}
    c5 = synth.reads(g, n_c5, 50, seed=4327, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    t3 = time.time()
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    # for the record-level comparison (round 6): the oracle's SampledSuffixArray::get walks ITS byte BWT / Occ from the product's samples (the samples themselves are
    # checked against the text above and in test_gpu_index_equals_host_index_100mbp), and its contig table is the text's one contig
    oidx.set_sampled_sa(pidx.sampled_sa()[0], 32, *pidx.sampled_sa()[1:])
    oidx.add_contig(0, G - 1, "chr1")
    t4 = time.time()
    print(f"oracle structures (byte BWT + Occ k=128, sampled SA) {t4 - t3:.1f} s")
    ctx.close()

    def check_records(ctx, rp, res, seqs, quals, offsets, what):
        """Device-built records (records_kernel / text_kernel / locate_kernel + host MAPQ and flags: mapad_hits_to_records_gpu) of a batch on this index against the
        oracle's intervals_to_record (mapping.rs:402-718, record.rs:269-449 restated) over the same hits with the same per-read stand-ins for rand::rng(), field by
        field, and the ungapped ones against the text itself."""
        t = time.time()
        recs, text = ctx.hits_to_records(res, seqs, quals, offsets, seed=0, as_arrays=True)
        t_dev = time.time() - t
        orecs, otext = oracle_records_from_product_hits(oidx, ob.make_params(rp), res, seqs, quals, offsets, n_threads=oracle_threads())
        n_bad, first, per_field = compare_records(canonical_records(recs, text, oracle_side=False), canonical_records(orecs, otext, oracle_side=True))
        checked, failed = check_ungapped_records_against_the_text(g, recs, text, seqs, offsets)
        mapped = recs["mapped"] != 0
        stats = dict(reads=len(recs), mapped=int(mapped.sum()), reverse=int((recs["reverse"][mapped] != 0).sum()), with_xa=int((recs["xa_len"][mapped] > 0).sum()),
                     multi_row=int((recs["x0"][mapped] > 1).sum()), gapped=int(mapped.sum()) - checked, differ=n_bad, text_check=(checked, failed))
        print(f"records [{what}]: device {t_dev:.2f} s, oracle {time.time() - t - t_dev:.1f} s: {stats}")
        assert n_bad == 0, (first, per_field)
        assert failed == 0 and checked > 0.5 * mapped.sum()
        if n > 2 ** 32 and len(recs) >= 20_000:
            assert stats["reverse"] > 0.3 * stats["mapped"]  # reverse-strand hits: suffix positions in the second half of the text, >= 2^32
        return stats

    for prm, (seqs, quals, offsets) in batches:
        rp = resolve_params(prm)
        ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
        res = ctx.map_batch(seqs, quals, offsets)
        rstats = check_records(ctx, rp, res, seqs, quals, offsets, "50 bp, " + ("no damage" if prm is NO_DAMAGE else "damage model"))
        if prm is NO_DAMAGE and rep_src:
            assert rstats["with_xa"] > 1000 and rstats["multi_row"] > 1000  # the reads from the repeated stretches: three rows each, XA entries, rows drawn by PrRange
        ctx.close()
        reads, qs = split_reads(seqs, quals, offsets)
        t5 = time.time()
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=oracle_threads(), keep_d=True)
        print(f"oracle mapped {len(reads)} reads in {time.time() - t5:.1f} s; hits above 2^32: {(res.hits_arr['lower'] >= 2 ** 32).sum()} of {res.n_hits}")
        assert_same_as_oracle(ores, res, offsets)
        if n > 2 ** 32:
            assert (res.hits_arr["lower"] >= 2 ** 32).sum() > 0.15 * res.n_hits  # the 64-bit half of the interval arithmetic is exercised
    if n_c5:
        seqs, quals, offsets = c5
        rp = dict(resolve_params(DAMAGE), **c5_limits)
        monkeypatch.setenv("MAPAD_CLASS_COUNTS", os.environ.get("MAPAD_TEST_C5_CLASS_COUNTS", "8192,4096,2048,1024,0,0,0,0,0,0"))
        monkeypatch.setenv("MAPAD_SET_ARENAS", "0")  # (round 5: an idle set of base arenas would hold these reads whole — 1 M nodes — and nothing would reach the full-limit stage)
        ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
        ctx.set_tail_pops(0)  # this case is about the GPU's own stages (growth through the size classes, the full-limit stage): nothing goes to the host tail
        t5 = time.time()
        res = ctx.map_batch(seqs, quals, offsets)
        t6 = time.time()
        ctx.close()
        reads, qs = split_reads(seqs, quals, offsets)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=oracle_threads(), keep_d=True)
        pops = res.counters["n_pop"].astype(np.int64)
        print(f"C5 mix: {n_c5} reads of 35-100 bp on the GPU in {t6 - t5:.1f} s, oracle in {time.time() - t6:.1f} s; pops mean {pops.mean():.0f} max {pops.max()}, "
              f"{res.n_second_pass} arena migrations, {res.n_third_pass} reads through the full-limit stage, {int((pops > c5_limits['edit_tree_limit']).sum())} reads past the tree limit")
        assert_same_as_oracle(ores, res, offsets)
        if "MAPAD_TEST_C5_CLASS_COUNTS" not in os.environ and n_c5 >= 20_000 and n > 2 ** 32:
            assert res.n_third_pass > 0 and res.n_second_pass > 0  # the knobs above did send reads through growth and through the full-limit stage
    # C5 at the reference's REAL limits (STACK_LIMIT 2 000 000 / EDIT_TREE_LIMIT 10 000 000, mapping.rs:52-54,1358-1380; stack_limit / edit_tree_limit left at 0 =
    # the defaults): the heaviest reads of a larger sample of the mix — those that search under eviction for millions of pops — as a batch of their own,
    # bit for bit against the oracle incl. the event counters.  Past the pop budget they are finished by the library's host threads (csrc/host_tail.hpp).
    n_pre = int(os.environ.get("MAPAD_TEST_C5_PRESELECT", 100_000))
    n_heavy = int(os.environ.get("MAPAD_TEST_C5_HEAVY", 2_000))
    if n_pre and n > 2 ** 32:
        monkeypatch.delenv("MAPAD_CLASS_COUNTS", raising=False)
        monkeypatch.delenv("MAPAD_SET_ARENAS", raising=False)
        seqs, quals, offsets = synth.reads(g, n_pre, 50, seed=4328, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
        rp = resolve_params(DAMAGE)
        ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
        t5 = time.time()
        res = ctx.map_batch(seqs, quals, offsets)
        info = ctx.tail_info()
        t6 = time.time()
        c5_stats = check_records(ctx, rp, res, seqs, quals, offsets, "C5 mix at the real limits")  # (the hits of the heaviest reads are compared with the oracle's below)
        assert c5_stats["gapped"] > 0.01 * c5_stats["mapped"] or n_pre < 20_000  # reads with indels took part
        pops = res.counters["n_pop"].astype(np.int64)
        heavy = np.sort(np.argsort(pops)[-n_heavy:])
        lens = np.diff(offsets.astype(np.int64))
        h_off = np.zeros(len(heavy) + 1, np.uint64)
        h_off[1:] = np.cumsum(lens[heavy])
        h_seqs = np.concatenate([seqs[int(offsets[i]):int(offsets[i + 1])] for i in heavy])
        h_quals = np.concatenate([quals[int(offsets[i]):int(offsets[i + 1])] for i in heavy])
        hres = ctx.map_batch(h_seqs, h_quals, h_off)
        hinfo = ctx.tail_info()
        t7 = time.time()
        ctx.close()
        reads, qs = split_reads(h_seqs, h_quals, h_off)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=oracle_threads(), keep_d=True)
        t8 = time.time()
        print(f"C5 at the real limits: {n_pre} reads in {t6 - t5:.1f} s ({info['reads']} finished on the host, {info['host_pops'] / max(pops.sum(), 1):.1%} of the pops); the {len(heavy)} heaviest "
              f"(pops {pops[heavy].min()} .. {pops[heavy].max()}) again in {t7 - t6:.1f} s ({hinfo['reads']} on the host), oracle {t8 - t7:.1f} s; "
              f"{int((pops[heavy] >= 2_000_000).sum())} reads with >= 2 M pops")
        assert_same_as_oracle(ores, hres, h_off)
        assert np.array_equal(hres.counters["n_pop"], res.counters["n_pop"][heavy]) and np.array_equal(np.diff(hres.hit_begin.astype(np.int64)), np.diff(res.hit_begin.astype(np.int64))[heavy])
        if n_pre >= 100_000:
            assert hinfo["reads"] > 0 and pops[heavy].max() > 2_000_000  # reads at the limits were in the batch, and the host tail took the heaviest
