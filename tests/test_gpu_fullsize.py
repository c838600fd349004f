"""BASELINE.json's full sizes on the GPU through properties that do not need an oracle run of the same size (C2: 48 Mbp, 1 M x 50 bp, and the
damage model of C3 on the same genome): a permutation of the reads permutes the results, shards concatenate to the whole, a repeat is
identical, the event counters add up, and every hit is a well-formed interval.  (bench.py checks a 200 K-read sample of the same workloads
bit for bit against the oracle.)"""
import hashlib

import numpy as np
import pytest

import mapad_amd
from mapad_amd import synth
from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve

pytestmark = pytest.mark.gpu

GENOME_BP, N_READS = 48_000_000, 1_000_000


def _mix(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)"""
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _per_read_digests(res, n):
    """one 64-bit checksum per read over its hits in order (interval, score bits, edit track in order) — independent of where the hits sit in
    the batch arrays; numpy only, so that a million reads take a second"""
    hb = res.hit_begin.astype(np.int64)
    hits = res.hits_arr
    nh = len(hits)
    if nh == 0:
        return np.zeros(n, dtype=np.uint64)
    ops = res.ops.astype(np.uint64)
    idx = np.arange(len(ops), dtype=np.uint64)
    with np.errstate(over="ignore"):
        s1 = np.concatenate([[np.uint64(0)], np.cumsum(_mix(ops + np.uint64(1)), dtype=np.uint64)])
        s2 = np.concatenate([[np.uint64(0)], np.cumsum(_mix(ops + np.uint64(1)) * idx, dtype=np.uint64)])
        o = hits["ops_offset"].astype(np.int64)
        e = o + hits["n_ops"].astype(np.int64)
        track = (s2[e] - s2[o]) - (s1[e] - s1[o]) * (o.astype(np.uint64) - np.uint64(1))  # sum over k of mix(op_k) * (k + 1)
        key = _mix(hits["lower"]) ^ _mix(hits["lower_rev"] + np.uint64(0x9E3779B97F4A7C15)) ^ _mix(hits["size"] * np.uint64(3)) \
            ^ _mix(hits["score"].view(np.uint32).astype(np.uint64) << np.uint64(17)) ^ _mix(track)
        rank = np.arange(nh, dtype=np.int64) - np.repeat(hb[:-1], np.diff(hb))  # position of a hit inside its read's heap array
        key = _mix(key * (rank.astype(np.uint64) + np.uint64(1)))
        csum = np.concatenate([[np.uint64(0)], np.cumsum(key, dtype=np.uint64)])
        return csum[hb[1:n + 1]] - csum[hb[:n]]


def _take(seqs, quals, offsets, idx):
    lens = np.diff(offsets.astype(np.int64))
    o = np.zeros(len(idx) + 1, dtype=np.uint64)
    o[1:] = np.cumsum(lens[idx])
    start = offsets.astype(np.int64)[idx]
    pos = np.repeat(start - o[:-1].astype(np.int64), lens[idx]) + np.arange(int(o[-1]), dtype=np.int64)
    return seqs[pos], quals[pos], o


@pytest.mark.parametrize("name,prm,kw", [("c2", NO_DAMAGE, dict(qual=40)), ("c3", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)))])
def test_full_size_batch_properties(name, prm, kw):
    g = synth.genome(GENOME_BP, seed=1234)
    idx = mapad_amd.Index.build([("chr1", g)], seed=1234, device=0)
    n = len(idx)
    seqs, quals, offsets = synth.reads(g, N_READS, 50, seed=4321 + (2 if name == "c2" else 3), **kw)
    ctx = mapad_amd.Context(idx, mapad_amd.make_params(resolve(prm)), 0)
    ctx.set_fetch_d_arrays(False)
    whole = ctx.map_batch(seqs, quals, offsets)
    hb = whole.hit_begin.astype(np.int64)
    hits = whole.hits_arr
    # well-formed: hit_begin is a prefix sum, intervals inside the text, scores are log-probabilities, edit tracks inside the ops array
    assert hb[0] == 0 and (np.diff(hb) >= 0).all() and hb[-1] == whole.n_hits == len(hits)
    assert (hits["size"] >= 1).all() and (hits["lower"] + hits["size"] <= n).all() and (hits["lower_rev"] + hits["size"] <= n).all()
    assert (hits["score"] <= 0).all() and np.isfinite(hits["score"]).all()
    assert (hits["ops_offset"].astype(np.int64) + hits["n_ops"] <= len(whole.ops)).all() and (hits["n_ops"] >= 50).all()
    assert 0.8 < (np.diff(hb) > 0).mean() < 0.95  # 90 % of the reads come from the genome
    c = whole.counters
    assert (c["n_pop"] >= c["e_search"]).all() and (c["n_pop"] - c["e_search"] <= 1).all()  # only the pop that ends a search is not extended
    assert (c["n_push"] <= c["n_node"] + 1).all() and (c["n_hits"] >= np.diff(hb)).all()
    want = _per_read_digests(whole, N_READS)
    # a repeat is identical down to the array layout
    again = ctx.map_batch(seqs, quals, offsets)
    assert np.array_equal(again.hit_begin, whole.hit_begin) and again.hits_arr.tobytes() == hits.tobytes() and np.array_equal(again.ops, whole.ops)
    assert again.counters.tobytes() == c.tobytes()
    # a permutation of the reads permutes the results (the order reads are scheduled in, and what shares a wavefront with what, changes nothing)
    rng = np.random.default_rng(7)
    perm = rng.permutation(N_READS)
    sub = perm[:300_000]
    s2, q2, o2 = _take(seqs, quals, offsets, sub)
    part = ctx.map_batch(s2, q2, o2)
    got = _per_read_digests(part, len(sub))
    assert np.array_equal(got, want[sub])
    assert np.array_equal(part.counters["n_pop"], c["n_pop"][sub]) and np.array_equal(part.counters["e_darray"], c["e_darray"][sub])
    # shards concatenate to the whole: a checksum of the per-read checksums
    cut = 437_501
    a = ctx.map_batch(seqs[:int(offsets[cut])], quals[:int(offsets[cut])], offsets[:cut + 1])
    b = ctx.map_batch(seqs[int(offsets[cut]):], quals[int(offsets[cut]):], offsets[cut:] - offsets[cut])
    both = np.concatenate([_per_read_digests(a, cut), _per_read_digests(b, N_READS - cut)])
    assert hashlib.sha256(both.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()
    assert a.n_hits + b.n_hits == whole.n_hits
    ctx.close()
