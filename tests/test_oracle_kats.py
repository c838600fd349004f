"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(tests/golden/*.json, transcribed from /root/reference by tests/golden/make_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as ob
from kat_util import load, oracle_params, quals_for, resolve_params

VARIANT = int(os.environ.get("MAPAD_ORACLE_HEAP_VARIANT", "0"))
KATS = load("search_kats")


def _index_for(case):
    ref = KATS["ref10k"] if case["reference"] == "@ref10k" else case["reference"]
    return ob.OracleIndex.from_text(ref.encode(), "$ACGT", 3)  # src/utils.rs:12-33


def check_search_expectations(case, hits, sa, bam_fields):
    """hits: list of {"interval","score","ops"} in BinaryHeap array order; sa: full suffix array;
    bam_fields(hit_idx, backward) -> (cigar, md, nm).  Shared with the GPU parity tests."""
    e = case["expect"]
    n = len(sa)
    fwd = lambda h: [int(sa[i]) for i in range(h["interval"][0], h["interval"][0] + h["interval"][2])]
    rev = lambda h: [int(sa[i]) for i in range(h["interval"][1], h["interval"][1] + h["interval"][2])]
    if "heap_scores" in e:
        assert [np.float32(h["score"]) for h in hits] == [np.float32(x) for x in e["heap_scores"]]
    if "positions_sorted" in e:
        assert sorted(p for h in hits for p in fwd(h)) == e["positions_sorted"]
    if "nonempty" in e:
        assert len(hits) > 0
    if "contains_position" in e:
        assert e["contains_position"] in [p for h in hits for p in fwd(h)]
    if "n_hits" in e:
        assert len(hits) == e["n_hits"]
    if "score0" in e:
        assert np.float32(hits[0]["score"]) == np.float32(e["score0"])
    if "score0_approx" in e:
        assert abs(float(hits[0]["score"]) - e["score0_approx"]) < 1e-6
    # BinaryHeap::peek()/pop() == array slot 0
    if "best_positions" in e:
        assert fwd(hits[0]) == e["best_positions"]
    if "best_score" in e:
        assert np.float32(hits[0]["score"]) == np.float32(e["best_score"])
    if "best_cigar" in e:
        assert bam_fields(0, False)[0] == e["best_cigar"]
    if "best_md" in e:
        assert bam_fields(0, False)[1] == e["best_md"]
    if "best_positions_fwd_rev" in e:
        got = [[p, "F"] for p in fwd(hits[0]) if p < n // 2] + [[p, "B"] for p in rev(hits[0]) if p < n // 2]
        assert got == e["best_positions_fwd_rev"]
    if "best_md_backward" in e:
        assert bam_fields(0, True)[1] == e["best_md_backward"]
    if "best_nm_backward" in e:
        assert bam_fields(0, True)[2] == e["best_nm_backward"]


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_search_kat(case):
    idx = _index_for(case)
    p = oracle_params(case["params"], heap_variant=VARIANT)
    res = idx.map_batch(p, [case["pattern"].encode()], [quals_for(case["pattern"], case["qual"])])
    check_search_expectations(case, res.hits(0), idx.sa(), lambda h, b: res.bam_fields(0, h, backward=b))


def test_d_array_kat():
    k = load("d_array_kat")
    idx = ob.OracleIndex.from_text(k["reference"].encode())
    p = oracle_params(k["params"])
    d = idx.d_array(p, k["pattern"].encode(), k["qual"], split=k["split"])
    assert d.tolist() == k["d_composite"]
    for a, b, v in k["get"]:
        assert idx.d_array_get(p, k["pattern"].encode(), k["qual"], k["split"], a, b) == v
    # bi_d_array.rs:291-302: get() index arithmetic
    sp = k["split"]
    g = lambda a, b: idx.d_array_get(p, k["pattern"].encode(), k["qual"], sp, a, b)
    assert g(1, 4) == d[1] + d[sp + 2]
    assert g(2, 3) == d[2] + d[sp + 3]
    assert g(0, 6) == d[0] + d[sp]


SDM = load("sdm_kats")


@pytest.mark.parametrize("block", SDM["get"], ids=[b["name"] for b in SDM["get"]])
def test_sdm_get(block):
    p = oracle_params({**block["params"], "bound": "test"})
    L = ob.lib()
    for v, i, ln, f, t, q in block["asserts"]:
        got = L.mo_sdm_get(C.byref(p), i, ln, ord(f), ord(t), q)
        assert abs(got - v) < block["tolerance"], (v, got, i, ln, f, t, q)


def test_sdm_wo_deam_exact():
    k = SDM["wo_deam"]
    p = oracle_params({**k["params"], "bound": "test"})
    L = ob.lib()
    for a, b in k["equal_pairs"]:
        ga = L.mo_sdm_get(C.byref(p), a[0], a[1], ord(a[2]), ord(a[3]), a[4])
        gb = L.mo_sdm_get(C.byref(p), b[0], b[1], ord(b[2]), ord(b[3]), b[4])
        assert ga == gb


@pytest.mark.parametrize("lib_kind", ["single_stranded", "double_stranded"])
def test_sdm_display(lib_kind):
    """sequence_difference_models.rs:1305-1339: the Display strings, value by value ({:.2})."""
    k = SDM["display"][lib_kind]
    p = oracle_params({**k["params"], "bound": "test"})
    L = ob.lib()
    fmt = lambda x: "%.2f" % x
    assert fmt(L.mo_sdm_repr_mm(C.byref(p))) == k["ordinary_mm"]
    assert fmt(L.mo_sdm_get(C.byref(p), 25, 50, ord("C"), ord("T"), 37)) == k["central"]
    assert [fmt(L.mo_sdm_get(C.byref(p), i, 50, ord("C"), ord("T"), 37)) for i in range(10)] == k["five_prime_c_to_t"]
    if lib_kind == "single_stranded":
        assert [fmt(L.mo_sdm_get(C.byref(p), i, 50, ord("C"), ord("T"), 37)) for i in range(49, 39, -1)] == k["three_prime_c_to_t"]
    else:
        assert [fmt(L.mo_sdm_get(C.byref(p), i, 50, ord("G"), ord("A"), 37)) for i in range(49, 39, -1)] == k["three_prime_g_to_a"]


def test_discrete_bound_tables():
    k = load("bounds_kats")
    L = ob.lib()
    for t in k["discrete_get"]:
        for ln, v in t["values"]:
            assert L.mo_discrete_get(t["poisson"], t["err"], ln) == float(v), (t, ln)
    for t in k["display"]:
        steps, prev = [], None
        for ln in range(17, 257):
            v = L.mo_discrete_get(t["poisson"], t["err"], ln)
            if prev is None or abs(v - prev) > np.finfo(np.float32).eps:
                steps.append([ln, int(v)])
                prev = v
        assert steps == t["steps"]


def test_prrange():
    k = load("misc_kats")["prrange"]
    for s, e, seed in k["permutations"]:
        r = ob.prrange(s, e, seed)
        assert sorted(r.tolist()) == list(range(s, e))
    for s, e, seed, cnt in k["counts"]:
        assert len(ob.prrange(s, e, seed)) == cnt
    for s, e, seed in k["invalid"]:
        assert ob.prrange(s, e, seed) is None
    to = 40  # the reference sweeps 0..=100 (prrange.rs:249-259); 0..=40 keeps the CPU suite quick, same property
    for s in range(0, to + 1):
        for e in range(s + 1, to + 1):
            for seed in range(0, to + 1, 3):
                assert len(ob.prrange(s, e, seed, 128)) == e - s


def test_tree_slab_semantics():
    """backtrack_tree.rs:131-196 via a small script interface (0 clear, 1 add(parent), 2 remove(id), 3 count ancestors, 4 len)."""
    script = np.array([[0, 0], [1, 0], [1, 1], [1, 2], [1, 3], [3, 4], [2, 2], [3, 4], [4, 0]], dtype=np.int32)
    out = np.zeros(16, dtype=np.int64)
    n = ob.lib().mo_tree_script(script.ctypes.data_as(C.c_void_p), len(script), out.ctypes.data_as(C.c_void_p))
    assert out[:n].tolist() == [0, 1, 2, 3, 4, 4, 4, 2, 4]
    # LIFO key reuse of the slab free list (slab 0.4): a removed key is handed out again first
    script = np.array([[0, 0], [1, 0], [1, 0], [1, 0], [2, 2], [2, 1], [1, 0], [1, 0], [1, 0], [4, 0]], dtype=np.int32)
    n = ob.lib().mo_tree_script(script.ctypes.data_as(C.c_void_p), len(script), out.ctypes.data_as(C.c_void_p))
    assert out[:n].tolist() == [0, 1, 2, 3, 3, 2, 1, 2, 4, 5]


def _integration_setup(variant=VARIANT):
    k = load("integration")
    # src/index/indexing.rs:43-144: uppercase, IUPAC replacement (short runs -> random base kept in .tos), contig map
    text, contigs, orig = "", [], {}
    for c in k["contigs"]:
        seq = c["seq"].upper()
        start = len(text)
        for i, ch in enumerate(seq):
            if ch not in "ACGT":
                orig[start + i] = ch
        seq = "".join(ch if ch in "ACGT" else k["n_replacement"] for ch in seq)
        contigs.append((start, start + len(seq) - 1, c["name"]))
        text += seq
    idx = ob.OracleIndex.from_text(text.encode(), "$ACGTX", 128)
    for s, e, nme in contigs:
        idx.add_contig(s, e, nme)
    for pos, ch in orig.items():
        idx.set_original_symbol(pos, ch)
    idx.sample_sa(32)
    p = oracle_params(k["params"], heap_variant=variant)
    return k, idx, p


def integration_reads(k):
    """Input-side conversion of src/map/record.rs:157-160: reads flagged 0x10 are un-reversed before mapping."""
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads, quals = [], []
    for r in k["reads"]:
        s, q = r["seq"].encode(), np.frombuffer(r["qual"].encode(), dtype=np.uint8) - 33
        if r["flags"] & 0x10:
            s, q = s.translate(comp)[::-1], q[::-1].copy()
        reads.append(s)
        quals.append(q)
    return reads, quals


def test_integration_expectation():
    """tests/integration_tests.rs:174-215 + shared_expectation (:464-868), decoded record fields."""
    k, idx, p = _integration_setup()
    reads, quals = integration_reads(k)
    res = idx.map_batch(p, reads, quals)
    recs = res.records(flags=[r["flags"] for r in k["reads"]])
    got = sorted(zip([r["name"] for r in k["reads"]], recs), key=lambda t: t[0].encode())
    assert [g[0] for g in got] == [e["name"] for e in k["expected_sorted_by_name"]]
    for (name, g), e in zip(got, k["expected_sorted_by_name"]):
        assert int(g["flags"]) == e["flags"], name
        assert int(g["mapq"]) == e["mapq"], name
        assert g["seq"] == e["seq"] and g["qual"] == e["qual"], name
        if e["tid"] is None:
            assert int(g["tid"]) == -1 and int(g["pos"]) == -1 and g["cigar"] == "*", name
            continue
        assert int(g["tid"]) == e["tid"] and int(g["pos"]) + 1 == e["pos"], (name, g)
        assert g["cigar"] == e["cigar"] and g["md"] == e["md"], (name, g)
        assert int(g["x0"]) == e["x0"] and int(g["x1"]) == e["x1"] and g["xt"] == e["xt"], (name, g)
        assert (g["xa"] if g["xa"] != "*" else None) == e["xa"], (name, g)
        if e["xs"] is None:
            assert g["xs_bits"] == "*", name
        else:
            assert np.uint32(int(g["xs_bits"], 16)).view(np.float32) == np.float32(e["xs"]), name
