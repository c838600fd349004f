"""CPU tests of the product's host side and of the kernels' per-read logic compiled for the host (tests/emu):
C-ABI surface, indexer (SA-IS, BWT, block layout, sampled SA, on-disk format), scoring plugins, heap/slab/D-array logic
and post-processing — all against the CPU oracle and the reference's golden vectors.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mapad_amd
from mapad_amd import binding as mb
from mapad_amd import synth
from oracle import binding as ob

import emu_util
from kat_util import load, quals_for, resolve_params
from parity_util import CONTINUOUS, DAMAGE, DOUBLE_STRANDED, IGNORE_BQ, NO_DAMAGE, VINDIJA, assert_same_as_oracle, split_reads
from test_oracle_kats import KATS, SDM, check_search_expectations, integration_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the boundary ---------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mapad_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mapad_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    L = mb.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/mapad_amd.h but not exported"
    assert declared == set(mb.SYMBOLS), declared ^ set(mb.SYMBOLS)
    assert b"gfx950" in L.mapad_version()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_no_gpu_fails_loudly_without_fallback():
    idx = mapad_amd.Index.build([("c", b"ACGTACGTTTGACCA")])
    with pytest.raises(mapad_amd.MapadError) as e:
        mapad_amd.Context(idx, mapad_amd.make_params(resolve_params(NO_DAMAGE)), 0)
    assert e.value.code == -5


# ---- indexer -----------------------------------------------------------------------------------------------------------
def _random_text(rng, n, kind):
    pick = lambda alphabet, k: np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), k)].tobytes()
    if kind == "uniform":
        return pick(b"ACGT", n)
    if kind == "repeat":
        unit = pick(b"ACGT", int(rng.integers(1, 7)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == "poly":
        return pick(b"AAAAAAAC", n)
    raise ValueError(kind)


@pytest.mark.parametrize("wide", [False, True], ids=["sa32", "sa64"])
@pytest.mark.parametrize("kind", ["uniform", "repeat", "poly"])
def test_index_build_matches_naive_suffix_sort(kind, wide, monkeypatch):
    if wide:  # the 64-bit suffix sorter that texts of >= 2^31 rows (3 Gbp references) take
        monkeypatch.setenv("MAPAD_INDEX_FORCE_64", "1")
    rng = np.random.default_rng(12345)
    for n in [1, 2, 3, 7, 31, 32, 33, 255, 256, 257, 1000, 5000]:
        text = _random_text(rng, n, kind)
        p = mapad_amd.Index.build([("c", text)])
        o = ob.OracleIndex.from_text(text, "$ACGTX", 128)
        assert len(p) == 2 * n + 2 == len(o)
        assert np.array_equal(p.bwt(), o.bwt()), (kind, n)
        sa = o.sa()
        sample, er, ev = p.sampled_sa()
        assert np.array_equal(sample, sa[::32])
        bwt = o.bwt()
        want = [(i, int(sa[i])) for i in range(len(sa)) if bwt[i] == 0 and i % 32 != 0]  # extra rows (index/mod.rs:112-118)
        assert list(zip(er.tolist(), ev.tolist())) == want
        _, _, less, sent = p.device_view()
        assert np.array_equal(less[:7], o.less(7))
        assert sent.tolist() == [i for i in range(len(bwt)) if bwt[i] == 0]
        for row in rng.integers(0, len(sa), 20):
            assert p.sa_get(int(row)) == int(sa[int(row)])
        rows = rng.integers(0, len(sa) + 3, 50).astype(np.uint64)
        want_pos = [int(sa[int(r)]) if r < len(sa) else 0xFFFFFFFFFFFFFFFF for r in rows]
        assert [int(x) for x in p.sa_get_batch(rows)] == want_pos


def test_index_ambiguous_bases_and_contigs(monkeypatch):
    k = load("misc_kats")["run_apply"]
    # run_apply with min_run_len 20 (indexing.rs:98-107): short runs replaced (original kept), long runs -> X
    seq = "acgtNNacgtRYacg" + "N" * 25 + "ACGTACGTAC"
    monkeypatch.setenv("MAPAD_INDEX_FIXED_REPLACEMENT", "A")
    p = mapad_amd.Index.build([("one", seq.encode()), ("two", b"GGGTTTAAACCC")])
    text = seq.upper()
    repl = "".join(ch if ch in "ACGT" else "A" for ch in text[:15]) + "X" * 25 + text[40:]
    full = repl + "GGGTTTAAACCC"
    o = ob.OracleIndex.from_text(full.encode(), "$ACGTX", 128)
    assert np.array_equal(p.bwt(), o.bwt())
    assert p.contigs() == [("one", 0, len(seq) - 1), ("two", len(seq), len(seq) + 11)]
    with pytest.raises(mapad_amd.MapadError):
        mapad_amd.Index.build([("bad", b"ACGT!ACGT")])  # non-IUPAC symbol (indexing.rs:71)
    # 'X' survives reverse complementation (indexing.rs:447-450)
    assert k["revcomp"] == ["GATTXACA", "TGTXAATC"]
    sa = o.sa()
    for row in range(0, len(sa), 7):
        assert p.sa_get(row) == int(sa[row])  # LF walks through 'X' rows


def test_index_on_disk_roundtrip(tmp_path):
    g = synth.genome(20_000, seed=8)
    g[5000:5030] = ord("N")  # long run -> X
    g[100] = ord("N")
    p = mapad_amd.Index.build([("chrA", g[:12000]), ("chrB", g[12000:])], seed=7)
    prefix = str(tmp_path / "ref.fa")
    p.save(prefix)
    for ext in ("tbw", "tle", "toc", "trt", "tsa", "tpi", "tos"):
        assert os.path.getsize(f"{prefix}.{ext}") > 10
        assert open(f"{prefix}.{ext}", "rb").read(10) == b"\xff\x06\x00\x00sNaPpY"  # snappy frame stream identifier
    q = mapad_amd.Index.open(prefix)
    assert np.array_equal(p.bwt(), q.bwt()) and p.contigs() == q.contigs()
    a, b = p.sampled_sa(), q.sampled_sa()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for row in (0, 1, 31, 32, 33, 12345, len(p) - 1):
        assert p.sa_get(row) == q.sa_get(row)
    # version gate (versioned_index.rs:31-40): flip the version byte of .tle -> MAPAD_ERR_INDEX_VERSION
    raw = bytearray(open(prefix + ".tle", "rb").read())
    raw[10 + 8] = 4  # first payload byte of the first (uncompressed) chunk = Item.version
    import zlib  # noqa: F401  (crc is CRC32C, not zlib's; corrupting the version also breaks the CRC -> parse error is acceptable)
    open(prefix + ".tle", "wb").write(raw)
    with pytest.raises(mapad_amd.MapadError) as e:
        mapad_amd.Index.open(prefix)
    assert e.value.code in (-3, -4)
    with pytest.raises(mapad_amd.MapadError) as e:
        mapad_amd.Index.open(str(tmp_path / "missing"))
    assert e.value.code == -2


# ---- scoring plugins (trait surface of the C ABI) -----------------------------------------------------------------------
@pytest.mark.parametrize("block", SDM["get"], ids=[b["name"] for b in SDM["get"]])
def test_product_sdm_get_golden(block):
    p = mapad_amd.make_params(resolve_params({**block["params"], "bound": "test"}))
    L = mb.lib()
    for v, i, ln, f, t, q in block["asserts"]:
        assert abs(L.mapad_sdm_get(C.byref(p), i, ln, ord(f), ord(t), q) - v) < block["tolerance"]


def test_product_plugins_bit_equal_to_oracle():
    rng = np.random.default_rng(7)
    L, O = mb.lib(), ob.lib()
    models = [SDM["get"][1]["params"], SDM["get"][2]["params"], {"model": "vindija_pwm"}, {"model": "test", "deam_score": -0.5, "mm_score": -1.0, "match_score": 0.0},
              dict(SDM["get"][1]["params"], ignore_base_quality=1)]
    bounds = [{"bound": "discrete", "poisson_threshold": 0.03, "base_error_rate": 0.02}, {"bound": "continuous", "cutoff": -0.5, "exponent": 1.1},
              {"bound": "test", "threshold": -3.0, "repr_mm_bound": -1.5}]
    f32 = lambda x: np.float32(x).tobytes()
    for m in models:
        for b in bounds:
            rp = resolve_params({**m, **b})
            pp, po = mapad_amd.make_params(rp), ob.make_params(rp)
            assert f32(L.mapad_sdm_representative_mismatch_penalty(C.byref(pp))) == f32(O.mo_sdm_repr_mm(C.byref(po)))
            for _ in range(300):
                ln = int(rng.integers(1, 300)); i = int(rng.integers(0, ln)); q = int(rng.integers(0, 94))
                fr, to = int(rng.choice(list(b"ACGT"))), int(rng.choice(list(b"ACGTN")))
                assert f32(L.mapad_sdm_get(C.byref(pp), i, ln, fr, to, q)) == f32(O.mo_sdm_get(C.byref(po), i, ln, fr, to, q))
                for om in (0, 1):
                    assert f32(L.mapad_sdm_min_penalty(C.byref(pp), i, ln, to, q, om)) == f32(O.mo_sdm_min_penalty(C.byref(po), i, ln, to, q, om))
                v = float(np.float32(rng.uniform(-40, 0))); r = float(np.float32(rng.uniform(-40, 0)))
                assert L.mapad_mb_reject(C.byref(pp), v, ln) == O.mo_mb_reject(C.byref(po), v, ln)
                assert L.mapad_mb_reject_iterative(C.byref(pp), v, r) == O.mo_mb_reject_iterative(C.byref(po), v, r)
                assert f32(L.mapad_mb_remaining_frac_of_repr_mm(C.byref(pp), v, ln)) == f32(O.mo_mb_remaining_frac(C.byref(po), v, ln))
            for ln in (1, 17, 50, 100, 1024):
                assert L.mapad_sdm_alignment_start(C.byref(pp), ln) == O.mo_sdm_alignment_start(C.byref(po), ln)


def test_params_from_cli_derivation():
    """build_alignment_parameters (main.rs:418-499) with the README example values."""
    p = mapad_amd.params_from_cli(library="single_stranded", five_prime_overhang=0.5, three_prime_overhang=0.5, ds_deamination_rate=0.02,
                                  ss_deamination_rate=1.0, divergence=0.02, poisson_prob=0.03, indel_rate=0.001, gap_extension_penalty=0.5)
    L = mb.lib()
    repr_mm = L.mapad_sdm_representative_mismatch_penalty(C.byref(p))
    assert "%.2f" % repr_mm == "-7.20"
    assert np.float32(p.divergence) == np.float32(0.02) / np.float32(3.0)
    assert np.float32(p.penalty_gap_extend) == np.float32(0.5) * np.float32(repr_mm)
    assert abs(p.penalty_gap_open - np.log2(0.001)) < 1e-5 and p.bound_kind == 0 and p.chunk_size == 250000
    p2 = mapad_amd.params_from_cli(poisson_prob=None, as_cutoff=0.6, as_cutoff_exponent=1.2)
    assert p2.bound_kind == 1 and np.float32(p2.cutoff) == np.float32(-0.6)


# ---- the kernels' per-read logic on the host (same source as the device code) ----------------------------------------------
@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_kernel_logic_search_kat(case):
    ref = KATS["ref10k"] if case["reference"] == "@ref10k" else case["reference"]
    rp = resolve_params(case["params"])
    pidx = mapad_amd.Index.build([("ref", ref.encode())])
    oidx = ob.OracleIndex.from_text(ref.encode(), "$ACGTX", 128)
    q = quals_for(case["pattern"], case["qual"])
    seqs = np.frombuffer(case["pattern"].encode(), dtype=np.uint8)
    offsets = np.array([0, len(seqs)], dtype=np.uint64)
    res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, q, offsets)
    ores = oidx.map_batch(ob.make_params(rp), [case["pattern"].encode()], [q], keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    hits = res.hits(0)
    for h, oh in zip(hits, ores.hits(0)):
        h["ops"] = oh["ops"]
    check_search_expectations(case, hits, oidx.sa(), lambda h, b: ores.bam_fields(0, h, backward=b))


@pytest.mark.parametrize("name,prm,kw,n", [
    ("no_damage_q40", NO_DAMAGE, dict(qual=40), 400),
    ("damage_q20_40", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 300),
    ("mixed_len_indels", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 150),
    ("continuous_bound", CONTINUOUS, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 300),
    ("continuous_mixed_len", CONTINUOUS, dict(qual_range=(20, 40), len_range=(35, 70), indel_frac=0.05), 150),
    ("double_stranded", DOUBLE_STRANDED, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 300),
    ("ignore_base_quality", IGNORE_BQ, dict(qual_range=(2, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 300),
    ("vindija_pwm", VINDIJA, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 70), indel_frac=0.05), 200),
])
@pytest.mark.parametrize("step", ["lane_parallel_commit", "payload_cache"])
def test_kernel_logic_synthetic(name, prm, kw, n, step, monkeypatch):
    # the two build options of the search step that change how a frame's children reach the heap (search_core.hpp): the quad kernel's default — children pushed
    # side by side, emulated lane by lane here — and the payload cache of heap slots 1 and 2 with the sequential pushes
    monkeypatch.setenv("MAPAD_EMU_PAYLOAD_CACHE", "1" if step == "payload_cache" else "0")
    g = synth.genome(150_000, seed=99)
    seqs, quals, offsets = synth.reads(g, n, 50, seed=7 + len(name), **kw)
    rp = resolve_params(prm)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("step", ["lane_parallel_commit", "payload_cache"])
def test_kernel_logic_heap_variants_follow_the_matching_oracle_variant(variant, step, monkeypatch):
    # csrc/heap_core.hpp: MAPAD_HEAP_VARIANT — the readings of the min-max-heap crate's tie rules that the reference's tests cannot tell apart (scan order of a
    # trickle-down stride, pop_max on a tie of slots 1 and 2).  The product built with reading v equals the oracle run with reading v: every KAT that searches,
    # and the 35-100 bp mix with indels (where the readings differ from one another in edit tracks and event counters), incl. the eviction path.
    monkeypatch.setenv("MAPAD_EMU_PAYLOAD_CACHE", "1" if step == "payload_cache" else "0")
    for case in KATS["cases"]:
        ref = KATS["ref10k"] if case["reference"] == "@ref10k" else case["reference"]
        rp = resolve_params(case["params"])
        pidx = mapad_amd.Index.build([("ref", ref.encode())])
        q = quals_for(case["pattern"], case["qual"])
        seqs = np.frombuffer(case["pattern"].encode(), dtype=np.uint8)
        offsets = np.array([0, len(seqs)], dtype=np.uint64)
        res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, q, offsets, heap_variant=variant)
        ores = ob.OracleIndex.from_text(ref.encode(), "$ACGTX", 128).map_batch(ob.make_params(dict(rp, heap_variant=variant)), [case["pattern"].encode()], [q], keep_d=True)
        assert_same_as_oracle(ores, res, offsets)
    g = synth.genome(150_000, seed=99)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    differs = 0
    for prm, kw, n, limits in ((DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 200, {}),
                               (NO_DAMAGE, dict(qual=40), 300, {}),
                               (DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 60, {"stack_limit": 60, "edit_tree_limit": 400})):
        seqs, quals, offsets = synth.reads(g, n, 50, seed=21 + n, **kw)
        rp = dict(resolve_params(prm), **limits)
        reads, qs = split_reads(seqs, quals, offsets)
        res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, heap_variant=variant)
        ores = oidx.map_batch(ob.make_params(dict(rp, heap_variant=variant)), reads, qs, n_threads=8, keep_d=True)
        assert_same_as_oracle(ores, res, offsets)
        o0 = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8)
        differs += int((o0.counters != ores.counters).any(axis=1).sum())
    assert differs > 0  # the readings are not the same algorithm: the switch does something


@pytest.mark.parametrize("subtree", [0, 1])
def test_kernel_logic_with_the_heap_levels_in_subtree_blocks(subtree):
    """csrc/heap_core.hpp: MAPAD_SUBTREE_HEAP — the host build of the step with the arena's heap levels in either physical layout: deep heaps (35-100 bp reads with
    indels), arena migrations (physical ranges copied: HeapLayout::phys_end) and the overflow recovery's pop_min sifts, against the oracle."""
    g = synth.genome(150_000, seed=99)
    seqs, quals, offsets = synth.reads(g, 250, 50, seed=19, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    reads, qs = split_reads(seqs, quals, offsets)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    for rp, caps in ((resolve_params(DAMAGE), dict(node_cap=64, heap_cap=64)), (dict(resolve_params(DAMAGE), stack_limit=300, edit_tree_limit=100000), {})):
        res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, subtree=subtree, **caps)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
        assert ores.counters[:, 3].max() > 2000 and (not caps or res.n_second_pass > 50)
        assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("step", ["lane_parallel_commit", "payload_cache"])
def test_kernel_logic_second_pass_and_limit_recovery(step, monkeypatch):
    monkeypatch.setenv("MAPAD_EMU_PAYLOAD_CACHE", "1" if step == "payload_cache" else "0")
    g = synth.genome(60_000, seed=5)
    seqs, quals, offsets = synth.reads(g, 150, 50, seed=11)
    reads, qs = split_reads(seqs, quals, offsets)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    rp = resolve_params(NO_DAMAGE)
    res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, node_cap=16, heap_cap=16)
    assert res.n_second_pass > 0 and res.n_third_pass > 0  # arena migrations (x4, twice) and full-limit re-runs
    assert_same_as_oracle(oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True), res, offsets)
    for limits in ({"stack_limit": 40, "edit_tree_limit": 100000}, {"stack_limit": 100000, "edit_tree_limit": 120},
                   {"stack_limit": 40, "edit_tree_limit": 100000, "stack_limit_abort": 1}):
        rp2 = dict(rp, **limits)
        res = emu_util.map_batch(pidx, mapad_amd.make_params(rp2), seqs, quals, offsets)
        ores = oidx.map_batch(ob.make_params(rp2), reads, qs, n_threads=8, keep_d=True)
        assert ores.counters[:, 3].max() > 40  # the limits really bite
        assert_same_as_oracle(ores, res, offsets)
        if limits.get("stack_limit_abort"):
            assert (res.status == 2).any()


def test_host_tail_search_on_host_threads(monkeypatch):
    """The search of a host-tail worker (csrc/host_tail.hpp: tail_search — search_core.hpp's step with the payload cache, worker pinning, the prefetches between
    evictions) on four threads, without a GPU: pops and status of every read equal the oracle's — with limits small enough that reads go on under pop_min eviction and
    with the abort switch — and nothing a read gets back (status, counters, hits, score bits, edit tracks: one digest per read) moves with the prefetch settings or the
    pinning."""
    g = synth.genome(80_000, seed=21)
    seqs, quals, offsets = synth.reads(g, 300, 50, seed=12, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    reads, qs = split_reads(seqs, quals, offsets)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    sel = np.arange(len(reads))
    for limits in ({}, {"stack_limit": 60, "edit_tree_limit": 100000}, {"stack_limit": 100000, "edit_tree_limit": 150}, {"stack_limit": 60, "edit_tree_limit": 100000, "stack_limit_abort": 1}):
        rp = dict(resolve_params(DAMAGE), **limits)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8)
        digests = []
        for pin, prefetch in (("1", "11"), ("0", "00"), ("1", "10"), ("0", "01")):
            monkeypatch.setenv("MAPAD_TAIL_PIN", pin)
            monkeypatch.setenv("MAPAD_TAIL_PREFETCH", prefetch)
            _, pops, status, dig, _ = emu_util.tail_search(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, sel)
            assert np.array_equal(pops, ores.counters[:, 3]), (limits, prefetch)
            digests.append(dig)
            if limits.get("stack_limit_abort"):
                assert (status == 2).any() and set(status.tolist()) <= {0, 2}
            else:
                assert not status.any()
        assert all(np.array_equal(digests[0], d) for d in digests[1:])
        if limits:
            assert ores.counters[:, 3].max() >= 100  # the searches run long enough to get to the limits


def test_kernel_logic_long_and_degenerate_reads():
    g = synth.genome(50_000, seed=3)
    long_read = g[5000:5300].copy()
    long_read[[10, 150, 290]] = ord("A")
    reads = [b"A", b"NNNNNNNNNNNNNNNNNNNNNNNNN", g[1000:1016].tobytes(), g[2000:2050].tobytes(), long_read.tobytes(), g[7000:8000].tobytes(),
             synth.revcomp(g[9000:9130]).tobytes()]
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    quals = np.full(len(seqs), 40, np.uint8)
    pidx = mapad_amd.Index.build([("chr1", g)])
    rp = resolve_params(NO_DAMAGE)
    res = emu_util.map_batch(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    ores = oidx.map_batch(ob.make_params(rp), reads, [quals[int(offsets[i]):int(offsets[i + 1])] for i in range(len(reads))], keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    assert [len(res.hits(i)) for i in range(len(reads))][3:] == [1, 1, 1, 1]


@pytest.mark.parametrize("max_n,levels", [(40, 3), (300, 2), (300, 7), (5000, 4), (5000, 1000), (70000, 25)])
def test_lane_parallel_commit_equals_sequential_pushes_on_random_heaps(max_n, levels):
    """The quad kernel pushes a frame's children side by side (search_core.hpp: MAPAD_PAR_COMMIT): stayers stored at once, movers in order.  The argument that this
    leaves the heap of the sequential pushes (a stayer stays whatever its siblings do) is checked here slot by slot on random min-max heaps — few score levels
    (ties everywhere, like the no-damage model), many (like the damage model), heaps that straddle the near / arena boundary (63 slots) and level boundaries."""
    for seed in range(4):
        assert emu_util.lib().emu_par_commit_selftest(seed + 17 * levels, 4000, max_n, levels) == 0


@pytest.mark.parametrize("subtree", [1, 0])
def test_heap_layout_is_a_bijection_with_aligned_pairs_and_one_block_per_stride(subtree):
    """heap_core.hpp: HeapLayout — where a logical slot of the arena's heap levels lives.  For every slot below 2^21 of the quads' (63 near slots) and the pairs' (31)
    layouts: physical entries are distinct, never inside the shadow of the near levels, covered by what an arena migration copies (phys_end, monotone), sibling pairs
    are 16-byte aligned neighbours, and (subtree blocks) a max-level entry's children and grandchildren share one 64-byte block."""
    assert emu_util.heap_selftest_lib(subtree).heap_layout_properties(1 << 21) == 0


@pytest.mark.parametrize("subtree,variant", [(1, 0), (0, 0), (1, 3)])
@pytest.mark.parametrize("max_n,levels", [(100, 3), (3000, 2), (3000, 500), (200_000, 6), (1_200_000, 40)])
def test_heap_in_its_physical_layout_equals_the_oracles_heap_under_random_operations(subtree, variant, max_n, levels):
    """The layout is a logical -> physical slot change: under random pushes, pop_max and pop_min (sifts from the root through the arena levels — the search's
    evictions) the product's heap equals the oracle's plain-vector min-max heap entry for entry, up to 2^20 entries (heap level 20), for the default and the last
    reading of the tie rules."""
    L = emu_util.heap_selftest_lib(subtree, variant)
    ops = min(6 * max_n, 3_000_000)
    assert L.heap_random_ops_selftest(11 + levels, ops, max_n, levels, max(ops // 7, 1)) == 0


@pytest.mark.parametrize("bits", [20, 33, 40, 47])
def test_row_to_block_split_is_exact(bits):
    """fmd_device.hpp: block_pos — BWT row -> (64-byte block of 96 rows, row inside it) by 32-bit arithmetic instead of a 64-bit division: exact on random rows, around
    every power of two and around the multiples of 96 * 2^16 where the split of the quotient carries."""
    assert emu_util.lib().emu_block_pos_selftest(7 + bits, 2_000_000, bits) == 0


# ---- post-search ----------------------------------------------------------------------------------------------------------
def check_integration_records(k, recs):
    """shared_expectation of tests/integration_tests.rs:464-868 on decoded record fields."""
    got = sorted(zip([r["name"] for r in k["reads"]], recs, k["reads"]), key=lambda t: t[0].encode())
    assert [g[0] for g in got] == [e["name"] for e in k["expected_sorted_by_name"]]
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for (name, g, rd), e in zip(got, k["expected_sorted_by_name"]):
        assert g["flags"] == e["flags"] and g["mapq"] == e["mapq"], (name, g)
        # SEQ/QUAL as written to BAM: mapping input orientation (record.rs:157-160), reversed again on the reverse strand (mapping.rs:795-819)
        s, q = rd["seq"].encode(), rd["qual"]
        if rd["flags"] & 0x10:
            s, q = s.translate(comp)[::-1], q[::-1]
        if g["reverse"]:
            s, q = s.translate(comp)[::-1], q[::-1]
        assert s.decode() == e["seq"] and q == e["qual"], name
        if e["tid"] is None:
            assert not g["mapped"] and g["tid"] == -1 and g["pos"] == -1
            continue
        assert g["tid"] == e["tid"] and g["pos"] + 1 == e["pos"] and g["cigar"] == e["cigar"] and g["md"] == e["md"], (name, g)
        assert g["x0"] == e["x0"] and g["x1"] == e["x1"] and g["xt"] == e["xt"] and (g["xa"] or None) == e["xa"], (name, g)
        assert (g["xs"] is None) == (e["xs"] is None) and (e["xs"] is None or g["xs"] == np.float32(e["xs"])), (name, g)


def test_postproc_integration_expectation(monkeypatch):
    k = load("integration")
    monkeypatch.delenv("MAPAD_INDEX_FIXED_REPLACEMENT", raising=False)  # StdRng(1234) itself must draw the base the reference's expectation implies
    pidx = mapad_amd.Index.build([(c["name"], c["seq"].encode()) for c in k["contigs"]], seed=1234)
    params = mapad_amd.make_params(resolve_params(k["params"]))
    reads, quals = integration_reads(k)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs, qs = np.frombuffer(b"".join(reads), dtype=np.uint8), np.concatenate(quals)
    res = emu_util.map_batch(pidx, params, seqs, qs, offsets)
    recs = mapad_amd.hits_to_records(pidx, params, res, seqs, qs, offsets, in_flags=[r["flags"] for r in k["reads"]])
    check_integration_records(k, recs)


def test_postproc_matches_oracle_on_multicontig_genome():
    g = synth.genome(120_000, seed=21)
    # make repeats so that multi-mapping (X0 > 1, XA, low MAPQ) occurs
    g[40_000:40_400] = g[10_000:10_400]
    g[80_000:80_400] = g[10_000:10_400]
    contigs = [("c1", g[:50_000]), ("c2", g[50_000:90_000]), ("c3", g[90_000:])]
    pidx = mapad_amd.Index.build(contigs)
    oidx = ob.OracleIndex.from_text(g.tobytes(), "$ACGTX", 128)
    s = 0
    for nme, c in contigs:
        oidx.add_contig(s, s + len(c) - 1, nme)
        s += len(c)
    oidx.sample_sa(32)
    seqs, quals, offsets = synth.reads(g, 300, 50, seed=77)
    rep = synth.reads(g[10_000:10_400], 60, 50, seed=78, exo_frac=0.0)
    seqs = np.concatenate([seqs, rep[0]]); quals = np.concatenate([quals, rep[1]])
    offsets = np.concatenate([offsets, rep[2][1:] + offsets[-1]])
    rp = resolve_params(NO_DAMAGE)
    params = mapad_amd.make_params(rp)
    res = emu_util.map_batch(pidx, params, seqs, quals, offsets)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8)
    orecs = ores.records(flags=np.zeros(len(reads), np.uint16))
    recs = mapad_amd.hits_to_records(pidx, params, res, seqs, quals, offsets, seed=0)
    n_multi = 0
    for i, (o, g_) in enumerate(zip(orecs, recs)):
        mapped = o["cigar"] != "*"
        assert g_["mapped"] == mapped and g_["flags"] == int(o["flags"]) and g_["mapq"] == int(o["mapq"]), (i, o, g_)
        if not mapped:
            continue
        best_rows = max(int(h["interval"][2]) for h in ores.hits(i))
        assert g_["cigar"] == o["cigar"] and g_["md"] == o["md"] and g_["nm"] == int(o["nm"]), (i, o, g_)
        assert g_["x0"] == int(o["x0"]) and g_["x1"] == int(o["x1"]) and g_["xt"] == o["xt"], (i, o, g_)
        assert np.float32(g_["as"]).tobytes() == np.uint32(int(o["as_bits"], 16)).tobytes()
        if best_rows <= 2:  # the reported row of a >= 3-row interval is drawn at random in the reference (F5); <= 2 rows is deterministic
            assert g_["tid"] == int(o["tid"]) and g_["pos"] == int(o["pos"]) and (g_["xa"] or "*") == o["xa"], (i, o, g_)
        n_multi += int(o["x0"]) > 1
    assert n_multi > 10


def test_records_equal_the_oracles_intervals_to_record_over_the_same_hits_and_the_text_itself():
    """The record-level audit's machinery (profiles/audit_c4.py --records, tests/test_gpu_index.py at n > 2^32) on a small multi-contig text with repeats: every field of
    every record the product builds (host path here) — flags, tid, POS, MAPQ, strand, AS / XS bits, NM, X0, X1, XT, CIGAR, MD, XA — equals the oracle's
    intervals_to_record run over the product's hits with the same per-read stand-ins for rand::rng() (seed 0), incl. the rows drawn from >= 3-row intervals;
    and the ungapped records agree with the text itself (NM = mismatches at POS on the reported strand)."""
    from parity_util import canonical_records, check_ungapped_records_against_the_text, compare_records, oracle_records_from_product_hits, records_digest
    g = synth.genome(150_000, seed=41)
    g[60_000:60_300] = g[5_000:5_300]
    g[110_000:110_300] = g[5_000:5_300]
    g[130_000:130_200] = synth.revcomp(g[5_050:5_250])  # the same stretch on the other strand
    contigs = [("c1", g[:70_000]), ("c2", g[70_000:120_000]), ("c3", g[120_000:])]
    pidx = mapad_amd.Index.build(contigs)
    oidx = ob.OracleIndex.from_text(g.tobytes(), "$ACGTX", 128)
    starts, s0 = [], 0
    for nme, c in contigs:
        oidx.add_contig(s0, s0 + len(c) - 1, nme)
        starts.append(s0)
        s0 += len(c)
    sample, er, ev = pidx.sampled_sa()
    oidx.set_sampled_sa(sample, 32, er, ev)  # the product's samples under the oracle's own LF walk, as at 3 Gbp
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=79, qual_range=(20, 40), len_range=(35, 80), indel_frac=0.05)
    rep = synth.reads(g[5_000:5_300], 300, 50, seed=80, exo_frac=0.0)
    seqs = np.concatenate([seqs, rep[0]]); quals = np.concatenate([quals, rep[1]])
    offsets = np.concatenate([offsets, rep[2][1:] + offsets[-1]])
    rp = resolve_params(NO_DAMAGE)
    params = mapad_amd.make_params(rp)
    res = emu_util.map_batch(pidx, params, seqs, quals, offsets)
    recs, text = mapad_amd.hits_to_records(pidx, params, res, seqs, quals, offsets, seed=0, as_arrays=True)
    prod = canonical_records(recs, text, oracle_side=False)
    n = len(offsets) - 1
    ora_parts = [oracle_records_from_product_hits(oidx, ob.make_params(rp), res, seqs, quals, offsets, lo, hi, n_threads=3) for lo, hi in ((0, 700), (700, n))]  # chunked like the audit
    for (lo, hi), (orecs, otext) in zip(((0, 700), (700, n)), ora_parts):
        sub = canonical_records(recs[lo:hi], text, oracle_side=False)
        n_bad, first, per_field = compare_records(sub, canonical_records(orecs, otext, oracle_side=True))
        assert n_bad == 0, (first, per_field)
    assert records_digest(prod) == records_digest(canonical_records(recs.copy(), text.copy(), oracle_side=False))
    f = prod[0]
    assert (f["x0"] >= 3).sum() > 100 and (prod[1][2][0] > 0).sum() > 100 and f["reverse"].sum() > 300  # multi-row intervals, XA entries and both strands took part
    checked, failed = check_ungapped_records_against_the_text(g, recs, text, seqs, offsets, contig_starts=starts)
    assert checked > 1200 and failed == 0
    bad = recs.copy()
    bad["pos"][np.flatnonzero(recs["mapped"])[0]] += 1
    assert compare_records(canonical_records(bad[:700], text, False), canonical_records(*ora_parts[0], True))[0] == 1


def test_postproc_thread_count_does_not_change_records(monkeypatch):
    """hits -> records runs on host threads over contiguous read ranges; the record array and text offsets must not depend on it."""
    g = synth.genome(200_000, seed=31)
    g[150_000:150_300] = g[20_000:20_300]  # some multi-mapping
    seqs, quals, offsets = synth.reads(g, 9000, 50, seed=5)
    pidx = mapad_amd.Index.build([("c1", g[:120_000]), ("c2", g[120_000:])])
    params = mapad_amd.make_params(resolve_params(NO_DAMAGE))
    res = emu_util.map_batch(pidx, params, seqs, quals, offsets)
    monkeypatch.setenv("MAPAD_POSTPROC_THREADS", "1")
    one = mapad_amd.hits_to_records(pidx, params, res, seqs, quals, offsets, seed=3)
    monkeypatch.setenv("MAPAD_POSTPROC_THREADS", "4")
    four = mapad_amd.hits_to_records(pidx, params, res, seqs, quals, offsets, seed=3)
    assert one == four and sum(r["mapped"] for r in one) > 7000


def test_iupac_replacement_follows_rand_stdrng(monkeypatch):
    """`mapad index` replaces ambiguity codes with rand 0.9 `StdRng::seed_from_u64(seed)` + `choose` (indexing.rs:30,78-92).  The restatement
    (host_index.hpp: StdRngCompat) against an independent Python restatement of the same published algorithms — PCG32 key expansion, ChaCha12,
    Canon's method — over a text with every ambiguity code, and against the one draw the reference pins: seed 1234 turns a lone N into 'A'."""
    monkeypatch.delenv("MAPAD_INDEX_FIXED_REPLACEMENT", raising=False)
    M64 = (1 << 64) - 1

    def key_of(seed):
        out, state = [], seed
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & M64
            x, rot = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF, state >> 59
            out.append(((x >> rot) | (x << ((32 - rot) & 31))) & 0xFFFFFFFF)
        return out

    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & 0xFFFFFFFF

    def block(key, counter):
        s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + key + [counter & 0xFFFFFFFF, counter >> 32, 0, 0]
        w = s[:]

        def qr(a, b, c, d):
            w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rotl(w[d] ^ w[a], 16)
            w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rotl(w[b] ^ w[c], 12)
            w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rotl(w[d] ^ w[a], 8)
            w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rotl(w[b] ^ w[c], 7)
        for _ in range(6):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return [(w[i] + s[i]) & 0xFFFFFFFF for i in range(16)]

    def stream(seed):
        key, c = key_of(seed), 0
        while True:
            yield from block(key, c)
            c += 1

    sets = {"R": "AG", "Y": "CT", "K": "GT", "M": "AC", "S": "CG", "W": "AT", "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
    for seed in (1234, 0, 7):
        codes = "NRYKMSWBDHVU" * 40
        seq = "ACGT".join(codes)  # every code is a run of one
        want, words = [], stream(seed)
        for ch in seq:
            if ch in "ACGT":
                want.append(ch)
            elif ch == "U":
                want.append("T")
            else:
                n = len(sets[ch])
                m = next(words) * n
                r, lo = m >> 32, m & 0xFFFFFFFF
                if lo > ((-n) & 0xFFFFFFFF):
                    if lo + ((next(words) * n) >> 32) > 0xFFFFFFFF:
                        r += 1
                want.append(sets[ch][r])
        p = mapad_amd.Index.build([("c", seq.encode())], seed=seed)
        o = ob.OracleIndex.from_text("".join(want).encode(), "$ACGTX", 128)
        assert np.array_equal(p.bwt(), o.bwt()), seed
    p = mapad_amd.Index.build([("c", b"ACGTACGTNACGTTTGA")], seed=1234)
    assert np.array_equal(p.bwt(), ob.OracleIndex.from_text(b"ACGTACGTAACGTTTGA", "$ACGTX", 128).bwt())
