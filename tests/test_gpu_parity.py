"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI vs the CPU oracle and the golden vectors."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mapad_amd
from mapad_amd import synth
from oracle import binding as ob

from kat_util import load, quals_for, resolve_params
from parity_util import CONTINUOUS, DAMAGE, DOUBLE_STRANDED, IGNORE_BQ, NO_DAMAGE, VINDIJA, assert_same_as_oracle, split_reads
from test_oracle_kats import KATS, check_search_expectations, integration_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rerun_in_heavy_build(request):
    """heavy_kernel (one wavefront per read, MAPAD_HEAVY=1) is compiled only into libmapad_amd.heavy.so (mapad_amd/build.py: -DMAPAD_HEAVY_KERNEL; built by
    __graft_entry__.build()) since round 6.  A test case that asks for it runs itself again in a child process that loads that library (a process loads one
    library).  Returns True in the parent — the child has run the case —, False in the child."""
    if os.environ.get("MAPAD_HEAVY_BUILD_CHILD"):
        return False
    from mapad_amd import build
    if not os.path.exists(build.lib_path(heavy=True)):
        build.build(heavy=True)
    env = dict(os.environ, MAPAD_AMD_LIB=build.lib_path(heavy=True), MAPAD_HEAVY_BUILD_CHILD="1")
    pr = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", request.node.nodeid], cwd=ROOT, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    assert pr.returncode == 0 and " passed" in pr.stdout, pr.stdout[-3000:]
    return True


def _gpu_map(index, params, seqs, quals, offsets):
    ctx = mapad_amd.Context(index, params, 0)
    try:
        return ctx.map_batch(seqs, quals, offsets)
    finally:
        ctx.close()


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_search_kat_on_gpu(case):
    ref = KATS["ref10k"] if case["reference"] == "@ref10k" else case["reference"]
    rp = resolve_params(case["params"])
    pidx = mapad_amd.Index.build([("ref", ref.encode())])
    oidx = ob.OracleIndex.from_text(ref.encode(), "$ACGTX", 128)
    assert np.array_equal(pidx.bwt(), oidx.bwt())
    q = quals_for(case["pattern"], case["qual"])
    seqs = np.frombuffer(case["pattern"].encode(), dtype=np.uint8)
    offsets = np.array([0, len(seqs)], dtype=np.uint64)
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, q, offsets)
    ores = oidx.map_batch(ob.make_params(rp), [case["pattern"].encode()], [q], keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    # and the reference's own expectations, evaluated on the GPU result
    hits = res.hits(0)
    for h, oh in zip(hits, ores.hits(0)):
        h["ops"] = oh["ops"]
    check_search_expectations(case, hits, oidx.sa(), lambda h, b: ores.bam_fields(0, h, backward=b))


@pytest.mark.parametrize("name,prm,kw", [
    ("no_damage_q40", NO_DAMAGE, dict(qual=40)),
    ("damage_q20_40", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))),
    ("mixed_len_indels", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05)),
    # the kernel variants of the other plugin settings (mismatch_bounds.rs:76-120, sequence_difference_models.rs:125-144,286-287)
    ("continuous_bound", CONTINUOUS, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))),
    ("continuous_mixed_len", CONTINUOUS, dict(qual_range=(20, 40), len_range=(35, 70), indel_frac=0.05)),
    ("double_stranded", DOUBLE_STRANDED, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))),
    ("ignore_base_quality", IGNORE_BQ, dict(qual_range=(2, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))),
    # VindijaPwm (sequence_difference_models.rs:336-396): what a dispatcher's task sheet may name (worker.rs:57-75); bidirectional search from the read's middle
    ("vindija_pwm", VINDIJA, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 70), indel_frac=0.05)),
])
@pytest.mark.parametrize("lanes_per_read", ["4", "2", "1"])
def test_synthetic_batch_matches_oracle(name, prm, kw, lanes_per_read, monkeypatch):
    monkeypatch.setenv("MAPAD_LANES_PER_READ", lanes_per_read)
    g = synth.genome(300_000, seed=99)
    n = 3000 if "len_range" not in kw else 600
    seqs, quals, offsets = synth.reads(g, n, 50, seed=7 + len(name), **kw)
    rp = resolve_params(prm)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("env", [{"MAPAD_ORDER": "0"}, {"MAPAD_POOL_BUDGET_GB": "1", "MAPAD_TIER0_NODES": "256"}, {"MAPAD_NEAR_LDS": "0"},
                                 {"MAPAD_TIER0_NODES": "32", "MAPAD_CLASS_COUNTS": "4,4,4,4,4,4,4,4,4,4", "MAPAD_MAX_WAITS": "0"},
                                 {"MAPAD_ORDER_CHUNK_LOG2": "10"}, {"MAPAD_HIT_POOL": "64"},
                                 {"MAPAD_HEAVY": "1", "MAPAD_TIER0_NODES": "256"}, {"MAPAD_HEAVY": "1", "MAPAD_POOL_BUDGET_GB": "1", "MAPAD_TIER0_NODES": "256"},
                                 {"MAPAD_HEAVY": "1", "MAPAD_HEAVY_FAST": "0", "MAPAD_TIER0_NODES": "64"},
                                 {"MAPAD_TIER0_NODES": "64", "MAPAD_WIDE_COPY_NODES": "16"}, {"MAPAD_TIER0_NODES": "32", "MAPAD_WIDE_COPY_NODES": "1", "MAPAD_CLASS_COUNTS": "64,16,8,8,8,8,8,8,8,8"}],
                         ids=["input_order", "tiny_pool_budget", "near_data_in_hbm", "give_up_and_restart", "order_chunks_of_1024",
                              "hit_pool_overflow_retry", "heavy_wavefronts", "heavy_wavefronts_tiny_pools", "heavy_wavefronts_general_steps",
                              "wavefront_wide_migrations", "wavefront_wide_migrations_busy_pools"])
def test_scheduling_and_memory_variants_do_not_change_results(env, monkeypatch, request):
    """The cost-class order, the size of the arena pools and where the near data lives only change when and where a read is
    processed, never its result."""
    if "MAPAD_HEAVY" in env and _rerun_in_heavy_build(request):
        return
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = synth.genome(300_000, seed=21)
    seqs, quals, offsets = synth.reads(g, 2500, 50, seed=5, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("heavy", ["0", "1"])
@pytest.mark.parametrize("class_counts", ["512,512,512,512,512,512,512,512,512", "8,2", "8,2+sets"])
def test_small_arena_growth_last_pass_and_limits(monkeypatch, class_counts, heavy, request):
    """Reads that outgrow their arena migrate into the size-class pools (owner-word acquire / release); when a pool is dry the
    read is re-run by the full-limit pass.  Tiny STACK/EDIT_TREE limits exercise the overflow recovery of mapping.rs:1358-1380."""
    if heavy == "1" and _rerun_in_heavy_build(request):
        return
    g = synth.genome(100_000, seed=5)
    seqs, quals, offsets = synth.reads(g, 400, 50, seed=11)
    reads, qs = split_reads(seqs, quals, offsets)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    monkeypatch.setenv("MAPAD_HEAVY", heavy)  # 1: a read that outgrows its base arena is continued by a wavefront of its own (heavy_kernel.hpp)
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")  # classes: 64, 128, 256, ..., 16384 nodes, full limits
    # "8,2": reads wait for the few arenas; those that need > 128 nodes are re-run with the full limits.  "+sets" (round 5, the default): a read that finds the
    # classes dry or too small takes an idle SET of base arenas as one arena (GrowPools: entry kClasses) — here 16 x 32-node slots' worth of HBM hold a few thousand
    # nodes, and no read is left for the full-limit pass.
    sets = class_counts.endswith("+sets")
    class_counts = class_counts.split("+")[0]
    monkeypatch.setenv("MAPAD_SET_ARENAS", "1" if sets else "0")
    monkeypatch.setenv("MAPAD_CLASS_COUNTS", class_counts)
    rp = resolve_params(NO_DAMAGE)
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    assert res.n_second_pass > 0  # migrations
    sets_used = sets and heavy == "0"  # (set arenas are off when reads are suspended to heavy wavefronts: their hit staging travels in the grown arena)
    if sets_used:
        assert res.n_third_pass <= 3  # (a set holds a few thousand nodes here: at most a read or two of this batch need more)
    else:
        assert (res.n_third_pass > 0) == (class_counts == "8,2")  # reads re-run with the full limits (by a host thread, host tail on: tests/test_gpu_tail.py) only when no class can hold them
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    for limits in ({"stack_limit": 40, "edit_tree_limit": 100000}, {"stack_limit": 100000, "edit_tree_limit": 120},
                   {"stack_limit": 40, "edit_tree_limit": 100000, "stack_limit_abort": 1}):
        rp2 = dict(rp, **limits)
        res = _gpu_map(pidx, mapad_amd.make_params(rp2), seqs, quals, offsets)
        ores = oidx.map_batch(ob.make_params(rp2), reads, qs, n_threads=8, keep_d=True)
        assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("heavy", ["0", "1"])
def test_arena_handoff_stress(monkeypatch, heavy, request):
    """Partitioned (per-XCD) pools with tiny base arenas and few grown arenas per XCD: arenas change owners constantly, under uneven
    load, and every word of every result is checked (a late store of an old owner landing in a new owner's arena would show here)."""
    if heavy == "1" and _rerun_in_heavy_build(request):
        return
    monkeypatch.setenv("MAPAD_HEAVY", heavy)
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    monkeypatch.setenv("MAPAD_CLASS_COUNTS", "128,128,128,64,64,64,64,64,64,16")  # >= 64: split per XCD, 8-16 arenas each
    g = synth.genome(400_000, seed=77)
    seqs, quals, offsets = synth.reads(g, 20000, 50, seed=13, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    for _ in range(2):
        res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
        assert res.n_second_pass > (20000 if heavy == "0" else 5000)  # arena migrations: more than one per read (heavy: reads that find the pools dry go to the full-limit stage instead)
        assert_same_as_oracle(ores, res, offsets)


@pytest.mark.parametrize("depth,tiny", [(2, False), (3, True), (6, "sets")])
def test_batches_in_flight_do_not_change_results(depth, tiny, monkeypatch):
    """mapad_ctx_set_pipeline_depth: batch k + 1 is submitted while batch k is still running (own streams and buffers, shared base arenas and
    size-class pools); every batch must come out exactly as it does alone.  "sets": one wavefront per CU's worth of base-arena sets, so the
    wavefronts of six launches in flight outnumber the sets and wait for one another's exit."""
    import ctypes as C
    mapad_amd.lib()
    # the HIP runtime the library itself has loaded (the very file: a second copy of the runtime would not see the device), for device-resident inputs without torch
    paths = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    hip = C.CDLL(paths[0] if paths else "libamdhip64.so")

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 8)))
        if rc != 0:
            free_b, total_b = C.c_size_t(), C.c_size_t()
            hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b))
            raise AssertionError(f"hipMalloc of {a.nbytes} bytes failed with {rc}; free {free_b.value >> 20} MiB of {total_b.value >> 20} MiB")
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0  # hipMemcpyHostToDevice
        return p.value

    if tiny == "sets":
        monkeypatch.setenv("MAPAD_TIER0_WAVES_PER_CU", "1")
        monkeypatch.setenv("GPU_MAX_HW_QUEUES", "12")  # read at the first HIP call of the process; harmless if that is over
    elif tiny:  # reads migrate constantly and the concurrent launches compete for the same few arenas
        monkeypatch.setenv("MAPAD_TIER0_NODES", "64")
        monkeypatch.setenv("MAPAD_CLASS_COUNTS", "256,128,64,64,64,64,64,16,16,16")
    g = synth.genome(300_000, seed=31)
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)], device=0)
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    batches = [synth.reads(g, n, 50, seed=40 + i, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)) for i, n in enumerate((6000, 1500, 9000, 3000, 4500, 700))]
    want = []
    for seqs, quals, offsets in batches:
        reads, qs = split_reads(seqs, quals, offsets)
        want.append(oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True))
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    on_dev = [tuple(to_device(a) for a in b) for b in batches]
    ctx.set_pipeline_depth(depth)
    ctx.prepare_lengths([50])
    for lo in range(0, len(batches), depth):
        group = list(range(lo, min(lo + depth, len(batches))))
        for i in group:  # submitted back to back, nothing fetched in between
            ctx.map_batch_device(on_dev[i][0], on_dev[i][1], on_dev[i][2], len(batches[i][2]) - 1, 50)
        for i in group:
            ctx.select_batch(group[-1] - i)
            assert_same_as_oracle(want[i], ctx.fetch(), batches[i][2])
    hist = ctx.kernel_history()
    assert hist.shape == (len(batches), 4) and (np.diff(hist, axis=1) >= 0).all()
    ctx.close()
    for b in on_dev:
        for p in b:
            hip.hipFree(C.c_void_p(p))


@pytest.mark.parametrize("prm", [DAMAGE, CONTINUOUS], ids=["discrete", "continuous"])
def test_heavy_wavefronts_on_mixed_lengths(monkeypatch, prm, request):
    """Reads of 35-100 bp with indels, small base arenas, MAPAD_HEAVY=1: most reads are suspended by their quad and finished by a wavefront of their
    own (deep sifts through the speculative block, pushes through the ancestor table, the general single-lane step at the last position)."""
    if _rerun_in_heavy_build(request):
        return
    monkeypatch.setenv("MAPAD_HEAVY", "1")
    monkeypatch.setenv("MAPAD_TIER0_NODES", "128")
    g = synth.genome(300_000, seed=17)
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=23, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    rp = resolve_params(prm)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    assert res.n_second_pass > 100
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)


def test_integration_expectation_on_gpu(monkeypatch):
    """tests/integration_tests.rs: FASTA -> index -> 17 reads -> record fields, all through the C ABI with the GPU search."""
    k = load("integration")
    monkeypatch.delenv("MAPAD_INDEX_FIXED_REPLACEMENT", raising=False)  # StdRng(1234) itself must draw the base the reference's expectation implies
    pidx = mapad_amd.Index.build([(c["name"], c["seq"].encode()) for c in k["contigs"]], seed=1234)
    rp = resolve_params(k["params"])
    params = mapad_amd.make_params(rp)
    reads, quals = integration_reads(k)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    qs = np.concatenate(quals)
    ctx = mapad_amd.Context(pidx, params, 0)
    res = ctx.map_batch(seqs, qs, offsets)
    recs = mapad_amd.hits_to_records(pidx, params, res, seqs, qs, offsets, in_flags=[r["flags"] for r in k["reads"]])
    ctx.close()
    from test_host_logic import check_integration_records
    check_integration_records(k, recs)


def test_empty_and_edge_batches():
    g = synth.genome(50_000, seed=3)
    pidx = mapad_amd.Index.build([("chr1", g)])
    params = mapad_amd.make_params(resolve_params(NO_DAMAGE))
    ctx = mapad_amd.Context(pidx, params, 0)
    r = ctx.map_batch(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert r.n_reads == 0 and r.n_hits == 0
    # ragged: a 1-base read, a read of N's, a read shorter than the 17 bp minimum of the Discrete bound
    long_read = g[5000:5300].copy()
    long_read[[10, 150, 290]] = ord("A")
    reads = [b"A", b"NNNNNNNNNNNNNNNNNNNNNNNNN", g[1000:1016].tobytes(), g[2000:2050].tobytes(), long_read.tobytes(), g[7000:8000].tobytes(),
             synth.revcomp(g[9000:9130]).tobytes()]
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    quals = np.full(len(seqs), 40, np.uint8)
    res = ctx.map_batch(seqs, quals, offsets)
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    ores = oidx.map_batch(ob.make_params(resolve_params(NO_DAMAGE)), reads, [quals[int(offsets[i]):int(offsets[i + 1])] for i in range(len(reads))], keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    ctx.close()


def test_long_reads_up_to_the_reference_limit():
    """Reads of 2 kb and 10 kb (the reference maps up to i16::MAX bases, record.rs:144-150) beside short ones: D-array chains and the read's position
    data in HBM instead of LDS, 15-bit start / length fields in the frame, score tables per length — against the oracle, bit for bit."""
    g = synth.genome(400_000, seed=41)
    rng = np.random.default_rng(4)

    def mutate(a, n_sub, indel=False):
        a = a.copy()
        for p in rng.choice(len(a) - 200, n_sub, replace=False) + 100:
            a[p] = ord("ACGT"[("ACGT".index(chr(a[p])) + 1) % 4])
        if indel:
            a = np.concatenate([a[:len(a) // 2], a[len(a) // 2 + 1:]])  # one base deleted from the read
        return a
    reads = [g[50_000:52_000].tobytes(), mutate(g[100_000:102_000], 3).tobytes(), synth.revcomp(mutate(g[150_000:152_000], 2)).tobytes(),
             mutate(g[200_000:210_000], 4).tobytes(), mutate(g[300_000:302_001], 1, indel=True).tobytes(), g[7_000:7_050].tobytes(), g[9_000:9_300].tobytes()]
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    quals = np.full(len(seqs), 40, np.uint8)
    rp = resolve_params(NO_DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    res = _gpu_map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets)
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    ores = oidx.map_batch(ob.make_params(rp), reads, [quals[int(offsets[i]):int(offsets[i + 1])] for i in range(len(reads))], keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    assert (np.diff(res.hit_begin.astype(np.int64)) >= 1).sum() >= 5  # the long reads map
