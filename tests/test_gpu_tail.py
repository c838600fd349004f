"""GPU tests of the host tail (csrc/host_tail.hpp): reads that pass a pop budget on the GPU are handed — by the running kernel, through page-locked
records — to the library's host threads, which map them from scratch with the kernel's own search step; their results join the batch before the
order-preserving collect.  Where a read was finished must not show anywhere: every result is compared bit for bit (hits, BinaryHeap order, score bits,
edit tracks, D arrays, the six event counters) with the CPU oracle and with the same batch mapped with the tail switched off.
Reference behaviour at stake: the overflow recovery of src/map/mapping.rs:1358-1380 (the reads that get here at the real limits are those that hit it)."""
import numpy as np
import pytest

import mapad_amd
from mapad_amd import synth
from oracle import binding as ob

from kat_util import resolve_params
from parity_util import CONTINUOUS, DAMAGE, NO_DAMAGE, assert_same_as_oracle, split_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hand_over_whatever_the_backlog(monkeypatch):
    """These tests are about the hand-over itself (which reads leave, what the host makes of them): every read past the budget leaves, whatever the host has waiting —
    round 5's behaviour.  The gate of round 6 (a read past the budget stays on the GPU while the host's backlog is long) has its own test below."""
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG_BUDGET", "4294967295")


def _map(index, params, seqs, quals, offsets, tail_pops):
    ctx = mapad_amd.Context(index, params, 0)
    try:
        ctx.set_tail_pops(tail_pops)
        res = ctx.map_batch(seqs, quals, offsets)
        return res, ctx.tail_info()
    finally:
        ctx.close()


def _same(a, b):
    assert np.array_equal(a.hit_begin, b.hit_begin) and np.array_equal(a.hits_arr, b.hits_arr) and np.array_equal(a.ops, b.ops)
    assert np.array_equal(a.status, b.status) and np.array_equal(a.counters, b.counters)


@pytest.mark.parametrize("name,prm,kw,n", [
    ("no_damage", NO_DAMAGE, dict(qual=40), 3000),
    ("damage_mixed_len_indels", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 800),
    ("continuous_bound", CONTINUOUS, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 2000),
])
def test_reads_finished_on_the_host_equal_the_oracle(name, prm, kw, n):
    g = synth.genome(300_000, seed=77)
    seqs, quals, offsets = synth.reads(g, n, 50, seed=3 + len(name), **kw)
    rp = resolve_params(prm)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    params = mapad_amd.make_params(rp)
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=48)  # a budget most reads with a mismatch exceed
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    pops = ores.counters[:, 3]
    assert info["reads"] == int((pops > 48).sum()) > n // 20  # exactly the reads beyond the budget went to the host, and they are many
    assert info["host_pops"] == int(pops[pops > 48].sum()) and info["gpu_pops"] >= 48 * info["reads"]
    assert (res.status & 16).sum() == 0  # no read is left marked "handed over"
    off, info_off = _map(pidx, params, seqs, quals, offsets, tail_pops=0)
    assert info_off["reads"] == 0
    _same(res, off)


def test_reads_past_the_budget_stay_on_the_gpu_while_the_host_is_busy(monkeypatch):
    """Round 6: every hand-over trigger looks at the host's backlog (the word the launch's dispatcher keeps current).  With no room at all
    (MAPAD_TAIL_BACKLOG_BUDGET=0) every read past the budget is refused, goes on on the GPU, asks again every 16 384 pops and finishes there; with the
    default (8 waiting reads per worker) any number between none and all of them leave, as the host's pace has it.  Nothing of that shows in the results."""
    g = synth.genome(300_000, seed=77)
    seqs, quals, offsets = synth.reads(g, 3000, 50, seed=12, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    params = mapad_amd.make_params(rp)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    past = int((ores.counters[:, 3] > 64).sum())
    every, info_every = _map(pidx, params, seqs, quals, offsets, tail_pops=64)
    assert info_every["reads"] == past > 500
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG_BUDGET", "0")
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=64)
    assert info["reads"] == 0, info
    assert_same_as_oracle(ores, res, offsets)
    _same(res, every)
    assert (res.status & 16).sum() == 0
    monkeypatch.delenv("MAPAD_TAIL_BACKLOG_BUDGET")
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=64)
    assert 0 <= info["reads"] <= past
    _same(res, every)


def test_reads_below_the_budget_leave_while_a_worker_is_idle(monkeypatch):
    """Round 6: from MAPAD_TAIL_POPS_IDLE pops on a read leaves although it is short of the budget — while fewer reads than MAPAD_TAIL_BACKLOG_IDLE (default: the workers)
    wait or run on the host, i.e. while a worker is idle; the ring slot is claimed under that test (the kernel counts the records the dispatcher has not picked up yet
    itself), so the hundreds of reads that ask in the same microsecond cannot overrun it.  With no room (0) nobody leaves.  Nothing of that shows in the results."""
    g = synth.genome(300_000, seed=77)
    seqs, quals, offsets = synth.reads(g, 3000, 50, seed=12, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    params = mapad_amd.make_params(rp)
    off, info_off = _map(pidx, params, seqs, quals, offsets, tail_pops=0)
    past = int((off.counters["n_pop"] > 64).sum())
    assert info_off["reads"] == 0 and past > 500
    monkeypatch.setenv("MAPAD_TAIL_POPS_IDLE", "64")
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=1 << 20)  # a budget no read of this batch reaches
    assert 0 < info["reads"] == info["reads_idle_tier"] <= past, info
    assert info["gpu_pops"] >= 64 * info["reads"]
    _same(res, off)
    assert (res.status & 16).sum() == 0
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG_IDLE", "0")
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=1 << 20)
    assert info["reads"] == 0, info
    _same(res, off)
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG_IDLE", "4294967295")  # whatever the backlog: every read past 64 pops
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=1 << 20)
    assert info["reads"] == info["reads_idle_tier"] == past and info["gpu_pops"] == 64 * past, (info, past)  # each left at its first ask
    _same(res, off)


def test_limit_recovery_and_abort_on_the_host():
    """Tiny STACK_LIMIT / EDIT_TREE_LIMIT: the reads that reach the host run into the overflow recovery (pop_min eviction, slab key reuse) or the abort there."""
    g = synth.genome(100_000, seed=5)
    seqs, quals, offsets = synth.reads(g, 600, 50, seed=11)
    reads, qs = split_reads(seqs, quals, offsets)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    for limits in ({"stack_limit": 40, "edit_tree_limit": 100000}, {"stack_limit": 100000, "edit_tree_limit": 120}, {"stack_limit": 40, "edit_tree_limit": 100000, "stack_limit_abort": 1}):
        rp = dict(resolve_params(NO_DAMAGE), **limits)
        res, info = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=30)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
        assert ores.counters[:, 3].max() > 40 and info["reads"] > 20
        assert_same_as_oracle(ores, res, offsets)
        if limits.get("stack_limit_abort"):
            assert (res.status == 2).any()


def test_full_ring_leaves_reads_on_the_gpu(monkeypatch):
    """More reads pass the budget than the ring has records: the rest stays on the GPU; nothing changes."""
    monkeypatch.setenv("MAPAD_TAIL_RING", "16")
    g = synth.genome(200_000, seed=9)
    seqs, quals, offsets = synth.reads(g, 2000, 50, seed=2, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res, info = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=40)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert info["reads"] == 16 < int((ores.counters[:, 3] > 40).sum())
    assert_same_as_oracle(ores, res, offsets)


def test_hit_pools_too_small_for_the_hosts_results_are_grown_and_the_batch_rerun(monkeypatch):
    """The host's hits are appended to the batch's device pools; when they do not fit (here: pools of 64 hits), the collect reports it like a kernel-side overflow and
    mapad_map_batch re-runs the batch — GPU stages and host tail — with pools sized for everything."""
    monkeypatch.setenv("MAPAD_HIT_POOL", "64")
    g = synth.genome(200_000, seed=13)
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=4, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res, info = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=40)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert res.n_hits > 64 and info["reads"] == int((ores.counters[:, 3] > 40).sum()) > 50
    assert_same_as_oracle(ores, res, offsets)


def test_tail_with_batches_in_flight_and_an_uncollected_batch():
    """Three batches in flight, each with reads on the host; the collect of a batch waits for ITS host reads only.  A batch whose slot is reused before anybody
    collected it is dropped together with its host reads."""
    g = synth.genome(300_000, seed=31)
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    batches = [synth.reads(g, 1500, 50, seed=40 + i, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)) for i in range(5)]
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    try:
        ctx.set_pipeline_depth(3)
        ctx.set_tail_pops(64)
        got = {}
        for i, b in enumerate(batches):
            ctx.submit_batch(*b)
            if i >= 2:
                ctx.select_batch(2)
                got[i - 2] = (ctx.fetch(), ctx.tail_info())
                ctx.select_batch(0)
        for age, i in ((1, 3), (0, 4)):
            ctx.select_batch(age)
            got[i] = (ctx.fetch(), ctx.tail_info())
        # two more batches are submitted and never collected; then the slots are reused
        for b in batches[:4]:
            ctx.submit_batch(*b)
        ctx.select_batch(0)
        again = ctx.fetch()
    finally:
        ctx.close()
    for i, b in enumerate(batches):
        reads, qs = split_reads(*b)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
        assert_same_as_oracle(ores, got[i][0], b[2])
        assert got[i][1]["reads"] == int((ores.counters[:, 3] > 64).sum()) > 20
    _same(again, got[3][0])


def test_a_dry_arena_class_sends_reads_to_the_host(monkeypatch):
    """Round 5: a read that needs a grown arena of a scarce class while every one is taken asks for the host instead of queuing (DeviceGrow::acquire) — here every
    class counts as scarce, each has two arenas, and the pop budget is out of reach, so whatever goes to the host goes for that reason (or because no class holds it)."""
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    monkeypatch.setenv("MAPAD_CLASS_COUNTS", "2,2,2,2,2,2,2,2,2,2")
    monkeypatch.setenv("MAPAD_SET_ARENAS", "0")  # (idle sets of base arenas would take the reads the dry classes turn away: tests/test_gpu_parity.py covers those)
    monkeypatch.setenv("MAPAD_TAIL_MIN_CLASS", "0")
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG", "100000")  # the host takes whatever comes
    g = synth.genome(200_000, seed=21)
    seqs, quals, offsets = synth.reads(g, 4000, 50, seed=6, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    res, info = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=1 << 30)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    assert info["reads_dry_class"] > 50 and info["reads"] == info["reads_dry_class"] + info["reads_full_limit"]
    assert (res.status & 16).sum() == 0
    # a backlog limit of zero: the host is "busy" from the start, nobody is handed over for a dry class; the reads queue on the GPU as in round 4
    monkeypatch.setenv("MAPAD_TAIL_BACKLOG", "0")
    res0, info0 = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=1 << 30)
    assert info0["reads_dry_class"] == 0
    _same(res, res0)


def test_reads_no_growable_arena_holds_go_to_the_host_instead_of_the_full_limit_stage(monkeypatch):
    """The classes end at 128 nodes: a read that needs more would be re-run by the full-limit stage (heavy_kernel, 336 MB arenas); with the host tail on it goes to a
    host thread.  n_third_pass counts such reads wherever they were finished; with the tail off the GPU's last stage takes them — same results."""
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    monkeypatch.setenv("MAPAD_CLASS_COUNTS", "512,512")
    monkeypatch.setenv("MAPAD_SET_ARENAS", "0")
    monkeypatch.setenv("MAPAD_TAIL_MIN_CLASS", "10")  # no hand-over for dry classes: only the reads no class can hold
    g = synth.genome(100_000, seed=5)
    seqs, quals, offsets = synth.reads(g, 1500, 50, seed=11)
    rp = resolve_params(NO_DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    res, info = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=1 << 30)
    assert_same_as_oracle(ores, res, offsets)
    assert info["reads_full_limit"] > 20 and info["reads"] == info["reads_full_limit"] == res.n_third_pass
    off, info_off = _map(pidx, mapad_amd.make_params(rp), seqs, quals, offsets, tail_pops=0)
    assert info_off["reads"] == 0 and off.n_third_pass == res.n_third_pass
    _same(res, off)


def test_hand_overs_reach_the_host_while_the_launch_is_running():
    """The ring is host-coherent page-locked memory and every word of a record is written through (hand_to_host): the dispatcher thread sees records — and the
    workers start on them — during the launch, not when it ends (with ordinary page-locked memory that was luck)."""
    g = synth.genome(2_000_000, seed=41)
    seqs, quals, offsets = synth.reads(g, 400_000, 50, seed=8, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    pidx = mapad_amd.Index.build([("chr1", g)], device=0)
    res, info = _map(pidx, mapad_amd.make_params(resolve_params(DAMAGE)), seqs, quals, offsets, tail_pops=4096)
    assert info["reads"] > 100, info
    assert info["seen_live"] > info["reads"] // 2, info  # the launch runs for a good 100 ms; the first hand-overs come within the first few
    assert (res.status & 16).sum() == 0


@pytest.mark.parametrize("name,prm,kw,n", [
    ("damage", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 3000),
    ("mixed_len_indels", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 800),
    ("continuous_bound", CONTINUOUS, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 2000),
])
def test_reads_handed_over_with_their_search_are_continued_on_the_host(name, prm, kw, n, monkeypatch):
    """Round 5: a read that sits in a grown arena when it is handed over leaves the arena to the host (heap top written out of LDS, SearchState in the record); a
    worker copies heap and nodes over PCIe, releases the arena and goes on where the GPU stopped (host_tail.hpp: TailState).  32-node base arenas make every read
    of consequence grow before the 150-pop budget; the results — event counters included, which now add GPU pops and host pops of one read — equal the oracle's
    and those of the same batch mapped from scratch on the host (MAPAD_TAIL_CONTINUE=0) and entirely on the GPU."""
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    g = synth.genome(300_000, seed=77)
    seqs, quals, offsets = synth.reads(g, n, 50, seed=3 + len(name), **kw)
    rp = resolve_params(prm)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    params = mapad_amd.make_params(rp)
    res, info = _map(pidx, params, seqs, quals, offsets, tail_pops=150)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    pops = ores.counters[:, 3]
    assert info["reads"] == int((pops > 150).sum()) > n // 50
    assert info["continued"] > info["reads"] // 2 and info["handed_over_with_state"] >= info["continued"], info  # (reads with a hit already found go from scratch)
    assert info["host_pops"] < int(pops[pops > 150].sum()) - 100 * info["continued"]  # the host did not repeat the GPU's pops of the continued reads
    assert (res.status & 16).sum() == 0
    monkeypatch.setenv("MAPAD_TAIL_CONTINUE", "0")
    scratch, info0 = _map(pidx, params, seqs, quals, offsets, tail_pops=150)
    assert info0["continued"] == 0 and info0["reads"] == info["reads"]
    _same(res, scratch)
    off, _ = _map(pidx, params, seqs, quals, offsets, tail_pops=0)
    _same(res, off)


def test_arenas_of_reads_nobody_came_for_are_released(monkeypatch):
    """Batches with reads handed over WITH their state are submitted and never collected; their slots are reused.  The grown arenas those reads still hold must go
    back to the pools (drop_tail): with two arenas per class a leak would leave later batches without any — they would still finish (dry classes send reads to
    the host) but with every class dry from the start, which the migration count shows."""
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    monkeypatch.setenv("MAPAD_CLASS_COUNTS", "8,8,8,8,8,8,8,8,8,8")
    monkeypatch.setenv("MAPAD_SET_ARENAS", "0")
    g = synth.genome(200_000, seed=31)
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    b = synth.reads(g, 1500, 50, seed=44, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    try:
        ctx.set_pipeline_depth(2)
        ctx.set_tail_pops(150)
        for _ in range(12):  # each submission reuses a slot whose batch nobody collected
            ctx.submit_batch(*b)
        ctx.select_batch(0)
        res = ctx.fetch()
        info = ctx.tail_info()
    finally:
        ctx.close()
    reads, qs = split_reads(*b)
    ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, b[2])
    assert info["handed_over_with_state"] > 0 and res.n_second_pass > 500  # reads still find grown arenas: nothing leaked


def test_an_arena_relayout_waits_for_the_host_tails_of_uncollected_batches(monkeypatch):
    """a batch with more reads than the arenas were laid out for makes the context free and re-allocate its pools.  Reads of ANOTHER slot's uncollected batch
    that were handed over with their state still own grown arenas in those pools until a host worker has copied them (and the workers read the pool descriptors
    unlocked): the re-layout first finishes and merges the host tails of every slot (finish_tails), and the earlier batch's results — collected afterwards — are
    still the oracle's."""
    monkeypatch.setenv("MAPAD_TIER0_NODES", "32")
    g = synth.genome(300_000, seed=77)
    rp = resolve_params(DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    small = synth.reads(g, 1500, 50, seed=61, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    big = synth.reads(g, 12_000, 50, seed=62, qual=40, subst_rate=0.0, exo_frac=0.0)  # (exact reads: a few dozen pops each, nothing for the host — the re-layout is what this batch is for)
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    try:
        ctx.set_pipeline_depth(2)
        ctx.set_tail_pops(150)
        ctx.submit_batch(*small)   # reads go to the host with their state; nobody collects yet
        ctx.submit_batch(*big)     # more reads than the pools were sized for: the arenas are laid out anew
        ctx.select_batch(1)
        first, info = ctx.fetch(), ctx.tail_info()
        ctx.select_batch(0)
        second = ctx.fetch()
    finally:
        ctx.close()
    assert info["handed_over_with_state"] > 50, info
    for b, res in ((small, first), (big, second)):
        reads, qs = split_reads(*b)
        ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
        assert_same_as_oracle(ores, res, b[2])


def test_hand_over_ring_is_not_fooled_by_records_of_earlier_launches():
    """the ring's `ready` words used to be cleared by count, and a small batch between two large ones left records of the first large batch marked ready when
    the third ran — the dispatcher would have taken a stale record for a new one.  A record is ready when its word holds THIS launch's number now.  Same context, same
    slot: 2 000 hand-overs, then a batch of 100 reads (ring of 100 records), then another large batch — every batch equals the oracle and hands over exactly its own reads."""
    g = synth.genome(300_000, seed=77)
    rp = resolve_params(NO_DAMAGE)
    pidx = mapad_amd.Index.build([("chr1", g)])
    oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    try:
        ctx.set_tail_pops(48)
        for n, seed in ((3000, 71), (100, 72), (3000, 73), (100, 74), (2500, 75)):
            b = synth.reads(g, n, 50, seed=seed, qual=40)
            res = ctx.map_batch(*b)
            info = ctx.tail_info()
            reads, qs = split_reads(*b)
            ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
            assert_same_as_oracle(ores, res, b[2])
            assert info["reads"] == int((ores.counters[:, 3] > 48).sum()) and (res.status & 16).sum() == 0
    finally:
        ctx.close()
