"""The N > 1 path on CPU: 2 processes over gloo shard a chunk, map their slices (with the host build of the kernels' logic
standing in for the GPU) and gather the hit records on rank 0; the merged result must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _map_slice(lo, hi):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_util
    import mapad_amd
    from mapad_amd import presets, synth
    g = synth.genome(60_000, seed=31)
    idx = mapad_amd.Index.build([("chr1", g)])
    seqs, quals, offsets = synth.reads(g, 101, 50, seed=32, len_range=(35, 70))
    sub_off = offsets[lo:hi + 1] - offsets[lo]
    s, e = int(offsets[lo]), int(offsets[hi])
    res = emu_util.map_batch(idx, mapad_amd.make_params(presets.resolve(presets.DAMAGE)), seqs[s:e], quals[s:e], sub_off)
    hits = res.hits_arr.view(np.int32).reshape(-1, 10).copy() if res.n_hits else np.zeros((0, 10), np.int32)
    return res.hit_begin.copy(), hits, res.ops.view(np.int32).copy()


def _as_completion_ordered_pools(hit_begin, hits, ops, seed):
    """What the search kernel leaves behind: reads finish in any order, each appending its hits / ops at the bump cursors."""
    rng = np.random.default_rng(seed)
    n = len(hit_begin) - 1
    count = np.diff(hit_begin.astype(np.int64)).astype(np.int32)
    first = np.zeros(n, np.int32)
    pool, ops_pool, ops_base = [], [], 0
    for r in rng.permutation(n):
        first[r] = len(pool)
        for h in hits[int(hit_begin[r]):int(hit_begin[r + 1])]:
            h = h.copy()
            ops_pool.append(ops[int(h[8]):int(h[8]) + int(h[7])])
            h[8] = ops_base
            ops_base += int(h[7])
            pool.append(h)
    return count, first, (np.stack(pool) if pool else np.zeros((0, 10), np.int32)), (np.concatenate(ops_pool) if ops_pool else np.zeros(0, np.int32))


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import hashlib
    from mapad_amd.distributed import collect_in_read_order, gather_hit_records, merge_gathered, shard_bounds
    lo, hi = shard_bounds(101, world, rank)
    own = _map_slice(lo, hi)
    own_digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in own)).hexdigest()
    # the rank's raw result is completion-ordered; the collect (device kernels on the GPU) lays it out in read order before the gather
    hit_begin, hits, ops = collect_in_read_order(*_as_completion_ordered_pools(*own, seed=100 + rank))
    meta = dist.new_group(backend="gloo")  # the size exchange on its own CPU group, as bench.py does beside RCCL
    parts = gather_hit_records(torch.from_numpy(hit_begin.view(np.int32)), torch.from_numpy(hits.reshape(-1)), torch.from_numpy(ops), rank, world, meta_group=meta)
    digests = [None] * world
    dist.all_gather_object(digests, own_digest)
    if rank == 0:
        hb, h, o, per_rank = merge_gathered(parts)
        assert per_rank == digests  # every shard arrived as the rank's own result
        np.savez(out_path, hit_begin=hb, hits=h, ops=o)
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from mapad_amd.distributed import shard_bounds
    # slices tile the chunk in rank order, also for ragged splits
    for n, w in [(101, 2), (7, 8), (0, 4), (250000, 8)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    out = str(tmp_path / "merged.npz")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_util
    emu_util.lib()  # built once here, not by both ranks at the same time
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    hb, hits, ops = _map_slice(0, 101)
    assert np.array_equal(got["hit_begin"], hb)
    assert np.array_equal(got["hits"], hits) and np.array_equal(got["ops"], ops)  # incl. rebased ops offsets
    assert hb[-1] > 50


def _fake_records(n, seed, p_unmapped=0.2):
    """n record structs in the layout of csrc/text_core.hpp: DevRecord (22 words) with their text and pair pools, as a rank's mapad_records_device leaves them."""
    rng = np.random.default_rng(seed)
    recs = np.zeros((n, 22), np.int32)
    text, pairs = bytearray(), []
    for i in range(n):
        if rng.random() < p_unmapped:
            recs[i, 0:2] = -1; recs[i, 2] = -1  # unmapped: pos -1, tid -1, offsets 0
            continue
        cig, md, xa = b"%dM" % rng.integers(30, 100), b"%d" % rng.integers(30, 100), (b"chr1,+%d,50M,50,0,1,-1.50;" % rng.integers(1, 10 ** 6)) if rng.random() < 0.3 else b""
        recs[i, 0] = rng.integers(0, 1 << 30); recs[i, 2] = 0; recs[i, 3] = 1
        recs[i, 12] = len(text); recs[i, 13:16] = (len(cig), len(md), len(xa))
        text += cig + md + xa
        k = int(rng.integers(0, 3))
        recs[i, 18] = len(pairs) // 2; recs[i, 19] = k
        pairs += [float(x) for x in rng.normal(size=2 * k)]
    text += b"\0" * (-len(text) % 4)
    return recs, np.frombuffer(bytes(text), np.uint8).view(np.int32).copy(), np.array(pairs, np.float32).view(np.int32).copy()


def _text_of(recs, text, i):
    o, c, m, x = (int(v) for v in recs[i, 12:16])
    return bytes(text[o:o + c + m + x])


def _records_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from mapad_amd.distributed import gather_hit_records, merge_gathered_records, records_digest
    own = _fake_records(300 + 17 * rank, seed=5 + rank)
    own_digest = records_digest(*own)
    meta = dist.new_group(backend="gloo")
    parts = gather_hit_records(torch.from_numpy(own[0].reshape(-1)), torch.from_numpy(own[1]), torch.from_numpy(own[2]), rank, world, meta_group=meta)
    digests = [None] * world
    dist.all_gather_object(digests, own_digest)
    if rank == 0:
        recs, text, pairs, per_rank = merge_gathered_records(parts)
        assert per_rank == digests
        np.savez(out_path, recs=recs, text=text, pairs=pairs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_of_compact_records(tmp_path):
    """What bench.py sends since round 4: per read an 88-byte record plus its text and MAPQ pairs (<= 128 bytes per read).  Rank 0's merge must
    rebase every shard's text and pair offsets so that each read still finds its own CIGAR / MD / XA bytes and pairs."""
    out = str(tmp_path / "records.npz")
    mp.spawn(_records_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    shards = [_fake_records(300, seed=5), _fake_records(317, seed=6)]
    assert got["recs"].shape == (617, 22)
    base = 0
    for recs, text, pairs in shards:
        tb, pf = text.view(np.uint8), pairs.view(np.float32)
        for i in range(recs.shape[0]):
            g = got["recs"][base + i]
            assert np.array_equal(np.delete(g, [12, 18]), np.delete(recs[i], [12, 18]))
            if recs[i, 3]:
                assert _text_of(got["recs"], got["text"], base + i) == _text_of(recs, tb, i)
                k = int(recs[i, 19])
                assert np.array_equal(got["pairs"][2 * int(g[18]):2 * int(g[18]) + 2 * k], pf[2 * int(recs[i, 18]):2 * int(recs[i, 18]) + 2 * k])
        base += recs.shape[0]
    per_read = (got["recs"].nbytes + got["text"].nbytes + got["pairs"].nbytes) / 617
    assert per_read <= 128
    # the digest follows the records' content, not the pools' layout: the same reads with their text and pairs laid out in another order hash alike
    from mapad_amd.distributed import records_digest
    recs, text, pairs = shards[0]
    tb, pf = text.view(np.uint8), pairs.view(np.float32)
    r2, t2, p2 = recs.copy(), bytearray(), []
    for i in reversed(range(recs.shape[0])):
        if not recs[i, 3]:
            continue
        piece = _text_of(recs, tb, i)
        r2[i, 12] = len(t2); t2 += piece
        k = int(recs[i, 19])
        r2[i, 18] = len(p2) // 2
        p2 += list(pf[2 * int(recs[i, 18]):2 * int(recs[i, 18]) + 2 * k])
    t2 += b"\0" * (-len(t2) % 4)
    assert records_digest(r2, np.frombuffer(bytes(t2), np.uint8).view(np.int32), np.array(p2, np.float32).view(np.int32)) == records_digest(recs, text, pairs)
    r2[5, 0] ^= 1
    assert records_digest(r2, np.frombuffer(bytes(t2), np.uint8).view(np.int32), np.array(p2, np.float32).view(np.int32)) != records_digest(recs, text, pairs)


# world size 8 — the node north_star names — with the shards an 8-way split really produces: empty shards (a chunk of fewer reads than ranks: shard_bounds(5, 8, r)), a
# one-read shard, a shard none of whose reads mapped (no text, no pairs: its two pool buffers are EMPTY tensors, which are never sent), and ordinary ones.
_SHARDS8 = {"ragged": [(300, 0.2), (0, 0.2), (5, 0.2), (317, 0.2), (1, 0.0), (0, 0.2), (40, 1.0), (64, 0.2)],
            "fewer_reads_than_ranks": [(1 if r < 5 else 0, 0.0) for r in range(8)]}  # = shard_bounds(5, 8, r), checked by the workers


def _records_worker8(rank, world, port, out_path, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from mapad_amd.distributed import gather_hit_records, merge_gathered_records, records_digest, shard_bounds
    n, p_un = _SHARDS8[case][rank]
    if case == "fewer_reads_than_ranks":
        lo, hi = shard_bounds(5, world, rank)
        assert hi - lo == n
    own = _fake_records(n, seed=50 + rank, p_unmapped=p_un)
    own_digest = records_digest(*own)
    meta = dist.new_group(backend="gloo")
    parts = gather_hit_records(torch.from_numpy(own[0].reshape(-1)), torch.from_numpy(own[1]), torch.from_numpy(own[2]), rank, world, meta_group=meta)
    digests = [None] * world
    dist.all_gather_object(digests, own_digest)
    if rank == 0:
        assert len(parts) == world
        recs, text, pairs, per_rank = merge_gathered_records(parts)
        assert per_rank == digests  # every shard — the empty ones too — arrived as the rank's own
        np.savez(out_path, recs=recs, text=text, pairs=pairs)
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ragged", "fewer_reads_than_ranks"])
def test_eight_rank_gather_with_empty_and_unmapped_shards(tmp_path, case):
    """N = 8 without the node (the `nccl` backend needs eight GPUs; RCCL refuses duplicate devices, so there is no one-GPU dry run of it): gather_hit_records +
    merge_gathered_records at world size 8 over gloo.  Reference return path: src/distributed/dispatcher.rs:223-247 (results of all workers merged in task order)."""
    out = str(tmp_path / f"records8_{case}.npz")
    mp.spawn(_records_worker8, args=(8, _free_port(), out, case), nprocs=8, join=True)
    got = np.load(out)
    shards = [_fake_records(n, seed=50 + r, p_unmapped=p) for r, (n, p) in enumerate(_SHARDS8[case])]
    total = sum(n for n, _ in _SHARDS8[case])
    assert got["recs"].shape == (total, 22) and total == (727 if case == "ragged" else 5)
    base = 0
    for recs, text, pairs in shards:
        tb, pf = text.view(np.uint8), pairs.view(np.float32)
        for i in range(recs.shape[0]):
            g = got["recs"][base + i]
            assert np.array_equal(np.delete(g, [12, 18]), np.delete(recs[i], [12, 18]))
            if recs[i, 3]:
                assert _text_of(got["recs"], got["text"], base + i) == _text_of(recs, tb, i)
                k = int(recs[i, 19])
                assert np.array_equal(got["pairs"][2 * int(g[18]):2 * int(g[18]) + 2 * k], pf[2 * int(recs[i, 18]):2 * int(recs[i, 18]) + 2 * k])
        base += recs.shape[0]
    if case == "ragged":
        assert not got["recs"][300 + 5 + 317 + 1:300 + 5 + 317 + 1 + 40, 3].any()  # the all-unmapped shard sits where rank 6's reads belong


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu_over_gloo(scaling):
    """bench.py's N > 1 path end to end on a one-GPU box: two ranks share device 0, rank 0 builds and saves the index, rank 1 loads the files,
    both map their shard with batches in flight, the read-ordered hit records are gathered on rank 0 (gloo, through host memory) and merged;
    every rank's gathered part must equal that rank's own result."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--config", "c2", "--genome-bp", "2000000", "--reads", "60000",
                        "--steps", "3", "--warmup", "1", "--scaling", scaling, "--no-cpu-baseline", "--no-extras", "--watchdog-s", "120"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if l.startswith("[rank") or "Error" in l)[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    g = line["gather"]
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and g["world_size_seen"] == 2 and g["ranks_identical_to_own_fetch"] == 2
    assert g["merged_reads"] == (120000 if scaling == "weak" else 60000)
    assert g["bytes_per_read"] <= 128 and g["merged_mapped"] > 0.8 * g["merged_reads"]  # compact records on the links, not hit intervals + edit tracks
    assert g["index"]["built_by"].startswith("rank 0") and g["index"]["load_s_per_rank"][1] is not None
    assert len(g["per_rank"]) == 2 and sum(r["reads"] for r in g["per_rank"]) == g["merged_reads"]


def test_merged_record_offsets_past_2_pow_31_do_not_overflow():
    """text_off / mq_off are u32 fields of an int32 record array; rank 0's rebase must not overflow int32 once the concatenated text pool passes
    2^31 bytes (8 ranks x 1.1 GB at C4), and must refuse pools the 32-bit offsets cannot address."""
    from mapad_amd.distributed import rebase_record_offsets
    recs, text, pairs = _fake_records(50, seed=1)
    mapped = recs[:, 3] != 0
    r = recs.copy()
    text_base, pair_base = 2_200_000_000, 2_300_000_000
    ends = rebase_record_offsets(r, mapped, text_base, pair_base, text.size * 4, pairs.size // 2)
    assert ends == (text_base + text.size * 4, pair_base + pairs.size // 2)
    u = r.view(np.uint32)
    assert np.array_equal(u[mapped, 12].astype(np.int64), recs[mapped, 12].astype(np.int64) + text_base) and (u[mapped, 12] > 2 ** 31).all()
    assert np.array_equal(u[mapped, 18].astype(np.int64), recs[mapped, 18].astype(np.int64) + pair_base)
    assert np.array_equal(r[~mapped], recs[~mapped]) and np.array_equal(np.delete(r, [12, 18], axis=1), np.delete(recs, [12, 18], axis=1))
    with pytest.raises(OverflowError):
        rebase_record_offsets(recs.copy(), mapped, 0xFFFFFFF0, 0, 64, 0)
