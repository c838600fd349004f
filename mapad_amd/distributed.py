"""Multi-GPU plumbing of the read-mapping path: reads shard across ranks, the index is replicated, and the only exchange is
the final gather of hit records on rank 0 (the reference's ResultSheet return path, src/distributed/mod.rs:21-34,
dispatcher.rs:223-247, done here with torch.distributed point-to-point transfers: RCCL over xGMI on GPUs, gloo in the CPU tests).

Each chunk is cut into `world` contiguous slices so that concatenating the per-rank results in rank order reproduces the input
order (rayon's order-preserving collect, src/map/mapping.rs:288).
"""
import numpy as np


def shard_bounds(n_reads, world, rank):
    """[lo, hi) of the contiguous slice of a chunk of n_reads reads that `rank` maps."""
    base, extra = divmod(n_reads, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_hit_records(hit_count, hits, ops, rank, world, device=None, meta_group=None):
    """Gathers three int32 buffers per rank on rank 0: either the hit records (hit_begin as int32 view of uint64[n_reads_local + 1], hits
    int32[n_hits_local * 10], ops int32[n_ops_local]) or — what bench.py sends since round 4 — the compact record fields of mapad_records_device
    (records int32[n_reads_local * 22], text bytes padded to int32, MAPQ pairs as int32 views of f32), see merge_gathered_records.

    Tensors may live on the GPU (nccl/RCCL) or the CPU (gloo).  Returns on rank 0 a list, indexed by source rank, of
    (hit_count, hits, ops) tensors; None elsewhere.  Sizes are exchanged first (all_gather of three int64), then every peer
    sends its three buffers straight to rank 0 — a fan-in over point-to-point links, no ring.
    meta_group: a CPU (gloo) group for the size exchange.  On GPUs every collective is a kernel that needs free wave slots; the persistent
    search wavefronts of the batches in flight hold them until a launch runs out of reads, and a host that waits for three integers would
    wait that long.  The bulk transfers are not waited for by the host, so they may start late.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return [(hit_count, hits, ops)]
    dev = hit_count.device if device is None else device
    mdev = torch.device("cpu") if meta_group is not None else dev
    sizes = torch.tensor([hit_count.numel(), hits.numel(), ops.numel()], dtype=torch.int64, device=mdev)
    all_sizes = [torch.zeros(3, dtype=torch.int64, device=mdev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=meta_group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    if rank == 0:
        out, reqs = [(hit_count, hits, ops)], []
        for r in range(1, world):
            bufs = [torch.empty(int(all_sizes[r][i]), dtype=torch.int32, device=dev) for i in range(3)]
            out.append(tuple(bufs))
            reqs += [dist.P2POp(dist.irecv, b, r) for b in bufs if b.numel()]
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        return out
    reqs = [dist.P2POp(dist.isend, b, 0) for b in (hit_count, hits, ops) if b.numel()]
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    return None


def collect_in_read_order(hit_count, hit_first, pool, ops_pool):
    """numpy restatement of the device-side collect (compact_* kernels of csrc/mapad_amd.hip) for the CPU tests of the N > 1 path:
    per-read (count, first index) into a completion-ordered hit pool -> hit_begin (uint64[n+1]), hits in read order (n_hits x 10 int32,
    word 8 = offset into the returned ops) and ops in read order."""
    hit_count = np.asarray(hit_count, dtype=np.int64)
    pool = np.asarray(pool).reshape(-1, 10)
    hit_begin = np.zeros(len(hit_count) + 1, dtype=np.uint64)
    hit_begin[1:] = np.cumsum(hit_count)
    hits = np.zeros((int(hit_begin[-1]), 10), dtype=np.int32)
    ops, k, base = [], 0, 0
    for r in range(len(hit_count)):
        for j in range(int(hit_count[r])):
            h = pool[int(hit_first[r]) + j].copy()
            n_ops, off = int(h[7]), int(h[8])
            ops.append(np.asarray(ops_pool[off:off + n_ops]))
            h[8] = base
            base += n_ops
            hits[k] = h
            k += 1
    return hit_begin, hits, (np.concatenate(ops) if ops else np.zeros(0, np.int32)).astype(np.int32)


def merge_gathered(parts):
    """rank-ordered (hit_begin as int32 view of uint64[n_r + 1], hits[10 x int32 per hit], ops) of read-ordered shards -> the chunk's
    hit_begin (uint64[n+1]), hits (n_hits x 10 int32), ops — every hit's ops offset (word 8) rebased into the concatenated ops array —
    and the sha256 of each rank's part, which must equal the digest of that rank's own fetched result."""
    import hashlib
    begins, hits, ops, digests = [np.zeros(1, np.uint64)], [], [], []
    hit_base = ops_base = 0
    for b, h, o in parts:
        b = np.ascontiguousarray(np.asarray(b.cpu())).view(np.uint64)
        h = np.ascontiguousarray(np.asarray(h.cpu())).reshape(-1, 10)
        o = np.ascontiguousarray(np.asarray(o.cpu()))
        dg = hashlib.sha256()
        for a in (b, h, o):
            dg.update(a.tobytes())
        digests.append(dg.hexdigest())
        assert int(b[-1]) == h.shape[0], "a shard's hit_begin does not match its hit records"
        begins.append(b[1:] + np.uint64(hit_base))
        h = h.copy()
        h[:, 8] += ops_base
        hit_base += h.shape[0]
        ops_base += int(o.size)
        hits.append(h)
        ops.append(o)
    return np.concatenate(begins), np.concatenate(hits), np.concatenate(ops), digests


RECORD_WORDS = 22  # csrc/text_core.hpp: DevRecord, 88 bytes
# word indices: i64 pos (0-1), tid 2, mapped 3, reverse 4, as 5, xs 6, nm 7, x0 8, x1 9, has_xs 10, xt 11, text_off 12, cigar_len 13, md_len 14, xa_len 15,
# best_size 16, read_len 17, mq_off 18, mq_n 19, error 20, (padding 21)
_REC_TEXT_OFF, _REC_CIGAR_LEN, _REC_MQ_OFF, _REC_MQ_N = 12, 13, 18, 19


def records_digest(recs, text, pairs):
    """sha256 of a shard's records that does not depend on where the text kernel put a read's bytes in its pools (wavefronts claim pool space with atomic
    adds, so two runs over the same batch lay the pools out differently): the records with text_off / mq_off zeroed, then every mapped read's
    [CIGAR][MD][XA] bytes and its (score, size) pairs, in read order."""
    import hashlib
    r = np.ascontiguousarray(np.asarray(recs)).reshape(-1, RECORD_WORDS)
    t = np.ascontiguousarray(np.asarray(text)).view(np.uint8)
    p = np.ascontiguousarray(np.asarray(pairs)).view(np.uint32)
    mapped = r[:, 3] != 0

    def in_read_order(pool, starts, lens):
        lens = np.where(mapped, lens, 0).astype(np.int64)
        total = int(lens.sum())
        if total == 0:
            return pool[:0]
        first = np.cumsum(lens) - lens  # where a read's piece starts in the output
        idx = np.repeat(starts.astype(np.int64) - first, lens) + np.arange(total, dtype=np.int64)
        return pool[idx]

    canon = r.copy()
    canon[:, _REC_TEXT_OFF] = 0
    canon[:, _REC_MQ_OFF] = 0
    h = hashlib.sha256()
    h.update(canon.tobytes())
    h.update(in_read_order(t, r[:, _REC_TEXT_OFF].view(np.uint32), r[:, _REC_CIGAR_LEN].astype(np.int64) + r[:, _REC_CIGAR_LEN + 1] + r[:, _REC_CIGAR_LEN + 2]).tobytes())
    h.update(in_read_order(p, 2 * r[:, _REC_MQ_OFF].view(np.uint32).astype(np.int64), 2 * r[:, _REC_MQ_N].astype(np.int64)).tobytes())
    return h.hexdigest()


def rebase_record_offsets(r, mapped, text_base, pair_base, text_bytes, n_pairs):
    """Adds a shard's bases to the text_off / mq_off words of its mapped records, in place, and returns the bases of the next shard.  The two offsets are u32
    fields of the int32 record array: the add happens on a uint32 view (as int32 it overflows once the concatenated text passes 2^31 bytes — 8 ranks x 1.1 GB
    at C4), and the merged pools must stay addressable by them."""
    text_end, pair_end = text_base + text_bytes, pair_base + n_pairs
    if text_end > 0xFFFFFFFF or pair_end > 0xFFFFFFFF:
        raise OverflowError(f"merged record pools exceed the records' 32-bit offsets ({text_end} text bytes, {pair_end} pairs): merge fewer ranks' steps at a time")
    u = r.view(np.uint32)
    u[mapped, _REC_TEXT_OFF] += np.uint32(text_base)
    u[mapped, _REC_MQ_OFF] += np.uint32(pair_base)
    return text_end, pair_end


def merge_gathered_records(parts):
    """rank-ordered (records int32[n_r * 22], text as int32 (bytes padded to a multiple of 4), pairs as int32 views of f32[2 * n_pairs_r]) of read-ordered
    shards -> the chunk's records (n x 22 int32, text_off / mq_off rebased into the concatenated pools), text bytes, pairs (float32) and the
    records_digest of each rank's part (which must equal the digest of that rank's own copy).  <= 128 bytes per read cross the links."""
    recs, texts, pairs, digests = [], [], [], []
    text_base = pair_base = 0
    for r, t, p in parts:
        r = np.ascontiguousarray(np.asarray(r.cpu())).reshape(-1, RECORD_WORDS)
        t = np.ascontiguousarray(np.asarray(t.cpu()))
        p = np.ascontiguousarray(np.asarray(p.cpu()))
        digests.append(records_digest(r, t, p))
        r = r.copy()
        mapped = r[:, 3] != 0
        text_end, pair_end = rebase_record_offsets(r, mapped, text_base, pair_base, int(t.size) * 4, int(p.size) // 2)
        text_base, pair_base = text_end, pair_end
        recs.append(r)
        texts.append(t.view(np.uint8))
        pairs.append(p.view(np.float32))
    return np.concatenate(recs), np.concatenate(texts), np.concatenate(pairs), digests
