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


def gather_hit_records(hit_count, hits, ops, rank, world, device=None):
    """Gathers (hit_count int32[n_reads_local], hits int32[n_hits_local * 10], ops int32[n_ops_local]) on rank 0.

    Tensors may live on the GPU (nccl/RCCL) or the CPU (gloo).  Returns on rank 0 a list, indexed by source rank, of
    (hit_count, hits, ops) tensors; None elsewhere.  Sizes are exchanged first (all_gather of three int64), then every peer
    sends its three buffers straight to rank 0 — a fan-in over point-to-point links, no ring.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return [(hit_count, hits, ops)]
    dev = hit_count.device if device is None else device
    sizes = torch.tensor([hit_count.numel(), hits.numel(), ops.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
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


def merge_gathered(parts):
    """rank-ordered (hit_count, hits[10 x int32 per hit], ops) -> global hit_begin (uint64[n+1]), hits (n_hits x 10 int32) and ops, with
    every hit's ops_offset (word 8) rebased into the concatenated ops array.  Hits inside a rank's pool may be stored in any
    order; `hit_first` is not needed here because per-rank pools are first compacted into read order by the caller."""
    counts = np.concatenate([np.asarray(p[0].cpu()) for p in parts]).astype(np.uint64)
    hit_begin = np.zeros(len(counts) + 1, dtype=np.uint64)
    hit_begin[1:] = np.cumsum(counts)
    hits, ops, base = [], [], 0
    for _, h, o in parts:
        h = np.asarray(h.cpu()).reshape(-1, 10).copy()
        h[:, 8] += base
        base += int(o.numel())
        hits.append(h)
        ops.append(np.asarray(o.cpu()))
    return hit_begin, np.concatenate(hits), np.concatenate(ops)
