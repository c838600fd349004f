#!/usr/bin/env python3
"""bench.py — mapped reads/s of the mapAD hot path on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5] [--genome-bp G] [--reads R]

`--gpus N` (N > 1) without a torchrun environment starts the N ranks itself (torch.distributed.run as a child process, before
anything touches the GPU); under the driver's torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  A world size that differs from
--gpus, or fewer visible devices than ranks, is an error, never a silent 1-GPU run.

Workloads (synthetic genome = i.i.d. ACGT from splitmix64(1234), reads 90 % endogenous with 2 % substitutions, 10 % exogenous;
`-p 0.03 -D 0.02 -i 0.001 -x 1.0`, gap_dist_ends 5, max_num_gaps_open 2; the index is built on the GPU in the setup phase):
  c1  5 386 bp genome (phiX174-size), 1 k x 50 bp, no-damage model, Phred 40 — with the single-thread oracle figure
  c2  48 Mbp genome (chr21-size), 1 M x 50 bp per GPU, no-damage model, Phred 40            [BASELINE.json configs[1]]
  c3  the same genome, single-stranded library f = t = 0.5, d = 0.02, s = 1.0, Phred 20-40
  c4  3 Gbp genome (hg19-size, n = 6e9 rows > 2^32), 10 M x 50 bp per GPU, no-damage model, Phred 40   [default: BASELINE.json configs[3],
      the largest single-GPU configuration and the north star's target]
  c5  the read mix of C5 (35-100 bp, 5 % of the reads with an indel, damage model) on the 48 Mbp genome, 1 M reads per step
One "step" = one pass of the hot path (D-array kernel, ordering, search kernel + its retry / full-limit launches) over the batch;
reads, index and score tables are resident in HBM before the timed region.  With N > 1 every rank holds a replica of the index, maps
its own shard (weak scaling) and, inside every step, lays its hits out in read order on the device, turns them into record fields there (coordinates,
CIGAR / MD / XA text, MAPQ inputs: <= 128 bytes per read) and sends those to rank 0 (RCCL point-to-point over xGMI); after the timed region rank 0 merges
the shards and checks them against every rank's own copy.

Also on the line: `roofline` (dominant kernel: algorithmic bytes from the kernels' event counters / HIP-event time; `random_access` = the requests the L2 sent to
memory per second, from the committed PMC passes, against this chip's measured rate of independent random reads — the bound the path actually runs against), `cpu_baseline`
(the C++ oracle on the host cores over a bounded sample, N = 1 only; its hits must equal the GPU's), `e2e` (host buffers in, host
results out: H2D + kernels + device-side collect + D2H), `sa_locate` and `post_search` (the next rows of the path), `tail` (reads finished by host threads),
`secondary` (C4 only: short C2 / C3 runs as processes of their own, and C5's read mix — 200 000 reads at the reference's real limits — on this run's 3 Gbp index).
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per batch in flight; must be set before torch or the library touch the GPU

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
RANDOM_LINE_GBPS = 6200.0  # GB/s that independent random 128-byte line fills deliver on this chip (48.5 G lines/s x 128 B; profiles/r05/calib_random_access.txt; the guide: ~6.2-6.3 TB/s)
RANDOM_ACCESS_CEILING_G = 45.0  # G requests/s: independent random 64-128-byte reads from an 8 GiB table, measured (profiles/calib/fetch_calib.hip: 42.8-48.5)

CONFIGS = {  # genome bp, reads per GPU
    "c1": (5_386, 1_000), "c2": (48_000_000, 1_000_000), "c3": (48_000_000, 1_000_000), "c4": (3_000_000_000, 10_000_000), "c5": (48_000_000, 1_000_000),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class DevArray:
    """__cuda_array_interface__ view of a raw device pointer so torch can wrap library-owned HBM buffers without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def spawn_ranks(n, same_device=False):
    """Start n ranks as children of this (GPU-free) process and exit with their status."""
    import torch
    have = torch.cuda.device_count()  # does not initialise the GPU
    if have < (1 if same_device else n):
        log(f"bench.py: --gpus {n} but only {have} device(s) visible")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def host_cpus():
    """CPUs this process may really use: os.cpu_count() capped by the cgroup's CPU-time quota (the GPU boxes show 256 CPUs and grant 16)."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, round(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, round(q / p)))
        except Exception:
            pass
    return n


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_reads(synth, genome, n_reads, seed, **kw):
    """synth.reads in chunks of 1 M reads (bounded host memory at 10 M reads)."""
    chunk = 1_000_000
    if n_reads <= 2 * chunk:
        return synth.reads(genome, n_reads, 50, seed=seed, **kw)
    parts, base = [], 0
    for k, lo in enumerate(range(0, n_reads, chunk)):
        s, q, o = synth.reads(genome, min(chunk, n_reads - lo), 50, seed=seed + 7919 * k, **kw)
        parts.append((s, q, o[:-1] + np.uint64(base)))
        base += int(o[-1])
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts] + [np.array([base], np.uint64)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--genome-bp", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the e2e / cli / sa_locate / post_search legs")
    ap.add_argument("--no-secondary", action="store_true", help="C4 only: skip the short C2 / C3 runs (BASELINE.json configs[1], configs[2]) reported under `secondary`")
    ap.add_argument("--no-cli", action="store_true", help="skip the command-line leg (FASTQ -> BAM; at C4 it writes and re-reads the 3 Gbp index files)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: test mode for boxes with one GPU — every rank uses device 0 and the gather goes through host memory")
    ap.add_argument("--depth", type=int, default=None, help="batches in flight (1 = every step runs alone on the stream); default 4 (round 4, C2 / C3 at 3 / 4 / 6 in flight: 5.71 / 6.01 / 5.98 M and 2.47 / 2.53 / 2.56 M reads/s), C4: 2 (its 10 M-read launches "
                    "run one after the other, the next batch's D arrays beside the search), C5: 6 (heavy-tailed reads: more tails to overlap)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --reads per GPU (total work grows with N); strong = ONE chunk of --reads reads cut into N contiguous slices (SURVEY 8e)")
    ap.add_argument("--watchdog-s", type=float, default=600.0, help="N > 1: exit 3 if no step or gather completes for this long (a starved transfer must not hang the node)")
    ap.add_argument("--search-waves-per-cu", type=int, default=0, help="resident search wavefronts per CU (0 = the library's default, the same for every N)")
    ap.add_argument("--reserved-cus", type=int, default=-1, help="CUs the search launches leave free for RCCL's transfer kernels (default 0: see profiles/r05/rccl_standin.txt)")
    ap.add_argument("--own-index", action="store_true", help="N > 1: every rank builds its own index instead of loading the files rank 0 wrote")
    args = ap.parse_args()
    if args.depth is None:
        args.depth = {"c4": 2, "c5": 6}.get(args.config, 4)
    genome_bp = args.genome_bp or CONFIGS[args.config][0]
    n_reads = args.reads or CONFIGS[args.config][1]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, same_device=args.dist_backend == "gloo")  # never returns
    # stdout carries the one JSON line and nothing else: whatever a library prints there (gloo's connection banner, ...) goes to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.dist_backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}")
        sys.exit(2)

    import torch
    import torch.distributed as dist

    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.distributed import gather_hit_records, merge_gathered_records, records_digest
    from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve as resolve_params

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in mapad_amd")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world == args.gpus
    xdev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the exchanged tensors live
    # sizes travel over a CPU group: a GPU collective is a kernel that needs wave slots, and the persistent search wavefronts hold them (distributed.py)
    meta_group = dist.new_group(backend="gloo") if world > 1 and args.dist_backend == "nccl" else None
    # (N > 1 runs the search launches like N = 1 — as many wavefronts per CU as fit: a launch leaves 12 KB of LDS and one wavefront's registers free on every CU,
    #  which is what RCCL's transfer kernels need; --search-waves-per-cu 8 is the fall-back should the watchdog ever report a starved transfer)
    if args.search_waves_per_cu:
        os.environ["MAPAD_TIER0_WAVES_PER_CU"] = str(args.search_waves_per_cu)
    if world > 1:  # rank 0 receives every rank's records into buffers of its own (1.1 GB per rank and step at C4): the size-class pools leave room for them on every GPU
        os.environ.setdefault("MAPAD_POOL_BUDGET_GB", "48")
    if world > 1 and args.dist_backend == "gloo":  # test mode: the ranks share ONE GPU — each takes its share of the chip's wavefront slots and of the HBM for its pools
        os.environ.setdefault("MAPAD_TIER0_WAVES_PER_CU", str(max(2, 12 // world)))
        os.environ["MAPAD_POOL_BUDGET_GB"] = str(max(4, 64 // world))

    # ---- watchdog (N > 1): a rank that makes no progress for --watchdog-s seconds ends the job with a fresh exit (never an exec) ---------------
    import threading
    beat = {"t": time.time(), "what": "start"}

    def heartbeat(what):
        beat["t"], beat["what"] = time.time(), what

    def watchdog():
        while True:
            time.sleep(2.0)
            if time.time() - beat["t"] > args.watchdog_s:
                log(f"[rank {rank}] WATCHDOG: no progress for {args.watchdog_s:.0f} s after '{beat['what']}'; giving up")
                os._exit(3)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    # ---- workload --------------------------------------------------------------------------------------------------------
    t0 = time.time()
    genome = synth.genome(genome_bp, seed=1234)
    t_genome = time.time() - t0
    heartbeat("genome")
    t0 = time.time()
    index_how, t_index_save, t_index_load = "built on this rank's GPU", None, None
    shared = None
    if world > 1 and not args.own_index:
        # one index for the node: rank 0 builds it on its GPU and writes the seven files (mapad_index_save), the other ranks load them
        # (mapad_index_open) — the way a `mapad map` process finds its index — instead of N suffix sorts and N host text preparations on one host
        import shutil
        import tempfile
        shared = os.path.join(tempfile.gettempdir(), f"mapad_bench_index_{os.environ.get('MASTER_PORT', '0')}_{genome_bp}")
        need = 3 * genome_bp + (2 << 30)
        ok = torch.tensor([1 if shutil.disk_usage(tempfile.gettempdir()).free > need else 0], dtype=torch.int64, device=xdev)
        dist.broadcast(ok, 0)
        if int(ok.item()) == 0:
            shared = None
            log(f"[rank {rank}] not enough scratch space for shared index files: every rank builds its own index")
    if shared is None:
        index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=local_rank)  # suffix sorting on the GPU (csrc/index_gpu.hip)
        t_index = time.time() - t0
    else:
        if rank == 0:
            index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=local_rank)
            t_index = time.time() - t0
            heartbeat("index built")
            t1 = time.time()
            os.makedirs(shared, exist_ok=True)
            index.save(os.path.join(shared, "ref"))
            t_index_save = time.time() - t1
            heartbeat("index saved")
        dist.barrier()
        if rank != 0:
            t1 = time.time()
            index = mapad_amd.Index.open(os.path.join(shared, "ref"))
            t_index_load = time.time() - t1
            t_index = time.time() - t0
            index_how = "loaded from the files rank 0 wrote"
        heartbeat("index ready")
        dist.barrier()
        if rank == 0:
            shutil.rmtree(shared, ignore_errors=True)
    heartbeat("index")
    if args.config in ("c1", "c2", "c4"):
        prm, kw = NO_DAMAGE, dict(qual=40)
    elif args.config == "c3":
        prm, kw = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
    else:
        prm, kw = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    cfg_id = int(args.config[1])
    rp = resolve_params(prm)
    params = mapad_amd.make_params(rp)
    n_chunk = n_reads  # --reads: per GPU (weak) or the whole chunk (strong)
    if world > 1 and args.scaling == "strong":
        from mapad_amd.distributed import shard_bounds
        a_seqs, a_quals, a_offsets = make_reads(synth, genome, n_chunk, 4321 + cfg_id, **kw)  # the same chunk on every rank ...
        lo, hi = shard_bounds(n_chunk, world, rank)                                            # ... of which this rank maps a contiguous slice
        b0, b1 = int(a_offsets[lo]), int(a_offsets[hi])
        seqs, quals, offsets = a_seqs[b0:b1].copy(), a_quals[b0:b1].copy(), (a_offsets[lo:hi + 1] - a_offsets[lo]).astype(np.uint64)
        del a_seqs, a_quals, a_offsets
        n_reads = hi - lo
    else:
        seqs, quals, offsets = make_reads(synth, genome, n_reads, 4321 + cfg_id + 1000 * rank, **kw)
    heartbeat("reads")
    log(f"[rank {rank}] genome {genome_bp} bp in {t_genome:.1f}s, index (n = {len(index)}) {index_how} in {t_index:.1f}s, {n_reads} reads")

    stream = torch.cuda.current_stream(dev)
    ctx = mapad_amd.Context(index, params, local_rank)
    ctx.set_stream(ctypes.c_void_p(stream.cuda_stream))
    # N > 1 over RCCL: the search launches leave a few CUs to the transfer kernels (csrc: create_slot_stream; profiles/r05/rccl_standin.txt for the choice)
    # Default 0 (profiles/r05/rccl_standin.txt): a kernel of RCCL's shape issued beside the pipelined C4 loop ran at the boundary between two search launches — one
    # step late, at no cost to the search — while eight CUs kept free for it (one per XCD) made it run within 0.2 s but cost the search 6.7 %.  The gather has its
    # own stream and only holds back the launch that reuses its batch slot (below), so a late transfer does not stall the pipeline.
    reserved_cus = max(args.reserved_cus, 0)
    if reserved_cus:
        ctx.set_reserved_cus(reserved_cus)
    ctx.set_pipeline_depth(args.depth)
    lens = np.diff(offsets.astype(np.int64))
    max_len = int(lens.max())
    ctx.prepare_lengths(sorted(set(lens.tolist())))
    ctx.set_fetch_d_arrays(False)
    ctx.reserve(n_reads, int(offsets[-1]), max_len)  # every batch slot's buffers up front: an allocation inside the pipeline would wait for running kernels
    d_seqs = torch.from_numpy(seqs).to(dev)
    d_quals = torch.from_numpy(quals).to(dev)
    d_offsets = torch.from_numpy(offsets.view(np.int64)).to(dev)
    torch.cuda.synchronize(dev)

    gather_bytes = []  # per gathered step: bytes this rank put on the links

    def record_views():
        """The selected batch's record fields on the device (collect, then the coordinate and text kernels over the device-resident hits): 88-byte records,
        text pool, MAPQ pairs — as int32 tensors over the library's buffers (the text padded to whole words)."""
        p_rec, p_text, p_pairs, n_text, n_pairs = ctx.records_device(0)
        recs = torch.as_tensor(DevArray(p_rec, (n_reads * 22,), "<i4"), device=dev)
        text = torch.as_tensor(DevArray(p_text, ((n_text + 3) // 4 + 1,), "<i4"), device=dev)[:(n_text + 3) // 4]
        pairs = torch.as_tensor(DevArray(p_pairs, (max(n_pairs, 1) * 2,), "<i4"), device=dev)[:n_pairs * 2]
        return recs, text, pairs

    # The gather runs on a stream of its own (round 5): RCCL's transfer kernels need whole CUs (248-256 VGPRs, 37.6 KB of LDS per block) and so start when the search
    # beside them leaves some free — on the CUs mapad_ctx_set_reserved_cus keeps free, or when a launch thins out.  On the caller's stream a late gather would hold
    # back everything submitted behind it; on its own stream it only has to be over before ITS batch slot is launched again (`depth` submissions later), which the
    # step loop enforces with an event.  gather.issue_to_done_ms (per step, on the line) shows how late it ran.
    gather_stream = torch.cuda.Stream(dev) if world > 1 and args.dist_backend == "nccl" else None
    gather_events = []  # per gathered step: (issued, done) events on the gather stream

    def gather_hits():
        """The only exchange of the path: every rank's record fields in read order (<= 128 bytes per read once the SA lookup is on the device) go to rank 0."""
        if world == 1:
            return None
        recs, text, pairs = record_views()
        gather_bytes.append(4 * (recs.numel() + text.numel() + pairs.numel()))
        if args.dist_backend == "gloo":
            torch.cuda.synchronize(dev)
            recs, text, pairs = recs.cpu(), text.cpu(), pairs.cpu()
            return gather_hit_records(recs, text, pairs, rank, world, device=xdev, meta_group=meta_group)
        gather_stream.wait_stream(torch.cuda.current_stream(dev))  # the record kernels of this batch have been queued on the caller's stream
        with torch.cuda.stream(gather_stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(gather_stream)
            out = gather_hit_records(recs, text, pairs, rank, world, device=xdev, meta_group=meta_group)
            e1.record(gather_stream)
        gather_events.append((e0, e1))
        return out

    tail_steps = []  # per collected step: the host tail's figures (csrc/host_tail.hpp)

    host_trace = []  # (what, seconds) of the library calls of the step loop, when MAPAD_BENCH_TRACE is set: where the host thread waits

    def timed(what, fn, *a):
        if not os.environ.get("MAPAD_BENCH_TRACE"):
            return fn(*a)
        t = time.perf_counter()
        r = fn(*a)
        host_trace.append((what, round(time.perf_counter() - t, 4)))
        return r

    def collect_step():
        """The selected batch's order-preserving collect on the device — which first waits for the host threads that finish the reads the GPU handed
        over (none for the 50 bp workloads) — and, N > 1, the gather of its records."""
        if world == 1:
            timed("collect", ctx.compact_device)
            g = None
        else:
            g = gather_hits()
        tail_steps.append(ctx.tail_info())
        return g

    def run_steps(k):
        """k steps back to back, `depth` batches in flight: behind the submission of step i the oldest batch in flight (step i - depth + 1) is collected
        (and, N > 1, its records gathered)."""
        last = None
        d = args.depth
        for i in range(k):
            if gather_stream is not None and gather_events:  # the batch slot this submission reuses is the one whose records were gathered last (step i - depth): that transfer must be over
                torch.cuda.current_stream(dev).wait_event(gather_events[-1][1])
            timed("submit", ctx.map_batch_device, d_seqs.data_ptr(), d_quals.data_ptr(), d_offsets.data_ptr(), n_reads, max_len)
            heartbeat(f"step {i} submitted")
            if i >= d - 1:  # `depth` batches are in flight: collect the oldest (the host waits for it — and for its host tail — while the others map)
                ctx.select_batch(d - 1)
                last = collect_step()
                ctx.select_batch(0)
        for age in range(min(d - 1, k) - 1, -1, -1):  # drain, oldest first
            ctx.select_batch(age)
            last = collect_step()
        if k:
            ctx.select_batch(0)
        heartbeat("steps done")
        return last

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    ctx.kernel_history()  # drop the warm-up launches' time stamps
    tail_steps.clear()
    gather_events.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    host_trace.clear()
    gathered = run_steps(args.steps)
    if host_trace:
        log("[host trace] " + " ".join(f"{w}:{t}" for w, t in host_trace[:24]))
    tail_timed = list(tail_steps)
    if args.config in ("c1", "c2", "c3", "c4") and any(t["reads"] for t in tail_timed):
        log(f"[rank {rank}] NOTE: {sum(t['reads'] for t in tail_timed)} reads of a 50 bp workload passed the pop budget and were finished on the host (none are expected to)")
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    rank_rates = None
    if world > 1:
        mine = torch.tensor([elapsed, float(n_reads)], dtype=torch.float64, device=xdev)
        every = [torch.zeros(2, dtype=torch.float64, device=xdev) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_rates = [{"rank": r, "reads": int(v[1].item()), "elapsed_s": round(float(v[0].item()), 4), "reads_per_s": round(float(v[1].item()) * args.steps / float(v[0].item()), 1)} for r, v in enumerate(every)]
        elapsed = max(float(v[0].item()) for v in every)
        total_reads_per_step = int(sum(float(v[1].item()) for v in every))
    else:
        total_reads_per_step = n_reads
    heartbeat("timed region done")
    hist = ctx.kernel_history().astype(np.float64)  # per launch: ms from the first launch's start to its four event marks
    assert hist.shape[0] == args.steps

    def union_ms(a, b):
        """length of the union of the intervals [a_i, b_i] (launches of different batches overlap)"""
        total, end = 0.0, -1.0
        for lo, hi in sorted(zip(a.tolist(), b.tolist())):
            if hi > end:
                total += hi - max(lo, end)
                end = hi
        return total

    # one more launch, alone on the chip: the per-kernel durations that `rocprofv3 --stats` reports for un-overlapped launches
    ctx.map_batch_device(d_seqs.data_ptr(), d_quals.data_ptr(), d_offsets.data_ptr(), n_reads, max_len)
    torch.cuda.synchronize(dev)
    solo_ms = [float(x) for x in ctx.kernel_ms()]
    ctx.kernel_history()

    res = ctx.fetch()
    counters = ctx.last_counters()

    # ---- N > 1: rank 0 rebuilds the read-ordered hit list of the whole chunk and checks it against every rank's own result ---------
    gather_check = None
    if world > 1:
        ctx.select_batch(1 if args.depth > 1 else 0)  # the last timed step's batch: still in its slot (the solo launch went to the next one)
        # (a digest of the records' content: where a read's text sits in the pools differs from run to run of the text kernel)
        own = records_digest(*[t.cpu().numpy() for t in record_views()]) if args.depth > 1 else None
        ctx.select_batch(0)
        if own is None:  # depth 1: the slot has been launched again; the solo launch maps the same reads, so its records are the same
            own = records_digest(*[t.cpu().numpy() for t in record_views()])
        all_own = [None] * world
        dist.all_gather_object(all_own, own)
        loads = [None] * world
        dist.all_gather_object(loads, None if t_index_load is None else round(t_index_load, 1))
        if rank == 0:
            m_recs, m_text, m_pairs, per_rank = merge_gathered_records(gathered)
            ok = [per_rank[r] == all_own[r] for r in range(world)]
            gather_check = {"world_size_seen": dist.get_world_size(), "ranks_identical_to_own_fetch": int(sum(ok)), "merged_reads": int(m_recs.shape[0]),
                            "merged_mapped": int((m_recs[:, 3] != 0).sum()), "merged_text_bytes": int(m_text.size), "merged_pairs": int(m_pairs.size // 2),
                            "payload": "per read an 88-byte record (position, contig, strand, AS / XS / NM / X0 / X1 / XT, text and pair offsets) + its CIGAR / MD / XA text + the (score, size) pairs of the mapping quality",
                            "bytes_per_read": round(sum(gather_bytes) / max(len(gather_bytes), 1) / max(n_reads, 1), 1), "per_rank": rank_rates,
                            # rank 0, per timed step: from the gather's issue (behind the step's record kernels) to the arrival of the last peer's records; a transfer
                            # that had to wait for CUs shows here as a step's length instead of a few tens of ms
                            "issue_to_done_ms": [round(a.elapsed_time(b), 1) for a, b in gather_events[:args.steps]] if gather_events else None,
                            "reserved_cus": reserved_cus,
                            "exchange": "RCCL point-to-point fan-in to rank 0 over xGMI, issued behind the next step's submission" if args.dist_backend == "nccl" else "gloo through host memory (test mode)",
                            "index": {"built_by": "rank 0, saved, loaded by the others" if shared is not None else "every rank", "save_s": None if t_index_save is None else round(t_index_save, 1), "load_s_per_rank": loads}}
            if not all(ok):
                log("GATHER FAILURE: a rank's gathered records differ from its own result")

    # ---- roofline of the dominant kernel (rank 0) -------------------------------------------------------------------------
    e_search, e_darray, n_push, n_pop, n_node, n_hit_events = [int(x) for x in counters]
    # the reads the host threads finished: their events are in the per-read counters (that is what makes the counters a parity check), but the kernel did not
    # execute them — they do not count towards the kernel's algorithmic bytes (nor does what the GPU spent on those reads before it gave them up)
    tail_last = ctx.tail_info()
    e_search -= tail_last["host_e_search"]; n_push -= tail_last["host_n_push"]; n_node -= tail_last["host_n_node"]
    n_pop_all, n_pop = n_pop, n_pop - tail_last["host_pops"]
    total_bases = int(offsets[-1])
    bytes_darray = 256 * e_darray + 6 * total_bases                      # 2 x 128-B index blocks per extension + read/qual in, D out
    bytes_search = 256 * e_search + 40 * (n_push + n_pop) + 8 * n_node    # + 40-B frames through the heap, 8-B tree nodes
    # effective launch duration = union of the K launches' intervals / K (equal to the plain mean when nothing overlaps)
    ms_darray = union_ms(hist[:, 0], hist[:, 1]) / args.steps
    ms_search = union_ms(hist[:, 1], hist[:, 2]) / args.steps
    ms_pass2 = float((hist[:, 3] - hist[:, 2]).mean())
    per_launch_search = float((hist[:, 2] - hist[:, 1]).mean())
    dominant = "search_kernel" if ms_search >= ms_darray else "darray_kernel"
    dom_bytes, dom_ms = (bytes_search, ms_search) if dominant == "search_kernel" else (bytes_darray, ms_darray)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # HBM traffic of the PMC passes (profiles/collect.sh) — reported only while the library's gfx950 machine code is the code those passes ran
    traffic, traffic_stale, traffic_darray, hbm_requests, ea = None, None, None, None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            from mapad_amd import build as mbuild_
            entry = json.load(open(tp)).get(f"{args.config}:{genome_bp}:{n_reads}", {})
            if entry.get(dominant) is not None:
                if entry.get("kernel_code_sha16"):  # the device machine code of the loaded library is the code the PMC passes ran
                    traffic_stale = entry["kernel_code_sha16"] != mbuild_.kernel_code_hash()
                else:
                    traffic_stale = entry.get("kernel_source_sha16") != mbuild_.source_hash()
                traffic = None if traffic_stale else entry.get(dominant)
                traffic_darray = None if traffic_stale else entry.get("darray_kernel")
                hbm_requests = None if traffic_stale else entry.get(dominant + "_hbm_requests")
                ea = None if traffic_stale else {k: entry.get(dominant + "_" + k) for k in ("hbm_read_requests", "hbm_write_64B_units", "ea_rdreq", "ea_wrreq", "ea_wrreq_64B")}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_stale": traffic_stale,
                "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms": round(float(dom_ms), 4),
                "kernel_ms_is": "union of the K launches' HIP-event intervals / K" + (f" ({args.depth} batches in flight: launch k+1 runs beside the tail of launch k)" if args.depth > 1 else ""),
                "search_kernel_ms_per_launch_overlapped": round(per_launch_search, 4),
                "launch_marks_ms": [[round(float(x), 2) for x in row] for row in hist],  # per launch: D-array start, search start, search end, last-pass end
                "solo_launch": {"darray_ms": round(solo_ms[0], 4), "search_ms": round(solo_ms[1], 4), "search_GB/s": round(bytes_search / (solo_ms[1] * 1e-3) / 1e9, 2),
                                "frac": round(bytes_search / (solo_ms[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "what": "one launch alone on the chip, nothing else in flight (what rocprofv3 --stats sees with --depth 1)"},
                # darray_kernel: its SURVEY 8(d) byte count is NOT a bandwidth (the top levels of every read's D-array chains hit the same index blocks: cache
                # resident, `algorithmic_GB/s` can exceed the HBM peak) — what it moves through HBM is the PMC traffic, and `hbm_GB/s` = that over its solo duration;
                # `ms` is its phase in the pipelined loop (beside the previous batch's search), `solo_ms` the same launch alone on the chip
                "all_kernels": {"darray_kernel": {"ms": round(float(ms_darray), 4), "solo_ms": round(solo_ms[0], 4), "algorithmic_bytes": bytes_darray,
                                                  "algorithmic_GB/s": round(bytes_darray / (solo_ms[0] * 1e-3) / 1e9, 2) if solo_ms[0] else None, "not_a_bandwidth": True,
                                                  "traffic": traffic_darray, "hbm_GB/s": round(traffic_darray / (solo_ms[0] * 1e-3) / 1e9, 2) if traffic_darray and solo_ms[0] else None,
                                                  "bound": "L2 / instruction issue (the index blocks of the first extensions are shared by all reads)"},
                                "search_kernel": {"ms": round(float(ms_search), 4), "bytes": bytes_search, "GB/s": round(bytes_search / (ms_search * 1e-3) / 1e9, 2)},
                                "search_kernel_last_pass": {"ms": round(float(ms_pass2), 4), "arena_migrations": res.n_second_pass, "reads": res.n_third_pass}},
                # the bound this path actually runs against (SURVEY 8(d)'s "random-access line rate"): requests the L2 sends to memory per second — the PMC passes'
                # FETCH_SIZE + WRITE_SIZE in 64-byte request units (profiles/traffic.json) over a launch alone on the chip — against the rate at which independent
                # random 64 / 128-byte reads of an 8 GiB table complete on this chip (profiles/calib/fetch_calib.hip, profiles/r05/calib_random_access.txt: 48.5 /
                # 42.8 G/s; 32-byte records gathered per lane 40.3 G/s): a rate of accesses, whatever their size
                "random_access": None if not hbm_requests or dominant != "search_kernel" else {
                    "hbm_requests_per_launch": hbm_requests, "achieved": round(hbm_requests / (solo_ms[1] * 1e-3) / 1e9, 2), "ceiling": RANDOM_ACCESS_CEILING_G, "unit": "G requests/s",
                    "frac": round(hbm_requests / (solo_ms[1] * 1e-3) / 1e9 / RANDOM_ACCESS_CEILING_G, 4),
                    "what": "read + write requests behind the L2 per launch (PMC: (FETCH_SIZE + WRITE_SIZE) / 64 B) / solo launch duration; ceiling = measured read-only random-access rate "
                            "of this chip.  This fraction counts a 32- or 64-byte write like a 128-byte line fill and divides by a read-only ceiling: read it beside the two below",
                    # (round-5 verdict, weak 2b) the same launch by BYTES: PMC traffic over the solo launch against what random 128-byte line fills deliver on this chip
                    # (48.5 G lines/s x 128 B = 6.2 TB/s, profiles/r05/calib_random_access.txt) ...
                    "frac_by_bytes": None if not traffic else round(traffic / (solo_ms[1] * 1e-3) / 1e9 / RANDOM_LINE_GBPS, 4), "random_line_GB/s": RANDOM_LINE_GBPS,
                    # ... and by the memory side's own request counts (TCC_EA0_RDREQ + TCC_EA0_WRREQ: a write request is 32 or 64 bytes, so there are more of them than WRITE_SIZE / 64)
                    "frac_by_ea_requests": None if not (ea and ea.get("ea_wrreq")) else round((ea["ea_rdreq"] + ea["ea_wrreq"]) / (solo_ms[1] * 1e-3) / 1e9 / RANDOM_ACCESS_CEILING_G, 4),
                    "requests_per_pop": None if not (ea and ea.get("hbm_read_requests") and n_pop) else {
                        "read_128B_lines": round(ea["hbm_read_requests"] / n_pop, 3), "write_64B_units": round(ea["hbm_write_64B_units"] / n_pop, 3),
                        "write_requests_TCC_EA0_WRREQ": None if not ea.get("ea_wrreq") else round(ea["ea_wrreq"] / n_pop, 3),
                        "of_them_64_byte": None if not ea.get("ea_wrreq") else round(ea["ea_wrreq_64B"] / n_pop, 3)}},
                # secondary bound of SURVEY 8(d): dependent random index blocks per second (2 per extension)
                "index_lines_per_s": {"search_kernel": round(2 * e_search / (ms_search * 1e-3), 1), "darray_kernel": round(2 * e_darray / (ms_darray * 1e-3), 1)},
                "events": {"E_search": e_search, "E_darray": e_darray, "N_push": n_push, "N_pop": n_pop, "N_node": n_node}}

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1) -------------------------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob  # the checker / CPU baseline: only ever used in this leg
        cores = 1 if args.config == "c1" else host_cpus()  # C1 is the single-thread figure
        oidx = ob.OracleIndex.from_bwt(index.bwt(), "$ACGTX", 128)  # byte BWT + Occ k = 128 like the reference (indexing.rs:188)
        op = ob.make_params(rp)

        def run(n):
            sub_off = offsets[:n + 1]
            reads = [seqs[int(sub_off[i]):int(sub_off[i + 1])].tobytes() for i in range(n)]
            qs = [quals[int(sub_off[i]):int(sub_off[i + 1])] for i in range(n)]
            t = time.perf_counter()
            r = oidx.map_batch(op, reads, qs, n_threads=cores)
            return r, time.perf_counter() - t

        n0 = min(n_reads, 4 * cores + 512)
        _, dt0 = run(n0)
        n_sample = int(min(n_reads, max(n0, n0 / max(dt0, 1e-3) * args.cpu_seconds)))
        ores, dt = run(n_sample)
        cpu = {"value": round(n_sample / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {n_sample} reads of the same batch, {dt:.1f} s wall, C++ oracle (restatement of the reference algorithm: byte BWT, "
                         f"Occ k=128, min-max heap, slab tree), {cores} thread(s) = the CPUs this process may use ({os.cpu_count()} visible)"}
        # parity of the GPU result on that sample: hit counts, intervals, f32 score bits, edit tracks
        hb = res.hit_begin[:n_sample + 1]
        nh = int(hb[-1])
        ok = (np.array_equal(hb, ores.hit_offsets) and np.array_equal(res.hits_arr["lower"][:nh], ores.intervals[:, 0])
              and np.array_equal(res.hits_arr["lower_rev"][:nh], ores.intervals[:, 1])
              and np.array_equal(res.hits_arr["size"][:nh], ores.intervals[:, 2])
              and np.array_equal(res.hits_arr["score"][:nh].view(np.uint32), ores.scores.view(np.uint32)))
        n_ops = int(ores.op_offsets[-1])
        ok = ok and np.array_equal(res.ops[:n_ops], ores.ops)
        c = res.counters[:n_sample]
        got = np.stack([c["e_search"], c["e_darray"], c["n_push"], c["n_pop"], c["n_node"], c["n_hits"]], axis=1).astype(np.uint64)
        ok_c = bool(np.array_equal(got, ores.counters))
        parity = {"reads_checked": n_sample, "bit_identical_hits": bool(ok), "identical_event_counters": ok_c,
                  "hit_intervals_at_or_above_2^32": int((res.hits_arr["lower"][:nh] >= 2 ** 32).sum())}
        if not (ok and ok_c):
            log("PARITY FAILURE: GPU hits differ from the oracle on the sample")
        if not (args.config == "c4" and not args.no_extras and not args.no_secondary):
            del oidx  # (kept for the C5 leg of `secondary`, which checks its sample against the same oracle index)

    extras = rank == 0 and world == 1 and not args.no_extras
    # ---- end to end through the host entry point: pageable host buffers in, host result out --------------------------------------
    e2e = None
    if extras:
        ctx.set_stream(None)
        from mapad_amd import binding as mb
        import ctypes as C
        ts = []
        for _ in range(3):
            out = C.POINTER(mb.BatchResultC)()
            t = time.perf_counter()
            rc = mb.lib().mapad_map_batch(ctx.h, seqs.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), n_reads, C.byref(out))
            ts.append(time.perf_counter() - t)  # the C call: results are in (page-locked) host memory when it returns
            assert rc == 0
            r2 = mb.BatchResult(out, mb.lib().mapad_batch_result_free)
        same = r2.n_hits == res.n_hits and digest(r2.hit_begin, r2.hits_arr, r2.ops) == digest(res.hit_begin, res.hits_arr, res.ops)
        dt = min(ts[1:])
        h2d = 2 * total_bases + 8 * (n_reads + 1)
        d2h = 8 * (n_reads + 1) + 40 * r2.n_hits + 4 * r2.n_ops + 28 * n_reads
        e2e = {"reads_per_s": round(n_reads / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "h2d_bytes": h2d, "d2h_bytes": d2h, "identical_to_device_path": bool(same),
               "what": "mapad_map_batch: host reads in (H2D), D arrays + ordering + search, device-side collect, hit records + edit tracks + counters out (D2H)"}
        r2.close()
        ctx.set_stream(ctypes.c_void_p(stream.cuda_stream))

    # ---- the next row of the path, outside the timed region: SA locate of the hits' rows on the device -------------
    locate = None
    if extras:
        small = res.hits_arr["size"] <= 8
        lo, sz = res.hits_arr["lower"][small].astype(np.uint64), res.hits_arr["size"][small].astype(np.uint64)
        rows = np.unique(np.concatenate([(lo + np.uint64(k))[sz > k] for k in range(8)]))
        pos = ctx.sa_locate(rows)
        pos = ctx.sa_locate(rows)  # second call: SA samples already resident
        l_ms, l_rows, l_steps = ctx.locate_info()
        l_bytes = 128 * l_steps + 16 * l_rows  # one index block per LF step, row in, sample + position out
        locate = {"rows": int(l_rows), "lf_steps": int(l_steps), "kernel_ms": round(l_ms, 4), "rows_per_s": round(l_rows / (l_ms * 1e-3), 1) if l_ms else None,
                  "algorithmic_GB/s": round(l_bytes / (l_ms * 1e-3) / 1e9, 2) if l_ms else None}
        if not args.no_cpu_baseline:
            n_s = int(min(rows.size, 300000))
            t = time.perf_counter()
            want = index.sa_get_batch(rows[:n_s])
            dt = time.perf_counter() - t
            locate["cpu_baseline"] = {"value": round(n_s / dt, 1), "unit": "rows/s", "cores": 1, "kind": "port",
                                      "sample": f"first {n_s} rows, host restatement of SampledSuffixArray::get on one thread"}
            locate["identical_positions"] = bool(np.array_equal(want, pos[:n_s]))

    # ---- hits -> record fields (intervals_to_bam minus BAM encoding; rows a14-a17 of SURVEY 8a), outside the timed region ----------
    post = None
    if extras:
        from mapad_amd import binding as mb
        import ctypes as C

        def records_call():
            out = C.POINTER(mb.RecordsC)()
            t = time.perf_counter()
            rc = mb.lib().mapad_hits_to_records_gpu(ctx.h, res._cptr, seqs.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p),
                                                    offsets.ctypes.data_as(C.c_void_p), None, 0, C.byref(out))
            dt = time.perf_counter() - t
            assert rc == 0
            mb.lib().mapad_records_free(out)
            return dt

        dt_all = records_call()
        post = {"reads_per_s": round(n_reads / dt_all, 1), "wall_s": round(dt_all, 3), "host_threads": min(os.cpu_count() or 1, 64),
                "what": "mapad_hits_to_records_gpu: coordinates (SA walks, contigs, X0/X1, XA candidates) and CIGAR/MD/XA text by kernels over the device-resident hits; flags and MAPQ (exp2f/log10f) on host threads"}

    ctx.close()  # the command-line leg below starts a process with a context of its own on the same GPU: this one's 180 GB of pools must be gone
    mapped_fraction = round(float((np.diff(res.hit_begin.astype(np.int64)) > 0).mean()), 4)
    del res
    # ---- the other single-GPU configurations of BASELINE.json, briefly: C2 (no damage) and C3 (damage model) on the 48 Mbp genome ----------------------
    # (own processes with contexts of their own, after this one's pools are gone; each line is this script's with --no-extras: reads/s, roofline, parity sample)
    secondary = None
    if extras and args.config == "c4" and not args.no_secondary:
        secondary = {}
        # C5's read mix (35-100 bp, 5 % of the reads with an indel, damage model, Phred 20-40) on the 3 Gbp index of this run at the reference's real limits
        # (STACK_LIMIT 2 M frames, EDIT_TREE_LIMIT 10 M nodes): one batch of 200 000 reads through mapad_map_batch, i.e. GPU stages + host tail + collect + fetch.
        # The search cost of such reads is heavy-tailed (a few per thousand make millions of pops): what the line shows is how the GPU and the host threads share them.
        try:
            t = time.perf_counter()
            n5 = int(os.environ.get("MAPAD_BENCH_C5_READS", "200000"))
            s5, q5, o5 = make_reads(synth, genome, n5, 4321 + 5, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
            rp5 = resolve_params(DAMAGE)

            def c5_run(local_world):
                """one mapad_map_batch of the C5 batch; local_world: this process as one rank of that many on its node (its share of the host tail's workers)"""
                workers = mb.lib().mapad_tail_set_local_world(local_world)
                ctx5 = mapad_amd.Context(index, mapad_amd.make_params(rp5), local_rank)
                try:
                    ctx5.set_fetch_d_arrays(False)
                    ctx5.prepare_lengths(sorted(set(np.diff(o5.astype(np.int64)).tolist())))
                    t1 = time.perf_counter()
                    res5 = ctx5.map_batch(s5, q5, o5)
                    dt5 = time.perf_counter() - t1
                    ti5 = ctx5.tail_info()
                    k5 = [float(x) for x in ctx5.kernel_ms()]
                    c5 = [int(x) for x in ctx5.last_counters()]
                finally:
                    ctx5.close()
                pops_all = c5[3]
                ev = {"E_search": c5[0] - ti5["host_e_search"], "N_push": c5[2] - ti5["host_n_push"], "N_pop": c5[3] - ti5["host_pops"], "N_node": c5[4] - ti5["host_n_node"]}
                b5 = 256 * ev["E_search"] + 40 * (ev["N_push"] + ev["N_pop"]) + 8 * ev["N_node"]
                leg = {"reads_per_s": round(n5 / dt5, 1), "wall_s": round(dt5, 3), "reads": n5, "steps": 1, "host_tail_workers": int(workers),
                       "workload": f"C5 read mix: {n5} x 35-100 bp reads, 5 % with an indel, ss 50% deamination model, Phred 20-40, -p 0.03, on the {genome_bp} bp index of this run; "
                                   "STACK_LIMIT / EDIT_TREE_LIMIT at the reference's values (2 000 000 / 10 000 000); one mapad_map_batch call (H2D, D arrays, search stages, host tail, collect, D2H)",
                       "pops_per_read": round(pops_all / n5, 1),
                       "roofline": {"kernel": "search_kernel", "kernel_ms": round(k5[1], 2), "algorithmic_bytes_per_launch": b5, "achieved": round(b5 / (k5[1] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                                    "unit": "GB/s", "frac": round(b5 / (k5[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                                    "note": "events of the reads the host finished are not the kernel's and are left out; the launch lasts as long as its slowest read (a serial chain at ~5.5 us per pop)"},
                       "tail": {"reads": ti5["reads"], "pops_share": round(ti5["host_pops"] / max(pops_all, 1), 5), "gpu_pops_before_hand_over": ti5["gpu_pops"], "host_s_per_step": round(ti5["host_us"] / 1e6, 3),
                                "host_thread_s_per_step": round(ti5["host_thread_us"] / 1e6, 3), "threads": ti5["threads"], "budget_pops": ti5["budget"], "reads_below_budget_idle_worker": ti5.get("reads_idle_tier", 0), "reads_dry_class": ti5["reads_dry_class"],
                                "reads_full_limit": ti5["reads_full_limit"], "seen_while_launch_ran": ti5["seen_live"], "continued_from_gpu_state": ti5["continued"]},
                       "arena_migrations": res5.n_second_pass}
                return leg, res5

            leg, res5 = c5_run(0)
            if not args.no_cpu_baseline:
                reads5 = [s5[int(o5[i]):int(o5[i + 1])].tobytes() for i in range(n5)]
                quals5 = [q5[int(o5[i]):int(o5[i + 1])] for i in range(n5)]
                op5 = ob.make_params(rp5)
                n0 = min(n5, 4 * cores + 256)
                t2 = time.perf_counter()
                oidx.map_batch(op5, reads5[:n0], quals5[:n0], n_threads=cores)
                n_s = int(min(n5, max(n0, n0 / max(time.perf_counter() - t2, 1e-3) * 6.0)))
                t2 = time.perf_counter()
                o5r = oidx.map_batch(op5, reads5[:n_s], quals5[:n_s], n_threads=cores)
                dto = time.perf_counter() - t2
                hb = res5.hit_begin[:n_s + 1]
                nh, no = int(hb[-1]), int(o5r.op_offsets[-1])
                ok5 = (np.array_equal(hb, o5r.hit_offsets) and np.array_equal(res5.hits_arr["lower"][:nh], o5r.intervals[:, 0]) and np.array_equal(res5.hits_arr["lower_rev"][:nh], o5r.intervals[:, 1])
                       and np.array_equal(res5.hits_arr["size"][:nh], o5r.intervals[:, 2]) and np.array_equal(res5.hits_arr["score"][:nh].view(np.uint32), o5r.scores.view(np.uint32))
                       and np.array_equal(res5.ops[:no], o5r.ops))
                cc = res5.counters[:n_s]
                got5 = np.stack([cc["e_search"], cc["e_darray"], cc["n_push"], cc["n_pop"], cc["n_node"], cc["n_hits"]], axis=1).astype(np.uint64)
                st5 = res5.status[:n_s]
                leg["cpu_baseline"] = {"value": round(n_s / dto, 1), "unit": "reads/s", "cores": cores, "kind": "port", "sample": f"first {n_s} reads of the batch, {dto:.1f} s wall, C++ oracle"}
                leg["parity"] = {"reads_checked": n_s, "bit_identical_hits": bool(ok5), "identical_event_counters": bool(np.array_equal(got5, o5r.counters)),
                                 "heaviest_read_checked_pops": int(o5r.counters[:, 3].max()), "status_words_clean": bool(((st5 & 16) == 0).all())}
                if not (ok5 and leg["parity"]["identical_event_counters"]):
                    log("PARITY FAILURE (C5 leg): GPU + host-tail hits differ from the oracle on the sample")
            leg["wall_s_leg"] = round(time.perf_counter() - t, 1)
            secondary["c5"] = leg
            # The same batch as ONE RANK OF EIGHT sees it (round-5 verdict): eight ranks of a node share its CPUs, so this rank's host tail gets an eighth of the workers
            # (mapad_tail_set_local_world(8): 2 of a 16-CPU box's 14).  Every hand-over trigger looks at the host's backlog, so what the two workers cannot take stays on the GPU.
            try:
                t = time.perf_counter()
                leg8, res8 = c5_run(8)
                same = all(np.array_equal(a, b) for a, b in ((res8.hit_begin, res5.hit_begin), (res8.hits_arr, res5.hits_arr), (res8.ops, res5.ops), (res8.status, res5.status), (res8.counters, res5.counters)))
                leg8["parity"] = {"identical_to_secondary_c5": bool(same), "what": "hit offsets, hit records, edit tracks, status words and the six event counters of all reads equal the c5 leg's (which carries the oracle sample)"}
                if not same:
                    log("PARITY FAILURE (C5 rank-of-8 leg): results differ from the c5 leg's")
                leg8["wall_s_leg"] = round(time.perf_counter() - t, 1)
                secondary["c5_rank_of_8"] = leg8
                del res8
            except Exception as e:
                secondary["c5_rank_of_8"] = {"skipped": f"{type(e).__name__}: {e}"}
            finally:
                mb.lib().mapad_tail_set_local_world(0)
            del res5
        except Exception as e:
            secondary["c5"] = {"skipped": f"{type(e).__name__}: {e}"}
        try:
            del oidx
        except Exception:
            pass
        for cfg in ("c2", "c3"):
            try:
                t = time.perf_counter()
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "5", "--warmup", "1", "--no-extras", "--cpu-seconds", "4"],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                d = json.loads(pr.stdout.strip().splitlines()[-1])
                r = d["roofline"]
                secondary[cfg] = {"reads_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"], "batches_in_flight": d["config"]["batches_in_flight"],
                                  "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "traffic", "traffic_stale", "kernel_ms", "algorithmic_bytes_per_launch", "random_access")},
                                  "solo_launch": r["solo_launch"], "cpu_baseline": d["cpu_baseline"], "parity": d["parity"], "tail": d["tail"], "wall_s": round(time.perf_counter() - t, 1)}
            except Exception as e:
                secondary[cfg] = {"skipped": f"{type(e).__name__}: {e}"}
    # ---- the command line end to end: FASTQ in, BAM out (reader, GPU mapping, records, BAM encoding + BGZF, all overlapped) -----------------
    cli = None
    if extras and args.config in ("c2", "c3", "c4") and not args.no_cli:
        import re
        import shutil
        import tempfile
        from mapad_amd import build as mbuild
        tmp = tempfile.mkdtemp(prefix="mapad_cli_")
        try:
            n_src = min(n_reads, 4_000_000)
            n_cli = n_src * max(1, (24_000_000 if genome_bp >= 1_000_000_000 else 8_000_000) // n_src)  # (3 Gbp: 24 launches of 1 M reads, below)  # the batch's first reads, repeated up to 8 M: 32 chunks of 250 000, so that the fill and drain of the pipeline are a small part of the run
            fa, fq, bam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fastq"), os.path.join(tmp, "out.bam")
            rec = np.empty((n_cli, 114), np.uint8)  # "@NNNNNNNN\n" + 50 bases + "\n+\n" + 50 qualities + "\n"
            rec[:, 0] = ord("@")
            ids = np.arange(n_cli)
            for k in range(8):
                rec[:, 8 - k] = 48 + (ids // 10 ** k) % 10
            reps = n_cli // n_src
            rec[:, 9] = 10; rec[:, 10:60] = np.tile(seqs[:50 * n_src].reshape(n_src, 50), (reps, 1)); rec[:, 60] = 10; rec[:, 61] = ord("+"); rec[:, 62] = 10
            rec[:, 63:113] = np.tile(quals[:50 * n_src].reshape(n_src, 50), (reps, 1)) + 33; rec[:, 113] = 10
            rec.tofile(fq)
            del rec
            exe = mbuild.build_cli()
            index_bytes = None
            if genome_bp <= 200_000_000:  # `mapad-amd index` from the FASTA (GPU suffix sorting inside the command)
                with open(fa, "wb") as f:
                    f.write(b">chr1\n")
                    f.write(genome.tobytes())
                    f.write(b"\n")
                t = time.perf_counter()
                subprocess.check_call([exe, "index", "-g", fa], stderr=subprocess.DEVNULL)
                t_idx, idx_how = time.perf_counter() - t, "mapad-amd index (FASTA -> 7 files)"
            else:  # the index of the timed region, written through mapad_index_save (the seven files of indexing.rs:110-208) and loaded by the command
                need = 4 * len(index) // 2 + (8 << 30)
                if shutil.disk_usage(tmp).free < need:
                    raise RuntimeError(f"not enough scratch space for the index files ({need >> 30} GiB)")
                t = time.perf_counter()
                index.save(fa)
                t_idx, idx_how = time.perf_counter() - t, "mapad_index_save of the index built for the timed region (7 files)"
            index_bytes = sum(os.path.getsize(fa + e) for e in (".tbw", ".tle", ".toc", ".trt", ".tsa", ".tpi", ".tos") if os.path.exists(fa + e))
            model = ["-f", "0.5", "-t", "0.5", "-d", "0.02", "-s", "1.0"] if args.config == "c3" else ["-f", "0", "-t", "0", "-d", "0", "-s", "0"]
            t = time.perf_counter()
            pr = subprocess.run([exe, "map", "-r", fq, "-g", fa, "-o", bam, "-l", "single_stranded", "-p", "0.03", "-D", "0.02", "-i", "0.001", "-x", "1.0",
                                 "--batch_size", "250000", "--force_overwrite"] + model, stderr=subprocess.PIPE, text=True)
            t_map = time.perf_counter() - t
            if pr.returncode != 0:
                raise RuntimeError(f"mapad-amd map exited with {pr.returncode}: {pr.stderr[-400:]}")
            m = re.search(r"(\d+) reads, (\d+) mapped; (\d+) device\(s\); index \+ contexts ([0-9.]+) s, mapping ([0-9.]+) s", pr.stderr)
            cli = {"reads_per_s": round(n_cli / float(m.group(5)), 1), "reads": n_cli, "mapping_s": float(m.group(5)), "index_load_and_contexts_s": float(m.group(4)),
                   "process_wall_s": round(t_map, 2), "mapped": int(m.group(2)), "bam_bytes": os.path.getsize(bam), "index_files_s": round(t_idx, 2), "index_files": idx_how,
                   "index_files_bytes": index_bytes,
                   "steady_state": (lambda mm: {"reads_per_s": float(mm.group(3)), "reads": int(mm.group(1)), "s": float(mm.group(2)), "pipeline_fill_s": float(mm.group(4)),
                                                "what": "the run without its fill: reads behind the first chunk / time from the first chunk's records written to the last chunk's"} if mm else None)(
                       re.search(r"steady state: (\d+) reads in ([0-9.]+) s behind the first chunk \((\d+) reads/s\); pipeline fill ([0-9.]+) s", pr.stderr)),
                   "stage_busy_s": (lambda mm: {"reader": float(mm.group(1)), "device_worker": float(mm.group(2)), "writer": float(mm.group(3))} if mm else None)(
                       re.search(r"reader ([0-9.]+) s, device worker 0 ([0-9.]+) s, writer ([0-9.]+) s", pr.stderr)),
                   "what": "mapad-amd map: FASTQ -> BAM, --batch_size 250000 (the reference's default; on a 3 Gbp index four chunks go to the device as one launch), up to 4 launches in flight on one GPU; reader, MAPQ and BGZF on host threads, coordinates and record text on the GPU"}
        except Exception as e:  # the leg is a report, not the metric: say why it is missing
            cli = {"skipped": f"{type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if rank == 0:
        total_reads = total_reads_per_step * args.steps
        model = "no-damage" if args.config in ("c1", "c2", "c4") else "ss 50% deamination"
        line = {
            "metric": "mapped reads/sec (50 bp, -p 0.03)", "value": round(total_reads / elapsed, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "u64+f32", "data": "synthetic",
            "config": {"workload": f"{args.config.upper()}: synthetic genome ({genome_bp} bp, n = {len(index)} BWT rows), "
                                   f"{n_chunk} x {'50' if args.config != 'c5' else '35-100'} bp reads {'per GPU' if not (world > 1 and args.scaling == 'strong') else 'in one chunk, cut into contiguous slices'}, -p 0.03, "
                                   f"{model} model{', 5 % of the reads with an indel' if args.config == 'c5' else ''}",
                       "reads_per_gpu": n_reads, "genome_bp": genome_bp, "index_bytes_hbm": int((len(index) + 255) // 256 * 128),
                       "batches_in_flight": args.depth,
                       "parallelism": f"reads sharded over {world} GPUs, index replicated, read-ordered hit records gathered on rank 0 (RCCL p2p)" if world > 1 else "1 GPU",
                       "mapped_fraction": mapped_fraction,
                       "index_build_s": round(t_index, 1), "index_build": "GPU suffix sorting (prefix doubling over radix sorts) + host text preparation"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            # the heavy tail (csrc/host_tail.hpp): reads past the pop budget on the GPU, finished by host threads with the kernel's own search step; inside the timed region
            "tail": {"reads": int(sum(t["reads"] for t in tail_timed)), "reads_per_step": round(sum(t["reads"] for t in tail_timed) / max(len(tail_timed), 1), 1),
                     "pops_share": round(tail_last["host_pops"] / max(n_pop_all, 1), 5), "gpu_pops_before_hand_over": tail_last["gpu_pops"],
                     "host_s_per_step": round(sum(t["host_us"] for t in tail_timed) / max(len(tail_timed), 1) / 1e6, 3),
                     # thread-seconds inside the reads: far below host_s x threads = the host waited for the GPU's hand-overs; close to it = the host's CPUs set the pace
                     "host_thread_s_per_step": round(sum(t.get("host_thread_us", 0) for t in tail_timed) / max(len(tail_timed), 1) / 1e6, 3), "budget_pops": tail_last["budget"],
                     # why reads went to the host: past the pop budget; below it while a worker was idle (round 6); their arena class was dry while the host had room (round 5); no growable arena could hold them
                     "reads_below_budget_idle_worker": int(sum(t.get("reads_idle_tier", 0) for t in tail_timed)),
                     "reads_dry_class": int(sum(t.get("reads_dry_class", 0) for t in tail_timed)), "reads_full_limit": int(sum(t.get("reads_full_limit", 0) for t in tail_timed)),
                     "seen_while_launch_ran": int(sum(t.get("seen_live", 0) for t in tail_timed)),
                     "where": f"{tail_last['threads']} host threads, search_core.hpp compiled for the host (the kernel's source; reads handed over from a grown arena continue from the GPU's state, others start over), overlapped with the GPU's bulk" if tail_last["budget"] else "off"},
            "secondary": secondary, "e2e": e2e, "cli": cli, "sa_locate": locate, "post_search": post,
        }
        if gather_check is not None:
            line["gather"] = gather_check
        print(json.dumps(line), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
