#!/usr/bin/env python3
"""bench.py — mapped reads/s of the mapAD hot path on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--genome-bp G] [--reads R] [--config c2|c3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1] "C2"): synthetic chr21-size genome (48 Mbp, i.i.d. ACGT, splitmix64 seed 1234),
1 M synthetic 50 bp reads per GPU (90 % endogenous with 2 % substitutions, 10 % exogenous, Phred 40), `-p 0.03`, no-damage
model (-l single_stranded -f 0 -t 0 -d 0 -s 0 -D 0.02 -i 0.001 -x 1.0), gap_dist_ends 5, max_num_gaps_open 2.
One "step" = one pass of the hot path (D-array kernel + search kernel + large-arena pass) over the batch; reads, index and
score tables are resident in HBM before the timed region.  With N > 1 every rank holds a replica of the index, maps its own
shard of reads (weak scaling: 1 M reads per GPU) and the hit records are gathered on rank 0 over RCCL inside each step.

Also reported on the same line:
  roofline     — dominant kernel, algorithmic bytes (SURVEY §8d formula from the kernels' event counters) / HIP-event time
  cpu_baseline — the C++ oracle (oracle/, a restatement of the reference algorithm; the Rust reference cannot be built
                 here) on all host cores over a bounded sample of the same reads; its hits must equal the GPU's.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class DevArray:
    """__cuda_array_interface__ view of a raw device pointer so torch can wrap library-owned HBM buffers without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-bp", type=int, default=48_000_000)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    n_gpus = world

    import torch
    import torch.distributed as dist

    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve as resolve_params

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in mapad_amd")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    # ---- workload --------------------------------------------------------------------------------------------------------
    t0 = time.time()
    genome = synth.genome(args.genome_bp, seed=1234)
    index = mapad_amd.Index.build([("chr1", genome)], seed=1234)
    t_index = time.time() - t0
    if args.config == "c2":
        prm, kw, cfg_id = NO_DAMAGE, dict(qual=40), 2
    elif args.config == "c3":
        prm, kw, cfg_id = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 3
    else:  # the read mix of C5 (35-100 bp, 5 % of the reads with an indel, damage model) on this genome
        prm, kw, cfg_id = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05), 5
    rp = resolve_params(prm)
    params = mapad_amd.make_params(rp)
    seqs, quals, offsets = synth.reads(genome, args.reads, 50, seed=4321 + cfg_id + 1000 * rank, **kw)
    n_reads = args.reads
    log(f"[rank {rank}] genome {args.genome_bp} bp, index built in {t_index:.1f}s, {n_reads} reads")

    stream = torch.cuda.current_stream(dev)
    ctx = mapad_amd.Context(index, params, local_rank)
    ctx.set_stream(ctypes.c_void_p(stream.cuda_stream))
    max_len = int(np.diff(offsets.astype(np.int64)).max())
    ctx.prepare_lengths(sorted(set(np.diff(offsets.astype(np.int64)).tolist())))
    ctx.set_fetch_d_arrays(False)
    d_seqs = torch.from_numpy(seqs).to(dev)
    d_quals = torch.from_numpy(quals).to(dev)
    d_offsets = torch.from_numpy(offsets.view(np.int64)).to(dev)
    torch.cuda.synchronize(dev)

    def gather_hits():
        """Final gather of the hit records on rank 0 (the only exchange of the path): per-read hit count + first-hit index,
        hit pool (10 x u32 per hit) and edit-operation pool, straight out of the library's HBM buffers."""
        if world == 1:
            return None
        from mapad_amd.distributed import gather_hit_records
        p_cnt, p_first, p_hits, p_ops, p_cur = ctx.device_result_ptrs()
        cur = torch.as_tensor(DevArray(p_cur, (2,), "<i8"), device=dev).cpu()
        n_hits, n_ops = int(cur[0]), int(cur[1])
        cnt_first = torch.as_tensor(DevArray(p_cnt, (n_reads,), "<u4"), device=dev).view(torch.int32)
        first = torch.as_tensor(DevArray(p_first, (n_reads,), "<u4"), device=dev).view(torch.int32)
        hits = torch.as_tensor(DevArray(p_hits, (max(n_hits, 1) * 10,), "<u4"), device=dev).view(torch.int32)[:n_hits * 10]
        ops = torch.as_tensor(DevArray(p_ops, (max(n_ops, 1),), "<u4"), device=dev).view(torch.int32)[:n_ops]
        return gather_hit_records(torch.cat([cnt_first, first]), hits, ops, rank, world, device=dev)

    def step():
        ctx.map_batch_device(d_seqs.data_ptr(), d_quals.data_ptr(), d_offsets.data_ptr(), n_reads, max_len)
        return gather_hits()

    kernel_ms = []
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gathered = None
    for _ in range(args.steps):
        gathered = step()
        kernel_ms.append(ctx.kernel_ms())  # HIP events on the launch stream; waits for this step's last kernel
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # torch <-> library interop used by the multi-GPU gather: wrap the library's cursor words without a copy and cross-check them
    p_cur = ctx.device_result_ptrs()[4]
    cur_view = torch.as_tensor(DevArray(p_cur, (2,), "<i8"), device=dev).cpu().numpy()

    # ---- roofline of the dominant kernel (rank 0) -------------------------------------------------------------------------
    res = ctx.fetch()
    counters = ctx.last_counters()
    assert int(cur_view[0]) == res.n_hits and int(cur_view[1]) == res.n_ops, "device-pointer interop check failed"
    e_search, e_darray, n_push, n_pop, n_node, n_hit_events = [int(x) for x in counters]
    total_bases = int(offsets[-1])
    bytes_darray = 256 * e_darray + 6 * total_bases                      # 2 x 128-B index blocks per extension + read/qual in, D out
    bytes_search = 256 * e_search + 40 * (n_push + n_pop) + 8 * n_node    # + 40-B frames through the heap, 8-B tree nodes
    km = np.array(kernel_ms, dtype=np.float64)
    ms_darray, ms_search, ms_pass2 = km.mean(axis=0)
    dominant = "search_kernel" if ms_search >= ms_darray else "darray_kernel"
    dom_bytes, dom_ms = (bytes_search, ms_search) if dominant == "search_kernel" else (bytes_darray, ms_darray)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            key = f"{args.config}:{args.genome_bp}:{args.reads}"
            traffic = tj.get(key, {}).get(dominant)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms": round(float(dom_ms), 4),
                "all_kernels": {"darray_kernel": {"ms": round(float(ms_darray), 4), "bytes": bytes_darray, "GB/s": round(bytes_darray / (ms_darray * 1e-3) / 1e9, 2)},
                                "search_kernel": {"ms": round(float(ms_search), 4), "bytes": bytes_search, "GB/s": round(bytes_search / (ms_search * 1e-3) / 1e9, 2)},
                                "search_kernel_last_pass": {"ms": round(float(ms_pass2), 4), "arena_migrations": res.n_second_pass, "reads": res.n_third_pass}},
                "whole_step_GB/s": round((bytes_darray + bytes_search) / ((ms_darray + ms_search + ms_pass2) * 1e-3) / 1e9, 2),
                # secondary bound of SURVEY 8(d): dependent random 128-byte index lines per second (2 per extension)
                "index_lines_per_s": {"search_kernel": round(2 * e_search / (ms_search * 1e-3), 1), "darray_kernel": round(2 * e_darray / (ms_darray * 1e-3), 1)},
                "events": {"E_search": e_search, "E_darray": e_darray, "N_push": n_push, "N_pop": n_pop, "N_node": n_node}}

    # ---- CPU baseline + parity on a bounded sample (rank 0) ------------------------------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding as ob  # the checker / CPU baseline: only ever used in this leg
        cores = os.cpu_count() or 1
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
                         f"Occ k=128, min-max heap, slab tree), {cores} threads"}
        # parity of the GPU result on that sample: hit counts, intervals, f32 score bits, edit tracks
        hb = res.hit_begin[:n_sample + 1]
        nh = int(hb[-1])
        ok = (np.array_equal(hb, ores.hit_offsets) and np.array_equal(res.hits_arr["lower"][:nh], ores.intervals[:, 0])
              and np.array_equal(res.hits_arr["size"][:nh], ores.intervals[:, 2])
              and np.array_equal(res.hits_arr["score"][:nh].view(np.uint32), ores.scores.view(np.uint32)))
        n_ops = int(ores.op_offsets[-1])
        ok = ok and np.array_equal(res.ops[:n_ops], ores.ops)
        parity = {"reads_checked": n_sample, "bit_identical_hits": bool(ok)}
        if not ok:
            log("PARITY FAILURE: GPU hits differ from the oracle on the sample")

    # ---- the next row of the path, outside the timed region: SA locate of the hits' rows on the device -------------
    locate = None
    if rank == 0:
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
    if rank == 0:
        from mapad_amd import binding as mb
        import ctypes as C

        def records_call():
            out = C.POINTER(mb.RecordsC)()
            t = time.perf_counter()
            rc = mb.lib().mapad_hits_to_records_gpu(ctx.h, res._cptr, seqs.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p),
                                                    offsets.ctypes.data_as(C.c_void_p), None, 0, C.byref(out))
            dt = time.perf_counter() - t
            assert rc == 0
            n_mapped = sum(1 for i in range(0, int(out.contents.n), max(1, int(out.contents.n) // 1000)) if out.contents.recs[i].mapped)
            mb.lib().mapad_records_free(out)
            return dt, n_mapped

        dt_all, _ = records_call()
        post = {"reads_per_s": round(n_reads / dt_all, 1), "wall_s": round(dt_all, 3), "host_threads": min(os.cpu_count() or 1, 64),
                "what": "mapad_hits_to_records_gpu: SA locate kernel + coordinates, MAPQ, CIGAR/MD/XA strings on host threads"}
        if not args.no_cpu_baseline:
            os.environ["MAPAD_POSTPROC_THREADS"] = "1"
            dt_one, _ = records_call()
            del os.environ["MAPAD_POSTPROC_THREADS"]
            post["one_host_thread_reads_per_s"] = round(n_reads / dt_one, 1)

    if rank == 0:
        total_reads = n_reads * n_gpus * args.steps
        line = {
            "metric": "mapped reads/sec (50 bp, -p 0.03)", "value": round(total_reads / elapsed, 1), "unit": "reads/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64+f32", "data": "synthetic",
            "config": {"workload": f"{args.config.upper()}: synthetic genome ({args.genome_bp} bp, n = {len(index)}), "
                                   f"{n_reads} x {'50' if args.config != 'c5' else '35-100'} bp reads per GPU, -p 0.03, "
                                   f"{'no-damage' if args.config == 'c2' else 'ss 50% deamination'} model{', 5 % of the reads with an indel' if args.config == 'c5' else ''}",
                       "reads_per_gpu": n_reads, "genome_bp": args.genome_bp, "index_bytes_hbm": int((len(index) + 255) // 256 * 128),
                       "parallelism": f"reads sharded over {n_gpus} GPU(s), index replicated, hit records gathered on rank 0" if n_gpus > 1 else "1 GPU",
                       "mapped_fraction": round(float((np.diff(res.hit_begin.astype(np.int64)) > 0).mean()), 4),
                       "index_build_s": round(t_index, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "sa_locate": locate, "post_search": post,
        }
        if gathered is not None:
            line["config"]["gathered_hit_records"] = int(sum(int(g[1].numel()) // 10 for g in gathered))
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
