#!/usr/bin/env python3
"""1-GPU rehearsal of the multi-GPU gather (round-4 verdict, item 6): how long does a kernel of RCCL's shape wait for a CU beside the persistent search wavefronts,
and what does it cost the search?

The RCCL (`nccl`) exchange has never run here (one GPU per box).  Its transfer kernel in this image (ncclDevKernel_Generic, librccl.so.1.0.70200, gfx950 code
object) takes 248-256 VGPRs, 37 664 B of LDS and 256-512 threads per block; a search launch at 11 blocks per CU leaves one wave slot of <= 176 VGPRs and 13 KB of
LDS per CU.  profiles/dev/standin.hip is a copy kernel of that shape (256 threads, 248 VGPRs, 37 664 B of LDS).  This script runs the pipelined C4 loop of bench.py
(10 M reads per step, two batches in flight) and issues the stand-in — 1.1 GB, the size of a rank's records at C4, 16 blocks — on a side stream behind every
submission, for search launches of 11 / 10 / 8 blocks per CU and with 0 / 8 CUs kept free of the search (mapad_ctx_set_reserved_cus: CU-masked streams).

    python profiles/dev/rccl_standin.py [--genome-bp 3000000000] [--reads 10000000] [--steps 4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=3_000_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--bytes", type=int, default=1_100_000_000)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--settings", default="11:0,11:8,10:0,8:0,11:0:nostandin")
    args = ap.parse_args()
    import torch
    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import NO_DAMAGE, resolve
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = C.CDLL(os.path.join(ROOT, "profiles", "dev", "libstandin.so"))
    lib.standin_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    genome = synth.genome(args.genome_bp, seed=1234)
    index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=0)
    seqs, quals, offsets = synth.reads(genome, args.reads, 50, seed=4325, qual=40)
    d_seqs, d_quals, d_offsets = torch.from_numpy(seqs).to(dev), torch.from_numpy(quals).to(dev), torch.from_numpy(offsets.view(np.int64)).to(dev)
    src = torch.zeros(args.bytes // 4, dtype=torch.int32, device=dev)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream(dev)
    # the stand-in alone on the chip
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        with torch.cuda.stream(side):
            e0.record(side)
            assert lib.standin_launch(C.c_void_p(side.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), args.bytes, args.blocks, None) == 0
            e1.record(side)
        torch.cuda.synchronize(dev)
    solo_ms = e0.elapsed_time(e1)
    print(json.dumps({"standin_alone_ms": round(solo_ms, 2), "bytes": args.bytes, "blocks": args.blocks, "GB/s": round(args.bytes / solo_ms / 1e6, 1)}), flush=True)
    for setting in args.settings.split(","):
        parts = setting.split(":")
        per_cu, reserved = int(parts[0]), int(parts[1])
        with_standin = "nostandin" not in parts
        layout = "striped" if "striped" in parts else "blocked"
        os.environ["MAPAD_RESERVED_CU_LAYOUT"] = layout
        os.environ["MAPAD_SEARCH_BLOCKS_PER_CU"] = str(per_cu)
        stream = torch.cuda.current_stream(dev)
        ctx = mapad_amd.Context(index, mapad_amd.make_params(resolve(NO_DAMAGE)), 0)
        ctx.set_stream(C.c_void_p(stream.cuda_stream))
        if reserved:
            ctx.set_reserved_cus(reserved)
        ctx.set_pipeline_depth(2)
        ctx.prepare_lengths([50])
        ctx.set_fetch_d_arrays(False)
        ctx.reserve(args.reads, int(offsets[-1]), 50)
        torch.cuda.synchronize(dev)

        def loop(k, events):
            for i in range(k):
                ctx.map_batch_device(d_seqs.data_ptr(), d_quals.data_ptr(), d_offsets.data_ptr(), args.reads, 50)
                if with_standin:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(side):
                        a.record(side)
                        lib.standin_launch(C.c_void_p(side.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), args.bytes, args.blocks, None)
                        b.record(side)
                    events.append((a, b))
                if i >= 1:
                    ctx.select_batch(1); ctx.compact_device(); ctx.select_batch(0)
            ctx.select_batch(0); ctx.compact_device()

        loop(1, [])
        torch.cuda.synchronize(dev)
        ctx.kernel_history()
        ev = []
        t0 = time.perf_counter()
        loop(args.steps, ev)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        hist = ctx.kernel_history().astype(np.float64)
        out = {"search_blocks_per_cu": per_cu, "reserved_cus": reserved, "reserved_layout": layout if reserved else None, "standin": with_standin, "reads_per_s": round(args.reads * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 1),
               "search_ms_per_launch": [round(float(x), 1) for x in (hist[:, 2] - hist[:, 1])],
               "standin_issue_to_done_ms": [round(a.elapsed_time(b), 1) for a, b in ev], "standin_alone_ms": round(solo_ms, 1)}
        print(json.dumps(out), flush=True)
        ctx.close()
        del ctx


if __name__ == "__main__":
    main()
