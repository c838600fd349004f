#!/usr/bin/env python3
"""Dev harness: times the host tail's search (mapad_amd/csrc/host_tail.hpp: tail_search) on the heaviest reads of the C5 mix, on host threads, no GPU needed.
  python profiles/dev/tail_bench.py [--genome-bp G] [--reads N] [--threads T] [--probe-pops P] [--max-pops M] [--top K]
Phase 1 runs every read up to --probe-pops pops to find the heavy ones; phase 2 maps the --top heaviest fully (or up to --max-pops) and reports pops/s per thread."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def throttled():
    """seconds this cgroup has been throttled by its CPU-time quota so far"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=48_000_000)
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--probe-pops", type=int, default=1 << 17)
    ap.add_argument("--max-pops", type=int, default=0)
    ap.add_argument("--top", type=int, default=32)
    ap.add_argument("--interleave", type=int, default=1)
    ap.add_argument("--flags", default="")
    ap.add_argument("--single", type=int, default=0)
    ap.add_argument("--device", type=int, default=None, help="build the index on this GPU (seconds) instead of on the host")
    ap.add_argument("--variants", default="", help="comma-separated PIN:PREFETCH settings (MAPAD_TAIL_PIN, MAPAD_TAIL_PREFETCH) to run the heavy phase with, e.g. 0:10,1:10,1:21")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--probe2-pops", type=int, default=0, help="second probe: the --probe2-top heaviest reads of the first are run to this many pops, and --top is taken from them")
    ap.add_argument("--probe2-top", type=int, default=64)
    ap.add_argument("--sel", default="", help="comma-separated read numbers: skip the probe")
    ap.add_argument("--poisson", type=float, default=0.03, help="-p: smaller = more mismatches allowed = bigger searches (0.03 is C5's; on a 48 Mbp genome no read reaches the limits with it)")
    args = ap.parse_args()
    import mapad_amd
    from mapad_amd import binding as mb, synth
    from mapad_amd.presets import DAMAGE, resolve

    import emu_util
    L = emu_util.tail_bench_lib(tuple(args.flags.split()))

    prefix = f"/tmp/tail_bench_{args.genome_bp}"
    genome = synth.genome(args.genome_bp, seed=1234)
    t = time.time()
    if args.device is not None:
        index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=args.device)
    elif any(f.startswith(os.path.basename(prefix) + ".") for f in os.listdir("/tmp")):
        index = mapad_amd.Index.open(prefix)
    else:
        index = mapad_amd.Index.build([("chr1", genome)], seed=1234)
        index.save(prefix)
    print(f"index: {time.time() - t:.1f} s, {len(index)} rows", file=sys.stderr)
    seqs, quals, offsets = synth.reads(genome, args.reads, 50, seed=4321 + 5, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    params = mapad_amd.make_params(resolve(dict(DAMAGE, poisson_threshold=args.poisson)))
    blocks, nb, less, sent = index.device_view()
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8); quals = np.ascontiguousarray(quals, dtype=np.uint8); offsets = np.ascontiguousarray(offsets, dtype=np.uint64)

    def run(sel, max_pops):
        sel = np.ascontiguousarray(sel, dtype=np.uint32)
        pops = np.zeros(sel.size, dtype=np.uint64); status = np.zeros(sel.size, dtype=np.uint32); dig = np.zeros(sel.size, dtype=np.uint64); rs = np.zeros(sel.size, dtype=np.float64)
        secs = L.tail_bench(blocks, nb, len(index), less.ctypes.data, sent.ctypes.data, C.byref(params), seqs.ctypes.data, quals.ctypes.data, offsets.ctypes.data,
                            offsets.size - 1, sel.ctypes.data, sel.size, args.threads, max_pops, args.interleave, pops.ctypes.data, status.ctypes.data, dig.ctypes.data, rs.ctypes.data)
        run.read_secs = rs
        return secs, pops, status, dig

    secs, pops, status, _ = run(np.arange(args.reads if not args.sel else 1), args.probe_pops)
    print(f"probe: {args.reads} reads, {int(pops.sum())} pops in {secs:.1f} s ({pops.sum() / secs / args.threads / 1e6:.2f} M pops/s/thread); "
          f"{int((pops >= args.probe_pops).sum())} reads reach {args.probe_pops} pops", file=sys.stderr)
    if args.probe2_pops and not args.sel:
        cand = np.argsort(-pops.astype(np.int64), kind="stable")[:args.probe2_top]
        secs, p2, _, _ = run(cand, args.probe2_pops)
        print(f"probe 2: {cand.size} reads to {args.probe2_pops} pops in {secs:.1f} s; {int((p2 >= args.probe2_pops).sum())} reach it", file=sys.stderr)
        pops = np.zeros_like(pops)
        pops[cand] = p2
    heavy = np.argsort(-pops.astype(np.int64), kind="stable")[:args.top] if not args.sel else np.array([int(x) for x in args.sel.split(',')])
    if args.single:  # the heaviest reads one by one on one thread: per-read cost per pop
        saved, args.threads = args.threads, 1
        for r in heavy[:args.single]:
            secs, hp, hs, dig = run(np.array([r]), args.max_pops)
            print(f"  read {int(r)} (L = {int(offsets[r + 1] - offsets[r])}): {int(hp[0])} pops in {secs:.2f} s = {secs / max(int(hp[0]), 1) * 1e6:.3f} us/pop, status {int(hs[0])}")
        args.threads = saved
    for var in (args.variants.split(",") if args.variants else [""]):
        if var:
            os.environ["MAPAD_TAIL_PIN"], os.environ["MAPAD_TAIL_PREFETCH"] = var.split(":")
        for _ in range(args.repeat):
            thr0 = throttled()
            secs, hp, hs, dig = run(heavy, args.max_pops)
            print(f"  cgroup throttled {throttled() - thr0:.2f} s during this run", file=sys.stderr)
            print(f"heavy[{var or 'env'}]: {heavy.size} reads on {args.threads} threads, {int(hp.sum())} pops in {secs:.2f} s: {run.read_secs.sum() / hp.sum() * 1e6:.3f} us/pop (thread time); reads at >= 5 M pops: {run.read_secs[hp >= 5_000_000].sum() / max(int(hp[hp >= 5_000_000].sum()), 1) * 1e6:.3f} us/pop; "
                  f"max {int(hp.max())} pops; reads >= 5 M pops: {heavy[hp >= 5_000_000].tolist()}; digest {int(np.bitwise_xor.reduce(dig)):016x}", flush=True)


if __name__ == "__main__":
    main()
