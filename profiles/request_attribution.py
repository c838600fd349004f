#!/usr/bin/env python3
"""Which lines are a pop's requests?  (round-4 verdict, lever (c): "attribute the 4.95 requests/pop to {index, node, heap, hits} with a 128-B-line L2 model in
tests/emu (CPU, free)".)

Maps synthetic reads with the HOST build of the kernel's step (tests/emu: the same search_core.hpp / heap_core.hpp, lane-parallel commit emulated, no payload cache)
while every arena access runs through a private LRU line cache per read (tests/emu/emu.cpp: LineCache) — a read slot's share of the L2 (6 lines of 128 B) and of
the Infinity Cache (45 lines), and a generous 360 lines for comparison, in 64-byte and 128-byte lines.  Per pop and structure: accesses, read misses (requests to
the next level), write-backs.  Index lines are shared by all reads and are reported as touches per pop (2 per extension) only.

    python profiles/request_attribution.py [--genome-bp 48000000] [--reads 20000] [--out profiles/r05/request_attribution.json]

CPU only; test infrastructure (loads tests/emu)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
KINDS = ("index", "heap", "node", "hits", "other")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=48_000_000)
    ap.add_argument("--reads", type=int, default=20_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05", "request_attribution.json"))
    ap.add_argument("--heap-layout", type=int, default=0, help="1: the model sees the arena's heap levels in the subtree-contiguous candidate layout (tests/emu/emu.cpp: "
                                                               "subtree_slot) — what would a re-laid heap save?  The mapping itself is unchanged")
    args = ap.parse_args()
    os.environ["MAPAD_ATTR_HEAP_LAYOUT"] = str(args.heap_layout)
    os.environ["MAPAD_EMU_PAYLOAD_CACHE"] = "0"
    import emu_util
    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve

    t0 = time.time()
    g = synth.genome(args.genome_bp, seed=1234)
    idx = mapad_amd.Index.build([("chr1", g)], seed=1234)
    print(f"index over {args.genome_bp} bp (host) in {time.time() - t0:.1f} s", flush=True)
    L = emu_util.lib()
    L.emu_attr_begin.argtypes = [C.c_void_p, C.c_uint32]
    L.emu_attr_end.argtypes = [C.c_void_p]
    cfgs = [(6, 12), (6, 90), (6, 720), (7, 6), (7, 45), (7, 360)]  # (log2 line bytes, lines): L2 share, Infinity-Cache share, generous
    out = {"what": "per pop: arena accesses / read misses / write-backs of a private LRU line cache per read, by structure; host build of the kernel's step",
           "genome_bp": args.genome_bp, "reads": args.reads, "heap_layout_seen_by_the_model": "subtree-contiguous blocks (candidate)" if args.heap_layout else "implicit array (as built)", "mixes": {}}
    mixes = {"c2_like": (NO_DAMAGE, dict(qual=40)), "c3_like": (DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)))}
    for name, (prm, kw) in mixes.items():
        seqs, quals, offsets = synth.reads(g, args.reads, 50, seed=4321 + len(name), **kw)
        cfg = np.array(cfgs, np.uint32).reshape(-1)
        L.emu_attr_begin(cfg.ctypes.data, len(cfgs))
        t1 = time.time()
        res = emu_util.map_batch(idx, mapad_amd.make_params(resolve(prm)), seqs, quals, offsets, node_cap=1 << 17, heap_cap=1 << 17)
        buf = np.zeros(len(cfgs) * 15 + 3 + 32, np.uint64)
        L.emu_attr_end(buf.ctypes.data)
        pops = int(buf[len(cfgs) * 15])
        idx_touch = int(buf[len(cfgs) * 15 + 1])
        near = int(buf[len(cfgs) * 15 + 2])
        levels = buf[len(cfgs) * 15 + 3:]
        c = res.counters
        mix = {"pops": pops, "pops_per_read": round(pops / args.reads, 1), "index_line_touches_per_pop": round(idx_touch / pops, 3), "near_(LDS)_accesses_per_pop": round(near / pops, 2),
               "pushes_per_pop": round(float(c["n_push"].sum()) / pops, 3), "nodes_per_pop": round(float(c["n_node"].sum()) / pops, 3),
               "arena_heap_reads_per_pop_by_level": {str(l): round(int(v) / pops, 4) for l, v in enumerate(levels) if v}, "caches": {}}
        for i, (sh, cap) in enumerate(cfgs):
            b = buf[i * 15:(i + 1) * 15].reshape(5, 3)
            d = {}
            for k, kind in enumerate(KINDS):
                if b[k].sum():
                    d[kind] = {"accesses": round(int(b[k, 0]) / pops, 3), "read_misses": round(int(b[k, 1]) / pops, 3), "write_backs": round(int(b[k, 2]) / pops, 3)}
            d["arena_requests_per_pop"] = round(float(b[1:, 1].sum() + b[1:, 2].sum()) / pops, 3)
            mix["caches"][f"{1 << sh}B_lines_x{cap}"] = d
        print(name, f"{time.time() - t1:.1f} s", json.dumps(mix, indent=0)[:1500], flush=True)
        out["mixes"][name] = mix
        del res
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
