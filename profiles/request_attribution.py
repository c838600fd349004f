#!/usr/bin/env python3
"""Which lines are a pop's requests behind the L2?  (round-4 verdict, lever (c); corrected in round 6 after the round-5 verdict measured what the first model left out.)

Maps synthetic reads with the HOST build of the kernel's step (tests/emu: the same search_core.hpp / heap_core.hpp, lane-parallel commit emulated, no payload cache)
while every index and arena access runs through a private LRU line cache per read (tests/emu/emu.cpp: LineCache, 128-byte lines).  Round 6: INDEX lines go through
the same LRU as the arena's (in the real L2 they evict arena lines every pop), the two rank queries of a pop that fall into one line count once
(`index_distinct_lines_per_pop`), write-backs are counted per dirty 64-byte half of a line (the memory side's write requests are 32 or 64 bytes), and the capacity —
a read slot's share of the L2 — is the calibration knob: swept, and the row closest to the measured PMC figures (reads and 64-byte write units per pop behind the L2:
--pmc-read / --pmc-write) is marked.  --layouts 0,1: the implicit heap array and the subtree-block layout (csrc/heap_core.hpp: MAPAD_SUBTREE_HEAP), each with its own
host build.  --index-from-gpu: build the index with the GPU indexer (a 3 Gbp point on the GPU box's host; the mapping itself stays on the CPU).

    python profiles/request_attribution.py [--genome-bp 48000000] [--reads 20000] [--out profiles/r06/request_attribution.json]

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


def cfg_name(sh, cap):
    return ("lru" if not ((sh >> 8) & 1) else "random") + (f"+clean{sh >> 16}" if sh >> 16 else "") + f":{cap}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=48_000_000)
    ap.add_argument("--reads", type=int, default=20_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "request_attribution.json"))
    ap.add_argument("--layouts", default="0,1", help="MAPAD_SUBTREE_HEAP values to model (0: implicit array, 1: subtree blocks)")
    ap.add_argument("--caps", default="2,3,4,6,8,12,16,24,32,45,64", help="capacities (128-byte lines per read) to sweep")
    ap.add_argument("--mixes", default="c2_like,c3_like")
    ap.add_argument("--clean", default="0,2,3,4", help="LineCache::clean_after values to sweep (0 = write back on eviction only)")
    ap.add_argument("--index-from-gpu", action="store_true")
    ap.add_argument("--pmc", default="c2_like:2.71:1.60,c3_like:3.90:2.19", help="measured requests behind the L2 per pop of the implicit-array build, MIX:READS:WRITE_64B_UNITS "
                                                                              "(profiles/r06/ab_pmc_layout.txt: FETCH_SIZE / 64 B and WRITE_SIZE / 64 B per pop)")
    ap.add_argument("--pmc-subtree", default="c2_like:2.29:1.62,c3_like:3.06:2.13", help="the same of the subtree-block build: what the calibrated model is then asked to PREDICT")
    args = ap.parse_args()
    os.environ["MAPAD_EMU_PAYLOAD_CACHE"] = "0"
    import emu_util
    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve

    pmc = {m.split(":")[0]: (float(m.split(":")[1]), float(m.split(":")[2])) for m in args.pmc.split(",") if m}
    pmc_sub = {m.split(":")[0]: (float(m.split(":")[1]), float(m.split(":")[2])) for m in args.pmc_subtree.split(",") if m}
    t0 = time.time()
    g = synth.genome(args.genome_bp, seed=1234)
    idx = mapad_amd.Index.build([("chr1", g)], seed=1234, device=0 if args.index_from_gpu else None)
    print(f"index over {args.genome_bp} bp in {time.time() - t0:.1f} s", flush=True)
    caps = [int(c) for c in args.caps.split(",")]
    # replacement: LRU or a random victim (tests/emu/emu.cpp: LineCache::random_victim); dirty lines written back when evicted, or already once `clean` other lines were touched after them
    cfgs = [(7 | (pol << 8) | (cl << 16), c) for pol in (0, 1) for cl in [int(x) for x in args.clean.split(",")] for c in caps if cl < c]
    out = {"what": "per pop: accesses / read misses (requests to memory) / write-backs (per dirty 64-byte half line) of a private LRU of 128-byte lines per read, index and arena "
                   "lines alike, by structure; host build of the kernel's step; capacity swept",
           "genome_bp": args.genome_bp, "reads": args.reads, "layouts": {}}
    all_mixes = {"c2_like": (NO_DAMAGE, dict(qual=40)), "c3_like": (DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)))}
    for layout in [int(x) for x in args.layouts.split(",")]:
        L = emu_util.lib(0, layout)
        L.emu_attr_begin.argtypes = [C.c_void_p, C.c_uint32]
        L.emu_attr_end.argtypes = [C.c_void_p]
        lay = {}
        for name in args.mixes.split(","):
            prm, kw = all_mixes[name]
            seqs, quals, offsets = synth.reads(g, args.reads, 50, seed=4321 + len(name), **kw)
            cfg = np.array(cfgs, np.uint32).reshape(-1)
            L.emu_attr_begin(cfg.ctypes.data, len(cfgs))
            t1 = time.time()
            res = emu_util.map_batch(idx, mapad_amd.make_params(resolve(prm)), seqs, quals, offsets, node_cap=1 << 17, heap_cap=1 << 17, subtree=layout)
            buf = np.zeros(len(cfgs) * 15 + 4 + 32, np.uint64)  # (emu_attr_end: 15 words per configuration, then totals and heap levels)
            L.emu_attr_end(buf.ctypes.data)
            pops, idx_touch, near, idx_distinct = (int(buf[len(cfgs) * 15 + k]) for k in range(4))
            levels = buf[len(cfgs) * 15 + 4:]
            c = res.counters
            mix = {"pops": pops, "pops_per_read": round(pops / args.reads, 1), "index_block_touches_per_pop": round(idx_touch / pops, 3), "index_distinct_lines_per_pop": round(idx_distinct / pops, 3),
                   "pushes_per_pop": round(float(c["n_push"].sum()) / pops, 3), "nodes_per_pop": round(float(c["n_node"].sum()) / pops, 3),
                   "arena_heap_reads_per_pop_by_level": {str(l): round(int(v) / pops, 4) for l, v in enumerate(levels) if v}, "by_capacity": {}}
            best = None
            for i, (sh, cap) in enumerate(cfgs):
                b = buf[i * 15:(i + 1) * 15].reshape(5, 3)
                d = {kind: {"read_misses": round(int(b[k, 1]) / pops, 3), "write_backs_64B": round(int(b[k, 2]) / pops, 3)} for k, kind in enumerate(KINDS) if b[k].sum()}
                d["reads_per_pop"] = round(float(b[:, 1].sum()) / pops, 3)
                d["writes_per_pop"] = round(float(b[:, 2].sum()) / pops, 3)
                d["arena_reads_per_pop"] = round(float(b[1:, 1].sum()) / pops, 3)
                if name in pmc and layout == 0:
                    err = max(abs(d["reads_per_pop"] / pmc[name][0] - 1), abs(d["writes_per_pop"] / pmc[name][1] - 1))
                    d["max_rel_error_vs_pmc"] = round(err, 3)
                    if best is None or err < best[0]:
                        best = (err, cfg_name(sh, cap))
                mix["by_capacity"][cfg_name(sh, cap)] = d
            if best:
                mix["calibrated_capacity_lines"] = best[1]
                mix["calibration_max_rel_error"] = round(best[0], 3)
                mix["pmc_reads_writes_per_pop"] = list(pmc[name])
            print(f"layout {layout} {name} {time.time() - t1:.1f} s: distinct index lines/pop {mix['index_distinct_lines_per_pop']}; " +
                  "; ".join(f"{k}: {v['reads_per_pop']} r + {v['writes_per_pop']} w" for k, v in mix["by_capacity"].items()) +
                  (f"; calibrated: {best[1]} lines (max rel. error {best[0]:.2f})" if best else ""), flush=True)
            lay[name] = mix
            del res
        out["layouts"]["subtree_blocks" if layout else "implicit_array"] = lay
    # one setting for every mix: the configuration whose worst relative error (reads or writes per pop, any mix) against the implicit-array build's PMC figures is smallest —
    # and what that setting says about the subtree-block build, against ITS PMC figures (a prediction: the setting was not fitted to them)
    base = out["layouts"].get("implicit_array", {})
    names = [n for n in base if n in pmc]
    if names:
        keys = list(base[names[0]]["by_capacity"])
        worst = {k: max(base[n]["by_capacity"][k]["max_rel_error_vs_pmc"] for n in names) for k in keys if all(k in base[n]["by_capacity"] for n in names)}
        k_best = min(worst, key=worst.get)
        joint = {"setting": k_best, "worst_rel_error": worst[k_best],
                 "implicit_array": {n: {"model": [base[n]["by_capacity"][k_best]["reads_per_pop"], base[n]["by_capacity"][k_best]["writes_per_pop"]], "pmc": list(pmc[n]),
                                        "model_by_structure": {kk: vv for kk, vv in base[n]["by_capacity"][k_best].items() if isinstance(vv, dict)}} for n in names}}
        sub = out["layouts"].get("subtree_blocks", {})
        if sub:
            joint["subtree_blocks_predicted"] = {n: {"model": [sub[n]["by_capacity"][k_best]["reads_per_pop"], sub[n]["by_capacity"][k_best]["writes_per_pop"]], "pmc": list(pmc_sub.get(n, (None, None))),
                                                     "model_by_structure": {kk: vv for kk, vv in sub[n]["by_capacity"][k_best].items() if isinstance(vv, dict)}} for n in names if n in sub}
        out["joint_calibration"] = joint
        print("joint calibration:", json.dumps(joint)[:1500], flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
