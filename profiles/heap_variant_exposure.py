#!/usr/bin/env python3
"""How much of the output depends on the two details of the min-max heap's tie handling that the reference's own tests cannot pin (the crate
`min-max-heap` 1.3.1-alpha.0 @ tov/min-max-heap-rs#76a2141a is not in /root/reference)?

Runs the CPU oracle (oracle/mapad_oracle.hpp: MinMaxHeap::variant) with each of the four readings over the read mixes of BASELINE.json's configs — C2-like (50 bp, no
damage, Phred 40), C3-like (50 bp, ss damage model, Phred 20-40), C5-like (35-100 bp, 5 % of the reads with an indel, damage model) — on a synthetic genome, and
counts, against reading 0 (the product's default build), the reads whose hit intervals / score bits / edit tracks (-> CIGAR, MD, NM) / event counters differ.
The product carries the same switch (csrc/heap_core.hpp: MAPAD_HEAP_VARIANT; mapad_amd/build.py builds libmapad_amd.hvV.so), and each product reading equals the
matching oracle reading (tests/test_host_logic.py on the CPU, tests/test_gpu_heap_variants.py on the GPU): pinning the crate later is a one-flag change.

    python profiles/heap_variant_exposure.py [--reads 100000] [--genome-bp 4000000] [--threads N] [--out profiles/heap_variant_exposure.json]

CPU only (test infrastructure: imports oracle/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--genome-bp", type=int, default=4_000_000)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "heap_variant_exposure.json"))
    args = ap.parse_args()
    from mapad_amd import synth
    from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve
    from oracle import binding as ob

    g = synth.genome(args.genome_bp, seed=1234)
    t0 = time.time()
    oidx = ob.OracleIndex.from_text(g.tobytes(), "$ACGTX", 128)
    print(f"oracle index over {args.genome_bp} bp in {time.time() - t0:.1f} s", flush=True)
    mixes = {
        "c2_like": (NO_DAMAGE, dict(qual=40)),
        "c3_like": (DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))),
        "c5_like": (DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)),
    }
    out = {"what": "reads (of --reads per mix) whose oracle result under heap reading v differs from reading 0; readings: bit 0 = family scan order in a trickle-down "
                   "stride, bit 1 = pop_max takes slot 1 on a tie of slots 1 and 2",
           "genome_bp": args.genome_bp, "reads_per_mix": args.reads, "mixes": {}}
    for name, (prm, kw) in mixes.items():
        seqs, quals, offsets = synth.reads(g, args.reads, 50, seed=77 + len(name), **kw)
        n = len(offsets) - 1
        reads = [seqs[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n)]
        qs = [quals[int(offsets[i]):int(offsets[i + 1])] for i in range(n)]
        rp = resolve(prm)
        res = {}
        for v in range(4):
            t1 = time.time()
            res[v] = oidx.map_batch(ob.make_params(dict(rp, heap_variant=v)), reads, qs, n_threads=args.threads)
            print(f"{name} reading {v}: {n} reads in {time.time() - t1:.1f} s", flush=True)
        base = res[0]
        hb0, ob0 = base.hit_offsets.astype(np.int64), base.op_offsets.astype(np.int64)
        with_indel_track = np.zeros(n, bool)  # reads whose reading-0 result has an indel in some hit's edit track
        kinds = (base.ops >> 24).astype(np.uint8)  # 0 = insertion, 1 = deletion (oracle/binding.py: OP_KINDS)
        indel_pos = np.flatnonzero(kinds <= 1)
        hit_of_op = np.searchsorted(ob0, indel_pos, side="right") - 1
        read_of_hit = np.searchsorted(hb0, hit_of_op, side="right") - 1
        with_indel_track[np.unique(read_of_hit)] = True
        mix = {"mapped_reads": int((np.diff(hb0) > 0).sum()), "reads_with_an_indel_in_a_reported_track": int(with_indel_track.sum()), "vs_reading_0": {}}
        for v in (1, 2, 3):
            r = res[v]
            hb, obv = r.hit_offsets.astype(np.int64), r.op_offsets.astype(np.int64)
            d_hits = np.zeros(n, bool); d_scores = np.zeros(n, bool); d_tracks = np.zeros(n, bool)
            d_count = np.diff(hb) != np.diff(hb0)
            for i in range(n):  # (vectorising this is not worth it: a few hundred thousand short slices)
                a0, a1, b0, b1 = hb0[i], hb0[i + 1], hb[i], hb[i + 1]
                if a1 - a0 != b1 - b0:
                    d_hits[i] = True
                    continue
                if a1 == a0:
                    continue
                if not np.array_equal(base.intervals[a0:a1], r.intervals[b0:b1]):
                    d_hits[i] = True
                if not np.array_equal(base.scores[a0:a1].view(np.uint32), r.scores[b0:b1].view(np.uint32)):
                    d_scores[i] = True
                if not np.array_equal(base.ops[ob0[a0]:ob0[a1]], r.ops[obv[b0]:obv[b1]]) or not np.array_equal(np.diff(ob0[a0:a1 + 1]), np.diff(obv[b0:b1 + 1])):
                    d_tracks[i] = True
            d_counters = (base.counters != r.counters).any(axis=1)
            mix["vs_reading_0"][str(v)] = {"hit_count_differs": int(d_count.sum()), "hit_intervals_differ": int(d_hits.sum()), "score_bits_differ": int(d_scores.sum()),
                                           "edit_tracks_differ": int(d_tracks.sum()), "edit_tracks_differ_among_reads_with_an_indel_track": int((d_tracks & with_indel_track).sum()),
                                           "event_counters_differ": int(d_counters.sum()),
                                           "edit_tracks_differ_pct": round(100.0 * d_tracks.sum() / n, 4), "event_counters_differ_pct": round(100.0 * d_counters.sum() / n, 4)}
            print(name, v, mix["vs_reading_0"][str(v)], flush=True)
        out["mixes"][name] = mix
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
