#!/usr/bin/env python3
"""Full-size parity audit of BASELINE.json configs[3] (C4): ALL reads of the bench batch — 10 M x 50 bp on the 3 Gbp index — mapped on the GPU through the C ABI
and by the CPU oracle, compared read by read: hit count, BinaryHeap array order, (lower, lower_rev, size), f32 score bits, edit tracks and the six event counters.

  python profiles/audit_c4.py [--reads N] [--genome-bp G] [--chunk C] [--out profiles/r04/c4_full_parity.json]

The oracle (test infrastructure) runs on all host cores, chunk by chunk; the JSON carries the two sha256 digests (GPU side, oracle side) over the compared arrays,
the number of reads that differ (with the first few read numbers) and the wall times.  One-off: ~30 min on the GPU box, almost all of it the oracle.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def records_audit(args, ctx, index, genome, rp, res, seqs, quals, offsets):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding as ob
    from parity_util import canonical_records, check_ungapped_records_against_the_text, compare_records, oracle_records_from_product_hits, oracle_threads, records_digest
    n_reads = len(offsets) - 1
    t0 = time.time()
    recs, text = ctx.hits_to_records(res, seqs, quals, offsets, seed=0, as_arrays=True)
    t_dev = time.time() - t0
    t0 = time.time()
    oidx = ob.OracleIndex.from_bwt(index.bwt(), "$ACGTX", 128)
    sample, er, ev = index.sampled_sa()
    oidx.set_sampled_sa(sample, 32, er, ev)  # the oracle walks its own byte BWT / Occ from the product's samples (which test_gpu_index.py checks against the text and the host SA-IS)
    for name, s, e in index.contigs():
        oidx.add_contig(s, e, name)
    t_struct = time.time() - t0
    op = ob.make_params(rp)
    h_dev, h_ora = hashlib.sha256(), hashlib.sha256()
    bad, first_bad, per_field_total, t_ora = 0, [], {}, 0.0
    for lo in range(0, n_reads, args.chunk):
        hi = min(lo + args.chunk, n_reads)
        t = time.time()
        orecs, otext = oracle_records_from_product_hits(oidx, op, res, seqs, quals, offsets, lo, hi, n_threads=oracle_threads())
        t_ora += time.time() - t
        cd, co = canonical_records(recs[lo:hi], text, oracle_side=False), canonical_records(orecs, otext, oracle_side=True)
        h_dev.update(records_digest(cd).encode()); h_ora.update(records_digest(co).encode())
        nb, fb, pf = compare_records(cd, co)
        bad += nb
        first_bad += [lo + i for i in fb][:max(0, 10 - len(first_bad))]
        for k, v in pf.items():
            per_field_total[k] = per_field_total.get(k, 0) + v
        print(f"  records [{lo}, {hi}): {nb} differ; oracle {t_ora:.0f} s so far", file=sys.stderr, flush=True)
    t0 = time.time()
    checked, failed = check_ungapped_records_against_the_text(genome, recs, text, seqs, offsets)
    mapped = recs["mapped"] != 0
    out = {"config": f"C4: synthetic genome ({args.genome_bp} bp, n = {len(index)} BWT rows), {n_reads} x 50 bp reads (the batch of bench.py --config c4), -p 0.03, no-damage model",
           "what": "device-built records (mapad_hits_to_records_gpu: records_kernel, text_kernel, locate_kernel; MAPQ and flags on the host) of every read vs the oracle's intervals_to_record "
                   "(oracle/mapad_oracle.hpp restating mapping.rs:402-718, record.rs:269-449) over the same hits with the same per-read stand-ins for rand::rng() (seed 0)",
           "compared": "per read: flags, tid, POS, strand, MAPQ, AS bits, XS bits / presence, NM, X0, X1, XT, CIGAR, MD, XA",
           "reads": n_reads, "mapped": int(mapped.sum()), "reverse_strand (suffix positions >= 2^32 in the 6e9-row text)": int((recs["reverse"][mapped] != 0).sum()),
           "with_XA": int((recs["xa_len"][mapped] > 0).sum()), "X0_above_1": int((recs["x0"][mapped] > 1).sum()), "mapq_histogram": {str(k): int(v) for k, v in zip(*np.unique(recs["mapq"][mapped], return_counts=True))},
           "reads_that_differ": bad, "first_reads_that_differ": first_bad, "per_field_differences": {k: v for k, v in per_field_total.items() if v},
           "sha256_device": h_dev.hexdigest(), "sha256_oracle": h_ora.hexdigest(), "digests_equal": h_dev.hexdigest() == h_ora.hexdigest(),
           "ungapped_records_checked_against_the_text": checked, "of_them_NM_or_POS_or_strand_wrong": failed,
           "device_records_s": round(t_dev, 2), "oracle_structures_s": round(t_struct, 1), "oracle_records_s": round(t_ora, 1), "text_check_s": round(time.time() - t0, 1)}
    os.makedirs(os.path.dirname(args.records) or ".", exist_ok=True)
    json.dump(out, open(args.records, "w"), indent=1)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--genome-bp", type=int, default=3_000_000_000)
    ap.add_argument("--chunk", type=int, default=500_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "c4_full_parity.json"))
    ap.add_argument("--against", default=None, help="a committed audit of the same batch (e.g. profiles/r04/c4_full_parity.json): map on the GPU only and compare the digest of the "
                                                    "results with that audit's ORACLE digest (same arrays, same chunking) — a minute instead of the oracle's 25")
    ap.add_argument("--records", default=None, metavar="OUT.json",
                    help="record-level audit (round 6): the device-built records of ALL reads (mapad_hits_to_records_gpu: records_kernel / text_kernel / locate_kernel, MAPQ and flags "
                         "on the host) against the oracle's intervals_to_record (mapping.rs:402-718, record.rs:269-449 restated) run over the same hits with the same per-read "
                         "stand-ins for rand::rng() — flags, tid, POS, strand, MAPQ, AS / XS bits, NM, X0, X1, XT, CIGAR, MD, XA —, digests of both sides, and every ungapped "
                         "record against the text itself.  Combine with --against: the hits are then shown identical to the oracle's by that audit's digest")
    args = ap.parse_args()

    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import NO_DAMAGE, resolve
    from oracle import binding as ob
    import bench

    t0 = time.time()
    genome = synth.genome(args.genome_bp, seed=1234)
    index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=0)
    t_index = time.time() - t0
    seqs, quals, offsets = bench.make_reads(synth, genome, args.reads, 4321 + 4, qual=40)  # the batch bench.py --config c4 maps (rank 0)
    rp = resolve(NO_DAMAGE)
    ctx = mapad_amd.Context(index, mapad_amd.make_params(rp), 0)
    ctx.set_fetch_d_arrays(False)
    t1 = time.time()
    res = ctx.map_batch(seqs, quals, offsets)
    t_gpu = time.time() - t1
    tail = ctx.tail_info()
    print(f"index {t_index:.1f} s, GPU mapped {args.reads} reads in {t_gpu:.1f} s ({res.n_hits} hits)", file=sys.stderr, flush=True)
    if args.records:
        records_audit(args, ctx, index, genome, rp, res, seqs, quals, offsets)
    ctx.close()

    if args.against:
        prior = json.load(open(args.against))
        assert prior["reads_checked"] == args.reads and prior.get("digests_equal"), "the audit to compare with must be a complete one of the same batch"
        h = hashlib.sha256()
        hb_all = res.hit_begin.astype(np.int64)
        c = res.counters
        for lo in range(0, args.reads, args.chunk):
            hi = min(lo + args.chunk, args.reads)
            h0, h1 = int(hb_all[lo]), int(hb_all[hi])
            g_hits = res.hits_arr[h0:h1]
            o0 = int(g_hits["ops_offset"][0]) if h1 > h0 else 0
            n_ops = int(g_hits["n_ops"].astype(np.int64).sum())
            g_ctr = np.stack([c[k][lo:hi] for k in ("e_search", "e_darray", "n_push", "n_pop", "n_node", "n_hits")], axis=1).astype(np.uint64)
            for a in ((hb_all[lo:hi + 1] - h0).astype(np.uint64), g_hits["lower"], g_hits["lower_rev"], g_hits["size"], g_hits["score"].view(np.uint32), g_hits["n_ops"].astype(np.uint64), res.ops[o0:o0 + n_ops], g_ctr):
                h.update(np.ascontiguousarray(a).tobytes())
        out = {"config": prior["config"], "reads_checked": args.reads, "hits": int(res.n_hits), "sha256_gpu": h.hexdigest(), "sha256_oracle_of": args.against, "sha256_oracle": prior["sha256_oracle"],
               "digests_equal": h.hexdigest() == prior["sha256_oracle"], "gpu_map_batch_s": round(t_gpu, 1), "index_s": round(t_index, 1), "host_tail_reads": tail["reads"],
               "what": "this build's GPU results for all reads of the bench's C4 batch, digested like the audit named in sha256_oracle_of (hit offsets, intervals, score bits, edit-track lengths, "
                       "edit tracks, six event counters per 500 000-read chunk) and compared with the ORACLE digest committed there: equal digests = every read bit-identical to the oracle"}
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
        print(json.dumps(out))
        return 0 if out["digests_equal"] else 1
    oidx = ob.OracleIndex.from_bwt(index.bwt(), "$ACGTX", 128)
    op = ob.make_params(rp)
    cores = bench.host_cpus()
    h_gpu, h_ora = hashlib.sha256(), hashlib.sha256()
    bad_reads, first_bad, checked, hits_ge_2_32 = 0, [], 0, 0
    t_oracle = 0.0
    hb_all = res.hit_begin.astype(np.int64)
    c = res.counters
    for lo in range(0, args.reads, args.chunk):
        hi = min(lo + args.chunk, args.reads)
        sub = offsets[lo:hi + 1]
        reads = [seqs[int(sub[i]):int(sub[i + 1])].tobytes() for i in range(hi - lo)]
        qs = [quals[int(sub[i]):int(sub[i + 1])] for i in range(hi - lo)]
        t = time.time()
        o = oidx.map_batch(op, reads, qs, n_threads=cores)
        t_oracle += time.time() - t
        h0, h1 = int(hb_all[lo]), int(hb_all[hi])
        g_hb = hb_all[lo:hi + 1] - h0
        g_hits = res.hits_arr[h0:h1]
        o0 = int(g_hits["ops_offset"][0]) if h1 > h0 else 0
        n_ops = int(g_hits["n_ops"].astype(np.int64).sum())
        g_ops = res.ops[o0:o0 + n_ops]
        g_ctr = np.stack([c[k][lo:hi] for k in ("e_search", "e_darray", "n_push", "n_pop", "n_node", "n_hits")], axis=1).astype(np.uint64)
        gpu_side = (g_hb.astype(np.uint64), g_hits["lower"], g_hits["lower_rev"], g_hits["size"], g_hits["score"].view(np.uint32), g_hits["n_ops"].astype(np.uint64), g_ops, g_ctr)
        ora_side = (o.hit_offsets.astype(np.uint64), o.intervals[:, 0], o.intervals[:, 1], o.intervals[:, 2], o.scores.view(np.uint32), np.diff(o.op_offsets).astype(np.uint64), o.ops, o.counters.astype(np.uint64))
        for a in gpu_side:
            h_gpu.update(np.ascontiguousarray(a).tobytes())
        for a in ora_side:
            h_ora.update(np.ascontiguousarray(a).tobytes())
        same = all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(gpu_side, ora_side))
        if not same:  # find the reads that differ
            for i in range(hi - lo):
                a0, a1 = int(g_hb[i]), int(g_hb[i + 1])
                b0, b1 = int(o.hit_offsets[i]), int(o.hit_offsets[i + 1])
                ok = (a1 - a0 == b1 - b0) and np.array_equal(g_ctr[i], o.counters[i].astype(np.uint64))
                if ok and a1 > a0:
                    gh, oi = g_hits[a0:a1], slice(b0, b1)
                    ok = (np.array_equal(gh["lower"], o.intervals[oi, 0]) and np.array_equal(gh["lower_rev"], o.intervals[oi, 1]) and np.array_equal(gh["size"], o.intervals[oi, 2])
                          and np.array_equal(gh["score"].view(np.uint32), o.scores[oi].view(np.uint32)))
                    for k in range(a1 - a0):
                        if not ok:
                            break
                        go = int(gh["ops_offset"][k])
                        ok = np.array_equal(res.ops[go:go + int(gh["n_ops"][k])], o.ops[int(o.op_offsets[b0 + k]):int(o.op_offsets[b0 + k + 1])])
                if not ok:
                    bad_reads += 1
                    if len(first_bad) < 10:
                        first_bad.append(lo + i)
        checked = hi
        hits_ge_2_32 += int((g_hits["lower"] >= 2 ** 32).sum())
        print(f"  reads [{lo}, {hi}): {'identical' if same else 'DIFFERENT'}; oracle {t_oracle:.0f} s so far", file=sys.stderr, flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)  # progress on disk: a run that is cut short still says how far it got
        json.dump({"partial": True, "reads_checked": checked, "reads_that_differ": bad_reads, "first_reads_that_differ": first_bad, "oracle_s": round(t_oracle, 1),
                   "sha256_gpu_so_far": h_gpu.hexdigest(), "sha256_oracle_so_far": h_ora.hexdigest()}, open(args.out, "w"), indent=1)
    out = {"config": f"C4: synthetic genome ({args.genome_bp} bp, n = {len(index)} BWT rows), {args.reads} x 50 bp reads (seed 4321 + 4: the batch of bench.py --config c4), -p 0.03, no-damage model",
           "reads_checked": checked, "reads_that_differ": bad_reads, "first_reads_that_differ": first_bad, "hits": int(res.n_hits), "hit_intervals_at_or_above_2^32": hits_ge_2_32,
           "compared": "per read: hit count, BinaryHeap array order, lower, lower_rev, size, f32 score bits, edit tracks, the six event counters",
           "sha256_gpu": h_gpu.hexdigest(), "sha256_oracle": h_ora.hexdigest(), "digests_equal": h_gpu.hexdigest() == h_ora.hexdigest(),
           "pops_per_read": {"mean": round(float(res.counters["n_pop"].mean()), 1), "max": int(res.counters["n_pop"].max()), "reads_above_2^17": int((res.counters["n_pop"] > (1 << 17)).sum())},
           "gpu_map_batch_s": round(t_gpu, 1), "oracle_s": round(t_oracle, 1), "oracle_threads": cores, "index_s": round(t_index, 1), "host_tail_reads": tail["reads"],
           "oracle": "oracle/mapad_oracle.hpp (CPU restatement of the reference algorithm; test infrastructure), byte BWT + Occ k = 128 over the product index's BWT"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))
    return 0 if bad_reads == 0 and out["digests_equal"] else 1


if __name__ == "__main__":
    sys.exit(main())
