#!/usr/bin/env python3
"""Dev: the C5 read mix on the 3 Gbp index at the reference's real limits, one batch through the C ABI, with the host tail's per-read log (MAPAD_TAIL_LOG) and
the process's per-thread CPU times — where the host's time goes.  python profiles/dev/tail_1m.py [--reads N] [--genome-bp G]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cg(name):
    try:
        return open("/sys/fs/cgroup/" + name).read().strip().replace("\n", "; ")
    except Exception as e:
        return f"({e})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--genome-bp", type=int, default=3_000_000_000)
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "tail_reads.log"))
    ap.add_argument("--budgets", default="", help="comma-separated pop budgets (mapad_ctx_set_tail_pops) to map the batch with, one after the other; default: the library's")
    ap.add_argument("--ranks", default="", help="comma-separated LOCAL_WORLD[:BACKLOG_BUDGET[:BUDGET_POPS[:IDLE_POPS[:ENV=VAL;ENV=VAL...]]]] settings to map the batch with, one after the other: this process as one rank "
                                                "of LOCAL_WORLD on its node (mapad_tail_set_local_world: its share of the host tail's workers), MAPAD_TAIL_BACKLOG_BUDGET (`u` = unconditional "
                                                "hand-over past the budget, round 5's behaviour; empty = the default, 8 per worker), pop budget, MAPAD_TAIL_POPS_IDLE (pops from which a read leaves while a "
                                                "worker is idle; 0 = off); e.g. 8,8:u,4,1,1:::131072")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    import mapad_amd
    from mapad_amd import synth
    from mapad_amd.presets import DAMAGE, resolve
    t0 = time.time()
    genome = synth.genome(args.genome_bp, seed=1234)
    index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=0)
    print(f"genome + index {time.time() - t0:.1f} s", flush=True)
    seqs, quals, offsets = synth.reads(genome, args.reads, 50, seed=4321 + 5, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    import hashlib
    runs = [(int(b), None, None) for b in args.budgets.split(",")] if args.budgets else [(None, None, None)]
    if args.ranks:
        runs = []
        for spec in args.ranks.split(","):
            f = (spec.split(":") + ["", "", "", ""])[:5]
            runs.append((int(f[2]) if f[2] else None, int(f[0]), f[1], f[3], f[4]))
    runs = [(r + (None, None))[:5] for r in runs]
    extra_set = []
    for budget, lw, backlog, idle, extra in runs:
        for k in extra_set:
            os.environ.pop(k, None)
        extra_set = []
        for kv in (extra or "").split(";"):
            if "=" in kv:
                k, v = kv.split("=", 1)
                os.environ[k] = v
                extra_set.append(k)
        if extra:
            print(f"--- environment: {extra}", flush=True)
        if lw is not None:
            workers = mapad_amd.lib().mapad_tail_set_local_world(lw)
            os.environ.pop("MAPAD_TAIL_BACKLOG_BUDGET", None)
            if backlog:
                os.environ["MAPAD_TAIL_BACKLOG_BUDGET"] = "4294967295" if backlog == "u" else backlog
            os.environ.pop("MAPAD_TAIL_POPS_IDLE", None)
            if idle:
                os.environ["MAPAD_TAIL_POPS_IDLE"] = idle
            print(f"--- one rank of {lw}: {workers} host workers, backlog limit past the budget: {backlog or 'default (8 per worker)'}, idle-worker threshold: {idle or 'default'} pops", flush=True)
        log_path = args.log + (f".{budget}" if budget is not None else "") + (f".lw{lw}_{backlog or 'd'}_{idle or 'd'}" if lw is not None else "") + (("." + "".join(c if c.isalnum() else "_" for c in extra)) if extra else "")
        if os.path.exists(log_path):
            os.remove(log_path)
        os.environ["MAPAD_TAIL_LOG"] = log_path
        if budget is not None:
            os.environ["MAPAD_TAIL_POPS"] = str(budget)  # read when the context is created: the arena pools are sized for the budget
        ctx = mapad_amd.Context(index, mapad_amd.make_params(resolve(DAMAGE)), 0)
        ctx.set_fetch_d_arrays(False)
        print("memory before:", cg("memory.current"), "max", cg("memory.max"), flush=True)
        cpu0 = cg("cpu.stat")
        t1 = time.time()
        res = ctx.map_batch(seqs, quals, offsets)
        wall = time.time() - t1
        info = ctx.tail_info()
        info["kernel_ms"] = [round(float(x), 1) for x in ctx.kernel_ms()]
        h = hashlib.sha256()
        for a in (res.hit_begin, res.hits_arr, res.ops, res.status, res.counters):
            h.update(np.ascontiguousarray(a).tobytes())
        print(json.dumps({"budget": budget, "reads": args.reads, "wall_s": round(wall, 2), "hits": int(res.n_hits), "results_sha256": h.hexdigest()[:16], "tail": info}), flush=True)
        print("cpu.stat before:", cpu0)
        print("cpu.stat after: ", cg("cpu.stat"))
        log = np.loadtxt(log_path, ndmin=2) if os.path.exists(log_path) else np.zeros((0, 6))
        if log.size:
            pops, secs = log[:, 2], log[:, 4]
            print(f"host reads {len(log)}: pops {int(pops.sum())}, thread seconds {secs.sum():.1f}, mean {secs.sum() / max(pops.sum(), 1) * 1e6:.3f} us/pop; last read ends {float((log[:, 3] + secs).max()):.1f} s after the first hand-over")
            h_, e_ = np.histogram(log[:, 3], bins=[0, 5, 10, 15, 20, 30, 45, 60, 90, 120, 180, 240, 300, 400, 1000])
            print("  hand-overs by start time:", {f"{e_[i]:.0f}-{e_[i + 1]:.0f}": int(h_[i]) for i in range(len(h_)) if h_[i]}, flush=True)
        del res
        ctx.close()


if __name__ == "__main__":
    main()
