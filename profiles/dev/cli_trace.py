"""Kernel timeline of a `mapad-amd map` run from rocprofv3's kernel trace (gpurun_out/prof_cli): per kernel family the union of its intervals, and how
much of the mapping phase some search kernel / any kernel was running."""
import csv, glob, sys
f = glob.glob("gpurun_out/prof_cli/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
def fam(n):
    for k in ("search_kernel", "heavy_kernel", "darray_kernel", "order_", "compact_", "records_kernel", "text_kernel", "locate_kernel"):
        if k in n: return k
    return "other"
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
srch = [(s, e) for n, s, e in rows if fam(n) == "search_kernel" and e - s > 1_000_000]
t0, t1 = min(s for s, e in srch), max(e for s, e in srch)
print(f"mapping phase on the GPU: {(t1 - t0) / 1e6:.1f} ms, {len(srch)} search launches, mean {sum(e - s for s, e in srch) / len(srch) / 1e6:.1f} ms each")
by = {}
for n, s, e in rows:
    if s >= t0 and e <= t1: by.setdefault(fam(n), []).append((s, e))
for k, iv in sorted(by.items(), key=lambda kv: -union(kv[1])):
    print(f"  {k:16s} launches {len(iv):6d}  union {union(iv) / 1e6:9.1f} ms ({100 * union(iv) / (t1 - t0):5.1f} %)  sum {sum(e - s for s, e in iv) / 1e6:9.1f} ms")
allk = [x for iv in by.values() for x in iv]
print(f"  any kernel       union {union(allk) / 1e6:9.1f} ms ({100 * union(allk) / (t1 - t0):5.1f} %)")
# concurrency of search launches over time
ev = sorted([(s, 1) for s, e in srch] + [(e, -1) for s, e in srch]); cur = 0; last = t0; hist = {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + t - last; last = t; cur += d
print("  search launches running at once:", {k: f"{100 * v / (t1 - t0):.1f} %" for k, v in sorted(hist.items())})
