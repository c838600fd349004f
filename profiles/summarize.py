"""Turns the rocprofv3 output of profiles/collect.sh into the committed summaries (kernel stats CSV, PMC summary, traffic.json)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
key = sys.argv[3] if len(sys.argv) > 3 else "c2:48000000:1000000"


def short(name):
    if "darray_kernel" in name:
        return "darray_kernel"
    if "search_kernel" in name:
        return "search_kernel" if ", 0, " in name else "search_kernel_retry" if ", 2, " in name else "search_kernel_last_pass"
    if "heavy_kernel" in name:  # the full-limit stage (and, with MAPAD_HEAVY=1, the suspended reads): one wavefront per read
        return "search_kernel_last_pass"
    if "order_" in name:
        return "order_kernels"
    return None


for sub, suffix in (("stats", "kernel_stats"), ("stats_d1", "kernel_stats_depth1")):
    st = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(max(st, key=os.path.getmtime), os.path.join(ROOT, "profiles", f"{tag}_{suffix}.csv"))
# pipelined launches overlap: the effective duration of a launch is the union of the launches' intervals divided by their number
# (what bench.py reports as roofline.kernel_ms); computed here from the kernel trace of the default command
tr = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
overlap = {}
if tr:
    iv = collections.defaultdict(list)
    for r in csv.DictReader(open(max(tr, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        if k:
            iv[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, lst in iv.items():
        longest = max(h - l for l, h in lst)
        lst = sorted(x for x in lst if x[1] - x[0] > 0.01 * longest)  # drop the empty warm-up launches of mapad_ctx_reserve
        total, end = 0, -1
        for lo, hi in lst:
            if hi > end:
                total += hi - max(lo, end)
                end = hi
        overlap[k] = {"launches": len(lst), "mean_duration_ms": sum(h - l for l, h in lst) / len(lst) / 1e6, "union_ms": total / 1e6,
                      "union_per_launch_ms": total / len(lst) / 1e6}
    # the same from the --depth 1 trace: every launch alone on the chip; the plain `--stats` average of that run is diluted by the empty launches
    # that warm the batch slot (Calls counts them), this one is not
    solo = {}
    tr1 = glob.glob(os.path.join(out, "stats_d1", "**", "*kernel_trace.csv"), recursive=True)
    if tr1:
        iv1 = collections.defaultdict(list)
        for r in csv.DictReader(open(max(tr1, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            if k:
                iv1[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, lst in iv1.items():
            real = [x for x in lst if x > 0.01 * max(lst)]
            solo[k] = {"launches": len(real), "empty_launches_dropped": len(lst) - len(real), "mean_duration_ms": sum(real) / len(real) / 1e6}
    json.dump({"note": "kernel trace of the default bench command (batches pipelined): per kernel, mean duration of a launch, union of all launches' intervals, "
                       "and union / launches = the effective per-launch time bench.py uses for the roofline (includes the warm-up step and the final solo launch of bench.py, "
                       "which run with nothing beside them; the empty launches that warm the batch slots are dropped).  depth1: the --depth 1 run, every launch alone on "
                       "the chip = bench.py's roofline.solo_launch",
               "kernels": overlap, "depth1": solo}, open(os.path.join(ROOT, "profiles", f"{tag}_kernel_overlap.json"), "w"), indent=1)
counters = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_ea"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
summary = {"note": "rocprofv3 --pmc, separate passes, `bench.py --steps 1 --warmup 0 --no-cpu-baseline`.  FETCH_SIZE / WRITE_SIZE in KiB as "
                   "reported, summed over the launches of one batch (= one bench step).  gfx950 correction (MI355X_MICROARCH.md, HBM section; "
                   "checked for this access pattern by profiles/calib/fetch_calib.hip): FETCH_SIZE tallies a 128-byte request as 64 bytes, so "
                   "traffic = 2 * FETCH_SIZE + WRITE_SIZE.",
           "per_batch": {}}
traffic = {}
# per batch: the search kernel is launched several times per batch (every read, two retry launches that are normally empty), so
# counters are summed over a batch's launches; the number of batches is the number of darray_kernel launches
n_batches = {n: max(len(ids), 1) for n, ids in launches.get("darray_kernel", {}).items()}
for k, c in counters.items():
    per = {n: v / n_batches.get(n, 1) for n, v in c.items()}
    summary["per_batch"][k] = per
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        traffic[k] = int((2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024)
        # requests the L2 sent to memory (FETCH_SIZE and WRITE_SIZE are request counts x 64 B on gfx950, whatever a request's size): the path's accesses are
        # random, and what such a path saturates is the rate of requests, not bytes (profiles/calib/fetch_calib.hip; bench.py: roofline.random_access)
        traffic[k + "_hbm_requests"] = int((per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024 / 64)
        traffic[k + "_hbm_read_requests"] = int(per["FETCH_SIZE"] * 1024 / 64)           # every one a 128-byte line (profiles/r05/calib_ea_request_sizes.txt)
        traffic[k + "_hbm_write_64B_units"] = int(per["WRITE_SIZE"] * 1024 / 64)        # bytes written / 64
    if "TCC_EA0_WRREQ_sum" in per:  # the memory side's own request counts (round 6): a write request is 32 or 64 bytes, so there are more of them than WRITE_SIZE / 64
        traffic[k + "_ea_rdreq"] = int(per.get("TCC_EA0_RDREQ_sum", 0))
        traffic[k + "_ea_wrreq"] = int(per["TCC_EA0_WRREQ_sum"])
        traffic[k + "_ea_wrreq_64B"] = int(per.get("TCC_EA0_WRREQ_64B_sum", 0))
summary["traffic_bytes_per_batch"] = traffic
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
tp = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tp)) if os.path.exists(tp) else {}
sys.path.insert(0, ROOT)
from mapad_amd import build as _build  # noqa: E402
traffic["kernel_source_sha16"] = _build.source_hash()
traffic["kernel_code_sha16"] = _build.kernel_code_hash()  # bench.py reports this traffic only while the library's gfx950 machine code is this code
tj[key] = traffic
json.dump(tj, open(tp, "w"), indent=1)
print(json.dumps(summary["traffic_bytes_per_batch"]))
