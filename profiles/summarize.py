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
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
key = sys.argv[3] if len(sys.argv) > 3 else "c2:48000000:1000000"


def short(name):
    if "darray_kernel" in name:
        return "darray_kernel"
    if "search_kernel" in name:
        return "search_kernel" if ", 0, " in name else "search_kernel_retry" if ", 2, " in name else "search_kernel_last_pass"
    if "order_" in name:
        return "order_kernels"
    return None


st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(max(st, key=os.path.getmtime), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
counters = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("pmc_fetch", "pmc_write", "pmc_tcc"):
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
summary["traffic_bytes_per_batch"] = traffic
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
tp = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tp)) if os.path.exists(tp) else {}
tj[key] = traffic
json.dump(tj, open(tp, "w"), indent=1)
print(json.dumps(summary["traffic_bytes_per_batch"]))
