#!/bin/bash
# FETCH_SIZE / TCC / TCP counters of the three gathers of fetch_calib (run from the repo root on the GPU box); one counter group per pass
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $OUT
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $OUT/$name -o out --output-format csv -- $GRAFT_REPO_ROOT/profiles/calib/fetch_calib > $OUT/$name.log 2>&1; echo "$name rc=$?"
  python3 - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print("  ", k[0], k[1], v)
PY
done
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o out --output-format csv -- $GRAFT_REPO_ROOT/profiles/calib/fetch_calib | tail -4
python3 - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/calib/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  ", r["Name"].split("(")[0], "avg ns", r["AverageNs"])
PY
