#!/bin/bash
# Request sizes between the L2 and memory (TCC_EA0_RDREQ by size) for the gathers of fetch_calib, then for one C2 launch of the search; run from the repo root on the GPU box
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib_ea; mkdir -p $OUT
sumcsv() {
python3 - "$1" "$2" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sys.argv[2] and sys.argv[2] not in r["Kernel_Name"]: continue
        acc[(k[:60], r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print("  ", k[0], k[1], v)
PY
}
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $OUT/rd -o out --output-format csv -- $GRAFT_REPO_ROOT/profiles/calib/fetch_calib > $OUT/rd.log 2>&1; echo "calib rd rc=$?"; sumcsv $OUT/rd gather
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $OUT/c2rd -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras > $OUT/c2rd.log 2>&1; echo "c2 rd rc=$?"; sumcsv $OUT/c2rd "search_kernel<4, false, 0"
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $OUT/c2wr -o out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras > $OUT/c2wr.log 2>&1; echo "c2 wr rc=$?"; sumcsv $OUT/c2wr "search_kernel<4, false, 0"
