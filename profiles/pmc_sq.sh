#!/bin/bash
# SQ / TCP counter passes over one launch sequence of the bench workload (instruction mix, issue and wait cycles of the search kernel).
# Usage: bash profiles/pmc_sq.sh TAG [bench args].  Output: gpurun_out/prof_sq/<pass>/..., summary profiles/<TAG>_sq_summary.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_sq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout 280 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras $BENCH_ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"
}
BENCH_ARGS="$*"
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM
pass c SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
cd "$ROOT"; python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
c = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "search_kernel" in r["Kernel_Name"]:
            k = "search_kernel<" + r["Kernel_Name"].split("<")[1].split(">")[0] + ">"
        c[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {k: {m: v / max(len(n[k][m]), 1) for m, v in d.items()} for k, d in c.items()}
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras`; per launch (mean over the launches of a kernel)", "per_launch": res},
          open(os.path.join("profiles", f"{tag}_sq_summary.json"), "w"), indent=1)
for k, d in res.items():
    if "search_kernel<4, false, 0" in k or "search_kernel<4, 0, 0" in k: print(k, json.dumps(d))
PY
mkdir -p "$ROOT/gpurun_out/profiles_out"; cp "$ROOT"/profiles/${TAG}_sq_summary.json "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
rm -rf "$OUT"  # raw counter CSVs are large; the summary is what is kept
