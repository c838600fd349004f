#!/bin/bash
# Memory-pipeline counter passes (TA / TCP / TCC / UTCL1) over one launch of the bench workload: what the search kernel keeps busy between the CU and HBM.
# Usage: bash profiles/pmc_mem.sh TAG [bench args].  Output: profiles/<TAG>_mem_summary.json (per counter, the search launch of stage 0 only)
# Counters in their own runs with --kernel-trace only (no --stats, no other trace domain beside --pmc).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_mem
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
BENCH_ARGS="$*"
pass() {
  name=$1; shift
  if [ -n "$PMC_MEM_PASSES" ] && ! echo " $PMC_MEM_PASSES " | grep -q " $name "; then return; fi  # PMC_MEM_PASSES="e f g h": only these (the GRBM / TA / TCC groups time out on this pool)
  timeout 280 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras $BENCH_ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"
}
pass a GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY GRBM_UTCL2_BUSY
pass b TA_BUSY_avr TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass c TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
pass d SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
pass e TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass f TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass g TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass h TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
pass i TCC_BUSY_avr TCC_CYCLE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
pass j TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass k TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass l TCC_READ_REQ_LATENCY_sum TCC_WRITE_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
cd "$ROOT"; python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
c = collections.defaultdict(float); n = collections.defaultdict(set); dur = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "search_kernel<4, false, 0" not in k and "search_kernel<4, 0, 0" not in k:
            continue
        c[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for f in glob.glob(os.path.join(out, "*", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "search_kernel<4, false, 0" in r["Kernel_Name"] or "search_kernel<4, 0, 0" in r["Kernel_Name"]:
            dur.setdefault(os.path.basename(os.path.dirname(os.path.dirname(f))), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {m: v / max(len(n[m]), 1) for m, v in c.items()}
fails = {os.path.basename(p)[:-4]: open(p, errors="replace").read()[-300:] for p in glob.glob(os.path.join(out, "*.log")) if not glob.glob(os.path.join(out, os.path.basename(p)[:-4], "**", "*counter_collection.csv"), recursive=True)}
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras`; search_kernel<4,false,0,...> dispatches only, mean per launch",
           "per_launch": res, "search_ms_per_pass": dur, "failed_passes": fails}, open(os.path.join("profiles", f"{tag}_mem_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=0)); print("failed:", list(fails))
PY
mkdir -p "$ROOT/gpurun_out/profiles_out"; cp "$ROOT"/profiles/${TAG}_mem_summary.json "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
rm -rf "$OUT"/*/  # raw counter CSVs are large; the summary is what is kept
