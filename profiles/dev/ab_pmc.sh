#!/bin/bash
# Same-box A/B of library variants by their memory-side counters: requests behind the L2 per pop.
#   bash profiles/dev/ab_pmc.sh "cfg..." LIB LIB ...     LIB = default | NAME (mapad_amd/variant_NAME.so)
# Per (variant, config): two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE — counters in their own runs, --kernel-trace only) of one launch alone on the chip
# (`bench.py --steps 1 --warmup 0 --depth 1`).  Read requests = FETCH_SIZE KiB * 1024 / 64 (gfx950 tallies a 128-byte request as 64 bytes), write requests =
# WRITE_SIZE KiB * 1024 / 64, mean over the process's real launches; pops from the bench line's event counters.  One line per (variant, config) on stdout, JSON.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
CFGS=$1; shift
OUT=$ROOT/gpurun_out/ab_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  for cfg in $CFGS; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      (
      if [ "$v" != default ]; then export MAPAD_AMD_LIB=$ROOT/mapad_amd/variant_$v.so; fi
      timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d "$OUT/${v}_${cfg}_$ctr" -o out --output-format csv -- python3 "$ROOT/bench.py" --config $cfg --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras > "$OUT/${v}_${cfg}_$ctr.json" 2> "$OUT/${v}_${cfg}_$ctr.err"
      )
    done
    python3 - "$OUT" "$v" "$cfg" <<'PY'
import csv, glob, json, os, sys
out, v, cfg = sys.argv[1:4]
res = {"variant": v, "config": cfg}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    per_dispatch, ms = {}, []
    for f in glob.glob(os.path.join(out, f"{v}_{cfg}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "search_kernel<4, false, 0" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                per_dispatch[r["Dispatch_Id"]] = per_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    # `bench.py --steps 1 --warmup 0` maps the batch twice (the timed step and the solo launch): the mean over the real launches (the empty warm-up launches dropped)
    real = [x for x in per_dispatch.values() if x > 0.01 * max(per_dispatch.values())] if per_dispatch else []
    tot = sum(real) / len(real) if real else 0.0
    res["launches_" + ctr] = len(real)
    for f in glob.glob(os.path.join(out, f"{v}_{cfg}_{ctr}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "search_kernel<4, false, 0" in r["Kernel_Name"]:
                ms.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    res[ctr + "_KiB"] = tot
    res["search_ms_" + ctr] = round(max(ms), 2) if ms else None
    try:
        line = json.loads(open(os.path.join(out, f"{v}_{cfg}_{ctr}.json")).read().strip().splitlines()[-1])
        res["N_pop"] = line["roofline"]["events"]["N_pop"]
        res["algorithmic_bytes"] = line["roofline"]["algorithmic_bytes_per_launch"]
    except Exception as e:
        res["bench_line_error"] = str(e)
if res.get("N_pop"):
    p = res["N_pop"]
    res["read_requests_per_pop"] = round(res["FETCH_SIZE_KiB"] * 16 / p, 3)
    res["write_requests_per_pop"] = round(res["WRITE_SIZE_KiB"] * 16 / p, 3)
    res["requests_per_pop"] = round((res["FETCH_SIZE_KiB"] + res["WRITE_SIZE_KiB"]) * 16 / p, 3)
    res["traffic_over_algorithmic"] = round((2 * res["FETCH_SIZE_KiB"] + res["WRITE_SIZE_KiB"]) * 1024 / res["algorithmic_bytes"], 3)
print(json.dumps(res), flush=True)
PY
    rm -rf "$OUT/${v}_${cfg}_FETCH_SIZE" "$OUT/${v}_${cfg}_WRITE_SIZE"
  done
done
