#!/bin/bash
# Collects a round's profiles on the GPU box: rocprofv3 kernel statistics and the HBM traffic counters of a bench workload.
# Counters go in their own passes (no --stats / trace domains next to --pmc).  Usage: bash profiles/collect.sh [bench args]
# COLLECT_TAG / COLLECT_KEY name the summary files and the traffic.json entry (default r02 / the C2 workload).
#   stats     : the default bench command (batches pipelined: launches of consecutive steps overlap)
#   stats_d1  : --depth 1 (every launch alone on the chip: the per-kernel averages bench.py reports as `solo_launch`)
#   pmc_*     : --depth 1 --steps 1 --warmup 0, one counter group per pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps ${COLLECT_STEPS:-5} --warmup 1 $B "$@" > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_d1" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --depth 1 $B "$@" > "$OUT/stats_d1.log" 2>&1; echo "stats_d1 rc=$?"
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" "ea:TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d "$OUT/pmc_$name" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --depth 1 $B "$@" > "$OUT/pmc_$name.log" 2>&1; echo "$name rc=$?"
done
cd "$ROOT"; python3 profiles/summarize.py "$OUT" "${COLLECT_TAG:-r02}" "${COLLECT_KEY:-c2:48000000:1000000}"
mkdir -p "$ROOT/gpurun_out/profiles_out"; cp "$ROOT"/profiles/${COLLECT_TAG:-r02}_* "$ROOT/profiles/traffic.json" "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
rm -rf "$OUT"/pmc_* "$OUT"/stats*/*/*trace*.csv 2>/dev/null  # raw traces are large; the summaries are what is kept
