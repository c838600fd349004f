#!/bin/bash
# Collects the round's profiles on the GPU box: rocprofv3 kernel statistics and the HBM traffic counters of the bench workload.
# Counters go in their own passes (no --stats / trace domains next to --pmc).  Usage: bash profiles/collect.sh [bench args]
# COLLECT_TAG / COLLECT_KEY name the summary files and the traffic.json entry (default r01 / the C2 workload).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
timeout 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
if [ -n "$COLLECT_TCC" ]; then
  timeout 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/pmc_tcc" -o out --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pmc_tcc.log" 2>&1; echo "tcc rc=$?"
fi
cd "$ROOT"; python3 profiles/summarize.py "$OUT" "${COLLECT_TAG:-r01}" "${COLLECT_KEY:-c2:48000000:1000000}"
