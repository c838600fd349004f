#!/bin/bash
# per-kernel durations of one bench command: bash profiles/dev/kstats.sh TAG [bench args]   (env vars pass through)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/kstats_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT" -o out --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd "$ROOT"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r["Name"][:110].ljust(110), r["Calls"], "total ms", round(float(r["TotalDurationNs"]) / 1e6, 3), "avg ms", round(float(r["AverageNs"]) / 1e6, 3), "max ms", round(float(r["MaxNs"]) / 1e6, 3))
PY
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
