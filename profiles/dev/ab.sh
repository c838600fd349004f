#!/bin/bash
# quick A/B after a kernel change: parity tests, then bench lines.  usage: bash profiles/dev/ab.sh TAG cfg...
TAG=$1; shift
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "not kat" 2>&1 | tail -2
for cfg in "$@"; do
  steps=10; [ "$cfg" = c4 ] && steps=3
  python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r03/ab_${TAG}_$cfg.json 2> gpurun_out/r03/ab.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/ab_${TAG}_$cfg.json")); r=d["roofline"]; print("$TAG $cfg reads/s", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "solo", r["solo_launch"]["search_ms"], r["solo_launch"]["frac"])
PY
done
