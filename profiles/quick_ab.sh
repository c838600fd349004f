#!/bin/bash
# Quick check after a kernel change: the GPU parity tests, then one default bench line per workload (reads/s, ms per step, effective kernel ms,
# roofline fraction, solo launch ms).  Usage on the GPU box: bash profiles/quick_ab.sh [c2 c3 ...]
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in ${@:-c2 c3}; do
  python bench.py --config $cfg --no-cpu-baseline --no-extras > gpurun_out/r02/quick_$cfg.json 2> gpurun_out/r02/quick.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/quick_$cfg.json")); r=d["roofline"]; print("$cfg", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["solo_launch"]["search_ms"], r["all_kernels"]["darray_kernel"]["ms"])
PY
done
