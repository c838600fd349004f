mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in c2 c3 c5; do
  python bench.py --config $cfg --no-cpu-baseline --no-extras > gpurun_out/r02/ea_$cfg.json 2> gpurun_out/r02/sw.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/ea_$cfg.json")); r=d["roofline"]; print("$cfg", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["solo_launch"]["search_ms"])
PY
done
