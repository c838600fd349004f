mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputests.log 2>&1; tail -2 gpurun_out/r02/gputests.log
python bench.py > gpurun_out/r02/final_c2.json 2> gpurun_out/r02/final_c2.log
python bench.py --config c3 --no-extras > gpurun_out/r02/final_c3.json 2> gpurun_out/r02/final_c3.log
python bench.py --config c1 --no-extras > gpurun_out/r02/final_c1.json 2> gpurun_out/r02/final_c1.log
python bench.py --config c4 --steps 3 --no-extras > gpurun_out/r02/final_c4.json 2> gpurun_out/r02/final_c4.log
bash profiles/collect.sh 2>&1 | tail -7
python - <<PY
import json
for c in ("c1","c2","c3","c4"):
    d=json.load(open("gpurun_out/r02/final_%s.json"%c)); r=d["roofline"]
    print(c, d["value"], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["solo_launch"]["search_ms"], r["solo_launch"]["frac"], (d.get("cpu_baseline") or {}).get("value"), d.get("parity"))
PY
