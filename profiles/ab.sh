mkdir -p gpurun_out/r02
for v in pf nopf; do
  if [ $v = nopf ]; then export MAPAD_AMD_LIB=$PWD/mapad_amd/libmapad_nopf.so; else unset MAPAD_AMD_LIB; fi
  python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02/ab_c2_$v.json 2> gpurun_out/r02/ab.err
  python bench.py --config c3 --no-cpu-baseline --no-extras > gpurun_out/r02/ab_c3_$v.json 2>> gpurun_out/r02/ab.err
  python bench.py --config c5 --reads 1000000 --depth 3 --steps 3 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/r02/ab_c5_$v.json 2>> gpurun_out/r02/ab.err
done
python - <<PY
import json
for c in ("c2","c3","c5"):
    for v in ("pf","nopf"):
        d=json.load(open("gpurun_out/r02/ab_%s_%s.json"%(c,v))); r=d["roofline"]
        print(c, v, d["value"], d["ms_per_step"], r["kernel_ms"], r["solo_launch"]["search_ms"])
PY
