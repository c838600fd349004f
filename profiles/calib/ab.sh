#!/bin/bash
# usage: [AB_C3=1] [AB_ENV="K=V ..."] ab.sh "<extra hipcc flags A>" "<flags B>" ...  -- bench C2 (and C3) for each build variant
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));k=d['roofline']['all_kernels'];print(sys.argv[2],'|',d['value'],d['ms_per_step'],'D',k['darray_kernel']['ms'],'S',k['search_kernel']['ms'],'last',k['search_kernel_last_pass'])" "$1" "$2"; }
for v in "$@"; do
  MAPAD_EXTRA_FLAGS="$v" python -m mapad_amd.build --force >/dev/null 2>&1
  for e in "" $AB_ENV; do
    env $e timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_c2.json; show gpurun_out/ab_c2.json "C2 [$v] [$e]"
    if [ -n "$AB_C3" ]; then env $e timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_c3.json; show gpurun_out/ab_c3.json "C3 [$v] [$e]"; fi
  done
done
