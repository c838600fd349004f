#!/bin/bash
# same-box A/B of library variants and environment settings:
#   bash profiles/dev/ab3.sh "cfg..." spec spec ...     spec = LIB[:ENV=VAL[,ENV=VAL...]]   LIB = default | NAME (mapad_amd/variant_NAME.so)
# two repetitions, variants interleaved; one line per run: reads/s, effective kernel ms, roofline fraction, solo launch ms
CFGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  (
  if [ "$v" != default ]; then export MAPAD_AMD_LIB=$PWD/mapad_amd/variant_$v.so; fi
  IFS=','; for e in $envs; do export "$e"; done; unset IFS
  for cfg in $CFGS; do
    steps=8; [ "$cfg" = c4 ] && steps=${AB_C4_STEPS:-3}
    python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-extras $AB_EXTRA 2> gpurun_out/ab/err_${v}_$cfg.txt | python -c "
import json,sys
try:
    d=json.load(sys.stdin); r=d['roofline']; print('$spec $cfg rep$rep reads/s', d['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'solo', r['solo_launch']['search_ms'], flush=True)
except Exception as e:
    print('$spec $cfg rep$rep FAILED', e, flush=True)"
  done
  )
done
done
