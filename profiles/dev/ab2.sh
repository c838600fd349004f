#!/bin/bash
# same-box A/B of library variants: bash profiles/dev/ab2.sh "cfg..." variantA variantB ...   ("default" = the in-tree library)
CFGS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset MAPAD_AMD_LIB; else export MAPAD_AMD_LIB=$PWD/mapad_amd/variant_$v.so; fi
  for cfg in $CFGS; do
    steps=8; [ "$cfg" = c4 ] && steps=3
    python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-extras 2> /dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; print('$v $cfg rep$rep reads/s', d['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'solo', r['solo_launch']['search_ms'])"
  done
done
done
