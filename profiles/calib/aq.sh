for q in ${AQ_LIST:-16 8 4 2 1}; do
  MAPAD_EXTRA_FLAGS="-DMAPAD_ACTIVE_QUADS=$q" python -m mapad_amd.build --force >/dev/null 2>&1
  n=$((q*125000))
  timeout 300 python bench.py --reads $n --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/aq.json
  python -c "
import json,sys;d=json.load(open('gpurun_out/aq.json'));k=d['roofline']['all_kernels'];e=d['roofline']['events'];print('quads',sys.argv[1],'reads',sys.argv[2],'S ms',k['search_kernel']['ms'],'pops',e['N_pop'])" $q $n
done
