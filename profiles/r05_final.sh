#!/bin/bash
# Round-5 final measurements on the GPU box (run from the repo root): section profiles (a -DMAPAD_PROFILE_SECTIONS build: mapad_amd/variant_prof.so), the
# rocprofv3 summaries + PMC traffic of C4 / C2 / C3 (profiles/collect.sh -> profiles/r05_c*_*, profiles/traffic.json), the SQ counters of C4, and the default
# bench line.  Everything lands in gpurun_out/r05/ and gpurun_out/profiles_out/.
mkdir -p gpurun_out/r05
if [ -f mapad_amd/variant_prof.so ]; then
  for cfg in c4 c3 c2; do
    MAPAD_AMD_LIB=$PWD/mapad_amd/variant_prof.so timeout 300 python bench.py --config $cfg --depth 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/r05/sections_$cfg.err
    grep "\[sections\]" gpurun_out/r05/sections_$cfg.err > gpurun_out/r05/sections_${cfg}_round5.txt; head -4 gpurun_out/r05/sections_${cfg}_round5.txt
  done
fi
COLLECT_TAG=r05_c4 COLLECT_KEY=c4:3000000000:10000000 COLLECT_STEPS=3 bash profiles/collect.sh --config c4 2>&1 | tail -3
COLLECT_TAG=r05_c2 COLLECT_KEY=c2:48000000:1000000 bash profiles/collect.sh --config c2 2>&1 | tail -2
COLLECT_TAG=r05_c3 COLLECT_KEY=c3:48000000:1000000 bash profiles/collect.sh --config c3 2>&1 | tail -2
bash profiles/pmc_sq.sh r05_c4 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_c4_default.json 2> gpurun_out/r05/bench_c4_default.err; tail -c 600 gpurun_out/r05/bench_c4_default.json
# the C5 mix at 1 M reads on the 3 Gbp index (defaults) and the digest of all 10 M C4 reads against round 4's oracle digest, with the final kernels
timeout 500 python profiles/dev/tail_1m.py 2>&1 | grep -v "^\[tail\] read" | tail -12 > gpurun_out/r05/c5_3gbp_1m_final.txt; grep -o '"wall_s": [0-9.]*\|"results_sha256": "[0-9a-f]*"' gpurun_out/r05/c5_3gbp_1m_final.txt
timeout 600 python profiles/audit_c4.py --against profiles/r04/c4_full_parity.json --out gpurun_out/r05/c4_full_parity.json 2>&1 | tail -3
