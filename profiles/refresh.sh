#!/bin/bash
# Refreshes the per-workload profiles of a round on the GPU box: section profile (a -DMAPAD_PROFILE_SECTIONS build passed as MAPAD_AMD_LIB),
# SQ counters, and the rocprofv3 summaries of C3 and C4.
mkdir -p gpurun_out/r02
if [ -f mapad_amd/libmapad_prof.so ]; then
  MAPAD_AMD_LIB=$PWD/mapad_amd/libmapad_prof.so python bench.py --depth 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/r02/sections.err
  grep "\[sections\]" gpurun_out/r02/sections.err > gpurun_out/r02/sections_c2_final.txt; head -3 gpurun_out/r02/sections_c2_final.txt
fi
bash profiles/pmc_sq.sh r02_final 2>&1 | tail -3
COLLECT_TAG=r02_c3 COLLECT_KEY=c3:48000000:1000000 bash profiles/collect.sh --config c3 2>&1 | tail -2
COLLECT_TAG=r02_c4 COLLECT_KEY=c4:3000000000:10000000 COLLECT_STEPS=3 bash profiles/collect.sh --config c4 2>&1 | tail -2
cp profiles/r02_final_sq_summary.json gpurun_out/profiles_out/ 2>/dev/null
