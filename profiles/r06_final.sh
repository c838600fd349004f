#!/bin/bash
# Round-6 final measurements on the GPU box (run from the repo root) for the shipped kernels (build.kernel_code_hash() keys profiles/traffic.json): the rocprofv3 summaries +
# PMC traffic and request counts of C4 / C2 / C3 (profiles/collect.sh -> profiles/r06_c*_*, profiles/traffic.json), the SQ counters of C4 (profiles/pmc_sq.sh), the full
# GPU test suite and the default bench line.  Everything lands in gpurun_out/r06/ and gpurun_out/profiles_out/.
mkdir -p gpurun_out/r06
COLLECT_TAG=r06_c4 COLLECT_KEY=c4:3000000000:10000000 COLLECT_STEPS=3 bash profiles/collect.sh --config c4 2>&1 | tail -3
COLLECT_TAG=r06_c2 COLLECT_KEY=c2:48000000:1000000 bash profiles/collect.sh --config c2 2>&1 | tail -2
COLLECT_TAG=r06_c3 COLLECT_KEY=c3:48000000:1000000 bash profiles/collect.sh --config c3 2>&1 | tail -2
bash profiles/pmc_sq.sh r06_c4 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputests_full.log 2>&1; tail -2 gpurun_out/r06/gputests_full.log
timeout 600 python bench.py > gpurun_out/r06/bench_c4_default.json 2> gpurun_out/r06/bench_c4_default.err; tail -c 400 gpurun_out/r06/bench_c4_default.json
