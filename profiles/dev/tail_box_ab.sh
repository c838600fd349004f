#!/bin/bash
# One gpurun call: per-pop cost of the host tail on the GPU box's CPUs at the reference's limits (3 Gbp index built on the GPU; reads that reach the limits found
# by a two-stage probe), for the pin / prefetch settings of host_tail.hpp.  Output: gpurun_out/tail_ab.txt
mkdir -p gpurun_out
{
cat /sys/fs/cgroup/cpu.max; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; nproc
ls /sys/devices/system/cpu/cpu0/cache/; cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list
date +%s
timeout 600 python profiles/dev/tail_bench.py --genome-bp ${GENOME:-3000000000} --device 0 --reads ${READS:-1500} --probe-pops 65536 --probe2-pops 2000000 --probe2-top 64 --top 16 \
    --threads 16 --max-pops 8000000 --poisson 0.0001 --variants ${VARIANTS:-0:10,0:10,1:10,0:21,1:21,1:20,1:11,1:00}
date +%s
} > gpurun_out/tail_ab.txt 2>&1
tail -30 gpurun_out/tail_ab.txt
