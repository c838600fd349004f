mkdir -p gpurun_out/r04
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; nproc; lscpu | head -25; python3 - <<'PY'
import time, threading, ctypes, os, multiprocessing as mp
def burn(q, secs):
    t=time.time(); n=0
    while time.time()-t < secs:
        x=0
        for i in range(200000): x+=i*i
        n+=1
    q.put(n)
for k in (1, 8, 32, 64, 128, 256):
    q=mp.Queue(); ps=[mp.Process(target=burn,args=(q,2.0)) for _ in range(k)]
    [p.start() for p in ps]; tot=sum(q.get() for _ in ps); [p.join() for p in ps]
    print("procs", k, "units/s", tot/2.0, "per proc", tot/2.0/k, flush=True)
PY
) > gpurun_out/r04/cpu_probe.txt 2>&1
python bench.py --gpus 2 --dist-backend gloo --config c2 --genome-bp 2000000 --reads 60000 --steps 3 --warmup 1 --scaling weak --no-cpu-baseline --no-extras --watchdog-s 120 > gpurun_out/r04/two_rank.json 2> gpurun_out/r04/two_rank.err
grep -n "Traceback" -A25 gpurun_out/r04/two_rank.err | head -60
MAPAD_AMD_LIB=$PWD/mapad_amd/variant_prof.so python bench.py --config c4 --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras > gpurun_out/r04/prof_c4.json 2> gpurun_out/r04/prof_c4.err
grep "sections" gpurun_out/r04/prof_c4.err
for th in 1 16 256; do MAPAD_TAIL_THREADS=$th MAPAD_TAIL_POPS=16384 python bench.py --config c5 --genome-bp 200000000 --reads 30000 --steps 1 --warmup 0 --depth 1 --no-cpu-baseline --no-extras 2> gpurun_out/r04/tailth_$th.err | python3 -c "
import json,sys
d=json.load(sys.stdin); t=d['tail']; print('threads $th', d['ms_per_step'], t)"; done
