"""mapad-amd map on an 8 M-read FASTQ (C2 workload; CLI_GENOME_BP=3000000000 for C4's; CLI_READS=24000000 for the size of bench.py's leg) for several
--batch_size / --in_flight [/ --coalesce / pool sizes] settings: reads/s and stage busy times.  Cases: batch:in_flight[:coalesce=N][:parse_threads=N][:encode_threads=N][:ENV=VAL...]"""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np
from mapad_amd import build as mbuild, synth
n = int(float(os.environ.get("CLI_READS", "8000000")))
genome_bp = int(float(os.environ.get("CLI_GENOME_BP", "48000000")))
g = synth.genome(genome_bp, seed=1234)
parts = [synth.reads(g, n // 4, 50, seed=4323 + 1000 * k, qual=40) for k in range(4)]
seqs = np.concatenate([p[0] for p in parts]); quals = np.concatenate([p[1] for p in parts])
tmp = tempfile.mkdtemp(prefix="mapad_cli_")
fa, fq, bam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fastq"), os.path.join(tmp, "out.bam")
rec = np.empty((n, 114), np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
ids = np.arange(n)
for k in range(7):
    rec[:, 8 - k] = 48 + (ids // 10 ** k) % 10
rec[:, 9] = 10; rec[:, 10:60] = seqs.reshape(n, 50); rec[:, 60] = 10; rec[:, 61] = ord("+"); rec[:, 62] = 10
rec[:, 63:113] = quals.reshape(n, 50) + 33; rec[:, 113] = 10
rec.tofile(fq)
exe = mbuild.build_cli()
if genome_bp <= 200_000_000:
    open(fa, "wb").write(b">chr1\n" + g.tobytes() + b"\n")
    subprocess.check_call([exe, "index", "-g", fa], stderr=subprocess.DEVNULL)
else:  # as bench.py does at C4: built in this process on the GPU, written as the seven index files
    import mapad_amd
    idx = mapad_amd.Index.build([("chr1", g)], device=0)
    idx.save(fa)
    del idx
del g
base = [exe, "map", "-r", fq, "-g", fa, "-o", bam, "-l", "single_stranded", "-p", "0.03", "-D", "0.02", "-i", "0.001", "-x", "1.0", "--force_overwrite", "-f", "0", "-t", "0", "-d", "0", "-s", "0"]
cases = [(250000, 4, {}), (250000, 8, {}), (250000, 12, {}), (500000, 4, {}), (500000, 8, {}), (1000000, 4, {})]
if len(sys.argv) > 1:  # e.g. "250000:4:MAPAD_TIER0_WAVES_PER_CU=10"
    cases = []
    for a in sys.argv[1:]:
        f = a.split(":")
        cases.append((int(f[0]), int(f[1]), dict(x.split("=") for x in f[2:])))
for bs, fl, env in cases:
    cmd = base + ["--batch_size", str(bs), "--in_flight", str(fl)]
    if env.get("exe"):  # another build of the command (e.g. exe=profiles/dev/r04_bin/mapad-amd: round 4's, with its own library beside it)
        cmd[0] = os.path.abspath(env.pop("exe"))
    for opt in ("coalesce", "coalesce_steady", "parse_threads", "encode_threads"):
        if opt in env:
            cmd += ["--" + opt, env.pop(opt)]
    if env.pop("ROCPROF", None):  # kernel statistics of the run: the summary lands in gpurun_out/prof_cli
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(os.getcwd(), "gpurun_out", "prof_cli"), "-o", "out", "--output-format", "csv", "--"] + cmd
    pr = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
    m = re.search(r"mapping ([0-9.]+) s \((\d+) reads/s\)", pr.stderr)
    b = re.search(r"reader ([0-9.]+) s, device worker 0 ([0-9.]+) s, writer ([0-9.]+) s", pr.stderr)
    w = re.search(r"submit ([0-9.]+) s, fetch \(incl. waiting for the GPU\) ([0-9.]+) s, coordinates ([0-9.]+) s; records thread \(strings, MAPQ\) ([0-9.]+) s", pr.stderr)
    hp = re.search(r"steady state.*", pr.stderr)
    print("   worker:", w.groups() if w else None, "|", hp.group(0) if hp else "")
    print(bs, fl, env, m.group(1) if m else pr.stderr[-300:], m.group(2) if m else "", b.groups() if b else "", flush=True)
