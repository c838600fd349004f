import os, subprocess, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from mapad_amd import synth, build
cli = build.build_cli()
d = tempfile.mkdtemp()
g = synth.genome(4_000_000, seed=1)
fa, fq = os.path.join(d, "ref.fa"), os.path.join(d, "r.fastq")
open(fa, "w").write(">chr1\n" + g.tobytes().decode() + "\n")
seqs, quals, offsets = synth.reads(g, 100_000, 50, seed=2, qual=40)
with open(fq, "w") as f:
    for i in range(100_000):
        s, e = int(offsets[i]), int(offsets[i + 1])
        f.write(f"@r{i}\n{seqs[s:e].tobytes().decode()}\n+\n{'I' * (e - s)}\n")
subprocess.check_call([cli, "index", "-g", fa], stderr=subprocess.DEVNULL)
base = [cli, "map", "-r", fq, "-g", fa, "-l", "single_stranded", "-p", "0.03", "-f", "0.5", "-t", "0.5", "-d", "0.02", "-s", "1.0", "-i", "0.001", "-o", os.path.join(d, "o.bam"), "--force_overwrite"]
for name, env in (("default", {}), ("old_counts", {"MAPAD_CLASS_COUNTS": "24576,12288,6144,3072,1024,256,64,32,16,16"}), ("small", {"MAPAD_CLASS_COUNTS": "4096,2048,1024,512,256,64,32,16,16,16"}), ("default_again", {})):
    for extra in ([], ["--in_flight", "2"]):
        p = subprocess.run(base + extra, env=dict(os.environ, **env), stderr=subprocess.PIPE, text=True)
        print(name, extra, [l for l in p.stderr.splitlines() if "index + contexts" in l], flush=True)
