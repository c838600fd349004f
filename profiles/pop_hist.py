"""Per-read pop counts of one batch (GPU result counters): how the search work is spread over the reads of a workload.
usage: python profiles/pop_hist.py c5 [n_reads]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mapad_amd
from mapad_amd import synth
from mapad_amd.presets import DAMAGE, NO_DAMAGE, resolve

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
genome = synth.genome(48_000_000, seed=1234)
index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=0)
if cfg == "c2":
    prm, kw = NO_DAMAGE, dict(qual=40)
elif cfg == "c3":
    prm, kw = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
else:
    prm, kw = DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
seqs, quals, offsets = synth.reads(genome, n_reads, 50, seed=4321 + int(cfg[1]), **kw)
ctx = mapad_amd.Context(index, mapad_amd.make_params(resolve(prm)), 0)
t = time.time()
res = ctx.map_batch(seqs, quals, offsets)
dt = time.time() - t
c = res.counters
pops = c["n_pop"].astype(np.int64)
lens = np.diff(offsets.astype(np.int64))
order = np.argsort(-pops)
tot = int(pops.sum())
out = {"config": cfg, "reads": n_reads, "wall_s": round(dt, 2), "pops_total": tot, "pops_mean": round(tot / n_reads, 1),
       "top20": [[int(pops[i]), int(lens[i]), int(c["n_push"][i]), int(c["n_hits"][i])] for i in order[:20]],
       "quantiles": {str(q): int(np.quantile(pops, q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999)},
       "share_of_pops_in_reads_above": {str(th): [int((pops > th).sum()), round(float(pops[pops > th].sum()) / tot, 4)] for th in (10_000, 100_000, 1_000_000)}}
print(json.dumps(out))
