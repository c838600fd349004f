"""Development aid: time the heaviest reads of the C5 read mix alone (one launch, nothing else on the chip).
usage: python profiles/dev/heavy_lone.py [extract]   — `extract` maps 250 K C5 reads once and stores the 16 heaviest in profiles/dev/heavy_reads.npz"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mapad_amd
from mapad_amd import synth
from mapad_amd.presets import DAMAGE, resolve

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "heavy_reads.npz")
genome = synth.genome(48_000_000, seed=1234)
index = mapad_amd.Index.build([("chr1", genome)], seed=1234, device=0)
params = mapad_amd.make_params(resolve(DAMAGE))
if len(sys.argv) > 1 and sys.argv[1] == "extract":
    kw = dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
    seqs, quals, offsets = synth.reads(genome, 250_000, 50, seed=4321 + 5, **kw)
    ctx = mapad_amd.Context(index, params, 0)
    res = ctx.map_batch(seqs, quals, offsets)
    pops = res.counters["n_pop"].astype(np.int64)
    order = np.argsort(-pops)[:16]
    rs = [seqs[int(offsets[i]):int(offsets[i + 1])] for i in order]
    qs = [quals[int(offsets[i]):int(offsets[i + 1])] for i in order]
    off = np.zeros(len(rs) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in rs])
    np.savez(FIX, seqs=np.concatenate(rs), quals=np.concatenate(qs), offsets=off, pops=pops[order])
    print("stored", pops[order].tolist())
    ctx.close()
    sys.exit(0)
z = np.load(FIX)
n = int(os.environ.get("N_HEAVY", "1"))
seqs, quals, offsets = z["seqs"][:int(z["offsets"][n])], z["quals"][:int(z["offsets"][n])], z["offsets"][:n + 1]
ctx = mapad_amd.Context(index, params, 0)
ctx.set_fetch_d_arrays(False)
for rep in range(2):
    t = time.perf_counter()
    res = ctx.map_batch(seqs, quals, offsets)
    dt = time.perf_counter() - t
    pops = res.counters["n_pop"].astype(np.int64)
    print(f"{n} heaviest read(s): {dt:.3f} s, pops {pops.tolist()}, {dt / pops.max() * 1e6:.2f} us per pop of the heaviest, hits {int(res.hit_begin[-1])}, kernel ms {[round(float(x), 1) for x in ctx.kernel_ms()]}")
ctx.close()
