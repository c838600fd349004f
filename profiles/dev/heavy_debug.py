"""Development aid: one batch through the GPU path with small arenas (so that reads go through the heavy path) against the oracle, read by read."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    os.environ[k] = v
import mapad_amd
from mapad_amd import synth
from oracle import binding as ob
from parity_util import DAMAGE, split_reads
from kat_util import resolve_params

n_reads = int(os.environ.get("N_READS", "2500"))
g = synth.genome(300_000, seed=21)
seqs, quals, offsets = synth.reads(g, n_reads, 50, seed=5, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0))
rp = resolve_params(DAMAGE)
pidx = mapad_amd.Index.build([("chr1", g)])
oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
res = ctx.map_batch(seqs, quals, offsets)
reads, qs = split_reads(seqs, quals, offsets)
ores = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8, keep_d=True)
c = res.counters
got = np.stack([c["e_search"], c["e_darray"], c["n_push"], c["n_pop"], c["n_node"], c["n_hits"]], axis=1).astype(np.uint64)
bad = np.where((got != ores.counters).any(axis=1))[0]
print("migrations", res.n_second_pass, "full-limit reads", res.n_third_pass, "reads with different counters:", len(bad), "of", n_reads)
hb_ok = np.array_equal(res.hit_begin, ores.hit_offsets)
print("hit_begin equal:", hb_ok, "status nonzero:", int((res.status != 0).sum()) if hasattr(res, "status") else "?")
for i in bad[:12]:
    print(i, "gpu", got[i].tolist(), "oracle", ores.counters[i].tolist(), "hits gpu", int(res.hit_begin[i + 1] - res.hit_begin[i]), "oracle", int(ores.hit_offsets[i + 1] - ores.hit_offsets[i]))
if len(bad) == 0 and hb_ok:
    ok = (np.array_equal(res.hits_arr["lower"], ores.intervals[:, 0]) and np.array_equal(res.hits_arr["size"], ores.intervals[:, 2])
          and np.array_equal(res.hits_arr["score"].view(np.uint32), ores.scores.view(np.uint32)) and np.array_equal(res.ops, ores.ops))
    print("hits/ops identical:", ok)
ctx.close()
