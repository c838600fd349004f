#!/usr/bin/env python3
"""Dev: 20-second check on a GPU box that the host tail of the library as built gives the same batch as the GPU alone (no oracle involved)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mapad_amd  # noqa: E402
from mapad_amd import synth  # noqa: E402
from mapad_amd.presets import DAMAGE, resolve  # noqa: E402

t0 = time.time()
g = synth.genome(300_000, seed=77)
seqs, quals, offsets = synth.reads(g, 4000, 50, seed=9, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(35, 100), indel_frac=0.05)
idx = mapad_amd.Index.build([("chr1", g)])
out = {}
for limits in ({}, {"stack_limit": 60, "edit_tree_limit": 100000}):
    for pops in (0, 48):
        ctx = mapad_amd.Context(idx, mapad_amd.make_params(dict(resolve(DAMAGE), **limits)), 0)
        ctx.set_tail_pops(pops)
        res = ctx.map_batch(seqs, quals, offsets)
        info = ctx.tail_info()
        out[(bool(limits), pops)] = (res, info)
        ctx.close()
    a, b = out[(bool(limits), 0)][0], out[(bool(limits), 48)][0]
    same = all(np.array_equal(getattr(a, k), getattr(b, k)) for k in ("hit_begin", "hits_arr", "ops", "status", "counters"))
    print(f"limits {limits or 'default'}: host tail took {out[(bool(limits), 48)][1]['reads']} reads ({out[(bool(limits), 48)][1]['host_pops']} pops, {out[(bool(limits), 48)][1]['threads']} threads, {out[(bool(limits), 48)][1]['host_thread_us'] / 1e6:.3f} thread-s in {out[(bool(limits), 48)][1]['host_us'] / 1e6:.3f} s); "
          f"{a.n_hits} hits; identical to the GPU-only batch: {same}", flush=True)
    assert same and out[(bool(limits), 48)][1]["reads"] > 100 and out[(bool(limits), 0)][1]["reads"] == 0
print(f"OK in {time.time() - t0:.1f} s")
