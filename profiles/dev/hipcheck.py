"""Development aid: does a raw hipMalloc through ctypes still work after a sequence of library calls?"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import mapad_amd
from mapad_amd import synth
from kat_util import resolve_params
from parity_util import DAMAGE

hip = C.CDLL("libamdhip64.so")


def check(tag):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(4096))
    n = C.c_int(-1)
    rc2 = hip.hipGetDeviceCount(C.byref(n))
    fds = len(os.listdir("/proc/self/fd"))
    print(tag, "hipMalloc rc", rc, "deviceCount rc", rc2, n.value, "open fds", fds, flush=True)
    if rc == 0:
        hip.hipFree(p)


check("start")
g = synth.genome(200_000, seed=3)
idx = mapad_amd.Index.build([("chr1", g)], device=0)
check("after index")
seqs, quals, offsets = synth.reads(g, 5000, 50, seed=5, qual_range=(20, 40))
params = mapad_amd.make_params(resolve_params(DAMAGE))
for it in range(3):
    ctx = mapad_amd.Context(idx, params, 0)
    res = ctx.map_batch(seqs, quals, offsets)
    check(f"iter {it} after map")
    recs = ctx.hits_to_records(res, seqs, quals, offsets, seed=7)
    check(f"iter {it} after records")
    ctx.close()
    check(f"iter {it} after close")
