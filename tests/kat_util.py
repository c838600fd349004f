"""Helpers shared by the known-answer tests: fixture loading and symbolic-parameter resolution."""
import ctypes as C
import json
import os

import numpy as np

from oracle import binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def _log2f(x):
    # glibc log2f through the C library (numpy's SIMD log2 is not guaranteed to be bit-identical)
    libm = C.CDLL("libm.so.6")
    libm.log2f.restype = C.c_float
    libm.log2f.argtypes = [C.c_float]
    return libm.log2f(C.c_float(x))


def resolve_params(d, repr_mm_fn=None):
    """Fixture dict -> plain dict of numbers.  Symbolic entries:
    {"log2": x}, {"repr_mm_times": k}, {"repr_mm": true}, {"div3": x} (x_f32 / 3.0_f32)."""
    d = dict(d)
    for k, v in list(d.items()):
        if isinstance(v, dict) and "div3" in v:
            d[k] = float(np.float32(v["div3"]) / np.float32(3.0))
        elif isinstance(v, dict) and "log2" in v:
            d[k] = _log2f(v["log2"])
    symbolic = {k: v for k, v in d.items() if isinstance(v, dict)}
    if symbolic:
        base = {k: (0.0 if isinstance(v, dict) else v) for k, v in d.items()}
        if repr_mm_fn is None:
            repr_mm = ob.lib().mo_sdm_repr_mm(C.byref(ob.make_params(base)))
        else:
            repr_mm = repr_mm_fn(base)
        for k, v in symbolic.items():
            if "repr_mm_times" in v:
                d[k] = float(np.float32(repr_mm) * np.float32(v["repr_mm_times"]))
            elif "repr_mm" in v:
                d[k] = float(repr_mm)
            else:
                raise KeyError(v)
    return d


def oracle_params(d, **overrides):
    r = resolve_params(d)
    r.update(overrides)
    return ob.make_params(r)


def quals_for(pattern, q):
    return np.full(len(pattern), q, dtype=np.uint8) if isinstance(q, int) else np.asarray(q, dtype=np.uint8)
