"""Builds and wraps tests/emu/emu.cpp — the g++ build of the kernels' per-read logic (test-only, see the file header)."""
import ctypes as C
import os
import subprocess

import numpy as np

import mapad_amd
from mapad_amd import binding as mb

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "emu.cpp")
_OUT = os.path.join(_HERE, "emu", "_build", "libmapad_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [_SRC] + [os.path.join(_HERE, "..", "mapad_amd", "csrc", f) for f in os.listdir(os.path.join(_HERE, "..", "mapad_amd", "csrc")) if f.endswith((".hpp", ".hip"))]
        if not os.path.exists(_OUT) or any(os.path.getmtime(d) > os.path.getmtime(_OUT) for d in deps):
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin-log2f", "-fno-builtin-powf",
                                   "-fno-builtin-expf", "-fno-builtin-exp2f", "-fno-builtin-log10f", "-Wall",
                                   "-Wno-unused-function", "-Wno-unknown-pragmas", "-o", _OUT + f".tmp{os.getpid()}", _SRC])
            os.replace(_OUT + f".tmp{os.getpid()}", _OUT)  # atomically: the two ranks of test_distributed may both find the library stale
        L = C.CDLL(_OUT)
        L.emu_map_batch.restype = C.POINTER(mb.BatchResultC)
        L.emu_map_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(mb.Params), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.emu_result_free.restype = None
        L.emu_result_free.argtypes = [C.POINTER(mb.BatchResultC)]
        L.emu_par_commit_selftest.restype = C.c_uint64
        L.emu_par_commit_selftest.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def map_batch(index, params, seqs, quals, offsets, node_cap=4096, heap_cap=4096):
    """Runs the kernels' per-read logic on the host over the product's device-layout index."""
    blocks, nb, less, sent = index.device_view()
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    r = lib().emu_map_batch(blocks, nb, len(index), less.ctypes.data_as(C.c_void_p), sent.ctypes.data_as(C.c_void_p), C.byref(params),
                            seqs.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                            offsets.size - 1, node_cap, heap_cap)
    return mb.BatchResult(r, lib().emu_result_free)
