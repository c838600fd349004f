"""Builds and wraps tests/emu/emu.cpp — the g++ build of the kernels' per-read logic (test-only, see the file header)."""
import ctypes as C
import os
import subprocess

import numpy as np

import mapad_amd
from mapad_amd import binding as mb

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "emu.cpp")
_libs = {}


def lib(heap_variant=0, subtree=0):
    """heap_variant: the reading of the frontier heap's tie rules the emulation is compiled with (csrc/heap_core.hpp: MAPAD_HEAP_VARIANT); subtree: MAPAD_SUBTREE_HEAP
    (the arena's heap levels in subtree blocks: profiles/request_attribution.py models both layouts)."""
    key = (heap_variant, subtree)
    _lib = _libs.get(key)
    _OUT = os.path.join(_HERE, "emu", "_build", "libmapad_emu" + (f".hv{heap_variant}" if heap_variant else "") + (".sub" if subtree else "") + ".so")
    if _lib is None:
        deps = [_SRC] + [os.path.join(_HERE, "..", "mapad_amd", "csrc", f) for f in os.listdir(os.path.join(_HERE, "..", "mapad_amd", "csrc")) if f.endswith((".hpp", ".hip"))]
        if not os.path.exists(_OUT) or any(os.path.getmtime(d) > os.path.getmtime(_OUT) for d in deps):
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin-log2f", "-fno-builtin-powf",
                                   "-fno-builtin-expf", "-fno-builtin-exp2f", "-fno-builtin-log10f", "-Wall",
                                   "-Wno-unused-function", "-Wno-unknown-pragmas", f"-DMAPAD_HEAP_VARIANT={int(heap_variant)}", f"-DMAPAD_SUBTREE_HEAP={int(subtree)}", "-o", _OUT + f".tmp{os.getpid()}", _SRC])
            os.replace(_OUT + f".tmp{os.getpid()}", _OUT)  # atomically: the two ranks of test_distributed may both find the library stale
        L = C.CDLL(_OUT)
        L.emu_map_batch.restype = C.POINTER(mb.BatchResultC)
        L.emu_map_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(mb.Params), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.emu_result_free.restype = None
        L.emu_result_free.argtypes = [C.POINTER(mb.BatchResultC)]
        L.emu_block_pos_selftest.restype = C.c_uint64
        L.emu_block_pos_selftest.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.emu_par_commit_selftest.restype = C.c_uint64
        L.emu_par_commit_selftest.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _libs[key] = _lib = L
    return _lib


def map_batch(index, params, seqs, quals, offsets, node_cap=4096, heap_cap=4096, heap_variant=0, subtree=0):
    """Runs the kernels' per-read logic on the host over the product's device-layout index."""
    blocks, nb, less, sent = index.device_view()
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    r = lib(heap_variant, subtree).emu_map_batch(blocks, nb, len(index), less.ctypes.data_as(C.c_void_p), sent.ctypes.data_as(C.c_void_p), C.byref(params),
                            seqs.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                            offsets.size - 1, node_cap, heap_cap)
    return mb.BatchResult(r, lib(heap_variant, subtree).emu_result_free)


_HS_SRC = os.path.join(_HERE, "emu", "heap_selftest.cpp")
_hs = {}


def heap_selftest_lib(subtree=1, heap_variant=0):
    """tests/emu/heap_selftest.cpp: the product's min-max heap in the build's physical layout (csrc/heap_core.hpp: HeapLayout; subtree=0: the implicit array) against
    the oracle's MinMaxHeap."""
    key = (int(subtree), int(heap_variant))
    if key not in _hs:
        out = os.path.join(_HERE, "emu", "_build", f"libheap_selftest.s{key[0]}.hv{key[1]}.so")
        csrc = os.path.join(_HERE, "..", "mapad_amd", "csrc")
        deps = [_HS_SRC, os.path.join(_HERE, "..", "oracle", "mapad_oracle.hpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-misleading-indentation",
                                   f"-DMAPAD_SUBTREE_HEAP={key[0]}", f"-DMAPAD_HEAP_VARIANT={key[1]}", "-o", out + f".tmp{os.getpid()}", _HS_SRC])
            os.replace(out + f".tmp{os.getpid()}", out)
        L = C.CDLL(out)
        L.heap_layout_properties.restype = C.c_uint64
        L.heap_layout_properties.argtypes = [C.c_uint32]
        L.heap_random_ops_selftest.restype = C.c_uint64
        L.heap_random_ops_selftest.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _hs[key] = L
    return _hs[key]


_TB_SRC = os.path.join(_HERE, "emu", "tail_bench.cpp")
_TB_OUT = os.path.join(_HERE, "emu", "_build", "libtail_bench.so")
_tb = None


def tail_bench_lib(extra_flags=()):
    """tests/emu/tail_bench.cpp: the host tail's search (mapad_amd/csrc/host_tail.hpp: tail_search, worker pinning, prefetch settings) on host threads without a GPU."""
    global _tb
    if _tb is None or extra_flags:
        csrc = os.path.join(_HERE, "..", "mapad_amd", "csrc")
        deps = [_TB_SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hpp", ".hip"))]
        if extra_flags or not os.path.exists(_TB_OUT) or any(os.path.getmtime(d) > os.path.getmtime(_TB_OUT) for d in deps):
            os.makedirs(os.path.dirname(_TB_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O3", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin-log2f", "-fno-builtin-powf",
                                   "-fno-builtin-expf", "-fno-builtin-exp2f", "-fno-builtin-log10f", "-Wno-unused-function", "-Wno-unknown-pragmas", *extra_flags,
                                   "-o", _TB_OUT + f".tmp{os.getpid()}", _TB_SRC])
            os.replace(_TB_OUT + f".tmp{os.getpid()}", _TB_OUT)
        L = C.CDLL(_TB_OUT)
        L.tail_bench.restype = C.c_double
        L.tail_bench.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(mb.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _tb = L
    return _tb


def tail_search(index, params, seqs, quals, offsets, sel, threads=4, max_pops=0):
    """Maps the reads `sel` of the batch from scratch the way a host-tail worker does.  Returns (seconds, pops, status, digest, seconds per read); `digest` is an FNV
    hash per read over status, the five search counters, every hit (interval, score bits, track length) and the edit tracks."""
    blocks, nb, less, sent = index.device_view()
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sel = np.ascontiguousarray(sel, dtype=np.uint32)
    pops = np.zeros(sel.size, dtype=np.uint64)
    status = np.zeros(sel.size, dtype=np.uint32)
    dig = np.zeros(sel.size, dtype=np.uint64)
    rs = np.zeros(sel.size, dtype=np.float64)
    secs = tail_bench_lib().tail_bench(blocks, nb, len(index), less.ctypes.data, sent.ctypes.data, C.byref(params), seqs.ctypes.data, quals.ctypes.data, offsets.ctypes.data,
                                       offsets.size - 1, sel.ctypes.data, sel.size, threads, max_pops, 1, pops.ctypes.data, status.ctypes.data, dig.ctypes.data, rs.ctypes.data)
    return secs, pops, status, dig, rs
