"""Builds libmapad_amd.so (HIP kernels + C ABI) in-tree for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmapad_amd.so")
SOURCES = ["mapad_amd.hip", "index_gpu.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         # device code at -O2: fewer branches and scalar instructions in the search kernel than -O3 (measured 1-3 % faster on MI355X)
         "-Xarch_device", "-O2",
         # small per-thread arrays of the kernels around the search stay out of LDS (those kernels must fit beside the search wavefronts, which own it)
         "-mllvm", "-disable-promote-alloca-to-lds",
         # keep libm out of the constant folder / pow->exp10 rewrites: the reference evaluates these at run time with glibc
         "-fno-builtin-log2f", "-fno-builtin-powf", "-fno-builtin-expf", "-fno-builtin-exp2f", "-fno-builtin-log10f",
         "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]


CLI = os.path.join(HERE, "mapad-amd")
CLI_SRC = os.path.join(CSRC, "cli", "main.cpp")


def build_cli(force=False, verbose=False):
    """`mapad-amd` command line (index / map) on top of the C ABI; plain host C++ linked against the library."""
    cli_dir = os.path.join(CSRC, "cli")
    deps = [os.path.join(cli_dir, f) for f in os.listdir(cli_dir)] + [os.path.join(os.path.dirname(HERE), "include", "mapad_amd.h"), LIB]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(d) <= os.path.getmtime(CLI) for d in deps):
        return CLI
    cmd = [HIPCC, "-O2", "-std=c++17", "-x", "c++", CLI_SRC, "-o", CLI, "-L" + HERE, "-lmapad_amd", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "mapad_amd.h"))
    return out


def source_hash():
    """sha256 (16 hex digits) over the sources the library is built from (host and device parts alike); recorded beside kernel_code_hash()."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(p for p in _deps() if p.endswith((".hip", ".hpp", ".h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _elf_sections(b, base=0):
    import struct
    shoff, = struct.unpack_from("<Q", b, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
    sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, base + shoff + i * shentsize)  # noqa: E731
    stro = sh(shstrndx)[4]
    out = {}
    for i in range(shnum):
        s = sh(i)
        name = b[base + stro + s[0]:b.index(b"\0", base + stro + s[0])].decode()
        out.setdefault(name, []).append((base + s[4], s[5], s[1]))
    return out


def kernel_code_hash(lib=None):
    """sha256 (16 hex digits) over the gfx950 machine code of the built library: the .text and .rodata sections of every amdgcn code object in its
    `.hip_fatbin` offload bundles.  This is what ties measurements that are not taken live (the PMC traffic of profiles/traffic.json) to the kernels
    they were taken with: it changes with any change of the device code (sources, flags, compiler) and with nothing else — the symbol tables and bundle
    ids, which differ from build to build, and the host code are not in it.  None when the library is not built."""
    import hashlib
    import struct
    lib = lib or LIB
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as f:
        b = f.read()
    h, n = hashlib.sha256(), 0
    for off, size, _ in _elf_sections(b).get(".hip_fatbin", []):
        pos, end = off, off + size
        while pos < end:
            j = b.find(b"__CLANG_OFFLOAD_BUNDLE__", pos, end)
            if j < 0:
                break
            count, = struct.unpack_from("<Q", b, j + 24)
            p, nxt = j + 32, j + 32
            for _ in range(count):
                eo, es, tl = struct.unpack_from("<QQQ", b, p)
                triple = b[p + 24:p + 24 + tl].decode()
                p += 24 + tl
                nxt = max(nxt, j + eo + es)
                if "amdgcn" in triple and es:
                    secs = _elf_sections(b, j + eo)
                    for name in (".text", ".rodata"):
                        for o, s, t in secs.get(name, []):
                            if t != 8:  # not SHT_NOBITS
                                h.update(name.encode())
                                h.update(b[o:o + s])
                                n += s
            pos = nxt
    return h.hexdigest()[:16] if n else None


# build flavours beside the default library: name -> (file, extra -D flags)
#   hv1..hv3: the other readings of the frontier heap's tie rules (csrc/heap_core.hpp: MAPAD_HEAP_VARIANT)        tests/test_gpu_heap_variants.py
#   heavy   : heavy_kernel.hpp compiled in (one wavefront per read, MAPAD_HEAVY=1; off the default path since round 5)  tests/test_gpu_parity.py (child processes)
#   sub     : the arena's heap levels in subtree-contiguous 64-byte blocks (csrc/heap_core.hpp: MAPAD_SUBTREE_HEAP=1; measured slower, kept as an option) tests/test_gpu_heap_variants.py
FLAVOURS = {"hv1": ["-DMAPAD_HEAP_VARIANT=1"], "hv2": ["-DMAPAD_HEAP_VARIANT=2"], "hv3": ["-DMAPAD_HEAP_VARIANT=3"], "heavy": ["-DMAPAD_HEAVY_KERNEL"], "sub": ["-DMAPAD_SUBTREE_HEAP=1"]}


def _flavour(heap_variant=0, heavy=False, flavour=None):
    if flavour:
        return flavour
    if heavy:
        return "heavy"
    return f"hv{int(heap_variant)}" if heap_variant else None


def lib_path(heap_variant=0, heavy=False, flavour=None):
    """The library of one build flavour (FLAVOURS); no arguments = the default library."""
    f = _flavour(heap_variant, heavy, flavour)
    return LIB if not f else os.path.join(HERE, f"libmapad_amd.{f}.so")


def selected_variant():
    """MAPAD_HEAP_VARIANT in the environment picks the library a process loads (binding.lib()); default 0."""
    v = int(os.environ.get("MAPAD_HEAP_VARIANT", "0") or 0)
    if not 0 <= v <= 3:
        raise ValueError("MAPAD_HEAP_VARIANT must be 0..3")
    return v


def needs_build(heap_variant=0, heavy=False, flavour=None):
    lib = lib_path(heap_variant, heavy, flavour)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False, heap_variant=0, heavy=False, flavour=None):
    f = _flavour(heap_variant, heavy, flavour)
    lib = lib_path(flavour=f)
    if not force and not needs_build(flavour=f):
        return lib
    extra = FLAVOURS[f] if f else []
    tmp = lib + f".tmp{os.getpid()}"
    cmd = [HIPCC] + FLAGS + extra + os.environ.get("MAPAD_EXTRA_FLAGS", "").split() + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, lib)
    return lib


def build_all(force=False, verbose=False, flavours=(None,) + tuple(FLAVOURS)):
    """The default library and every flavour, side by side (one hipcc each; they share nothing but the sources)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=6) as ex:
        return list(ex.map(lambda f: build(force=force, verbose=verbose, flavour=f), flavours))


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_cli(force="--force" in sys.argv, verbose=True)
    print(LIB)
