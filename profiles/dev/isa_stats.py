"""Register and instruction statistics of the kernels in a built library (no GPU needed):
    python profiles/dev/isa_stats.py [lib.so] [kernel-name-substring ...]
Unbundles the gfx950 code object(s), reads the kernel descriptors' metadata (VGPRs, SGPR spills, scratch) and counts instruction classes in the disassembly."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    import struct
    b = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        j = b.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if j < 0:
            break
        count, = struct.unpack_from("<Q", b, j + 24)
        p = j + 32
        nxt = p
        for _ in range(count):
            eo, es, tl = struct.unpack_from("<QQQ", b, p)
            triple = b[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            nxt = max(nxt, j + eo + es)
            if "amdgcn" in triple and es:
                out.append(b[j + eo:j + eo + es])
        pos = nxt
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "mapad_amd", "libmapad_amd.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["search_kernel", "heavy_kernel", "darray_kernel"]
    for k, co in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            path = f.name
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        os.unlink(path)
        # metadata blocks: one per kernel
        for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            sym = name.group(1)
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            if not any(p in dem for p in pats):
                continue
            g = lambda key: (re.search(r"\." + key + r":\s+(\d+)", blk) or [None, "?"])[1]  # noqa: E731
            body = re.search(r"\n[0-9a-f]+ <" + re.escape(sym) + r">:\n(.*?)(?=\n\n[0-9a-f]+ <|\Z)", dis, re.S)
            counts = {}
            if body:
                for line in body.group(1).splitlines():
                    m = re.match(r"\s+(\w+)", line)
                    if m:
                        op = m.group(1)
                        cls = ("readlane" if "readlane" in op else "writelane" if "writelane" in op else "global_load" if op.startswith("global_load") else
                               "global_store" if op.startswith("global_store") else "global_atomic" if op.startswith("global_atomic") else "flat" if op.startswith("flat_") else
                               "scratch" if op.startswith("scratch_") else "ds" if op.startswith("ds_") else "s_waitcnt" if op == "s_waitcnt" else "valu" if op.startswith("v_") else
                               "salu" if op.startswith("s_") else "other")
                        counts[cls] = counts.get(cls, 0) + 1
            print(f"{dem[:100]}\n   vgpr {g('vgpr_count')} sgpr {g('sgpr_count')} sgpr_spill {g('sgpr_spill_count')} vgpr_spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}"
                  f"\n   " + " ".join(f"{k}={v}" for k, v in sorted(counts.items())))


if __name__ == "__main__":
    main()
