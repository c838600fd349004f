"""GPU parity of the other three readings of the frontier heap's tie rules (csrc/heap_core.hpp: MAPAD_HEAP_VARIANT; the default library is reading 0 and is what
every other GPU test runs).  The crate's source (`min-max-heap`, call sites src/map/mapping.rs:147,986,1058,1376) is not in /root/reference and the reference's own
tests pass under all four readings; the product can be built for each (mapad_amd/build.py: libmapad_amd.hvV.so, built by __graft_entry__.build()), and reading v of
the product must equal reading v of the oracle bit for bit — hits, BinaryHeap order, score bits, edit tracks, D arrays, the six event counters — on synthetic batches
of every read mix, through the quad kernel, the lane-parallel commit, arena growth and the host tail.  One process per reading: a process loads one library."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["MAPAD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MAPAD_ROOT"], "tests"))
import mapad_amd
from mapad_amd import build, synth
from oracle import binding as ob
from kat_util import resolve_params
from parity_util import CONTINUOUS, DAMAGE, NO_DAMAGE, assert_same_as_oracle, split_reads
v = build.selected_variant()
flavour = os.environ.get("MAPAD_TEST_FLAVOUR")  # a build flavour of reading 0 (mapad_amd/build.py: FLAVOURS), loaded through MAPAD_AMD_LIB
assert v == int(sys.argv[1]) and (flavour or os.path.basename(build.lib_path(v)) == f"libmapad_amd.hv{v}.so")
assert not flavour or (v == 0 and os.path.basename(os.environ["MAPAD_AMD_LIB"]) == f"libmapad_amd.{flavour}.so")
g = synth.genome(300_000, seed=99)
pidx = mapad_amd.Index.build([("chr1", g)])
oidx = ob.OracleIndex.from_bwt(pidx.bwt(), "$ACGTX", 128)
differs = 0
for name, prm, kw, n, env, tail in (
        ("no_damage", NO_DAMAGE, dict(qual=40), 3000, {}, None),
        ("damage", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 3000, {}, None),
        ("mixed_len_indels", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 800, {}, None),
        ("continuous", CONTINUOUS, dict(qual_range=(20, 40), len_range=(35, 70), indel_frac=0.05), 600, {}, None),
        ("tiny_arenas", DAMAGE, dict(qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0)), 3000, {"MAPAD_TIER0_NODES": "32", "MAPAD_CLASS_COUNTS": "128,128,128,64,64,64,64,64,64,16"}, None),
        ("host_tail", DAMAGE, dict(qual_range=(20, 40), len_range=(35, 100), indel_frac=0.05), 800, {}, 64),
        ("limits", NO_DAMAGE, dict(qual=40), 600, {"_limits": "1"}, None)):
    for k, val in env.items():
        if not k.startswith("_"):
            os.environ[k] = val
    seqs, quals, offsets = synth.reads(g, n, 50, seed=7 + len(name), **kw)
    rp = resolve_params(prm)
    if env.get("_limits"):
        rp = dict(rp, stack_limit=40, edit_tree_limit=100000)
    ctx = mapad_amd.Context(pidx, mapad_amd.make_params(rp), 0)
    if tail is not None:
        ctx.set_tail_pops(tail)
    res = ctx.map_batch(seqs, quals, offsets)
    info = ctx.tail_info()
    ctx.close()
    for k in env:
        os.environ.pop(k, None)
    reads, qs = split_reads(seqs, quals, offsets)
    ores = oidx.map_batch(ob.make_params(dict(rp, heap_variant=v)), reads, qs, n_threads=8, keep_d=True)
    assert_same_as_oracle(ores, res, offsets)
    if tail is not None:
        assert info["reads"] > 20
    o0 = oidx.map_batch(ob.make_params(rp), reads, qs, n_threads=8)
    differs += int((o0.counters != ores.counters).any(axis=1).sum())
    print(name, "ok", flush=True)
assert differs > 0 or flavour  # this reading is not reading 0
print("variant", v, "identical to the matching oracle reading;", differs, "reads differ from reading 0 in their event counters")
'''


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_product_reading_equals_the_matching_oracle_reading_on_the_gpu(variant, tmp_path):
    sys.path.insert(0, ROOT)
    from mapad_amd import build
    if not os.path.exists(build.lib_path(variant)):
        build.build(heap_variant=variant)  # (normally built by __graft_entry__.build() and shipped with the snapshot)
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, MAPAD_HEAP_VARIANT=str(variant), MAPAD_ROOT=ROOT)
    env.pop("MAPAD_AMD_LIB", None)
    pr = subprocess.run([sys.executable, str(script), str(variant)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert pr.returncode == 0, pr.stdout[-3000:]
    assert f"variant {variant} identical" in pr.stdout


def test_subtree_block_heap_layout_build_equals_the_oracle_on_the_gpu(tmp_path):
    """csrc/heap_core.hpp: MAPAD_SUBTREE_HEAP=1 — the arena's heap levels in subtree-contiguous 64-byte blocks (libmapad_amd.sub.so; measured slower than the implicit
    array and therefore not the default: profiles/r06/ab_layout.txt).  A logical -> physical slot change under the same min-max heap: the same batches as above —
    quad kernel, speculative sift, lane-parallel commit, arena growth and migrations, host tail with continuation from the arena, limits — bit for bit against the oracle."""
    sys.path.insert(0, ROOT)
    from mapad_amd import build
    if not os.path.exists(build.lib_path(flavour="sub")):
        build.build(flavour="sub")
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, MAPAD_HEAP_VARIANT="0", MAPAD_ROOT=ROOT, MAPAD_AMD_LIB=build.lib_path(flavour="sub"), MAPAD_TEST_FLAVOUR="sub")
    pr = subprocess.run([sys.executable, str(script), "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert pr.returncode == 0, pr.stdout[-3000:]
    assert "variant 0 identical" in pr.stdout
