"""`mapad-amd` command line: index files and record I/O on the CPU; the full `map` run (FASTA -> index -> BAM in -> BAM out)
against the reference's integration expectation on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import mapad_amd
from mapad_amd import build as mbuild
from mapad_amd import synth

from bam_util import read_bam, write_bam
from kat_util import load


def _cli():
    mapad_amd.lib()
    return mbuild.build_cli()


def _write_fasta(path, contigs):
    with open(path, "w") as f:
        for name, seq in contigs:
            f.write(f">{name} some description\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")


def test_cli_index_writes_the_seven_files(tmp_path):
    g = synth.genome(30_000, seed=4).tobytes().decode()
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, [("chrA", g[:20_000]), ("chrB", g[20_000:].lower())])
    subprocess.check_call([_cli(), "--seed", "99", "index", "-g", fa])
    for ext in ("tbw", "tle", "toc", "trt", "tsa", "tpi", "tos"):
        assert os.path.exists(f"{fa}.{ext}")
    a = mapad_amd.Index.open(fa)
    b = mapad_amd.Index.build([("chrA", g[:20_000].encode()), ("chrB", g[20_000:].encode())], seed=99)
    assert np.array_equal(a.bwt(), b.bwt()) and a.contigs() == b.contigs() == [("chrA", 0, 19_999), ("chrB", 20_000, 29_999)]


def test_cli_record_io_roundtrip(tmp_path):
    k = load("integration")
    inp, out = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    recs = [dict(r, tags=[("XI", "Z", "ACGACGT"), ("FF", "i", 3), ("AS", "i", 0), ("MD", "Z", "28"), ("RG", "Z", "A12345")]) for r in k["reads"]]
    write_bam(inp, "@HD\tVN:1.0\n@RG\tID:A12345\tSM:Sample1\n", [("chr1", 600)], recs)
    subprocess.check_call([_cli(), "recode", "-r", inp, "-o", out])
    _, _, got = read_bam(out)
    comp = str.maketrans("ACGTN", "TGCAN")
    assert len(got) == len(recs)
    for g, r in zip(got, recs):
        # input records flagged 0x10 are un-reversed on read (record.rs:157-160); mapAD-specific tags are dropped (mapping.rs:834-848)
        seq, qual = (r["seq"].translate(comp)[::-1], r["qual"][::-1]) if r["flags"] & 0x10 else (r["seq"], r["qual"])
        assert (g["name"], g["seq"], g["qual"]) == (r["name"], seq, qual)
        assert g["flags"] & 0x4 and g["tid"] == -1 and g["pos"] == -1 and g["cigar"] == ""
        assert g["tag_order"] == ["XI", "FF", "RG", "XD"]
    # FASTQ (gz) input
    fq = str(tmp_path / "in.fastq.gz")
    import gzip
    with gzip.open(fq, "wt") as f:
        for r in k["reads"][:5]:
            f.write(f"@{r['name']} extra\n{r['seq'].lower()}\n+\n{r['qual']}\n")
    subprocess.check_call([_cli(), "recode", "-r", fq, "-o", out, "-R", "RG01"])
    _, _, got = read_bam(out)
    assert [(g["name"], g["seq"], g["qual"]) for g in got] == [(r["name"], r["seq"], r["qual"]) for r in k["reads"][:5]]
    assert all(g["tags"]["RG"] == ("Z", "RG01") for g in got)


@pytest.mark.gpu
def test_cli_map_matches_integration_expectation(tmp_path, monkeypatch):
    """tests/integration_tests.rs:174-215 through the command line: header prefix and every decoded record field."""
    k = load("integration")
    monkeypatch.delenv("MAPAD_INDEX_FIXED_REPLACEMENT", raising=False)  # StdRng(1234) itself must draw the base the reference's expectation implies
    fa, inp, out = str(tmp_path / "test_genome.fa"), str(tmp_path / "input_reads.bam"), str(tmp_path / "out.bam")
    _write_fasta(fa, [(c["name"], c["seq"]) for c in k["contigs"]])
    header = ("@HD\tVN:1.0\n@RG\tID:A12345\tSM:Sample1\n@SQ\tSN:chr1\tLN:600\n"
              "@PG\tID:samtools\tPN:samtools\tVN:1.13\tCL:samtools view -h interesting_specimen.bam -o input_reads.bam\n"
              "@PG\tID:mapAD\tPN:mapAD\tCL:mapad map\tPP:samtools\tDS:An aDNA aware short-read mapper\tVN:0.0.33\n"
              "@PG\tID:mapAD.1\tPN:mapAD\tCL:mapad map\tPP:mapAD\tDS:An aDNA aware short-read mapper\tVN:0.0.33\n")
    tags = [("XI", "Z", "ACGACGT"), ("YI", "Z", ":BBBBGG"), ("FF", "i", 3), ("RG", "Z", "A12345")]
    recs = [dict(r, tags=tags if i < 7 else []) for i, r in enumerate(k["reads"])]
    write_bam(inp, header, [("chr1", 600)], recs)
    subprocess.check_call([_cli(), "--seed", "1234", "index", "-g", fa])
    # integration parameters (tests/integration_tests.rs:140-163): gap open = 1.5 * repr_mm is not reachable through -i,
    # so this run uses the CLI derivation (-i 0.001 -> log2) and is compared with the C-ABI result for the same parameters
    cmd = [_cli(), "map", "-r", inp, "-g", fa, "-o", out, "-l", "single_stranded", "-p", "0.03", "-f", "0.6", "-t", "0.55", "-d", "0.01", "-s", "1.0",
           "-D", "0.02", "-i", "0.001", "-x", "0.5", "--batch_size", "5"]
    subprocess.check_call(cmd)
    text, refs, got = read_bam(out)
    assert text.split("\n")[:5] == k["header_prefix"]
    assert refs == [("chr1", 600), ("Chromosome_02", 600), ("Chromosome_03", 84), ("Chromosome_04", 46)]
    lines = text.strip().split("\n")
    assert lines[5] == "@RG\tID:A12345\tSM:Sample1" and lines[6].startswith("@PG\tID:samtools") and lines[8].startswith("@PG\tID:mapAD.1")
    assert lines[9].startswith("@PG\tID:mapAD.2\tPN:mapAD") and lines[9].endswith("PP:mapAD.1")
    # same parameters through the C ABI
    from test_oracle_kats import integration_reads
    idx = mapad_amd.Index.open(fa)
    params = mapad_amd.params_from_cli(library="single_stranded", five_prime_overhang=0.6, three_prime_overhang=0.55, ds_deamination_rate=0.01,
                                       ss_deamination_rate=1.0, divergence=0.02, poisson_prob=0.03, indel_rate=0.001, gap_extension_penalty=0.5)
    reads, quals = integration_reads(k)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    seqs, qs = np.frombuffer(b"".join(reads), dtype=np.uint8), np.concatenate(quals)
    ctx = mapad_amd.Context(idx, params, 0)
    res = ctx.map_batch(seqs, qs, offsets)
    want = mapad_amd.hits_to_records(idx, params, res, seqs, qs, offsets, in_flags=[r["flags"] for r in k["reads"]], seed=1234)
    ctx.close()
    assert [g["name"] for g in got] == [r["name"] for r in k["reads"]]  # output order = input order (mapping.rs:288)
    n_mapped = 0
    for g, w_, r in zip(got, want, recs):
        assert g["flags"] == w_["flags"] and g["mapq"] == w_["mapq"] and g["tid"] == w_["tid"] and g["pos"] == w_["pos"]
        assert g["cigar"] == w_["cigar"]
        if w_["mapped"]:
            n_mapped += 1
            t = g["tags"]
            assert t["MD"] == ("Z", w_["md"]) and t["NM"] == ("i", w_["nm"]) and t["X0"] == ("i", w_["x0"]) and t["X1"] == ("i", w_["x1"])
            assert np.float32(t["AS"][1]) == w_["as"] and t["XT"] == ("A", w_["xt"]) and (t.get("XA", ("Z", ""))[1] == w_["xa"])
            assert ("XS" in t) == (w_["xs"] is not None)
            new = [x for x in g["tag_order"] if x not in ("XI", "YI", "FF", "RG")]
            assert new == [x for x in ["AS", "NM", "MD", "XA", "X0", "X1", "XS", "XT", "XD"] if x in t]  # aux order of mapping.rs:850-918
        else:
            assert g["tags"].keys() <= {"XI", "YI", "FF", "RG", "XD"}
        assert g["tag_order"][:len(r["tags"])] == [x[0] for x in r["tags"]]  # input tags first
    assert n_mapped >= 14


@pytest.mark.gpu
def test_cli_map_reads_cram_like_bam(tmp_path):
    """The integration reads as a CRAM 3.0 file (tests/cram_util.py) map to the records the same reads give as BAM input (header chain included)."""
    import cram_util as cu
    from test_cram import _unmapped_series
    k = load("integration")
    fa, bam_in, cram_in = str(tmp_path / "g.fa"), str(tmp_path / "in.bam"), str(tmp_path / "in.cram")
    _write_fasta(fa, [(c["name"], c["seq"]) for c in k["contigs"]])
    header = "@HD\tVN:1.0\n@RG\tID:A12345\tSM:Sample1\n@SQ\tSN:chr1\tLN:600\n@PG\tID:samtools\tPN:samtools\tVN:1.13\tCL:samtools view\n"
    tags = [("XI", "Z", "ACGACGT"), ("FF", "i", 3), ("RG", "Z", "A12345")]
    recs = [dict(r, tags=tags if i < 7 else [], flags=r["flags"] | 0x4) for i, r in enumerate(k["reads"])]  # unmapped input: a CRAM record's bases are then its own BA series
    write_bam(bam_in, header, [("chr1", 600)], recs)
    series = _unmapped_series()
    tag_lines = [[], [(b"XI", "Z"), (b"FF", "i")]]
    tag_encs = {(b"XI", "Z"): cu.ByteArrayStop(ord("\t"), 8), (b"FF", "i"): cu.ByteArrayLen(cu.Huffman({4: 0}), cu.External(9))}
    st = cu.SliceStreams()
    for i, r in enumerate(recs):
        series["BF"].put(st, r["flags"]); series["CF"].put(st, 3); series["RL"].put(st, len(r["seq"])); series["AP"].put(st, 0)
        series["RG"].put(st, 0 if i < 7 else -1)  # the read group travels as an index into the header's @RG lines
        series["RN"].put(st, r["name"].encode())
        series["MF"].put(st, 0); series["NS"].put(st, -1); series["NP"].put(st, 0); series["TS"].put(st, 0)
        series["TL"].put(st, 1 if i < 7 else 0)
        if i < 7:
            tag_encs[(b"XI", "Z")].put(st, cu.aux_value("Z", "ACGACGT")); tag_encs[(b"FF", "i")].put(st, cu.aux_value("i", 3))
        for b in r["seq"]:
            series["BA"].put(st, ord(b))
        for q in r["qual"]:
            series["QS"].put(st, ord(q) - 33)
    ch = cu.compression_header(series, tag_encs, tag_lines)
    blocks = [cu.block(cu.GZIP, cu.COMPRESSION_HEADER, 0, ch)] + cu.slice_blocks(-1, 0, 0, len(recs), 0, st, {6: cu.RANS1, 7: cu.RANS0})
    with open(cram_in, "wb") as f:
        f.write(cu.file_start(header) + cu.container(-1, 0, 0, len(recs), 0, 0, blocks, [0]) + cu.eof_container())
    subprocess.check_call([_cli(), "--seed", "1234", "index", "-g", fa])
    outs = []
    for inp in (bam_in, cram_in):
        out = str(tmp_path / (os.path.basename(inp) + ".out.bam"))
        subprocess.check_call([_cli(), "--seed", "7", "map", "-r", inp, "-g", fa, "-o", out, "-l", "single_stranded", "-p", "0.03", "-f", "0.6", "-t", "0.55", "-d", "0.01", "-s", "1.0",
                               "-D", "0.02", "-i", "0.001", "-x", "0.5"])
        outs.append((read_bam(out)[0].split("\n")[:-2], _decoded(out)))
    assert outs[0][0][:9] == outs[1][0][:9]  # header: everything but the @PG line of this run (its CL names the input file)
    assert outs[0][1] == outs[1][1] and sum(1 for r in outs[0][1][1] if r[2] >= 0) >= 14


def _decoded(path):
    """decoded records without the wall-time tag XD (SURVEY §8c parity definition: compare decoded records, ignore XD and @PG CL)"""
    text, refs, recs = read_bam(path)
    out = []
    for r in recs:
        tags = {k: v for k, v in r["tags"].items() if k != "XD"}
        out.append((r["name"], r["flags"], r["tid"], r["pos"], r["mapq"], r["cigar"], r["seq"], r["qual"], tuple(sorted(tags.items())), tuple(x for x in r["tag_order"] if x != "XD")))
    return refs, out


@pytest.mark.gpu
def test_cli_map_devices_chunks_and_unmappable_reads(tmp_path):
    """`--devices 0,0` (two contexts and worker threads, contiguous slices of every chunk) and small chunks (a pipeline of many batches)
    must write the same records as one device and one chunk; empty reads and reads beyond i16::MAX come out unmapped, in place; a 1 500 bp read maps."""
    g = synth.genome(120_000, seed=17)
    fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fastq")
    _write_fasta(fa, [("chr1", g[:70_000].tobytes().decode()), ("chr2", g[70_000:].tobytes().decode())])
    seqs, quals, offsets = synth.reads(g, 4000, 50, seed=23, qual_range=(20, 40), damage=dict(f=0.5, t=0.5, d=0.02, s=1.0), len_range=(30, 80))
    long_read = g[1000:2500].tobytes().decode()  # 1 500 bp: mapped (position data of such a chunk lives in HBM instead of LDS)
    too_long = (g.tobytes() * 1)[:33_000].decode()  # > i16::MAX (MAPAD_MAX_READ_LEN, record.rs:144-150): written as an unmapped record
    with open(fq, "w") as f:
        for i in range(4000):
            s, e = int(offsets[i]), int(offsets[i + 1])
            f.write(f"@r{i}\n{seqs[s:e].tobytes().decode()}\n+\n{''.join(chr(33 + q) for q in quals[s:e])}\n")
            if i == 777:
                f.write(f"@too_long\n{too_long}\n+\n{'I' * len(too_long)}\n")
            if i == 1200:
                f.write(f"@long\n{long_read}\n+\n{'I' * len(long_read)}\n")
            if i == 2500:
                f.write("@empty\n\n+\n\n")
    subprocess.check_call([_cli(), "index", "-g", fa])
    base = [_cli(), "map", "-r", fq, "-g", fa, "-l", "single_stranded", "-p", "0.03", "-f", "0.5", "-t", "0.5", "-d", "0.02", "-s", "1.0", "-i", "0.001", "--seed", "7"]
    outs = {}
    for tag, extra in (("one", ["--devices", "0", "--batch_size", "250000"]), ("two", ["--devices", "0,0", "--batch_size", "250000"]),
                       ("chunks", ["--devices", "0", "--batch_size", "301"]), ("chunks_serial", ["--devices", "0", "--batch_size", "301", "--in_flight", "1"]),
                       ("chunks_deep", ["--devices", "0,0", "--batch_size", "301", "--in_flight", "7"]), ("coalesced", ["--devices", "0", "--batch_size", "301", "--coalesce", "3"])):
        out = str(tmp_path / f"{tag}.bam")
        subprocess.check_call(base + ["-o", out] + extra)
        outs[tag] = _decoded(out)
    refs, one = outs["one"]
    assert refs == [("chr1", 70_000), ("chr2", 50_000)]
    assert len(one) == 4003 and [r[0] for r in one][778] == "too_long" and [r[0] for r in one][1202] == "long" and [r[0] for r in one][2503] == "empty"
    for name in ("too_long", "empty"):
        r = next(x for x in one if x[0] == name)
        assert r[1] & 0x4 and r[2] == -1 and r[3] == -1 and r[5] == ""
    lr = next(x for x in one if x[0] == "long")
    assert not (lr[1] & 0x4) and lr[2] == 0 and lr[3] == 1000 and lr[5] == "1500M"
    assert sum(1 for r in one if not (r[1] & 0x4)) > 3000
    assert outs["two"][1] == one  # identical BAM records whichever device mapped a read
    # chunk boundaries change nothing (round 4: the seed of a read counts in reads of the run, not of its chunk), nor does the number of chunks in flight
    # (1 = strictly serial, 7 = every chunk's stream on its own hardware queue), nor handing several chunks to the device as one launch (--coalesce)
    assert outs["chunks"][1] == one
    assert outs["chunks_serial"][1] == one and outs["chunks_deep"][1] == one and outs["coalesced"][1] == one
