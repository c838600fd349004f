#!/usr/bin/env python3
"""Generates tests/golden/*.json from the known-answer tests of the reference (mpieva/mapAD v0.45.0).

Run in the build container only (reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py

Every fixture is DATA: inputs (reference text, reads, parameters) and expected outputs that are literal
in the reference's own tests.  Each case records the reference file:line it was transcribed from.
Parameter values that the reference computes at run time are kept symbolic:
    {"log2": x}          -> x_f32.log2()
    {"repr_mm_times": k} -> k * sdm.get_representative_mismatch_penalty()   (f32 multiply)
    {"repr_mm": true}    -> sdm.get_representative_mismatch_penalty()
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def src(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


def test_model(deam, mm, match):
    return {"model": "test", "deam_score": deam, "mm_score": mm, "match_score": match}


def test_bound(threshold, repr_mm):
    return {"bound": "test", "threshold": threshold, "repr_mm_bound": repr_mm}


def gaps(open_, ext, gde, mgo):
    return {"penalty_gap_open": open_, "penalty_gap_extend": ext, "gap_dist_ends": gde, "max_num_gaps_open": mgo}


def search_kats():
    mapping = src("src/map/mapping.rs")
    cases = []

    def case(name, line, reference, params, pattern, qual, expect):
        cases.append({"name": name, "ref": f"src/map/mapping.rs:{line}", "reference": reference, "params": params,
                      "pattern": pattern, "qual": qual, "expect": expect})

    # test_inexact_search :1401
    case("inexact_search", 1401, "ACGTACGTACGTACGT", {**test_model(-0.5, -1.0, 0.0), **test_bound(-1.0, -1.0), **gaps(-2.0, -1.0, 0, 2)},
         "GTTC", 0, {"heap_scores": [-1.0], "positions_sorted": [2, 6, 10, 19, 23, 27]})
    # test_reverse_strand_search :1458
    case("reverse_strand_search", 1458, "GAAAAG", {**test_model(-10.0, -10.0, 0.0), **test_bound(-1.0, -10.0), **gaps(-20.0, -10.0, 0, 2)},
         "TTTT", 0, {"positions_sorted": [8]})
    # test_gapped_alignment :1512
    case("gapped_alignment", 1512, "TAT", {**test_model(-10.0, -10.0, 0.0), **test_bound(-3.0, -10.0), **gaps(-2.0, -1.0, 0, 2)},
         "TT", 0, {"positions_sorted": [0, 2, 5]})
    # test_gapped_alignment_read_end :1566
    p = {**test_model(-10.0, -10.0, 0.0), **test_bound(-6.0, -10.0), **gaps(-2.0, -1.0, 5, 2)}
    case("gapped_alignment_read_end_allowed", 1595, "AAAAAAGGGGAAAAAA", p, "AAAAAAAAAAAA", 0, {"nonempty": True})
    case("gapped_alignment_read_end_forbidden", 1618, "AAAAAAGGGGAAAAAA", p, "AGGGAAAAAA", 0, {"positions_sorted": []})
    # test_gap_open_limit :1642
    p = {**test_model(-10.0, -10.0, 0.0), **test_bound(-6.0, -10.0), **gaps(-2.0, -1.0, 5, 1)}
    r = "CTAGCCAGCGATTTACATGCTCTCGGAATATCGACATGTA"
    case("gap_open_limit_one_gap", 1673, r, p, "CTAGCCAGCGAACATGCTCTCGGAATATCGACATGTA", 0, {"contains_position": 0})
    case("gap_open_limit_two_gaps", 1698, r, p, "CTAGCCAGCGATTACATGCTCTCGGAATTCGACATGTA", 0, {"positions_sorted": []})
    # test_vindija_pwm_alignment :1724
    p = {"model": "vindija_pwm", "bound": "test", "threshold": -30.0, "repr_mm_bound": {"repr_mm": True}, **gaps(-200.0, -100.0, 0, 2)}
    case("vindija_pwm_1", 1749, "CCCCCC", p, "TTCCCT", 40, {"score0": -4.641691, "positions_sorted": [0]})
    case("vindija_pwm_2", 1776, "CCCCCC", p, "CCCCCC", 0, {"score0": 0.0, "positions_sorted": [0]})
    case("vindija_pwm_3", 1813, "AAAAAA", p, "AAGAAA", 0, {"score0_approx": -10.965062})
    # test_corner_cases :1874
    p = {"model": "vindija_pwm", "bound": "discrete", "poisson_threshold": 0.01, "base_error_rate": 0.02,
         "penalty_gap_open": {"repr_mm_times": 3.0}, "penalty_gap_extend": {"repr_mm_times": 0.6}, "gap_dist_ends": 0, "max_num_gaps_open": 2}
    case("corner_cases", 1874,
         "GTTGTATTTTTAGTAGAGACAGGGTTTCATCATGTTGGCCAGAAAAAAAAAAAAAAAAAAAATTTGTATTTTTAGTAGAGACAGGCTTTCATCATGTTGGCCAG", p,
         "GTTGTATTTTTAGTAGAGACAGGCTTTCATCATGTTGGCCAG", 40,
         {"heap_scores": [-10.936638, -39.474224, -10.965062], "positions_sorted": [0, 62, 63], "best_positions": [0]})
    # test_cigar_indels :1937
    p = {**test_model(-10.0, -10.0, 0.0), **test_bound(-4.0, -10.0), **gaps(-2.0, -1.0, 0, 2)}
    case("cigar_deletion", 1962, "GATTAGCA", p, "ATTACA", 0, {"best_cigar": "4M1D2M"})
    case("cigar_deletion2", 2012, "GATTACAG", p, "GATCAG", 0, {"best_score": -4.0, "best_cigar": "3M2D3M"})
    case("cigar_insertion", 2064, "GATTACA", p, "GATTAGCA", 0, {"best_score": -3.0, "best_cigar": "5M1I2M"})
    case("cigar_insertion2", 2115, "GATTACA", p, "GATTAGGCA", 0, {"best_score": -4.0, "best_cigar": "5M2I2M"})
    p5 = {**test_model(-10.0, -10.0, 0.0), "bound": "test", "threshold": -5.0, "repr_mm_bound": {"repr_mm": True}, **gaps(-2.0, -1.0, 0, 2)}
    case("cigar_insertion3", 2166, "GATTACA", p5, "GATTAGTGCA", 0, {"best_score": -5.0, "best_cigar": "5M3I2M"})
    # test_md_tag :2232
    p = {**test_model(-1.0, -2.0, 0.0), **test_bound(-1.0, -2.0), **gaps(-2.0, -1.0, 0, 2)}
    case("md_mutation", 2257, "GATTACA", p, "GATTATA", 40, {"best_md": "5C1"})
    p4 = {**test_model(-1.0, -2.0, 0.0), "bound": "test", "threshold": -4.0, "repr_mm_bound": {"repr_mm": True}, **gaps(-2.0, -1.0, 0, 2)}
    case("md_deletion", 2291, "GATTAGCA", p4, "ATTACA", 0, {"best_md": "4^G2"})
    case("md_deletion2", 2341, "GATTACAG", p4, "GATCAG", 0, {"best_md": "3^TA3"})
    case("md_insertion", 2376, "GATTACA", p4, "GATTAGCA", 0, {"best_md": "7"})
    case("md_insertion2", 2410, "GATTACA", p4, "GATTAGGCA", 0, {"best_md": "7"})
    # test_reverse_strand_search_2 :2443
    case("reverse_strand_search_2", 2443, "AAAGCGTTTGCG", {**test_model(-1.0, -1.0, 0.0), **test_bound(0.0, -1.0), **gaps(-3.0, -1.0, 0, 2)},
         "TTT", 0, {"best_positions_fwd_rev": [[6, "F"], [0, "B"]]})
    # test_edit_operations_reverse_strand :2516
    case("edit_operations_reverse_strand", 2516, "GATTACA", {**test_model(-1.0, -1.0, 0.0), **test_bound(-1.0, -1.0), **gaps(-3.0, -1.0, 0, 2)},
         "TAGT", 0, {"best_positions_fwd_rev": [[1, "B"]], "best_md_backward": "1T2", "best_nm_backward": 1})
    # test_n :2593
    sadna = {"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.475, "three_prime_overhang": 0.475,
             "ds_deamination_rate": 0.001, "ss_deamination_rate": 0.9, "divergence": {"div3": 0.02}, "ignore_base_quality": 0}
    p = {**sadna, "bound": "test", "threshold": -14.0, "repr_mm_bound": {"repr_mm": True},
         "penalty_gap_open": {"log2": 0.001}, "penalty_gap_extend": {"repr_mm": True}, "gap_dist_ends": 0, "max_num_gaps_open": 2}
    case("n_all", 2630, "GATTACAGATTACAGATTACA", p, "NNNNNNNNNN", 40, {"n_hits": 0})
    case("n_one", 2649, "GATTACAGATTACAGATTACA", p, "AGATNACAG", 40, {"n_hits": 1})
    # test_bench :2669 — 10 kb reference + 100 bp reads (also benches/benchmark.rs)
    m = re.search(r'fn test_bench\(\) \{\s*let ref_seq = "(.*?)"\.as_bytes', mapping, re.S)
    ref10k = re.sub(r"[\\\s]", "", m.group(1))
    assert len(ref10k) == 10000 and set(ref10k) <= set("ACGT"), len(ref10k)
    p = {**sadna, "bound": "discrete", "poisson_threshold": 0.04, "base_error_rate": 0.02,
         "penalty_gap_open": {"log2": 0.00001}, "penalty_gap_extend": {"repr_mm": True}, "gap_dist_ends": 5, "max_num_gaps_open": 2}
    body = mapping[m.end():]
    reads = re.findall(r'// (bench_\w+)\s*\{.*?let pattern = "([ACGT]+)".*?assert_eq!\(intervals\.len\(\), (\d+)\);', body, re.S)
    assert len(reads) == 7, len(reads)
    for nm, pat, cnt in reads:
        case(nm, 2802, "@ref10k", p, pat, 40, {"n_hits": int(cnt)})
    return {"ref10k": ref10k, "cases": cases}


def sdm_kats():
    text = src("src/map/sequence_difference_models.rs")
    tests = text[text.index("mod tests"):]
    out = []
    blocks = {
        "test_vindija_pwm": ({"model": "vindija_pwm"}, "vindija_pwm"),
        "test_simple_adna_model": ({"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.6, "three_prime_overhang": 0.55,
                                    "ds_deamination_rate": 0.01, "ss_deamination_rate": 1.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0}, "adna_model"),
        "test_simple_adna_model_ds": ({"model": "simple_adna", "library": "double_stranded", "five_prime_overhang": 0.475, "three_prime_overhang": 0.475,
                                       "ds_deamination_rate": 0.01, "ss_deamination_rate": 0.9, "divergence": {"div3": 0.02}, "ignore_base_quality": 0}, "adna_model"),
    }
    for fn, (params, var) in blocks.items():
        m = re.search(r"fn %s\(\) \{(.*?)\n    \}\n" % fn, tests, re.S)
        body = m.group(1).replace("read_length", "35") if fn == "test_vindija_pwm" else m.group(1)
        asserts = re.findall(r"assert_approx_eq!\(\s*(-?[\d._]+),\s*%s\.get\((\d+), (\d+), b'(\w)', b'(\w)', (\d+)\)\s*\);" % var, body)
        line = text[:text.index("fn %s()" % fn)].count("\n") + 1
        out.append({"name": fn, "ref": f"src/map/sequence_difference_models.rs:{line}", "params": params, "tolerance": 1e-6,
                    "asserts": [[float(v.replace("_", "")), int(i), int(l), f, t, int(q)] for v, i, l, f, t, q in asserts]})
    assert [len(b["asserts"]) for b in out] == [5, 400, 400], [len(b["asserts"]) for b in out]
    display = {
        "ref": "src/map/sequence_difference_models.rs:1305",
        "single_stranded": {"params": {"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.4, "three_prime_overhang": 0.3,
                                       "ds_deamination_rate": 0.02, "ss_deamination_rate": 1.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0},
                            "ordinary_mm": "-7.20", "central": "-5.25",
                            "five_prime_c_to_t": "-1.29 -2.48 -3.52 -4.30 -4.80 -5.05 -5.17 -5.22 -5.24 -5.25".split(),
                            "three_prime_c_to_t": "-1.68 -3.16 -4.27 -4.88 -5.13 -5.22 -5.24 -5.25 -5.25 -5.25".split()},
        "double_stranded": {"params": {"model": "simple_adna", "library": "double_stranded", "five_prime_overhang": 0.4, "three_prime_overhang": 0.4,
                                       "ds_deamination_rate": 0.02, "ss_deamination_rate": 1.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0},
                            "ordinary_mm": "-7.20", "central": "-5.25",
                            "five_prime_c_to_t": "-1.29 -2.48 -3.52 -4.30 -4.80 -5.05 -5.17 -5.22 -5.24 -5.25".split(),
                            "three_prime_g_to_a": "-1.29 -2.48 -3.52 -4.30 -4.80 -5.05 -5.17 -5.22 -5.24 -5.25".split()},
    }
    wo_deam = {"ref": "src/map/sequence_difference_models.rs:1279",
               "params": {"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.0, "three_prime_overhang": 0.0,
                          "ds_deamination_rate": 0.0, "ss_deamination_rate": 0.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0},
               "equal_pairs": [[[0, 25, "C", "T", 40], [13, 25, "T", "A", 40]], [[24, 25, "C", "T", 40], [13, 25, "T", "A", 40]],
                               [[13, 25, "C", "C", 40], [0, 25, "C", "C", 40]]]}
    return {"get": out, "display": display, "wo_deam": wo_deam}


def bounds_kats():
    return {
        "ref": "src/map/mismatch_bounds.rs:288-377",
        "discrete_get": [
            {"poisson": 0.04, "err": 0.02, "values": [[156, 6], [124, 6], [123, 5], [93, 5], [92, 4], [64, 4], [63, 3], [38, 3], [37, 2], [17, 2], [16, 0], [15, 0], [3, 0], [2, 0], [0, 0]]},
            {"poisson": 0.01, "err": 0.02, "values": [[207, 10], [176, 9], [146, 8], [117, 7], [90, 6], [64, 5], [42, 4], [22, 3], [17, 2], [8, 0], [1, 0]]},
        ],
        # Display tables: first read length at which the allowance changes (17..=256)
        "display": [
            {"poisson": 0.06, "err": 0.02, "steps": [[17, 1], [20, 2], [45, 3], [73, 4], [104, 5], [137, 6], [172, 7], [208, 8], [244, 9]]},
            {"poisson": 0.03, "err": 0.02, "steps": [[17, 2], [34, 3], [58, 4], [86, 5], [116, 6], [147, 7], [180, 8], [213, 9], [248, 10]]},
        ],
    }


def d_array_kat():
    return {"ref": "src/map/bi_d_array.rs:243-309", "reference": "GATTACA",
            "params": {**test_model(-1.0, -1.0, 0.0), "bound": "test", "threshold": 0.0, "repr_mm_bound": {"repr_mm": True},
                       "penalty_gap_open": {"log2": 0.00001}, "penalty_gap_extend": {"repr_mm": True}, "gap_dist_ends": 0, "max_num_gaps_open": 2},
            "pattern": "CCCCCCC", "qual": [10, 40, 40, 40, 40, 10, 40], "split": 3,
            "d_composite": [0.0, 0.0, -1.0, 0.0, 0.0, -1.0, -1.0], "get": [[2, 3, -2.0], [0, 6, 0.0]]}


def integration():
    text = src("tests/integration_tests.rs")
    fasta = re.search(r'let fasta_content = "(.*?)";', text, re.S).group(1)
    contigs = []
    for block in fasta.split(">")[1:]:
        lines = block.strip().split("\n")
        contigs.append({"name": lines[0].strip(), "seq": "".join(l.strip() for l in lines[1:])})
    sam = re.search(r'let sam_content = b"\\\n(.*?)";', text, re.S).group(1)
    reads = []
    for line in sam.split("\\n\\\n"):
        line = line.strip()
        if not line or line.startswith("@"):
            continue
        f = line.replace("\\\\", "\\").split("\\t")
        reads.append({"name": f[0], "flags": int(f[1]), "seq": f[9], "qual": f[10].rstrip("\\n")})
    assert len(reads) == 17, len(reads)
    exp_src = text[text.index("fn shared_expectation()"):]
    recs = []
    for blk in exp_src.split("BamFieldSubset {")[1:]:
        g = lambda pat, d=None: (re.search(pat, blk, re.S).group(1) if re.search(pat, blk, re.S) else d)
        cigar = "".join(f"{n}{ {'Match': 'M', 'Insertion': 'I', 'Deletion': 'D'}[k]}" for k, n in re.findall(r"Kind::(\w+), (\d+)\)", blk))
        tid = g(r"tid: Some\((\d+)")
        recs.append({
            "name": g(r'name: Some\(b"(.*?)"'), "flags": int(g(r"flags: (\d+)\.into")),
            "tid": int(tid) if tid is not None else None,
            "pos": int(g(r"pos: Some\((\d+)")) if g(r"pos: Some\((\d+)") else None,
            "mapq": int(g(r"mq: Some\((\d+)_u8")), "cigar": cigar, "seq": g(r'seq: b"(\w+)"'),
            "qual": g(r'qual: b"(.*?)"\s*\.iter').replace("\\\\", "\\"),
            "md": g(r'md: Some\("(.*?)"'), "x0": int(g(r"x0: Some\((\d+)")) if g(r"x0: Some\((\d+)") else None,
            "x1": int(g(r"x1: Some\((\d+)")) if g(r"x1: Some\((\d+)") else None,
            "xa": g(r'xa: Some\(\s*"(.*?)"'), "xs": float(g(r"xs: Some\((-?[\d.]+)")) if g(r"xs: Some\((-?[\d.]+)") else None,
            "xt": g(r"xt: Some\('(\w)'"),
        })
    assert len(recs) == 17, len(recs)
    params = {"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.6, "three_prime_overhang": 0.55,
              "ds_deamination_rate": 0.01, "ss_deamination_rate": 1.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0,
              "bound": "discrete", "poisson_threshold": 0.03, "base_error_rate": 0.02,
              "penalty_gap_open": {"repr_mm_times": 1.5}, "penalty_gap_extend": {"repr_mm_times": 0.5}, "gap_dist_ends": 5, "max_num_gaps_open": 2}
    return {"ref": "tests/integration_tests.rs:59-868", "contigs": contigs, "reads": reads, "params": params, "expected_sorted_by_name": recs,
            "note": "The reference replaces the single N of Chromosome_02 with StdRng(seed 1234).choose(ACGT); the expected MAPQ 37 of read "
                    "A00795_0135 (one real mismatch + the N column, MD 4C5N11) is only reachable if that draw was 'A' (any other base costs a "
                    "second mismatch -> MAPQ 20), so the fixture pins the replacement to 'A'.",
            "n_replacement": "A",
            "header_prefix": ["@HD\tVN:1.6\tSO:unsorted", "@SQ\tSN:chr1\tLN:600", "@SQ\tSN:Chromosome_02\tLN:600", "@SQ\tSN:Chromosome_03\tLN:84",
                              "@SQ\tSN:Chromosome_04\tLN:46"]}


def misc_kats():
    return {
        "prrange": {"ref": "src/map/prrange.rs:190-260",
                    "permutations": [[6100000000, 6100000005, 1234], [13, 23, 1234], [1, 2, 1234]],
                    "counts": [[5233065207, 5233065216, 400636091, 9]], "invalid": [[1, 0, 1234], [1, 1, 1234]], "seed_sweep_to": 100},
        "tree": {"ref": "src/map/backtrack_tree.rs:131-196"},
        "run_apply": {"ref": "src/index/indexing.rs:263-450", "input": "NNGATNTACANGATTNNACANNN",
                      "min_run_1": "XXGATXTACAXGATTXXACAXXX", "min_run_2": "XXGATATACAAGATTXXACAXXX",
                      "input2": "CYNTYYNNT", "min_run_2_input2": "CAATXXXXT", "revcomp": ["GATTXACA", "TGTXAATC"]},
        "frame_size": {"ref": "src/map/mod.rs:172-175", "bytes": 40},
    }


def main():
    for name, fn in [("search_kats", search_kats), ("sdm_kats", sdm_kats), ("bounds_kats", bounds_kats), ("d_array_kat", d_array_kat),
                     ("integration", integration), ("misc_kats", misc_kats)]:
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
