"""Parameter presets of the benchmark configurations in fixture style, plus the resolver of symbolic values.

Symbolic values mirror how the reference derives them at run time (src/main.rs:418-499):
    {"div3": x}          -> x_f32 / 3.0_f32              (divergence / 3)
    {"log2": x}          -> x_f32.log2()                 (penalty_gap_open = log2(indel_rate))
    {"repr_mm_times": k} -> k * representative mismatch penalty
    {"repr_mm": true}    -> representative mismatch penalty
"""
import ctypes as C

import numpy as np

# C2: `-l single_stranded -f 0 -t 0 -d 0 -s 0 -D 0.02 -p 0.03 -i 0.001 -x 1.0`
NO_DAMAGE = {"model": "simple_adna", "library": "single_stranded", "five_prime_overhang": 0.0, "three_prime_overhang": 0.0,
             "ds_deamination_rate": 0.0, "ss_deamination_rate": 0.0, "divergence": {"div3": 0.02}, "ignore_base_quality": 0,
             "bound": "discrete", "poisson_threshold": 0.03, "base_error_rate": 0.02,
             "penalty_gap_open": {"log2": 0.001}, "penalty_gap_extend": {"repr_mm_times": 1.0}, "gap_dist_ends": 5, "max_num_gaps_open": 2}
# C3: README example, single-stranded library with 50 % overhang parameters (Readme.md:147-150)
DAMAGE = dict(NO_DAMAGE, five_prime_overhang=0.5, three_prime_overhang=0.5, ds_deamination_rate=0.02, ss_deamination_rate=1.0)
# `-c 0.2 -e 1.2` instead of `-p`: the Continuous bound (mismatch_bounds.rs:76-120; cutoff = -c, main.rs:463-475); 50^1.2 * -0.2 = -21.9
CONTINUOUS = {k: v for k, v in dict(DAMAGE, bound="continuous", cutoff=-0.2, exponent=1.2).items() if k not in ("poisson_threshold", "base_error_rate")}
# `-l double_stranded -f 0.5`: the 3' overhang parameter equals the 5' one (main.rs:427-437), G->A at the 3' end (sequence_difference_models.rs:125-144)
DOUBLE_STRANDED = dict(DAMAGE, library="double_stranded")
# `--ignore_base_quality`: one quality level (sequence_difference_models.rs:286-287)
IGNORE_BQ = dict(DAMAGE, ignore_base_quality=1)
# the VindijaPwm difference model (sequence_difference_models.rs:336-396; "only for testing purposes" in the reference, reachable through the worker's task sheet):
# position-dependent C->T probabilities symmetric about the read's middle, alignment starting there (bidirectional search)
VINDIJA = {"model": "vindija_pwm", "bound": "discrete", "poisson_threshold": 0.03, "base_error_rate": 0.02,
           "penalty_gap_open": {"log2": 0.001}, "penalty_gap_extend": {"repr_mm_times": 1.0}, "gap_dist_ends": 5, "max_num_gaps_open": 2}


def _log2f(x):
    libm = C.CDLL("libm.so.6")
    libm.log2f.restype = C.c_float
    libm.log2f.argtypes = [C.c_float]
    return libm.log2f(C.c_float(x))


def resolve(d):
    """fixture-style dict -> plain dict of numbers, using the library's own SequenceDifferenceModel for repr_mm"""
    from . import binding as mb
    d = dict(d)
    for k, v in list(d.items()):
        if isinstance(v, dict) and "div3" in v:
            d[k] = float(np.float32(v["div3"]) / np.float32(3.0))
        elif isinstance(v, dict) and "log2" in v:
            d[k] = _log2f(v["log2"])
    symbolic = {k: v for k, v in d.items() if isinstance(v, dict)}
    if symbolic:
        base = mb.make_params({k: (0.0 if isinstance(v, dict) else v) for k, v in d.items()})
        repr_mm = mb.lib().mapad_sdm_representative_mismatch_penalty(C.byref(base))
        for k, v in symbolic.items():
            d[k] = float(np.float32(repr_mm) * np.float32(v["repr_mm_times"])) if "repr_mm_times" in v else float(repr_mm)
    return d
