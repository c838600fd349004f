// host_models.hpp — host side of the two scoring plugins: the values the GPU tables are filled from.
//
// Mirrors trait SequenceDifferenceModel and its three impls (src/map/sequence_difference_models.rs:14-424) and trait
// MismatchBound with Discrete / Continuous / TestBound (src/map/mismatch_bounds.rs:10-281), evaluated with the same f32
// operations as the Rust code lowers to: fused fmaf for mul_add, compiler-rt style square-and-multiply for powi, glibc
// log2f / powf / expf.  Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mapad_amd.h"
#include "common.hpp"

namespace mapad {
namespace host {

// f32::powi -> __powisf2
inline float powi(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}
inline float qual2prob(uint8_t q) { return std::pow(10.0f, -(float)q / 10.0f) / 3.0f; }  // :275-277

inline float simple_adna_get(const mapad_params_t& p, uint64_t i, uint64_t len, uint8_t from, uint8_t to, uint8_t q) {  // :117-207
    const int fp = (int)i + 1, tp = (int)(len - 1 - i) + 1;
    const float seq_err = qual2prob(p.ignore_base_quality ? 255 : q);
    const float e = std::fmaf(seq_err, -p.divergence, seq_err + p.divergence);
    const float no_err = std::fmaf(3.0f, -e, 1.0f);
    float v = e;
    const bool ct = from == 'C' && (to == 'C' || to == 'T');
    const bool ga = from == 'G' && (to == 'G' || to == 'A');
    if ((from == 'A' && to == 'A') || (from == 'T' && to == 'T')) v = no_err;
    else if (ct || ga) {
        float p_fwd, p_rev;
        if (p.library_prep == MAPAD_LIBRARY_SINGLE_STRANDED) {
            const float f = powi(p.five_prime_overhang, fp), t = powi(p.three_prime_overhang, tp);
            p_fwd = std::fmaf(f, -t, f + t);
            p_rev = 0.0f;
        } else {
            p_fwd = powi(p.five_prime_overhang, fp);
            p_rev = powi(p.five_prime_overhang, tp);
        }
        const float pr = ct ? p_fwd : p_rev;
        const float deam = std::fmaf(p.ss_deamination_rate, pr, p.ds_deamination_rate * (1.0f - pr));
        if (to == from) v = std::fmaf(4.0f * e, deam, no_err - deam);
        else v = std::fmaf(4.0f * e, -deam, e + deam);
    }
    const float eps = 1.1920929e-07f;  // f32::EPSILON
    return std::log2(v > eps ? v : eps);
}

inline float vindija_get(uint64_t i, uint64_t len, uint8_t from, uint8_t to) {  // :357-385
    static const float ppm[7] = {0.4f, 0.25f, 0.1f, 0.06f, 0.05f, 0.04f, 0.03f};
    const float sub = 0.0005f;
    float pr;
    if (from == 'C') {
        const uint64_t k = i < len - (i + 1) ? i : len - (i + 1);
        const float ct = k < 7 ? ppm[k] : 0.02f;
        pr = to == 'T' ? ct : to == 'C' ? 1.0f - ct : sub;
    } else pr = from == to ? 1.0f - sub : sub;
    return std::log2(pr);
}

inline float sdm_get(const mapad_params_t& p, uint64_t i, uint64_t len, uint8_t from, uint8_t to, uint8_t q) {
    switch (p.model_kind) {
        case MAPAD_MODEL_SIMPLE_ADNA: return simple_adna_get(p, i, len, from, to, q);
        case MAPAD_MODEL_VINDIJA_PWM: return vindija_get(i, len, from, to);
        default: return (from == 'C' && to == 'T') ? p.deam_score : (from == to) ? p.match_score : p.mm_score;  // :415-424
    }
}
inline float sdm_repr_mm(const mapad_params_t& p) {  // :16-31
    return sdm_get(p, 40, 80, 'T', 'A', 255) - sdm_get(p, 40, 80, 'T', 'T', 255);
}
inline float sdm_min_penalty(const mapad_params_t& p, uint64_t i, uint64_t len, uint8_t to, uint8_t q, bool only_mm) {  // :34-57
    static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};
    if (!only_mm && base_index(to) > 3) return 0.0f;
    float m = kF32Min;
    for (uint8_t b : ACGT) {
        if (only_mm && b == to) continue;
        const float v = sdm_get(p, i, len, b, to, q);
        m = v > m ? v : m;
    }
    return m;
}
inline int sdm_alignment_start(const mapad_params_t& p, uint64_t len) {  // :59-61, :209-211
    return p.model_kind == MAPAD_MODEL_SIMPLE_ADNA ? (int)(int16_t)len : (int)((int16_t)len / 2);
}

// Discrete::calculate_max_num_mismatches (mismatch_bounds.rs:217-241)
inline float discrete_allowed(uint64_t len, float thr, float err) {
    if (len < 17) return 0.0f;  // :245-247
    const float lambda = (float)len * err;
    const float eml = std::exp(-lambda);
    if (!(1.0f - eml > thr)) return 0.0f;
    uint64_t last = 1, fact = 1;
    float pw = 1.0f, sum = eml;
    for (uint64_t k = 1; k <= len; ++k) {
        pw *= lambda;
        fact *= k;
        sum += pw * eml / (float)fact;
        if (1.0f - sum > thr) last = k + 1; else break;
    }
    return (float)last;
}
inline float bound_repr_mm(const mapad_params_t& p) { return p.bound_kind == MAPAD_BOUND_TEST ? p.repr_mm_bound : sdm_repr_mm(p); }
// the per-length quantity the kernels compare against (common.hpp: DevParams::reject_thr)
inline float bound_threshold(const mapad_params_t& p, uint64_t len, float repr_mm) {
    switch (p.bound_kind) {
        case MAPAD_BOUND_DISCRETE: return discrete_allowed(len, p.poisson_threshold, p.base_error_rate) * repr_mm;  // :131-134
        case MAPAD_BOUND_CONTINUOUS: return std::pow((float)len, p.exponent);                                       // :107-119
        default: return p.threshold;
    }
}
inline bool mb_reject(const mapad_params_t& p, float v, uint64_t len) {
    const float t = bound_threshold(p, len, bound_repr_mm(p));
    return p.bound_kind == MAPAD_BOUND_CONTINUOUS ? (v / t) < p.cutoff : v < t;
}
inline bool mb_reject_iterative(const mapad_params_t& p, float v, float ref) {
    return p.bound_kind == MAPAD_BOUND_TEST ? false : v < ref + bound_repr_mm(p);
}
inline float mb_remaining_frac(const mapad_params_t& p, float v, uint64_t len) {  // :93-98, :140-144, :277-279
    const float repr = bound_repr_mm(p);
    switch (p.bound_kind) {
        case MAPAD_BOUND_DISCRETE: return std::fmaf(discrete_allowed(len, p.poisson_threshold, p.base_error_rate), repr, -v) / repr;
        case MAPAD_BOUND_CONTINUOUS: { const float s = std::pow((float)len, p.exponent); return (p.cutoff - v / s) / (repr / s); }
        default: return (p.threshold - v) / p.repr_mm_bound;
    }
}

// ---- tables the kernels read (common.hpp: DevParams) --------------------------------------------------------------------
struct HostTables {
    int nq = 1;                       // quality levels per position
    std::vector<int32_t> table_base;  // [kMaxReadLen + 1], -1 = absent
    std::vector<float> sdm;           // float4 entries
    std::vector<float> reject_thr;    // [kMaxReadLen + 1]
    float repr_mm = 0;
};
inline int quality_levels(const mapad_params_t& p) {
    return (p.model_kind == MAPAD_MODEL_SIMPLE_ADNA && !p.ignore_base_quality) ? 256 : 1;
}
// appends the table of one read length (all positions x quality levels x 5 read-base classes x 4 reference bases)
inline void add_length(const mapad_params_t& p, HostTables& t, int len) {
    if (t.table_base[len] >= 0) return;
    static const uint8_t TO[5] = {'A', 'C', 'G', 'T', 'N'}, FROM[4] = {'A', 'C', 'G', 'T'};
    t.table_base[len] = (int32_t)(t.sdm.size() / 4);
    const size_t base = t.sdm.size();
    t.sdm.resize(base + (size_t)len * t.nq * 5 * 4);
    // SimpleAncientDnaModel: the value depends on q only through qual2prob(q); positions only matter for C/G rows
    for (int i = 0; i < len; ++i)
        for (int q = 0; q < t.nq; ++q)
            for (int c = 0; c < 5; ++c)
                for (int f = 0; f < 4; ++f)
                    t.sdm[base + ((((size_t)i * t.nq + q) * 5 + c) * 4) + f] = sdm_get(p, (uint64_t)i, (uint64_t)len, FROM[f], TO[c], (uint8_t)q);
}
inline HostTables make_tables(const mapad_params_t& p) {
    HostTables t;
    t.nq = quality_levels(p);
    t.table_base.assign(kMaxReadLen + 1, -1);
    t.repr_mm = bound_repr_mm(p);
    t.reject_thr.resize(kMaxReadLen + 1);
    for (int l = 0; l <= kMaxReadLen; ++l) t.reject_thr[l] = bound_threshold(p, (uint64_t)l, t.repr_mm);
    return t;
}

}  // namespace host
}  // namespace mapad
