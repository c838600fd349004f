// mapad_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A plain C++17 restatement of the read-mapping hot path of mpieva/mapAD v0.45.0
// (reference tree /root/reference, Rust).  Each section cites the reference
// file:line it follows.  It keeps the reference's data-structure choices (byte BWT,
// k-sampled Occ, 40-byte frames in a min-max heap, slab edit tree, std-BinaryHeap of
// hits) so that it can double as the "CPU restatement" baseline in bench.py.
//
// ONLY tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
// code.  The product path (mapad_amd/) never includes, links or calls it.
//
// Pinning status: the reference cannot be built here (no Rust toolchain; three
// un-vendored crates: bio 1.5.0-mapAD @ jch-13/rust-bio#807e5a25, min-max-heap
// 1.3.1-alpha.0 @ tov/min-max-heap-rs#76a2141a, slab 0.4.12).  The oracle is pinned
// against every known-answer test the reference holds for this path (tests/golden/,
// transcribed from src/map/*.rs tests and tests/integration_tests.rs).  The min-max
// heap's tie behaviour is restated from the published algorithm (Atkinson et al. 1986)
// and the crate's documented structure; beyond those KATs tie order is "parity unpinned".
//
// Compile with: -O2 -std=c++17 -ffp-contract=off -fno-fast-math (f32 bit-exactness).
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mo {

// ---------------------------------------------------------------------------------
// f32 helpers
// ---------------------------------------------------------------------------------
// Rust f32::powi lowers to compiler-rt __powisf2 (square-and-multiply), not powf.
inline float powi_f32(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}
inline float f32_max(float a, float b) { return std::fmax(a, b); }  // IEEE maxNum
inline float f32_min(float a, float b) { return std::fmin(a, b); }  // IEEE minNum
constexpr float F32_MIN = -std::numeric_limits<float>::max();      // Rust f32::MIN
constexpr float F32_EPSILON = std::numeric_limits<float>::epsilon();

// bio::alphabets::dna::complement (bio 1.5): identity except the IUPAC pairs.
inline uint8_t dna_complement(uint8_t a) {
    switch (a) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k';
        case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
        default: return a;  // S, W, N, X, $ ... map to themselves
    }
}
inline std::vector<uint8_t> dna_revcomp(const std::vector<uint8_t>& s) {
    std::vector<uint8_t> r(s.size());
    for (size_t i = 0; i < s.size(); ++i) r[i] = dna_complement(s[s.size() - 1 - i]);
    return r;
}

enum class Direction : uint8_t { Forward, Backward };

// ---------------------------------------------------------------------------------
// SequenceDifferenceModel  (src/map/sequence_difference_models.rs:14-62)
// ---------------------------------------------------------------------------------
static const uint8_t DNA_UPPERCASE_ALPHABET[4] = {'A', 'C', 'G', 'T'};  // src/index/mod.rs:16

struct SequenceDifferenceModel {
    virtual ~SequenceDifferenceModel() = default;
    virtual float get(size_t i, size_t read_length, uint8_t from, uint8_t to, uint8_t base_quality) const = 0;
    // :16-31
    float get_representative_mismatch_penalty() const {
        const size_t read_length = 80;
        return get(read_length / 2, read_length, 'T', 'A', 255) - get(read_length / 2, read_length, 'T', 'T', 255);
    }
    // :34-57
    float get_min_penalty(size_t i, size_t read_length, uint8_t to, uint8_t base_quality, bool only_mismatches) const {
        if (!only_mismatches) {
            bool in_alphabet = false;
            for (uint8_t b : DNA_UPPERCASE_ALPHABET) in_alphabet |= (b == to);
            if (!in_alphabet) return 0.0f;
        }
        float acc = F32_MIN;
        for (uint8_t base : DNA_UPPERCASE_ALPHABET) {
            if (only_mismatches && base == to) continue;
            acc = f32_max(acc, get(i, read_length, base, to, base_quality));
        }
        return acc;
    }
    // :59-61 (default) — overridden by SimpleAncientDnaModel :209-211
    virtual int16_t find_alignment_start(size_t pattern_length) const { return (int16_t)pattern_length / 2; }
};

// :93-100, :103-334
struct SimpleAncientDnaModel : SequenceDifferenceModel {
    bool single_stranded = true;
    float five_prime_overhang = 0, three_prime_overhang = 0;  // DoubleStranded(x): five == three == x
    float ds_deamination_rate = 0, ss_deamination_rate = 0, divergence = 0;
    bool use_default_base_quality = false;
    float default_base_quality_prob = 0;
    std::vector<float> cache;

    static float qual2prob(uint8_t q) { return std::pow(10.0f, -float(q) / 10.0f) / 3.0f; }  // :275-277

    SimpleAncientDnaModel(bool ss, float five, float three, float ds_rate, float ss_rate, float div, bool ignore_bq)
        : single_stranded(ss), five_prime_overhang(five), three_prime_overhang(three), ds_deamination_rate(ds_rate),
          ss_deamination_rate(ss_rate), divergence(div) {
        use_default_base_quality = ignore_bq;  // :286-287
        if (ignore_bq) default_base_quality_prob = qual2prob(255);
        else { cache.resize(256); for (int q = 0; q < 256; ++q) cache[q] = qual2prob((uint8_t)q); }
    }

    float get(size_t i, size_t read_length, uint8_t from, uint8_t to, uint8_t base_quality) const override {
        const size_t fp_dist = i, tp_dist = read_length - 1 - i;  // :118-119
        auto deam = [&](float& c_to_t, float& g_to_a) {            // :124-155
            float p_fwd, p_rev;
            if (single_stranded) {
                const float f = powi_f32(five_prime_overhang, (int)fp_dist + 1);
                const float t = powi_f32(three_prime_overhang, (int)tp_dist + 1);
                p_fwd = std::fmaf(f, -t, f + t);
                p_rev = 0.0f;
            } else {
                p_fwd = powi_f32(five_prime_overhang, (int)fp_dist + 1);
                p_rev = powi_f32(five_prime_overhang, (int)tp_dist + 1);
            }
            c_to_t = std::fmaf(ss_deamination_rate, p_fwd, ds_deamination_rate * (1.0f - p_fwd));
            g_to_a = std::fmaf(ss_deamination_rate, p_rev, ds_deamination_rate * (1.0f - p_rev));
        };
        const float sequencing_error = use_default_base_quality ? default_base_quality_prob : cache[base_quality];  // :157-164
        const float e = std::fmaf(sequencing_error, -divergence, sequencing_error + divergence);                     // :167-168
        float v, c_to_t, g_to_a;
        switch (from) {  // :170-204
            case 'A': v = (to == 'A') ? std::fmaf(3.0f, -e, 1.0f) : e; break;
            case 'C':
                if (to == 'C') { deam(c_to_t, g_to_a); v = std::fmaf(4.0f * e, c_to_t, std::fmaf(3.0f, -e, 1.0f) - c_to_t); }
                else if (to == 'T') { deam(c_to_t, g_to_a); v = std::fmaf(4.0f * e, -c_to_t, e + c_to_t); }
                else v = e;
                break;
            case 'G':
                if (to == 'A') { deam(c_to_t, g_to_a); v = std::fmaf(4.0f * e, -g_to_a, e + g_to_a); }
                else if (to == 'G') { deam(c_to_t, g_to_a); v = std::fmaf(4.0f * e, g_to_a, std::fmaf(3.0f, -e, 1.0f) - g_to_a); }
                else v = e;
                break;
            case 'T': v = (to == 'T') ? std::fmaf(3.0f, -e, 1.0f) : e; break;
            default: v = e;
        }
        return std::log2(f32_max(v, F32_EPSILON));  // :205-206
    }
    int16_t find_alignment_start(size_t pattern_length) const override { return (int16_t)pattern_length; }  // :209-211
};

// :339-401
struct VindijaPwm : SequenceDifferenceModel {
    float ppm[7] = {0.4f, 0.25f, 0.1f, 0.06f, 0.05f, 0.04f, 0.03f};
    float ct_default = 0.02f, subst_default = 0.0005f;
    float get(size_t i, size_t read_length, uint8_t from, uint8_t to, uint8_t) const override {
        float p;
        if (from == 'C') {
            const size_t k = std::min(i, read_length - (i + 1));
            const float ct = k < 7 ? ppm[k] : ct_default;
            p = (to == 'T') ? ct : (to == 'C') ? 1.0f - ct : subst_default;
        } else {
            p = (from == to) ? 1.0f - subst_default : subst_default;
        }
        return std::log2(p);
    }
};

// :403-424
struct TestDifferenceModel : SequenceDifferenceModel {
    float deam_score, mm_score, match_score;
    TestDifferenceModel(float d, float m, float ma) : deam_score(d), mm_score(m), match_score(ma) {}
    float get(size_t, size_t, uint8_t from, uint8_t to, uint8_t) const override {
        if (from == 'C' && to == 'T') return deam_score;
        if (from == to) return match_score;
        return mm_score;
    }
};

// ---------------------------------------------------------------------------------
// MismatchBound (src/map/mismatch_bounds.rs:10-20)
// ---------------------------------------------------------------------------------
struct MismatchBound {
    virtual ~MismatchBound() = default;
    virtual bool reject(float value, size_t read_length) const = 0;
    virtual bool reject_iterative(float value, float reference) const = 0;
    virtual float remaining_frac_of_repr_mm(float value, size_t read_length) const = 0;
};

// :76-120
struct Continuous : MismatchBound {
    float cutoff, exponent, repr_mm;
    std::vector<float> cache;
    Continuous(float c, float e, float r) : cutoff(c), exponent(e), repr_mm(r) {
        cache.resize(256);
        for (int l = 0; l < 256; ++l) cache[l] = std::pow((float)l, exponent);
    }
    float scale(size_t l) const { return l < cache.size() ? cache[l] : std::pow((float)l, exponent); }
    bool reject(float v, size_t l) const override { return (v / scale(l)) < cutoff; }
    bool reject_iterative(float v, float ref) const override { return v < ref + repr_mm; }
    float remaining_frac_of_repr_mm(float v, size_t l) const override {
        const float s = scale(l);
        return (cutoff - v / s) / (repr_mm / s);
    }
};

// :122-261
struct Discrete : MismatchBound {
    static constexpr size_t MIN_READ_LENGTH = 17;
    float poisson_threshold, base_error_rate, repr_mm;
    std::vector<float> cache;
    Discrete(float p, float e, float r) : poisson_threshold(p), base_error_rate(e), repr_mm(r) {
        cache.resize(256);
        for (size_t i = 0; i < 256; ++i) cache[i] = calc(i + MIN_READ_LENGTH, p, e);
    }
    // :217-241
    static float calc(size_t read_length, float thr, float err) {
        const float lambda = (float)read_length * err;
        const float exp_minus_lambda = std::exp(-lambda);
        uint64_t last_k = 0;
        bool any = false;
        // k = 0 entry: (1, exp_minus_lambda)
        if (1.0f - exp_minus_lambda > thr) { last_k = 1; any = true; } else return 0.0f;
        float lambda_to_the_k = 1.0f, sum = exp_minus_lambda;
        uint64_t k_factorial = 1;
        for (uint64_t k = 1; k <= (uint64_t)read_length; ++k) {
            lambda_to_the_k *= lambda;
            k_factorial *= k;  // wraps like release-mode Rust; never reached in practice
            sum += lambda_to_the_k * exp_minus_lambda / (float)k_factorial;
            if (1.0f - sum > thr) last_k = k + 1; else break;
        }
        return any ? (float)last_k : 0.0f;
    }
    float get(size_t l) const {  // :243-260
        if (l < MIN_READ_LENGTH) return 0.0f;
        const size_t idx = l - MIN_READ_LENGTH;
        return idx < cache.size() ? cache[idx] : calc(l, poisson_threshold, base_error_rate);
    }
    bool reject(float v, size_t l) const override { return v < get(l) * repr_mm; }
    bool reject_iterative(float v, float ref) const override { return v < ref + repr_mm; }
    float remaining_frac_of_repr_mm(float v, size_t l) const override { return std::fmaf(get(l), repr_mm, -v) / repr_mm; }
};

// :263-281
struct TestBound : MismatchBound {
    float threshold, repr_mm_bound;
    TestBound(float t, float r) : threshold(t), repr_mm_bound(r) {}
    bool reject(float v, size_t) const override { return v < threshold; }
    bool reject_iterative(float, float) const override { return false; }
    float remaining_frac_of_repr_mm(float v, size_t) const override { return (threshold - v) / repr_mm_bound; }
};

// src/map/mod.rs:21-31
struct AlignmentParameters {
    float penalty_gap_open = 0, penalty_gap_extend = 0;
    uint8_t gap_dist_ends = 0, max_num_gaps_open = 0;
    bool stack_limit_abort = false;
    // src/map/mapping.rs:52-54 (constants there; runtime here so tests can exercise recovery)
    uint32_t stack_limit = 2000000, edit_tree_limit = 10000000;
};

// ---------------------------------------------------------------------------------
// Index primitives: suffix array, BWT, Less, Occ  (bio fork)
// ---------------------------------------------------------------------------------
// Plain lexicographic suffix sort of a rank-transformed text that ends in the sentinel 0.
// (Naive comparison sort — the oracle is only used on small/medium texts.)
inline std::vector<uint64_t> suffix_array_naive(const std::vector<uint8_t>& t) {
    const size_t n = t.size();
    std::vector<uint64_t> sa(n);
    std::iota(sa.begin(), sa.end(), 0);
    std::sort(sa.begin(), sa.end(), [&](uint64_t a, uint64_t b) {
        if (a == b) return false;
        const size_t la = n - a, lb = n - b, m = std::min(la, lb);
        const int c = std::memcmp(t.data() + a, t.data() + b, m);
        if (c != 0) return c < 0;
        return la < lb;  // shorter suffix first
    });
    return sa;
}
inline std::vector<uint8_t> bwt_from_sa(const std::vector<uint8_t>& t, const std::vector<uint64_t>& sa) {
    std::vector<uint8_t> b(t.size());
    for (size_t i = 0; i < sa.size(); ++i) b[i] = sa[i] > 0 ? t[sa[i] - 1] : t[t.size() - 1];
    return b;
}
// less[c] = #symbols < c ; length max_symbol + 2
inline std::vector<uint64_t> less_from_bwt(const std::vector<uint8_t>& bwt, int n_symbols) {
    std::vector<uint64_t> less(n_symbols + 1, 0);
    for (uint8_t c : bwt) less[c] += 1;
    uint64_t acc = 0;
    for (auto& v : less) { const uint64_t t = v; v = acc; acc += t; }
    return less;
}
// Occ with checkpoints every k rows: checkpoint j holds the inclusive count at row j*k.
struct Occ {
    uint32_t k = 1;
    int n_symbols = 0;
    std::vector<std::vector<uint64_t>> occ;  // [symbol][row / k]
    Occ() = default;
    Occ(const std::vector<uint8_t>& bwt, uint32_t k_, int n_sym) : k(k_), n_symbols(n_sym), occ(n_sym) {
        std::vector<uint64_t> cur(n_sym, 0);
        for (size_t i = 0; i < bwt.size(); ++i) {
            cur[bwt[i]] += 1;
            if (i % k == 0) for (int s = 0; s < n_sym; ++s) occ[s].push_back(cur[s]);
        }
    }
    // occurrences of a in bwt[0..=r]
    uint64_t get(const std::vector<uint8_t>& bwt, uint64_t r, uint8_t a) const {
        const uint64_t i = r / k;
        uint64_t c = occ[a][i];
        for (uint64_t p = i * k + 1; p <= r; ++p) c += (bwt[p] == a);
        return c;
    }
};

// src/map/fmd_index.rs:184-219
struct RtBiInterval {
    uint64_t lower = 0, lower_rev = 0, size = 0;
    RtBiInterval swapped() const { return {lower_rev, lower, size}; }
};

struct Counters {
    uint64_t e_search = 0;   // frames popped and extended (mapping.rs:1245)
    uint64_t e_darray = 0;   // single-base extensions in compute_part (bi_d_array.rs:138-141)
    uint64_t n_push = 0, n_pop = 0, n_node = 0;
    uint64_t n_hits = 0;
};

// src/map/fmd_index.rs:13-182
struct RtFmdIndex {
    std::vector<uint8_t> bwt;
    std::vector<uint64_t> less;
    Occ occ_table;
    uint64_t sentinel_occ[2] = {0, 0};
    // rank transform: ASCII -> rank, or -1
    int rank_of[256];
    std::vector<uint8_t> back_transform;  // rank -> ASCII (sorted keys)

    RtFmdIndex() { std::fill(rank_of, rank_of + 256, -1); }
    void finish(const std::string& alphabet_sorted /* e.g. "$ACGT" or "$ACGTX" */) {
        std::fill(rank_of, rank_of + 256, -1);
        back_transform.assign(alphabet_sorted.begin(), alphabet_sorted.end());
        for (size_t r = 0; r < alphabet_sorted.size(); ++r) rank_of[(uint8_t)alphabet_sorted[r]] = (int)r;
        int f = 0;  // :38-47
        for (size_t i = 0; i < bwt.size() && f < 2; ++i) if (bwt[i] == 0) sentinel_occ[f++] = i;
    }
    uint64_t occ(uint64_t r, uint8_t a) const { return occ_table.get(bwt, r, a); }  // :23-25
    RtBiInterval init_interval() const { return {0, 0, (uint64_t)bwt.size()}; }     // :67-73
    uint8_t get_rev(uint8_t rank) const { return back_transform[rank]; }            // :103-105

    // FmdExtIterator (:109-182) unrolled: fills out[0..4) for c = 4,3,2,1 (T,G,C,A).
    void extend_all(const RtBiInterval& in, RtBiInterval out[4]) const {
        auto sent = [&](uint64_t pos) -> uint64_t {  // :140-146
            for (int i = 0; i < 2; ++i) if (pos < sentinel_occ[i]) return (uint64_t)i;
            return 2;
        };
        const uint64_t o0 = in.lower == 0 ? 0 : sent(in.lower - 1);
        uint64_t s = sent(in.lower + in.size - 1) - o0;
        uint64_t l = in.lower_rev;
        for (int k = 0; k < 4; ++k) {
            const uint8_t c = (uint8_t)(4 - k);
            l += s;  // :163
            const uint64_t o = in.lower == 0 ? 0 : occ(in.lower - 1, c);
            s = occ(in.lower + in.size - 1, c) - o;
            out[k] = {less[c] + o, l, s};
        }
    }
    // :77-91
    RtBiInterval backward_ext(const RtBiInterval& in, uint8_t a) const {
        if (rank_of[a] < 0) return {0, 0, 0};
        const int r = rank_of[a];
        RtBiInterval out[4];
        // the iterator is lazy: it stops at the matching rank; results are identical
        extend_all(in, out);
        if (r < 1 || r > 4) throw std::runtime_error("backward_ext: symbol not extendable");
        return out[4 - r];
    }
    // :93-96
    RtBiInterval forward_ext(const RtBiInterval& in, uint8_t a) const {
        return backward_ext(in.swapped(), dna_complement(a)).swapped();
    }
};

// src/utils.rs:12-33 — test index: text + '$' + revcomp + '$', alphabet "$ACGT", Occ k = 3.
// (k is a parameter here: indexing.rs:188 uses 128.)
inline RtFmdIndex build_index_from_text(std::vector<uint8_t> reference, const std::string& alphabet_sorted,
                                        uint32_t occ_k, std::vector<uint64_t>* sa_out) {
    const auto rc = dna_revcomp(reference);
    reference.push_back('$');
    reference.insert(reference.end(), rc.begin(), rc.end());
    reference.push_back('$');
    RtFmdIndex idx;
    idx.finish(alphabet_sorted);
    for (auto& c : reference) {
        if (idx.rank_of[c] < 0) throw std::runtime_error("symbol not in alphabet");
        c = (uint8_t)idx.rank_of[c];
    }
    auto sa = suffix_array_naive(reference);
    idx.bwt = bwt_from_sa(reference, sa);
    idx.less = less_from_bwt(idx.bwt, (int)alphabet_sorted.size());
    idx.occ_table = Occ(idx.bwt, occ_k, (int)alphabet_sorted.size());
    idx.finish(alphabet_sorted);
    if (sa_out) *sa_out = std::move(sa);
    return idx;
}
// Build from a ready-made BWT (used by bench.py's cpu_baseline leg: same BWT as the GPU index).
inline RtFmdIndex build_index_from_bwt(std::vector<uint8_t> bwt, const std::string& alphabet_sorted, uint32_t occ_k) {
    RtFmdIndex idx;
    idx.bwt = std::move(bwt);
    idx.less = less_from_bwt(idx.bwt, (int)alphabet_sorted.size());
    idx.occ_table = Occ(idx.bwt, occ_k, (int)alphabet_sorted.size());
    idx.finish(alphabet_sorted);
    return idx;
}

// ---------------------------------------------------------------------------------
// BiDArray (src/map/bi_d_array.rs:18-225)
// ---------------------------------------------------------------------------------
struct BiDArray {
    std::vector<float> d_composite;
    size_t split = 0;

    // :104-198 — one offset chain; returns the first `n_take` items of the lazy iterator
    // (the reference only ever pulls `part.len()` items, so the last extension is never run).
    static std::vector<float> compute_part(const uint8_t* part, const uint8_t* quals, size_t part_len, Direction direction,
                                           size_t full_len, uint16_t initial_skip, const AlignmentParameters& ap,
                                           const RtFmdIndex& fmd, const SequenceDifferenceModel& sdm, size_t n_take,
                                           Counters* ctr) {
        std::vector<float> out;
        out.reserve(n_take);
        for (size_t i = 0; i < (size_t)initial_skip + 1 && out.size() < n_take; ++i) out.push_back(0.0f);
        float z = 0.0f;
        int16_t last_mismatch_pos = (int16_t)initial_skip - 1;
        RtBiInterval interval = fmd.init_interval();
        auto at = [&](size_t idx) { return direction == Direction::Forward ? idx : part_len - 1 - idx; };
        for (size_t index = initial_skip; index < part_len && out.size() < n_take; ++index) {
            const uint8_t base = part[at(index)];
            interval = direction == Direction::Forward ? fmd.forward_ext(interval, base) : fmd.backward_ext(interval, base);
            if (ctr) ctr->e_darray += 1;
            if (interval.size < 1) {
                float m = F32_MIN;
                for (size_t j = (size_t)(last_mismatch_pos + 1); j <= index; ++j) {
                    const uint8_t base_j = part[at(j)], qual_j = quals[at(j)];
                    const size_t idx_read = direction == Direction::Forward ? j : full_len - 1 - j;  // :116-121
                    const float best_mm = sdm.get_min_penalty(idx_read, full_len, base_j, qual_j, true);
                    const float optimal = sdm.get_min_penalty(idx_read, full_len, base_j, qual_j, false);
                    float v = best_mm - optimal;
                    if (std::min(idx_read, full_len - idx_read - 1) >= (size_t)ap.gap_dist_ends) v = f32_max(v, ap.penalty_gap_extend);
                    m = f32_max(m, v);
                }
                z += m;
                interval = fmd.init_interval();
                last_mismatch_pos = (int16_t)index;
            }
            out.push_back(z);
        }
        return out;
    }

    // :24-99
    BiDArray(const uint8_t* pattern, const uint8_t* quals, size_t len, size_t split_, const AlignmentParameters& ap,
             const RtFmdIndex& fmd, const SequenceDifferenceModel& sdm, Counters* ctr)
        : split(split_) {
        constexpr uint16_t MAX_OFFSET = 15;
        d_composite.assign(len, 0.0f);
        // "backward" D from pattern[..split] extended forwards
        {
            std::vector<std::vector<float>> parts;
            for (uint16_t o = 0; o < MAX_OFFSET; ++o)
                parts.push_back(compute_part(pattern, quals, split, Direction::Forward, len, o, ap, fmd, sdm, split, ctr));
            for (size_t i = 0; i < split; ++i) {
                float acc = 0.0f;
                for (auto& p : parts) acc = f32_min(acc, p[i]);
                d_composite[i] = acc;
            }
        }
        // "forward" D from pattern[split..] extended backwards
        {
            const size_t rest = len - split;
            std::vector<std::vector<float>> parts;
            for (uint16_t o = 0; o < MAX_OFFSET; ++o)
                parts.push_back(compute_part(pattern + split, quals + split, rest, Direction::Backward, len, o, ap, fmd, sdm, rest, ctr));
            for (size_t i = 0; i < rest; ++i) {
                float acc = 0.0f;
                for (auto& p : parts) acc = f32_min(acc, p[i]);
                d_composite[split + i] = acc;
            }
        }
    }

    // :200-224
    float get(int16_t backward_index, int16_t forward_index) const {
        const size_t n = d_composite.size();
        float d_rev = 0.0f;
        if (backward_index >= 0 && (size_t)backward_index < n) d_rev = d_composite[(size_t)backward_index];
        float d_fwd = 0.0f;
        const size_t sub = 1 + (size_t)forward_index;  // forward_index >= 0 by contract
        if (n >= sub) {
            const size_t idx = (n - sub) + split;
            if (idx < n) d_fwd = d_composite[idx];
        }
        return d_rev + d_fwd;
    }
};

// ---------------------------------------------------------------------------------
// Edit operations + backtrack tree (src/map/record.rs:225-237; src/map/backtrack_tree.rs)
// ---------------------------------------------------------------------------------
enum class OpKind : uint8_t { Insertion = 0, Deletion = 1, Match = 2, Mismatch = 3 };
struct EditOperation {
    OpKind kind = OpKind::Match;
    uint16_t pos = 0;
    uint8_t base = 0;
    bool operator==(const EditOperation& o) const { return kind == o.kind && pos == o.pos && base == o.base; }
};

// slab 0.4 semantics: LIFO free list, len() = occupied count.
struct Tree {
    struct Node { EditOperation value; uint32_t parent; bool occupied; uint32_t next_free; };
    std::vector<Node> entries;
    uint32_t next = 0;
    size_t len_ = 0;
    uint32_t insert(EditOperation v, uint32_t parent) {
        const uint32_t key = next;
        if (key == entries.size()) { entries.push_back({v, parent, true, 0}); next = key + 1; }
        else { next = entries[key].next_free; entries[key] = {v, parent, true, 0}; }
        len_ += 1;
        return key;
    }
    void remove(uint32_t key) {  // backtrack_tree.rs:50-54
        if (key == 0) return;
        if (key >= entries.size() || !entries[key].occupied) throw std::runtime_error("slab: invalid key");
        entries[key].occupied = false;
        entries[key].next_free = next;
        next = key;
        len_ -= 1;
    }
    uint32_t add_node(EditOperation v, uint32_t parent) { return insert(v, parent); }  // :60-64
    size_t len() const { return len_; }
    uint32_t clear() {  // :93-97
        entries.clear(); next = 0; len_ = 0;
        insert(EditOperation{}, 0);
        return 0;
    }
    // :112-123 — leaf -> root, root excluded, stops at a vacant slot
    template <class F> void ancestors(uint32_t id, F f) const {
        uint32_t state = id;
        while (state != 0) {
            if (state >= entries.size() || !entries[state].occupied) return;
            const Node& n = entries[state];
            state = n.parent;
            f(n.value);
        }
    }
};

// src/map/record.rs:465-500
inline std::vector<EditOperation> extract_edit_operations(uint32_t end_node, const Tree& tree, int16_t alignment_start) {
    std::map<uint16_t, std::vector<EditOperation>> buckets;
    tree.ancestors(end_node, [&](const EditOperation& op) { buckets[op.pos].push_back(op); });
    std::vector<EditOperation> out;
    for (auto& [pos, vec] : buckets) {
        if (pos < (uint16_t)alignment_start) out.insert(out.end(), vec.begin(), vec.end());
        else out.insert(out.end(), vec.rbegin(), vec.rend());
    }
    return out;
}

// src/map/record.rs:269-449
struct BamFields {
    std::vector<std::pair<char, uint32_t>> cigar;  // kind char 'M','I','D' + run length
    std::string md;
    uint16_t nm = 0;
};
struct OriginalSymbols {
    std::map<uint64_t, uint8_t> map;
    std::optional<uint8_t> get(uint64_t idx) const {
        auto it = map.find(idx);
        if (it == map.end()) return std::nullopt;
        return it->second;
    }
};
inline size_t track_effective_len(const std::vector<EditOperation>& t) {
    size_t n = 0;
    for (auto& op : t) n += op.kind == OpKind::Insertion ? 0 : 1;
    return n;
}
inline size_t track_read_len(const std::vector<EditOperation>& t) {
    size_t n = 0;
    for (auto& op : t) n += op.kind == OpKind::Deletion ? 0 : 1;
    return n;
}
inline char cigar_kind(OpKind k) { return k == OpKind::Insertion ? 'I' : k == OpKind::Deletion ? 'D' : 'M'; }

inline BamFields to_bam_fields(const std::vector<EditOperation>& track, Direction strand, uint64_t absolute_pos,
                               const OriginalSymbols& orig) {
    BamFields out;
    uint32_t num_matches = 0, num_operations = 1;
    uint16_t edit_distance = 0;
    std::optional<EditOperation> last;
    auto comp = [&](uint8_t b) { return strand == Direction::Forward ? b : dna_complement(b); };
    auto add_md = [&](std::optional<EditOperation> op, std::optional<EditOperation> lop, uint32_t k) -> uint32_t {  // :391-430
        if (!op) { out.md += std::to_string(k); return k; }
        switch (op->kind) {
            case OpKind::Match: k += 1; break;
            case OpKind::Mismatch: out.md += std::to_string(k); out.md.push_back((char)comp(op->base)); k = 0; break;
            case OpKind::Insertion: break;
            case OpKind::Deletion:
                if (lop && lop->kind == OpKind::Deletion) out.md.push_back((char)comp(op->base));
                else { out.md += std::to_string(k); out.md.push_back('^'); out.md.push_back((char)comp(op->base)); }
                k = 0;
                break;
        }
        return k;
    };
    const size_t n = track.size();
    for (size_t i = 0; i < n; ++i) {  // :301 — i enumerates all ops incl. insertions
        EditOperation op = strand == Direction::Forward ? track[i] : track[n - 1 - i];
        const auto o = orig.get(absolute_pos + i);
        switch (op.kind) {  // :302-320
            case OpKind::Insertion: break;
            case OpKind::Match: if (o) { op.kind = OpKind::Mismatch; op.base = *o; } break;
            case OpKind::Deletion: if (o) op.base = *o; break;
            case OpKind::Mismatch: if (o) op.base = *o; break;
        }
        if (op.kind != OpKind::Match) edit_distance += 1;  // :432-438
        num_matches = add_md(op, last, num_matches);
        if (last) {  // :333-381
            const bool same_run = cigar_kind(op.kind) == cigar_kind(last->kind);
            if (same_run) num_operations += 1;
            else { out.cigar.push_back({cigar_kind(last->kind), num_operations}); num_operations = 1; last = op; }
        } else {
            last = op;
        }
    }
    if (last) out.cigar.push_back({cigar_kind(last->kind), num_operations});
    add_md(std::nullopt, std::nullopt, num_matches);
    out.nm = edit_distance;
    return out;
}

// ---------------------------------------------------------------------------------
// Frames, hits and their heaps
// ---------------------------------------------------------------------------------
enum class GapState : uint8_t { Insertion, Deletion, Closed };  // src/map/mod.rs:93-98

// src/map/mod.rs:105-137 (40 bytes in the reference)
struct Frame {
    RtBiInterval current_interval;
    int16_t start = 0, len = 0;
    GapState gap_forwards = GapState::Closed, gap_backwards = GapState::Closed;
    uint8_t num_gaps_open = 0;
    float alignment_score = 0;
    uint32_t edit_node_id = 0;
};
static_assert(sizeof(Frame) == 40, "frame is 40 bytes like the reference's");

// src/map/mod.rs:34-61
struct HitInterval {
    RtBiInterval interval;
    float alignment_score = 0;
    std::vector<EditOperation> edit_operations;
};

// min-max-heap 1.3.1-alpha.0 (tov) — SURVEY Appendix A.2.  Keyed on alignment_score only.
// `variant` selects between readings of the two unverifiable tie details:
//   bit0 = 0: candidate scan order child1, child2, then the four grandchildren  (ascending index; default)
//   bit0 = 1: candidate scan order child1, gc(child1)x2, child2, gc(child2)x2
//   bit1 = 1: pop_max prefers slot 1 on ties (v[1] >= v[2])
struct MinMaxHeap {
    std::vector<Frame> v;
    int variant = 0;
    static bool is_min_level(size_t pos) { return (__builtin_clzll((unsigned long long)pos + 1) & 1) == 1; }
    void clear() { v.clear(); }
    size_t len() const { return v.size(); }

    void push(const Frame& f) {
        v.push_back(f);
        bubble_up(v.size() - 1);
    }
    void bubble_up(size_t pos) {
        Frame elt = v[pos];
        auto hop = [&](bool greater) {
            while (pos > 2) {
                const size_t gp = (pos - 3) / 4;
                const bool go = greater ? (elt.alignment_score > v[gp].alignment_score) : (elt.alignment_score < v[gp].alignment_score);
                if (!go) break;
                v[pos] = v[gp]; pos = gp;
            }
        };
        if (pos > 0) {
            const size_t parent = (pos - 1) / 2;
            if (is_min_level(pos)) {
                if (elt.alignment_score > v[parent].alignment_score) { v[pos] = v[parent]; pos = parent; hop(true); }
                else hop(false);
            } else {
                if (elt.alignment_score < v[parent].alignment_score) { v[pos] = v[parent]; pos = parent; hop(false); }
                else hop(true);
            }
        }
        v[pos] = elt;
    }
    // trickle down; `mx` = max flavour (strict >) else min flavour (strict <)
    void trickle_down(size_t pos, bool mx) {
        Frame elt = v[pos];
        const size_t n = v.size();
        auto better = [&](float a, float b) { return mx ? a > b : a < b; };
        while (2 * pos + 1 < n) {
            const size_t c1 = 2 * pos + 1, c2 = 2 * pos + 2;
            size_t best = c1; bool grandchild = false;
            auto check = [&](size_t idx, bool gc) {
                if (idx < n && better(v[idx].alignment_score, v[best].alignment_score)) { best = idx; grandchild = gc; }
            };
            if ((variant & 1) == 1) { check(2 * c1 + 1, true); check(2 * c1 + 2, true); check(c2, false); check(2 * c2 + 1, true); check(2 * c2 + 2, true); }
            else { check(c2, false); check(2 * c1 + 1, true); check(2 * c1 + 2, true); check(2 * c2 + 1, true); check(2 * c2 + 2, true); }
            if (!better(v[best].alignment_score, elt.alignment_score)) break;
            v[pos] = v[best]; pos = best;
            if (!grandchild) break;
            const size_t parent = (pos - 1) / 2;
            if (better(v[parent].alignment_score, elt.alignment_score)) std::swap(elt, v[parent]);
        }
        v[pos] = elt;
    }
    bool pop_max(Frame& out) {
        const size_t n = v.size();
        if (n == 0) return false;
        size_t idx;
        if (n == 1) idx = 0; else if (n == 2) idx = 1;
        else if (variant & 2) idx = (v[1].alignment_score >= v[2].alignment_score) ? 1 : 2;
        else idx = (v[1].alignment_score > v[2].alignment_score) ? 1 : 2;
        Frame item = v.back(); v.pop_back();
        if (idx < v.size()) { std::swap(item, v[idx]); trickle_down(idx, true); }
        out = item;
        return true;
    }
    bool pop_min(Frame& out) {
        if (v.empty()) return false;
        Frame item = v.back(); v.pop_back();
        if (!v.empty()) { std::swap(item, v[0]); trickle_down(0, false); }
        out = item;
        return true;
    }
};

// Rust std::collections::BinaryHeap — SURVEY Appendix A.3.  Keyed on alignment_score only.
struct HitHeap {
    std::vector<HitInterval> data;
    size_t len() const { return data.size(); }
    bool empty() const { return data.empty(); }
    const HitInterval* peek() const { return data.empty() ? nullptr : &data[0]; }
    void sift_up(size_t start, size_t pos) {
        HitInterval elt = std::move(data[pos]);
        while (pos > start) {
            const size_t parent = (pos - 1) / 2;
            if (elt.alignment_score <= data[parent].alignment_score) break;
            data[pos] = std::move(data[parent]); pos = parent;
        }
        data[pos] = std::move(elt);
    }
    void push(HitInterval h) { data.push_back(std::move(h)); sift_up(0, data.size() - 1); }
    void sift_down_range(size_t pos, size_t end) {
        HitInterval elt = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        while (child <= (end >= 2 ? end - 2 : 0) && end >= 2) {
            child += (data[child].alignment_score <= data[child + 1].alignment_score) ? 1 : 0;
            if (elt.alignment_score >= data[child].alignment_score) { data[pos] = std::move(elt); return; }
            data[pos] = std::move(data[child]); pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1 && elt.alignment_score < data[child].alignment_score) { data[pos] = std::move(data[child]); pos = child; }
        data[pos] = std::move(elt);
    }
    void sift_down_to_bottom(size_t pos) {
        const size_t end = data.size(), start = pos;
        HitInterval elt = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            child += (data[child].alignment_score <= data[child + 1].alignment_score) ? 1 : 0;
            data[pos] = std::move(data[child]); pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1) { data[pos] = std::move(data[child]); pos = child; }
        data[pos] = std::move(elt);
        sift_up(start, pos);
    }
    bool pop(HitInterval& out) {
        if (data.empty()) return false;
        HitInterval item = std::move(data.back()); data.pop_back();
        if (!data.empty()) { std::swap(item, data[0]); sift_down_to_bottom(0); }
        out = std::move(item);
        return true;
    }
    std::vector<HitInterval> into_sorted_vec() && {
        size_t end = data.size();
        while (end > 1) { end -= 1; std::swap(data[0], data[end]); sift_down_range(0, end); }
        return std::move(data);
    }
};

// src/map/mapping.rs:572-588
inline std::vector<float> compute_optimal_scores(const uint8_t* pattern, const uint8_t* quals, size_t len,
                                                 const SequenceDifferenceModel& sdm) {
    std::vector<float> v(len);
    for (size_t i = 0; i < len; ++i) v[i] = sdm.get_min_penalty(i, len, pattern[i], quals[i], false);
    return v;
}

// src/map/mapping.rs:932-987
inline void check_and_push_stack_frame(Frame frame, size_t pattern_len, int16_t alignment_start_pos, EditOperation op,
                                       Tree& tree, MinMaxHeap& stack, HitHeap& hits, const MismatchBound& mb,
                                       const AlignmentParameters& ap, Counters* ctr) {
    if (const HitInterval* best = hits.peek())
        if (mb.reject_iterative(frame.alignment_score, best->alignment_score)) return;
    if (frame.num_gaps_open > ap.max_num_gaps_open) return;
    frame.edit_node_id = tree.add_node(op, frame.edit_node_id);
    if (ctr) ctr->n_node += 1;
    if ((size_t)frame.len == pattern_len) {
        HitInterval h;
        h.interval = frame.current_interval;
        h.alignment_score = frame.alignment_score;
        h.edit_operations = extract_edit_operations(frame.edit_node_id, tree, alignment_start_pos);
        hits.push(std::move(h));
        if (ctr) ctr->n_hits += 1;
        return;
    }
    stack.push(frame);
    if (ctr) ctr->n_push += 1;
}

// src/map/mapping.rs:1012-1383
inline HitHeap k_mismatch_search(const uint8_t* pattern, const uint8_t* quals, size_t plen, const AlignmentParameters& ap,
                                 const RtFmdIndex& fmd, MinMaxHeap& stack, Tree& tree, const SequenceDifferenceModel& sdm,
                                 const MismatchBound& mb, Counters* ctr, std::vector<float>* d_out = nullptr) {
    const int16_t alignment_start_pos = sdm.find_alignment_start(plen);  // :1026
    BiDArray bi_d(pattern, quals, plen, (size_t)alignment_start_pos, ap, fmd, sdm, ctr);
    if (d_out) *d_out = bi_d.d_composite;
    const std::vector<float> optimal_penalties = compute_optimal_scores(pattern, quals, plen, sdm);
    HitHeap hits;
    stack.clear();
    const uint32_t root = tree.clear();
    {
        Frame f;
        f.current_interval = fmd.init_interval();
        f.start = alignment_start_pos; f.len = 0;
        f.gap_backwards = GapState::Closed; f.gap_forwards = GapState::Closed;
        f.num_gaps_open = 0; f.alignment_score = 0.0f; f.edit_node_id = root;
        stack.push(f);
        if (ctr) ctr->n_push += 1;
    }
    const int16_t L = (int16_t)plen;
    static const uint8_t REV_ACGT[4] = {'T', 'G', 'C', 'A'};  // b"ACGT".iter().rev()
    float mm_scores[4];
    Frame sf;
    while (stack.pop_max(sf)) {
        if (ctr) ctr->n_pop += 1;
        int16_t j, d_k, d_l; Direction direction;
        if (sf.start <= (int16_t)(L - sf.start - sf.len)) {  // :1077-1097
            j = sf.start + sf.len; direction = Direction::Forward; d_k = sf.start; d_l = sf.start + sf.len;
        } else {
            j = sf.start - 1; direction = Direction::Backward; d_k = sf.start - 1; d_l = sf.start + sf.len - 1;
        }
        RtBiInterval ext_interval;
        GapState next_ins_b, next_ins_f, next_del_b, next_del_f, next_closed_b, next_closed_f;
        float insertion_score, deletion_score; uint8_t num_gaps_open;
        const float optimal_penalty = optimal_penalties[(size_t)j];
        const float open_ext = ap.penalty_gap_open + ap.penalty_gap_extend;
        if (direction == Direction::Forward) {  // :1116-1153
            ext_interval = sf.current_interval.swapped();
            next_ins_b = sf.gap_backwards; next_ins_f = GapState::Insertion;
            next_del_b = sf.gap_backwards; next_del_f = GapState::Deletion;
            next_closed_b = sf.gap_backwards; next_closed_f = GapState::Closed;
            insertion_score = (sf.gap_forwards == GapState::Insertion ? ap.penalty_gap_extend : open_ext) + sf.alignment_score;
            deletion_score = (sf.gap_forwards == GapState::Deletion ? ap.penalty_gap_extend : open_ext) + sf.alignment_score;
            for (int k = 0; k < 4; ++k)
                mm_scores[k] = sdm.get((size_t)j, plen, dna_complement(REV_ACGT[k]), pattern[j], quals[j]) - optimal_penalty + sf.alignment_score;
            num_gaps_open = sf.gap_forwards == GapState::Closed ? sf.num_gaps_open + 1 : sf.num_gaps_open;
        } else {  // :1154-1191
            ext_interval = sf.current_interval;
            next_ins_b = GapState::Insertion; next_ins_f = sf.gap_forwards;
            next_del_b = GapState::Deletion; next_del_f = sf.gap_forwards;
            next_closed_b = GapState::Closed; next_closed_f = sf.gap_forwards;
            insertion_score = (sf.gap_backwards == GapState::Insertion ? ap.penalty_gap_extend : open_ext) + sf.alignment_score;
            deletion_score = (sf.gap_backwards == GapState::Deletion ? ap.penalty_gap_extend : open_ext) + sf.alignment_score;
            for (int k = 0; k < 4; ++k)
                mm_scores[k] = sdm.get((size_t)j, plen, REV_ACGT[k], pattern[j], quals[j]) - optimal_penalty + sf.alignment_score;
            num_gaps_open = sf.gap_backwards == GapState::Closed ? sf.num_gaps_open + 1 : sf.num_gaps_open;
        }
        const float lower_bound = bi_d.get(d_k, d_l);  // :1195
        if (const HitInterval* best = hits.peek())   // :1201-1208
            if (mb.reject_iterative(sf.alignment_score + lower_bound, best->alignment_score)) break;

        // Insertion in read (:1213-1242)
        if (!mb.reject(insertion_score + lower_bound, plen) && std::min<int16_t>(j, (int16_t)(L - j - 1)) >= (int16_t)ap.gap_dist_ends) {
            Frame c = sf;
            c.start = direction == Direction::Backward ? sf.start - 1 : sf.start;
            c.len = sf.len + 1;
            c.gap_backwards = next_ins_b; c.gap_forwards = next_ins_f;
            c.alignment_score = insertion_score; c.num_gaps_open = num_gaps_open;
            check_and_push_stack_frame(c, plen, alignment_start_pos, EditOperation{OpKind::Insertion, (uint16_t)j, 0}, tree, stack, hits, mb, ap, ctr);
        }
        // Extension (:1245-1339)
        RtBiInterval ext[4];
        fmd.extend_all(ext_interval, ext);
        if (ctr) ctr->e_search += 1;
        for (int k = 0; k < 4; ++k) {
            RtBiInterval ip = ext[k];
            if (ip.size < 1) continue;
            uint8_t c = fmd.get_rev((uint8_t)(4 - k));
            if (direction == Direction::Forward) { ip = ip.swapped(); c = dna_complement(c); }
            {   // Deletion in read (:1265-1302)
                const int16_t dist_5 = direction == Direction::Backward ? j + 1 : j;
                const int16_t dist_3 = L - dist_5;
                const int16_t dist = std::min(dist_5, dist_3);
                if (!mb.reject(deletion_score + lower_bound, plen) && dist >= (int16_t)ap.gap_dist_ends) {
                    Frame ch = sf;
                    ch.current_interval = ip;
                    ch.gap_backwards = next_del_b; ch.gap_forwards = next_del_f;
                    ch.alignment_score = deletion_score; ch.num_gaps_open = num_gaps_open;
                    check_and_push_stack_frame(ch, plen, alignment_start_pos, EditOperation{OpKind::Deletion, (uint16_t)j, c}, tree, stack, hits, mb, ap, ctr);
                }
            }
            // Match / mismatch (:1307-1338)
            if (!mb.reject(mm_scores[k] + lower_bound, plen)) {
                Frame ch = sf;
                ch.current_interval = ip;
                ch.start = direction == Direction::Backward ? sf.start - 1 : sf.start;
                ch.len = sf.len + 1;
                ch.gap_backwards = next_closed_b; ch.gap_forwards = next_closed_f;
                ch.alignment_score = mm_scores[k];
                EditOperation op = (c == pattern[j]) ? EditOperation{OpKind::Match, (uint16_t)j, 0} : EditOperation{OpKind::Mismatch, (uint16_t)j, c};
                check_and_push_stack_frame(ch, plen, alignment_start_pos, op, tree, stack, hits, mb, ap, ctr);
            }
        }
        // :1348-1355
        if (hits.len() > 9 || (hits.peek() && hits.peek()->interval.size > 1)) return hits;
        // :1358-1380
        if (stack.len() > (size_t)ap.stack_limit || tree.len() > (size_t)ap.edit_tree_limit) {
            if (ap.stack_limit_abort) return hits;
            const int64_t a = (int64_t)stack.len() - (int64_t)ap.stack_limit;
            const int64_t b = (int64_t)tree.len() - (int64_t)ap.edit_tree_limit;
            for (int64_t i = 0; i < std::max(a, b); ++i) {
                Frame m;
                if (stack.pop_min(m)) tree.remove(m.edit_node_id);
            }
        }
    }
    return hits;
}

// ---------------------------------------------------------------------------------
// PrRange (src/map/prrange.rs)
// ---------------------------------------------------------------------------------
struct PrRange {
    uint64_t start, l, m, a, x, seed, count;
    static bool is_prime(uint64_t n) {
        if (n <= 1) return false; if (n <= 3) return true;
        if (n % 2 == 0 || n % 3 == 0) return false;
        for (uint64_t i = 5; i * i <= n; i += 6) if (n % i == 0 || n % (i + 2) == 0) return false;
        return true;
    }
    static uint64_t next_prime(uint64_t n) {
        uint64_t p = n + 1;
        if (p <= 2) return 2;
        if (p % 2 == 0) p += 1;
        while (!is_prime(p)) p += 2;
        return p;
    }
    static bool checked_pow_mod(uint64_t base, uint64_t exponent, uint64_t modulus, uint64_t& out) {
        if (modulus == 1) { out = 0; return true; }
        unsigned __int128 chk = (unsigned __int128)(modulus - 1) * (modulus - 1);
        if (chk >> 64) return false;
        uint64_t result = 1; base %= modulus;
        while (exponent > 0) {
            if (exponent % 2 == 1) result = (result * base) % modulus;
            exponent >>= 1;
            base = (base * base) % modulus;
        }
        out = result; return true;
    }
    // PrimeFactorIterator :126-165 — restated as the same resumable loop nest
    struct PrimeFactorIterator {
        uint64_t n, i = 2, step = 1, last = 0;
        explicit PrimeFactorIterator(uint64_t n_) : n(n_) {}
        bool next(uint64_t& out) {
            if (n <= 3) return false;
            while (i * i <= n) {
                while (n > 1) {
                    while (n % i == 0) {
                        if (i > last) { out = i; last = i; return true; }
                        n /= i;
                    }
                    i += step;
                    step = 2;
                }
            }
            return false;
        }
    };
    static std::vector<uint64_t> prime_factors(uint64_t n0) {
        std::vector<uint64_t> out;
        PrimeFactorIterator it(n0);
        uint64_t f;
        while (it.next(f)) out.push_back(f);
        return out;
    }
    static bool is_primitive_root(uint64_t a, uint64_t n, bool& ok) {
        const uint64_t phi = n - 1;
        for (uint64_t p : prime_factors(phi)) {
            uint64_t r;
            if (!checked_pow_mod(a, phi / p, n, r)) { ok = false; return false; }
            if (r == 1) return false;
        }
        return true;
    }
    static std::optional<PrRange> try_new(uint64_t start, uint64_t end, uint64_t seed) {
        const uint64_t l = end > start ? end - start : 0;
        if (l == 0) return std::nullopt;
        const uint64_t m = next_prime(l);
        uint64_t a = 2; bool ok = true;
        while (!is_primitive_root(a, m, ok)) { if (!ok) return std::nullopt; a += 1; }
        seed = std::max<uint64_t>(seed % l, 1);
        return PrRange{start, l, m, a, seed, seed, 0};
    }
    bool next(uint64_t& out) {
        if (count == 0 && l == 1) { count += 1; out = start; return true; }
        while (true) {
            const uint64_t prev_x = x;
            x = (a * x) % m;
            if (count > 0 && prev_x == seed) return false;
            if (prev_x <= l) { count += 1; out = prev_x - 1 + start; return true; }
        }
    }
};

// ---------------------------------------------------------------------------------
// Sampled SA, contig map (src/index/mod.rs:44-196)
// ---------------------------------------------------------------------------------
struct FastaIdPosition { uint64_t start, end; std::string identifier; };
struct FastaIdPositions {
    std::vector<FastaIdPosition> id_position;
    // :55-75
    bool get_reference_identifier(uint64_t position, uint64_t pattern_length, uint32_t& tid, uint64_t& rel, const std::string*& name) const {
        for (size_t i = 0; i < id_position.size(); ++i) {
            const auto& id = id_position[i];
            if (id.start <= position && position + pattern_length - 1 <= id.end) {
                tid = (uint32_t)i; rel = position - id.start; name = &id.identifier; return true;
            }
        }
        return false;
    }
};
struct SampledSuffixArray {
    const RtFmdIndex* fmd = nullptr;
    std::vector<uint64_t> sample;
    uint64_t sampling_rate = 32;
    std::map<uint64_t, uint64_t> extra_rows;
    uint8_t sentinel = 0;
    // :88-128
    static SampledSuffixArray sample_from(const std::vector<uint64_t>& sa, const RtFmdIndex& fmd, uint64_t rate) {
        SampledSuffixArray s; s.fmd = &fmd; s.sampling_rate = rate; s.sentinel = 0;
        for (size_t i = 0; i < sa.size(); ++i) {
            if (i % rate == 0) s.sample.push_back(sa[i]);
            else if (fmd.bwt[i] == s.sentinel) s.extra_rows[i] = sa[i];
        }
        return s;
    }
    uint64_t len() const { return fmd->bwt.size(); }
    // :160-187
    bool get(uint64_t index, uint64_t& out) const {
        if (index >= len()) return false;
        uint64_t pos = index, offset = 0;
        while (true) {
            if (pos % sampling_rate == 0) { out = sample[pos / sampling_rate] + offset; return true; }
            const uint8_t c = fmd->bwt[pos];
            if (c == sentinel) { out = extra_rows.at(pos) + offset; return true; }
            pos = fmd->less[c] + fmd->occ(pos - 1, c);
            offset += 1;
        }
    }
};

// ---------------------------------------------------------------------------------
// Post-search (src/map/mapping.rs:402-718, 722-927 minus BAM byte encoding)
// ---------------------------------------------------------------------------------
struct OutRecord {
    uint16_t flags = 0;
    int32_t tid = -1;
    int64_t pos = -1;  // 0-based; -1 unmapped (BAM POS = pos + 1)
    uint8_t mapq = 0;
    bool mapped = false;
    bool reverse = false;
    std::string cigar, md, xa;
    std::string seq; std::vector<uint8_t> qual;  // as written (revcomp'ed / reversed on the reverse strand)
    float as = 0, xs = 0;
    int32_t nm = 0, x0 = 0, x1 = 0;
    bool has_xs = false;
    char xt = 0;
};

inline bool interval_cross_check(const RtBiInterval& a, const RtBiInterval& b) {  // :651-653
    return a.size == b.size && (a.lower == b.lower || a.lower_rev == b.lower_rev);
}

// :658-718
inline uint8_t estimate_mapping_quality(const HitInterval& best, uint64_t best_size, const std::vector<HitInterval>& others,
                                        const MismatchBound& mb) {
    const float MAX_MAPQ = 37.0f, MIN_MAPQ_UNIQ = 20.0f;
    float p;
    {
        const float prob_best = std::exp2(best.alignment_score);
        if (best_size > 1) p = 1.0f / (float)best_size;
        else {
            float acc = 0.0f;
            for (const auto& s : others) {
                if (interval_cross_check(best.interval, s.interval)) continue;
                acc = std::fmaf(std::exp2(s.alignment_score), (float)s.interval.size, acc);
            }
            p = prob_best / (prob_best + acc);
        }
        // Rust f32::clamp(0,1): NaN stays NaN
        if (p < 0.0f) p = 0.0f; if (p > 1.0f) p = 1.0f;
    }
    auto to_u8 = [](float x) -> uint8_t {  // `as u8` saturates, NaN -> 0
        if (std::isnan(x)) return 0; if (x <= 0.0f) return 0; if (x >= 255.0f) return 255; return (uint8_t)x;
    };
    const uint8_t mq = to_u8(std::round(f32_min(-10.0f * std::log10(1.0f - p), MAX_MAPQ)));
    if (mq == 37) {
        const float frac = f32_min(mb.remaining_frac_of_repr_mm(best.alignment_score, track_read_len(best.edit_operations)), 1.0f);
        return to_u8(std::round(std::fmaf(MAX_MAPQ - MIN_MAPQ_UNIQ, frac, MIN_MAPQ_UNIQ)));
    }
    return mq;
}

struct Coord { uint32_t tid; const std::string* contig; uint64_t rel, abs; Direction strand; size_t num_skipped; const HitInterval* hit; };

// :590-649 — eager version of the lazy iterator: all valid coordinates in PrRange order.
// `seed` plays the role of rng.next_u32() (non-deterministic in the reference; only matters for >= 3 rows).
inline bool interval2coordinate(const HitInterval& hit, const SampledSuffixArray& sa, const FastaIdPositions& idmap,
                                uint32_t seed, std::vector<Coord>& out) {
    const uint64_t strand_len = sa.len() / 2;
    const uint64_t eff = track_effective_len(hit.edit_operations);
    auto pr = PrRange::try_new(hit.interval.lower, hit.interval.lower + hit.interval.size, seed);
    if (!pr) return false;
    uint64_t row; size_t i = 0;
    while (pr->next(row)) {
        uint64_t p;
        if (sa.get(row, p)) {
            Direction strand = Direction::Forward;
            if (p >= strand_len) { p = sa.len() - p - eff - 1; strand = Direction::Backward; }
            uint32_t tid; uint64_t rel; const std::string* name;
            if (idmap.get_reference_identifier(p, eff, tid, rel, name)) out.push_back({tid, name, rel, p, strand, i, &hit});
        }
        ++i;
    }
    return true;
}

inline std::string cigar_string(const BamFields& f) {
    std::string s;
    for (auto& [k, n] : f.cigar) { s += std::to_string(n); s.push_back(k); }
    return s;
}

struct InRecord { std::string name; uint16_t flags = 4; std::vector<uint8_t> seq, qual; };

// :402-567 + flag/seq handling of :748-819.  `next_seed()` is called once per interval2coordinate().
template <class SeedFn>
inline OutRecord intervals_to_record(const InRecord& in, HitHeap hits_heap, const SampledSuffixArray& sa,
                                     const FastaIdPositions& idmap, const OriginalSymbols& orig, const MismatchBound& mb,
                                     SeedFn next_seed) {
    OutRecord rec;
    uint16_t flags = in.flags;
    flags &= ~(0x8 | 0x20 | 0x2 | 0x100 | 0x800);
    std::vector<HitInterval> intervals = std::move(hits_heap).into_sorted_vec();
    while (!intervals.empty()) {
        HitInterval best = std::move(intervals.back()); intervals.pop_back();
        std::vector<Coord> best_coords;
        if (!interval2coordinate(best, sa, idmap, next_seed(), best_coords)) throw std::runtime_error("invalid index");
        if (best_coords.empty()) continue;  // :541-543
        const Coord first = best_coords.front();
        const uint64_t upd_size = best.interval.size - first.num_skipped;
        // XA: remaining coords of best, then all coords of non-cross-checked suboptimals (descending score); take(2).
        // The reference's chain is lazy: suboptimal intervals are only converted (and the rng only advanced) while
        // fewer than 2 entries have been produced.
        std::vector<std::pair<Coord, const HitInterval*>> xa_items;
        for (size_t i = 1; i < best_coords.size() && xa_items.size() < 2; ++i) xa_items.push_back({best_coords[i], &best});
        for (size_t r = intervals.size(); r-- > 0 && xa_items.size() < 2;) {
            const HitInterval& sub = intervals[r];
            if (interval_cross_check(best.interval, sub.interval)) continue;
            std::vector<Coord> cs;
            if (!interval2coordinate(sub, sa, idmap, next_seed(), cs)) continue;
            for (auto& c : cs) { if (xa_items.size() >= 2) break; xa_items.push_back({c, &sub}); }
        }
        for (auto& [c, h] : xa_items) {
            const BamFields bf = to_bam_fields(h->edit_operations, c.strand, c.abs, orig);
            char buf[64];
            std::snprintf(buf, sizeof buf, "%.2f", (double)h->alignment_score);
            rec.xa += *c.contig + "," + (c.strand == Direction::Forward ? "+" : "-") + std::to_string(c.rel + 1) + "," +
                      cigar_string(bf) + "," + bf.md + "," + std::to_string(bf.nm) + "," + std::to_string(h->interval.size) + "," + buf + ";";
        }
        rec.x0 = upd_size > (uint64_t)INT32_MAX ? INT32_MAX : (int32_t)upd_size;
        uint64_t x1 = 0;
        for (auto& s : intervals) if (!interval_cross_check(best.interval, s.interval)) x1 += s.interval.size;
        rec.x1 = x1 > (uint64_t)INT32_MAX ? INT32_MAX : (int32_t)x1;
        rec.xs = intervals.empty() ? 0.0f : intervals.back().alignment_score;
        rec.has_xs = rec.x1 > 0;
        rec.xt = upd_size == 0 ? 'N' : upd_size == 1 ? 'U' : 'R';
        rec.mapq = estimate_mapping_quality(best, upd_size, intervals, mb);
        // create_bam_record (:722-927) fields
        const BamFields bf = to_bam_fields(best.edit_operations, first.strand, first.abs, orig);
        rec.cigar = cigar_string(bf); rec.md = bf.md; rec.nm = bf.nm;
        rec.mapped = true; rec.tid = (int32_t)first.tid; rec.pos = (int64_t)first.rel;
        rec.reverse = first.strand == Direction::Backward;
        rec.as = best.alignment_score;
        flags &= ~0x4;
        if (rec.reverse) flags |= 0x10; else flags &= ~0x10;
        rec.flags = flags;
        if (rec.reverse) {
            auto rc = dna_revcomp(in.seq);
            rec.seq.assign(rc.begin(), rc.end());
            rec.qual.assign(in.qual.rbegin(), in.qual.rend());
        } else { rec.seq.assign(in.seq.begin(), in.seq.end()); rec.qual = in.qual; }
        return rec;
    }
    // unmapped (:553-566, :765-776)
    flags |= 0x4; flags &= ~0x10; flags &= ~0x2;
    rec.flags = flags; rec.mapq = 0;
    rec.seq.assign(in.seq.begin(), in.seq.end()); rec.qual = in.qual;
    return rec;
}

}  // namespace mo
