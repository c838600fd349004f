// postproc_core.hpp — the numeric half of intervals_to_bam (src/map/mapping.rs:402-567) as code that runs on the device:
// which hit is reported, its coordinate (strand, contig, position), the XA candidates, X0 / X1.  One GPU thread per read
// (records_kernel in mapad_amd.hip); host_postproc.hpp keeps the host restatement of the same logic (the parity reference for this
// file) and everything that is text or libm: CIGAR / MD / XA strings and the mapping quality.
//
// Follows: into_sorted_vec, interval2coordinate (mapping.rs:590-649), PrRange (src/map/prrange.rs),
// SampledSuffixArray::get (src/index/mod.rs:160-187), FastaIdPositions::get_reference_identifier (:55-75),
// interval_cross_check (mapping.rs:651-653), the x0 / x1 bookkeeping (:430-431,494-508).
#pragma once
#include "fmd_device.hpp"
#include "search_core.hpp"

namespace mapad {

// ---- PrRange: Lehmer-LCG lazy permutation of an SA interval ---------------------------------------------------------------------
struct PrRangeHD {
    uint64_t start, l, m, a, x, seed, count;
    MAPAD_HD static bool is_prime(uint64_t n) {
        if (n <= 1) return false;
        if (n <= 3) return true;
        if (n % 2 == 0 || n % 3 == 0) return false;
        for (uint64_t i = 5; i * i <= n; i += 6) if (n % i == 0 || n % (i + 2) == 0) return false;
        return true;
    }
    MAPAD_HD static uint64_t next_prime(uint64_t n) {
        uint64_t p = n + 1;
        if (p <= 2) return 2;
        if (p % 2 == 0) p += 1;
        while (!is_prime(p)) p += 2;
        return p;
    }
    MAPAD_HD static bool pow_mod(uint64_t base, uint64_t e, uint64_t mod, uint64_t& out) {  // checked_pow_mod (prrange.rs:170-184)
        if (mod == 1) { out = 0; return true; }
        if (mod - 1 > 0xFFFFFFFFull) return false;  // (mod - 1)^2 would overflow u64
        uint64_t r = 1;
        base %= mod;
        while (e > 0) {
            if (e & 1) r = (r * base) % mod;
            e >>= 1;
            base = (base * base) % mod;
        }
        out = r;
        return true;
    }
    MAPAD_HD static bool is_primitive_root(uint64_t a, uint64_t n, bool& overflow) {  // :113-121 over the distinct prime factors of n - 1 (:126-165)
        const uint64_t phi = n - 1;
        uint64_t rest = phi, i = 2, step = 1, last = 0;
        for (;;) {
            // PrimeFactorIterator::next
            bool have = false;
            uint64_t f = 0;
            if (rest > 3) {
                while (!have && i * i <= rest) {
                    while (!have && rest > 1) {
                        while (rest % i == 0) {
                            if (i > last) { f = last = i; have = true; break; }
                            rest /= i;
                        }
                        if (have) break;
                        i += step;
                        step = 2;
                    }
                }
            }
            if (!have) return true;
            uint64_t r;
            if (!pow_mod(a, phi / f, n, r)) { overflow = true; return false; }
            if (r == 1) return false;
        }
    }
    MAPAD_HD static bool make(uint64_t start, uint64_t end, uint64_t seed, PrRangeHD& out) {  // try_new :43-71
        const uint64_t l = end > start ? end - start : 0;
        if (l == 0) return false;
        const uint64_t m = next_prime(l);
        uint64_t a = 2;
        for (;;) {
            bool overflow = false;
            if (is_primitive_root(a, m, overflow)) break;
            if (overflow) return false;
            a += 1;
        }
        const uint64_t s = (seed % l) > 1 ? (seed % l) : 1;
        out.start = start; out.l = l; out.m = m; out.a = a; out.x = s; out.seed = s; out.count = 0;
        return true;
    }
    MAPAD_HD bool next(uint64_t& out) {  // :19-35
        if (count == 0 && l == 1) { count = 1; out = start; return true; }
        for (;;) {
            const uint64_t prev = x;
            x = (a * x) % m;
            if (count > 0 && prev == seed) return false;
            if (prev <= l) { count += 1; out = prev - 1 + start; return true; }
        }
    }
};

// ---- what the records kernel sees of the index ------------------------------------------------------------------------------------
struct PostIndex {
    DevIndex ix;
    const uint64_t* sa_sample;
    const uint64_t* x_counts;      // per block: 'X' symbols before it; nullptr if the text has none
    uint64_t extra_row[2], extra_val[2];
    uint32_t sa_shift;
    const uint64_t* contig_start;  // [n_contigs] ascending, non-overlapping (src/index/mod.rs:30-35)
    const uint64_t* contig_end;    // inclusive
    uint32_t n_contigs;
};
// SampledSuffixArray::get by one thread
MAPAD_HD bool sa_get_hd(const PostIndex& Q, uint64_t row, uint64_t& out) {
    if (row >= Q.ix.n) return false;
    uint64_t pos = row, offset = 0;
    const uint64_t mask = (1ull << Q.sa_shift) - 1;
    for (;;) {
        if ((pos & mask) == 0) { out = Q.sa_sample[pos >> Q.sa_shift] + offset; return true; }
        const int code = bwt_code(Q.ix, pos);
        if (code == 0) { out = (pos == Q.extra_row[0] ? Q.extra_val[0] : Q.extra_val[1]) + offset; return true; }
        if (code >= 4) pos = Q.ix.less[code - 3] + occ_scalar(Q.ix, pos - 1, code - 4);
        else pos = Q.ix.less[5] + occ_x_scalar(Q.ix, Q.x_counts, pos - 1);  // 'X' (rank 5)
        offset += 1;
    }
}
// first contig with start <= p && p + len - 1 <= end; contigs tile the text in order, so it is the last one that starts at or before p
MAPAD_HD bool contig_of_hd(const PostIndex& Q, uint64_t p, uint64_t len, uint32_t& tid, uint64_t& rel) {
    uint32_t lo = 0, hi = Q.n_contigs;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (Q.contig_start[mid] <= p) lo = mid + 1; else hi = mid; }
    if (lo == 0) return false;
    const uint32_t c = lo - 1;
    if (p + len - 1 > Q.contig_end[c]) return false;
    tid = c; rel = p - Q.contig_start[c];
    return true;
}

// ---- per-read output of the records kernel ----------------------------------------------------------------------------------------
struct CoordOut { uint32_t hit; int32_t tid; uint64_t rel, abs; uint32_t backward, pad; };  // 32 bytes
struct CoordRec {
    uint32_t mapped, best;        // best = index (within the read's hits) of the reported alignment
    CoordOut first;               // its coordinate
    uint64_t x0, x1;              // :430-431, :497-508
    uint32_t n_xa, n_order;       // XA candidates; hits that remain after the reported one was popped
    CoordOut xa[2];
    uint8_t order[kMaxHits];      // those remaining hits, ascending score (into_sorted_vec order): MAPQ, XS are computed from them on the host
    uint32_t error;               // 1: PrRange could not be set up (the reference returns an error there)
};

MAPAD_HD uint64_t effective_len_hd(const uint32_t* ops, uint32_t n) { uint64_t k = 0; for (uint32_t i = 0; i < n; ++i) k += (ops[i] >> 24) != OP_INS; return k; }
MAPAD_HD bool cross_check_hd(const HitRec& a, const HitRec& b) { return a.size == b.size && (a.lower == b.lower || a.lower_rev == b.lower_rev); }
MAPAD_HD uint32_t seed_for_hd(uint64_t seed, uint64_t read_idx, uint32_t call) {
    uint64_t z = seed + (read_idx + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)call * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}

// interval2coordinate, eager up to max_items; returns false if the permutation could not be set up
MAPAD_HD bool coords_hd(const PostIndex& Q, const HitRec& h, uint32_t hit_idx, const uint32_t* ops, uint32_t seed, CoordOut* out, uint32_t& n_out, uint64_t* skipped,
                        uint32_t max_items) {
    const uint64_t strand_len = Q.ix.n / 2, eff = effective_len_hd(ops + h.ops_off, h.n_ops);
    PrRangeHD pr;
    n_out = 0;
    if (!PrRangeHD::make(h.lower, h.lower + h.size, seed, pr)) return false;
    uint64_t row, i = 0;
    while (n_out < max_items && pr.next(row)) {
        uint64_t p;
        if (sa_get_hd(Q, row, p)) {
            uint32_t backward = 0;
            if (p >= strand_len) { p = Q.ix.n - p - eff - 1; backward = 1; }
            uint32_t tid; uint64_t rel;
            if (contig_of_hd(Q, p, eff, tid, rel)) { out[n_out] = CoordOut{hit_idx, (int32_t)tid, rel, p, backward, 0}; if (skipped) skipped[n_out] = i; n_out += 1; }
        }
        ++i;
    }
    return true;
}

// The read's part of intervals_to_bam that needs the index: hits = the read's hit records (BinaryHeap array order), ops = the batch's op array.
MAPAD_HD void record_coords(const PostIndex& Q, const HitRec* hits, uint32_t n, const uint32_t* ops, uint64_t seed, uint64_t read_idx, CoordRec& rec) {
    rec.mapped = 0; rec.best = 0; rec.x0 = 0; rec.x1 = 0; rec.n_xa = 0; rec.n_order = 0; rec.error = 0;
    rec.first = CoordOut{0, -1, 0, 0, 0, 0}; rec.xa[0] = rec.first; rec.xa[1] = rec.first;
    if (n > (uint32_t)kMaxHits) n = (uint32_t)kMaxHits;
    // into_sorted_vec on hit indices: repeated swap(0, end) + sift_down_range choosing the right child on ties
    uint8_t d[kMaxHits];
    for (uint32_t i = 0; i < n; ++i) d[i] = (uint8_t)i;
    {
        uint32_t end = n;
        while (end > 1) {
            end -= 1;
            const uint8_t t0 = d[0]; d[0] = d[end]; d[end] = t0;
            uint32_t pos = 0, child = 1;
            const uint8_t elt = d[0];
            bool placed = false;
            while (child + 1 < end) {
                if (hits[d[child]].score <= hits[d[child + 1]].score) child += 1;
                if (hits[elt].score >= hits[d[child]].score) { placed = true; break; }
                d[pos] = d[child]; pos = child; child = 2 * pos + 1;
            }
            if (!placed && child + 1 == end && hits[elt].score < hits[d[child]].score) { d[pos] = d[child]; pos = child; }
            d[pos] = elt;
        }
    }
    uint32_t n_left = n, call = 0;
    while (n_left > 0) {  // :421 pop the best until one yields a coordinate
        const uint32_t bi = d[--n_left];
        const HitRec& best = hits[bi];
        CoordOut bc[3];
        uint64_t skipped[3];
        uint32_t n_bc = 0;
        if (!coords_hd(Q, best, bi, ops, seed_for_hd(seed, read_idx, call++), bc, n_bc, skipped, 3)) { rec.error = 1; return; }
        if (n_bc == 0) continue;  // :541-543
        rec.mapped = 1; rec.best = bi; rec.first = bc[0];
        rec.x0 = best.size - skipped[0];
        uint32_t n_xa = 0;  // XA (:436-491): the other coordinates of the best hit, then the suboptimal hits in descending score, take(2)
        for (uint32_t i = 1; i < n_bc && n_xa < 2; ++i) rec.xa[n_xa++] = bc[i];
        for (uint32_t k = n_left; k-- > 0 && n_xa < 2;) {
            const HitRec& sub = hits[d[k]];
            if (cross_check_hd(best, sub)) continue;
            CoordOut sc[2];
            uint32_t n_sc = 0;
            if (!coords_hd(Q, sub, d[k], ops, seed_for_hd(seed, read_idx, call++), sc, n_sc, nullptr, 2 - n_xa)) continue;
            for (uint32_t i = 0; i < n_sc && n_xa < 2; ++i) rec.xa[n_xa++] = sc[i];
        }
        rec.n_xa = n_xa;
        uint64_t x1 = 0;
        for (uint32_t k = 0; k < n_left; ++k) if (!cross_check_hd(best, hits[d[k]])) x1 += hits[d[k]].size;
        rec.x1 = x1;
        rec.n_order = n_left;
        for (uint32_t k = 0; k < n_left; ++k) rec.order[k] = d[k];
        return;
    }
}

}  // namespace mapad
