// host_postproc.hpp — hit intervals -> alignment record fields (the dispatcher side of the reference's worker/dispatcher
// split: src/distributed/dispatcher.rs:341-379 calls the same intervals_to_bam as run_inner).
//
// Mirrors intervals_to_bam (src/map/mapping.rs:402-567), interval2coordinate (:590-649), interval_cross_check (:651-653),
// estimate_mapping_quality (:658-718), the flag / SEQ / QUAL rules of create_bam_record (:748-819), PrRange
// (src/map/prrange.rs) and EditOperationsTrack::{to_bam_fields, effective_len, read_len} (src/map/record.rs:269-449).
// BAM byte encoding is done by the caller (bam_writer.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>
#include <cstdio>
#include <cstring>
#include <string>

#include "host_cpus.hpp"
#include <vector>

#include "../../include/mapad_amd.h"
#include "host_index.hpp"
#include "host_models.hpp"
#include "postproc_core.hpp"
#include "text_core.hpp"

namespace mapad {
namespace host {

// ---- PrRange: Lehmer-LCG lazy permutation of an SA interval (src/map/prrange.rs): the one restatement, shared with the records kernel (postproc_core.hpp) ----
using PrRange = PrRangeHD;

// ---- edit track helpers (record.rs:269-449) -----------------------------------------------------------------------------------
struct Track {
    const uint32_t* ops;
    uint32_t n;
    static uint32_t kind(uint32_t op) { return op >> 24; }
    static uint8_t base(uint32_t op) { return (uint8_t)(op >> 16); }
    uint64_t effective_len() const { uint64_t k = 0; for (uint32_t i = 0; i < n; ++i) k += kind(ops[i]) != OP_INS; return k; }
    uint64_t read_len() const { uint64_t k = 0; for (uint32_t i = 0; i < n; ++i) k += kind(ops[i]) != OP_DEL; return k; }
};
struct BamFields { std::string cigar, md; int32_t nm = 0; };

inline BamFields to_bam_fields(const Track& t, bool backward, uint64_t absolute_pos, const Index& ix) {
    BamFields out;
    uint32_t run_kind = 0xFF, run_len = 0, matches = 0;
    bool prev_del = false;  // last_edit_operation is a Deletion <=> the current CIGAR run is a deletion run
    auto cig = [](uint32_t k) -> char { return k == OP_INS ? 'I' : k == OP_DEL ? 'D' : 'M'; };
    auto flush = [&]() { if (run_len) { out.cigar += std::to_string(run_len); out.cigar.push_back(cig(run_kind)); } };
    for (uint32_t i = 0; i < t.n; ++i) {  // :301 `i` counts every operation, insertions included
        const uint32_t op = backward ? t.ops[t.n - 1 - i] : t.ops[i];
        uint32_t k = Track::kind(op);
        uint8_t b = Track::base(op), o;
        if (k != OP_INS && ix.original_symbol(absolute_pos + i, o)) { b = o; if (k == OP_MATCH) k = OP_MISMATCH; }  // :302-320
        if (k != OP_MATCH) out.nm += 1;
        const uint8_t shown = backward ? complement(b) : b;
        switch (k) {  // add_md_edit_operation :391-430
            case OP_MATCH: matches += 1; break;
            case OP_MISMATCH: out.md += std::to_string(matches); out.md.push_back((char)shown); matches = 0; break;
            case OP_INS: break;
            default:
                if (prev_del) out.md.push_back((char)shown);
                else { out.md += std::to_string(matches); out.md.push_back('^'); out.md.push_back((char)shown); }
                matches = 0;
        }
        const char c = cig(k);
        if (run_len && c == cig(run_kind)) run_len += 1;
        else { flush(); run_kind = k; run_len = 1; }
        prev_del = cig(run_kind) == 'D';
    }
    flush();
    out.md += std::to_string(matches);
    return out;
}

// ---- Rust BinaryHeap::into_sorted_vec on hit indices ----------------------------------------------------------------
inline std::vector<uint32_t> sorted_ascending(const mapad_hit_t* hits, uint32_t n) {
    std::vector<uint32_t> d(n);
    for (uint32_t i = 0; i < n; ++i) d[i] = i;
    auto score = [&](uint32_t i) { return hits[i].alignment_score; };
    uint32_t end = n;
    while (end > 1) {
        end -= 1;
        std::swap(d[0], d[end]);
        uint32_t pos = 0, child = 1;  // sift_down_range(0, end)
        const uint32_t elt = d[0];
        bool placed = false;
        while (child + 1 < end) {
            if (score(d[child]) <= score(d[child + 1])) child += 1;
            if (score(elt) >= score(d[child])) { placed = true; break; }
            d[pos] = d[child]; pos = child; child = 2 * pos + 1;
        }
        if (!placed && child + 1 == end && score(elt) < score(d[child])) { d[pos] = d[child]; pos = child; }
        d[pos] = elt;
    }
    return d;
}

struct Coord { uint32_t tid; uint64_t rel, abs; bool backward; uint64_t num_skipped; };

// interval2coordinate (:590-649), eager; `seed` stands in for rng.next_u32()
// Pre-computed suffix-array values (row -> position), e.g. from the device's locate kernel; rows that are absent fall back to
// Index::sa_get().  Sorted by row.
struct SaCache {
    std::vector<std::pair<uint64_t, uint64_t>> v;
    bool get(uint64_t row, uint64_t& out) const {
        auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(row, (uint64_t)0));
        if (it == v.end() || it->first != row) return false;
        out = it->second;
        return true;
    }
};

inline bool interval2coordinate(const Index& ix, const mapad_hit_t& h, const Track& t, uint32_t seed, std::vector<Coord>& out, size_t max_items,
                                const SaCache* cache = nullptr) {
    const uint64_t strand_len = ix.n / 2, eff = t.effective_len();
    PrRange pr;
    if (!PrRange::make(h.lower, h.lower + h.size, seed, pr)) return false;
    uint64_t row, i = 0;
    while (out.size() < max_items && pr.next(row)) {
        uint64_t p;
        if ((cache && cache->get(row, p)) || ix.sa_get(row, p)) {
            bool backward = false;
            if (p >= strand_len) { p = ix.n - p - eff - 1; backward = true; }
            uint32_t tid; uint64_t rel;
            if (ix.contig_of(p, eff, tid, rel)) out.push_back({tid, rel, p, backward, i});
        }
        ++i;
    }
    return true;
}
inline bool cross_check(const mapad_hit_t& a, const mapad_hit_t& b) { return a.size == b.size && (a.lower == b.lower || a.lower_rev == b.lower_rev); }

inline uint8_t f32_to_u8(float x) { return std::isnan(x) || x <= 0.0f ? 0 : x >= 255.0f ? 255 : (uint8_t)x; }  // `as u8` saturates

// estimate_mapping_quality (:658-718); others = the hits that remain after the best was popped
inline uint8_t mapping_quality(const mapad_params_t& prm, const mapad_hit_t& best, uint64_t best_size, uint64_t best_read_len, const mapad_hit_t* hits,
                               const std::vector<uint32_t>& others) {
    float p;
    const float prob_best = std::exp2(best.alignment_score);
    if (best_size > 1) p = 1.0f / (float)best_size;
    else {
        float acc = 0.0f;
        for (uint32_t i : others) {
            if (cross_check(best, hits[i])) continue;
            acc = std::fmaf(std::exp2(hits[i].alignment_score), (float)hits[i].size, acc);
        }
        p = prob_best / (prob_best + acc);
    }
    if (p < 0.0f) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    const float phred = -10.0f * std::log10(1.0f - p);
    const uint8_t mq = f32_to_u8(std::round(phred < 37.0f ? phred : 37.0f));
    if (mq == 37) {
        float frac = mb_remaining_frac(prm, best.alignment_score, best_read_len);
        if (!(frac < 1.0f)) frac = 1.0f;  // f32::min(frac, 1.0)
        return f32_to_u8(std::round(std::fmaf(17.0f, frac, 20.0f)));
    }
    return mq;
}

inline uint32_t seed_for(uint64_t seed, uint64_t read_idx, uint32_t call) {
    uint64_t z = seed + (read_idx + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)call * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}

struct RecordsOwner {
    mapad_records_t pub{};
    std::vector<mapad_record_t> recs;
    std::string text;
};

inline mapad_records_t* hits_to_records(const Index& ix, const mapad_params_t& prm, const mapad_batch_result_t& res, const uint8_t* seqs, const uint8_t* quals,
                                        const uint64_t* offsets, const uint16_t* in_flags, uint64_t seed, const SaCache* cache = nullptr) {
    (void)seqs; (void)quals; (void)offsets;
    auto own = std::unique_ptr<RecordsOwner>(new RecordsOwner());
    own->recs.resize(res.n_reads);
    // Reads are independent (the reference runs this stage in a rayon map, mapping.rs:153-271): contiguous ranges go to host threads,
    // each with its own text blob; blobs are concatenated in read order afterwards, so the output does not depend on the thread count.
    unsigned n_threads = 1;
    if (const char* e = std::getenv("MAPAD_POSTPROC_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(e));
    else n_threads = std::min<unsigned>(cpu_share(), 64u);
    n_threads = (unsigned)std::min<uint64_t>(n_threads, std::max<uint64_t>(1, res.n_reads / 2048));
    std::vector<std::string> texts(n_threads);
    std::vector<std::exception_ptr> errors(n_threads);
    auto work = [&](unsigned t) {
      try {
        const uint64_t r0 = res.n_reads * t / n_threads, r1 = res.n_reads * (t + 1) / n_threads;
        std::string& text = texts[t];
        auto put = [&](const std::string& s, uint32_t& off, uint32_t& len) { off = (uint32_t)text.size(); len = (uint32_t)s.size(); text += s; };
        for (uint64_t r = r0; r < r1; ++r) {
        mapad_record_t rec{};
        uint16_t flags = in_flags ? in_flags[r] : 0;
        flags &= (uint16_t)~(0x8 | 0x20 | 0x2 | 0x100 | 0x800);  // :750-755
        const mapad_hit_t* hits = res.hits + res.hit_begin[r];
        const uint32_t n = (uint32_t)(res.hit_begin[r + 1] - res.hit_begin[r]);
        std::vector<uint32_t> order = sorted_ascending(hits, n);  // :419
        uint32_t call = 0;
        bool mapped = false;
        while (!order.empty()) {  // :421
            const uint32_t bi = order.back();
            order.pop_back();
            const mapad_hit_t& best = hits[bi];
            const Track bt{res.ops + best.ops_offset, best.n_ops};
            std::vector<Coord> bc;
            if (!interval2coordinate(ix, best, bt, seed_for(seed, r, call++), bc, 3, cache)) throw std::runtime_error("Could not enumerate possible reference positions");
            if (bc.empty()) continue;  // :541-543
            const Coord first = bc.front();
            const uint64_t upd = best.size - first.num_skipped;  // :430-431
            // XA (:436-491): lazily chained, take(2)
            std::string xa;
            int n_xa = 0;
            auto emit = [&](const Coord& c, const mapad_hit_t& h) {
                const Track t{res.ops + h.ops_offset, h.n_ops};
                const BamFields bf = to_bam_fields(t, c.backward, c.abs, ix);
                char buf[64];
                std::snprintf(buf, sizeof buf, "%.2f", (double)h.alignment_score);
                xa += ix.contigs[c.tid].name + "," + (c.backward ? "-" : "+") + std::to_string(c.rel + 1) + "," + bf.cigar + "," + bf.md + "," + std::to_string(bf.nm) +
                      "," + std::to_string(h.size) + "," + buf + ";";
                n_xa += 1;
            };
            for (size_t i = 1; i < bc.size() && n_xa < 2; ++i) emit(bc[i], best);
            for (size_t k = order.size(); k-- > 0 && n_xa < 2;) {
                const mapad_hit_t& sub = hits[order[k]];
                if (cross_check(best, sub)) continue;
                std::vector<Coord> sc;
                const Track st{res.ops + sub.ops_offset, sub.n_ops};
                if (!interval2coordinate(ix, sub, st, seed_for(seed, r, call++), sc, (size_t)(2 - n_xa), cache)) continue;
                for (auto& c : sc) { if (n_xa >= 2) break; emit(c, sub); }
            }
            uint64_t x1 = 0;
            for (uint32_t i : order) if (!cross_check(best, hits[i])) x1 += hits[i].size;  // :497-508
            rec.x0 = upd > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)upd;
            rec.x1 = x1 > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)x1;
            rec.xs_score = order.empty() ? 0.0f : hits[order.back()].alignment_score;  // :510-513
            rec.has_xs = rec.x1 > 0;                                                    // :895
            rec.xt = upd == 0 ? 'N' : upd == 1 ? 'U' : 'R';
            rec.mapq = mapping_quality(prm, best, upd, bt.read_len(), hits, order);
            const BamFields bf = to_bam_fields(bt, first.backward, first.abs, ix);
            put(bf.cigar, rec.cigar_off, rec.cigar_len);
            put(bf.md, rec.md_off, rec.md_len);
            put(xa, rec.xa_off, rec.xa_len);
            rec.nm = bf.nm; rec.as_score = best.alignment_score;
            rec.mapped = 1; rec.reverse = first.backward; rec.tid = (int32_t)first.tid; rec.pos = (int64_t)first.rel;
            flags &= (uint16_t)~0x4;
            if (first.backward) flags |= 0x10; else flags &= (uint16_t)~0x10;
            mapped = true;
            break;
        }
        if (!mapped) {  // :553-566, :765-776
            flags |= 0x4; flags &= (uint16_t)~0x10; flags &= (uint16_t)~0x2;
            rec.mapq = 0; rec.tid = -1; rec.pos = -1;
        }
        rec.flags = flags;
        own->recs[r] = rec;
        }
      } catch (...) { errors[t] = std::current_exception(); }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (auto& e : errors) if (e) std::rethrow_exception(e);
    uint64_t total = 0;
    for (auto& t : texts) total += t.size();
    if (total > 0xFFFFFFFFull) throw std::runtime_error("record text of one batch exceeds 4 GiB");
    own->text.reserve(total);
    for (unsigned t = 0; t < n_threads; ++t) {
        const uint32_t base = (uint32_t)own->text.size();
        if (base) for (uint64_t r = res.n_reads * t / n_threads; r < res.n_reads * (t + 1) / n_threads; ++r) {
            mapad_record_t& rec = own->recs[r];
            if (rec.mapped) { rec.cigar_off += base; rec.md_off += base; rec.xa_off += base; }
        }
        own->text += texts[t];
    }
    own->pub.n = res.n_reads; own->pub.recs = own->recs.data(); own->pub.text = own->text.c_str(); own->pub.text_len = own->text.size();
    return &own.release()->pub;
}
// The text half of intervals_to_bam for coordinates that were computed on the device (postproc_core.hpp: record_coords): flags, CIGAR / MD /
// XA strings, XS / XT and the mapping quality (libm: exp2f, log10f — on the host like every other transcendental of this design).
inline mapad_records_t* records_from_coords(const Index& ix, const mapad_params_t& prm, const mapad_batch_result_t& res, const uint16_t* in_flags, const CoordRec* coords) {
    auto own = std::unique_ptr<RecordsOwner>(new RecordsOwner());
    own->recs.resize(res.n_reads);
    unsigned n_threads = 1;
    if (const char* e = std::getenv("MAPAD_POSTPROC_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(e));
    else n_threads = std::min<unsigned>(cpu_share(), 64u);
    n_threads = (unsigned)std::min<uint64_t>(n_threads, std::max<uint64_t>(1, res.n_reads / 2048));
    std::vector<std::string> texts(n_threads);
    std::vector<std::exception_ptr> errors(n_threads);
    auto work = [&](unsigned t) {
      try {
        const uint64_t r0 = res.n_reads * t / n_threads, r1 = res.n_reads * (t + 1) / n_threads;
        std::string& text = texts[t];
        auto put = [&](const std::string& s, uint32_t& off, uint32_t& len) { off = (uint32_t)text.size(); len = (uint32_t)s.size(); text += s; };
        std::vector<uint32_t> order;
        for (uint64_t r = r0; r < r1; ++r) {
            mapad_record_t rec{};
            const CoordRec& cr = coords[r];
            if (cr.error) throw std::runtime_error("Could not enumerate possible reference positions");
            uint16_t flags = in_flags ? in_flags[r] : 0;
            flags &= (uint16_t)~(0x8 | 0x20 | 0x2 | 0x100 | 0x800);  // :750-755
            const mapad_hit_t* hits = res.hits + res.hit_begin[r];
            if (cr.mapped) {
                const mapad_hit_t& best = hits[cr.best];
                const Track bt{res.ops + best.ops_offset, best.n_ops};
                std::string xa;
                for (uint32_t k = 0; k < cr.n_xa; ++k) {  // :436-491
                    const CoordOut& c = cr.xa[k];
                    const mapad_hit_t& h = hits[c.hit];
                    const Track tk{res.ops + h.ops_offset, h.n_ops};
                    const BamFields bf = to_bam_fields(tk, c.backward != 0, c.abs, ix);
                    char buf[64];
                    std::snprintf(buf, sizeof buf, "%.2f", (double)h.alignment_score);
                    xa += ix.contigs[(size_t)c.tid].name + "," + (c.backward ? "-" : "+") + std::to_string(c.rel + 1) + "," + bf.cigar + "," + bf.md + "," + std::to_string(bf.nm) +
                          "," + std::to_string(h.size) + "," + buf + ";";
                }
                order.assign(cr.order, cr.order + cr.n_order);
                rec.x0 = cr.x0 > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)cr.x0;
                rec.x1 = cr.x1 > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)cr.x1;
                rec.xs_score = order.empty() ? 0.0f : hits[order.back()].alignment_score;  // :510-513
                rec.has_xs = rec.x1 > 0;                                                    // :895
                rec.xt = cr.x0 == 0 ? 'N' : cr.x0 == 1 ? 'U' : 'R';
                rec.mapq = mapping_quality(prm, best, cr.x0, bt.read_len(), hits, order);
                const BamFields bf = to_bam_fields(bt, cr.first.backward != 0, cr.first.abs, ix);
                put(bf.cigar, rec.cigar_off, rec.cigar_len);
                put(bf.md, rec.md_off, rec.md_len);
                put(xa, rec.xa_off, rec.xa_len);
                rec.nm = bf.nm; rec.as_score = best.alignment_score;
                rec.mapped = 1; rec.reverse = cr.first.backward != 0; rec.tid = cr.first.tid; rec.pos = (int64_t)cr.first.rel;
                flags &= (uint16_t)~0x4;
                if (cr.first.backward) flags |= 0x10; else flags &= (uint16_t)~0x10;
            } else {  // :553-566, :765-776
                flags |= 0x4; flags &= (uint16_t)~0x10; flags &= (uint16_t)~0x2;
                rec.mapq = 0; rec.tid = -1; rec.pos = -1;
            }
            rec.flags = flags;
            own->recs[r] = rec;
        }
      } catch (...) { errors[t] = std::current_exception(); }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (auto& e : errors) if (e) std::rethrow_exception(e);
    uint64_t total = 0;
    for (auto& t : texts) total += t.size();
    if (total > 0xFFFFFFFFull) throw std::runtime_error("record text of one batch exceeds 4 GiB");
    own->text.reserve(total);
    for (unsigned t = 0; t < n_threads; ++t) {
        const uint32_t base = (uint32_t)own->text.size();
        if (base) for (uint64_t r = res.n_reads * t / n_threads; r < res.n_reads * (t + 1) / n_threads; ++r) {
            mapad_record_t& rec = own->recs[r];
            if (rec.mapped) { rec.cigar_off += base; rec.md_off += base; rec.xa_off += base; }
        }
        own->text += texts[t];
    }
    own->pub.n = res.n_reads; own->pub.recs = own->recs.data(); own->pub.text = own->text.c_str(); own->pub.text_len = own->text.size();
    return &own.release()->pub;
}
// What is left for the host when the text was built on the device (text_core.hpp: DevRecord, text pool, pair pool): flags (:750-776) and the
// mapping quality (estimate_mapping_quality, :658-718) from the pairs (score, (float)size) of the hits that count — libm, like every transcendental
// of this design.
inline uint8_t mapping_quality_from_pairs(const mapad_params_t& prm, float best_score, float best_size_f, uint64_t best_read_len, const float* pairs, uint32_t n_pairs) {
    float p;
    const float prob_best = std::exp2(best_score);
    if (best_size_f > 1.0f) p = 1.0f / best_size_f;
    else {
        float acc = 0.0f;
        for (uint32_t i = 0; i < n_pairs; ++i) acc = std::fmaf(std::exp2(pairs[2 * i]), pairs[2 * i + 1], acc);
        p = prob_best / (prob_best + acc);
    }
    if (p < 0.0f) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    const float phred = -10.0f * std::log10(1.0f - p);
    const uint8_t mq = f32_to_u8(std::round(phred < 37.0f ? phred : 37.0f));
    if (mq == 37) {
        float frac = mb_remaining_frac(prm, best_score, best_read_len);
        if (!(frac < 1.0f)) frac = 1.0f;
        return f32_to_u8(std::round(std::fmaf(17.0f, frac, 20.0f)));
    }
    return mq;
}
inline mapad_records_t* records_from_device_text(const mapad_params_t& prm, uint64_t n_reads, const uint16_t* in_flags, const DevRecord* dev, const char* text, size_t text_len,
                                                 const float* pairs) {
    auto own = std::unique_ptr<RecordsOwner>(new RecordsOwner());
    own->recs.resize(n_reads);
    own->text.assign(text, text_len);
    unsigned n_threads = (unsigned)std::min<uint64_t>(std::min<unsigned>(cpu_share(), 16u), std::max<uint64_t>(1, n_reads / 16384));
    std::vector<int> bad(n_threads, 0);
    auto work = [&](unsigned t) {
        for (uint64_t r = n_reads * t / n_threads; r < n_reads * (t + 1) / n_threads; ++r) {
            const DevRecord& d = dev[r];
            if (d.error) { bad[t] = (int)d.error; continue; }
            mapad_record_t rec{};
            uint16_t flags = in_flags ? in_flags[r] : 0;
            flags &= (uint16_t)~(0x8 | 0x20 | 0x2 | 0x100 | 0x800);  // :750-755
            if (d.mapped) {
                rec.mapped = 1; rec.reverse = (uint8_t)d.reverse; rec.tid = d.tid; rec.pos = d.pos;
                rec.as_score = d.as_score; rec.xs_score = d.xs_score; rec.nm = d.nm; rec.x0 = d.x0; rec.x1 = d.x1; rec.has_xs = (uint8_t)d.has_xs; rec.xt = (char)d.xt;
                rec.cigar_off = d.text_off; rec.cigar_len = d.cigar_len; rec.md_off = d.text_off + d.cigar_len; rec.md_len = d.md_len;
                rec.xa_off = d.text_off + d.cigar_len + d.md_len; rec.xa_len = d.xa_len;
                rec.mapq = mapping_quality_from_pairs(prm, d.as_score, d.best_size_f, d.read_len, pairs + 2 * (size_t)d.mq_off, d.mq_n);
                flags &= (uint16_t)~0x4;
                if (d.reverse) flags |= 0x10; else flags &= (uint16_t)~0x10;
            } else {  // :553-566, :765-776
                flags |= 0x4; flags &= (uint16_t)~0x10; flags &= (uint16_t)~0x2;
                rec.mapq = 0; rec.tid = -1; rec.pos = -1;
            }
            rec.flags = flags;
            own->recs[r] = rec;
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (int b : bad) if (b) throw std::runtime_error(b == 1 ? "Could not enumerate possible reference positions" : "record text pools too small");
    own->pub.n = n_reads; own->pub.recs = own->recs.data(); own->pub.text = own->text.c_str(); own->pub.text_len = own->text.size();
    return &own.release()->pub;
}
inline void free_records(mapad_records_t* r) { if (r) delete reinterpret_cast<RecordsOwner*>(r); }

}  // namespace host
}  // namespace mapad
