// text_core.hpp — the text half of intervals_to_bam as code that runs on the device: CIGAR / MD / NM of the reported alignment
// (EditOperationsTrack::to_bam_fields, src/map/record.rs:269-449), the XA entries (mapping.rs:436-491) and the inputs of the mapping quality.
// Integer work only: the mapping quality itself needs exp2f / log10f and is finished on the host from what this file leaves in the record
// (estimate_mapping_quality, mapping.rs:658-718) — the same rule as for the score tables.  host_postproc.hpp holds the host restatement of the
// same logic (to_bam_fields, records_from_coords): the parity reference for this file (tests/test_gpu_locate.py).
#pragma once
#include "postproc_core.hpp"

namespace mapad {

struct TextIndex {
    const uint64_t* os_pos;   // OriginalSymbols (src/index/mod.rs): text positions whose base was an ambiguity code, ascending
    const uint8_t* os_sym;    // ... and the code
    uint64_t n_os;
    const uint32_t* name_off; // contig names, concatenated: name i = names[name_off[i] .. name_off[i + 1])
    const char* names;
};

// counts, or writes behind `p`
struct TextSink {
    char* p;
    uint32_t n;
    bool write;
    MAPAD_HD void put(char c) { if (write) p[n] = c; n += 1; }
    MAPAD_HD void put_u64(uint64_t v) {
        char buf[20];
        int k = 0;
        do { buf[k++] = (char)('0' + (int)(v % 10)); v /= 10; } while (v);
        while (k) put(buf[--k]);
    }
    MAPAD_HD void put_f2(float x) {  // "{:.2}" / "%.2f": the exact value, rounded half to even at the second decimal
        double v = (double)x;
        const bool neg = __builtin_signbit(v);
        if (neg) v = -v;
        const uint64_t r = (uint64_t)__builtin_rint(v * 100.0);  // the product is exact: a 24-bit significand times 25 times 4
        if (neg) put('-');
        put_u64(r / 100); put('.'); put((char)('0' + (int)((r / 10) % 10))); put((char)('0' + (int)(r % 10)));
    }
};

MAPAD_HD uint8_t complement_hd(uint8_t a) {  // bio::alphabets::dna::complement
    switch (a) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k';
        case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
        default: return a;
    }
}

// to_bam_fields (record.rs:282-449): CIGAR into `cigar`, MD into `md` (either may be null), returns NM.  `i` counts every operation, insertions
// included, when the original symbol of a position is looked up (record.rs:301-320).
MAPAD_HD int32_t bam_fields_hd(const TextIndex& T, const uint32_t* ops, uint32_t n, bool backward, uint64_t abs_pos, TextSink* cigar, TextSink* md) {
    int32_t nm = 0;
    uint32_t run_kind = 0xFF, run_len = 0, matches = 0;
    bool prev_del = false;
    auto cig = [](uint32_t k) -> char { return k == OP_INS ? 'I' : k == OP_DEL ? 'D' : 'M'; };
    // first original symbol at or behind the alignment (binary search), then a merge: the positions asked for only grow
    uint64_t j = 0;
    if (T.n_os) {
        uint64_t lo = 0, hi = T.n_os;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (T.os_pos[mid] < abs_pos) lo = mid + 1; else hi = mid; }
        j = lo;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t op = backward ? ops[n - 1 - i] : ops[i];
        uint32_t k = op >> 24;
        uint8_t b = (uint8_t)(op >> 16);
        if (k != OP_INS && j < T.n_os) {
            const uint64_t at = abs_pos + i;
            while (j < T.n_os && T.os_pos[j] < at) ++j;
            if (j < T.n_os && T.os_pos[j] == at) { b = T.os_sym[j]; if (k == OP_MATCH) k = OP_MISMATCH; }
        }
        if (k != OP_MATCH) nm += 1;
        const char shown = (char)(backward ? complement_hd(b) : b);
        if (md) {  // add_md_edit_operation (record.rs:391-430)
            if (k == OP_MATCH) matches += 1;
            else if (k == OP_MISMATCH) { md->put_u64(matches); md->put(shown); matches = 0; }
            else if (k == OP_DEL) {
                if (prev_del) md->put(shown);
                else { md->put_u64(matches); md->put('^'); md->put(shown); }
                matches = 0;
            }
        }
        const char c = cig(k);
        if (run_len && c == cig(run_kind)) run_len += 1;
        else {
            if (cigar && run_len) { cigar->put_u64(run_len); cigar->put(cig(run_kind)); }
            run_kind = k; run_len = 1;
        }
        prev_del = cig(run_kind) == 'D';
    }
    if (cigar && run_len) { cigar->put_u64(run_len); cigar->put(cig(run_kind)); }
    if (md) md->put_u64(matches);
    return nm;
}

// One XA entry (mapping.rs:462-487): name,±pos,CIGAR,MD,NM,size,score;
MAPAD_HD void xa_entry_hd(const TextIndex& T, const CoordOut& c, const HitRec& h, const uint32_t* ops, TextSink& s) {
    for (uint32_t k = T.name_off[c.tid]; k < T.name_off[c.tid + 1]; ++k) s.put(T.names[k]);
    s.put(','); s.put(c.backward ? '-' : '+'); s.put_u64(c.rel + 1); s.put(',');
    bam_fields_hd(T, ops + h.ops_off, h.n_ops, c.backward != 0, c.abs, &s, nullptr);
    s.put(',');
    const int32_t nm = bam_fields_hd(T, ops + h.ops_off, h.n_ops, c.backward != 0, c.abs, nullptr, &s);
    s.put(','); s.put_u64((uint64_t)nm); s.put(','); s.put_u64(h.size); s.put(','); s.put_f2(h.score); s.put(';');
}

// What leaves the device per read (80 bytes) besides its text: the record fields that are integers, and what the host needs for the mapping quality.
struct DevRecord {
    int64_t pos;               // 0-based position on the contig, -1 if unmapped
    int32_t tid;
    uint32_t mapped, reverse;
    float as_score, xs_score;
    int32_t nm, x0, x1;
    uint32_t has_xs, xt;
    uint32_t text_off, cigar_len, md_len, xa_len;  // [CIGAR][MD][XA] behind text_off in the batch's text pool
    float best_size_f;         // (float)x0 before the cap: p = 1 / best_size when the best interval has more than one row (mapping.rs:668-670)
    uint32_t read_len;         // EditOperationsTrack::read_len of the reported alignment (remaining_frac_of_repr_mm)
    uint32_t mq_off, mq_n;     // the other hits that count for the mapping quality: (score, (float)size) pairs in the batch's pair pool
    uint32_t error;
};
static_assert(sizeof(DevRecord) == 88, "device record layout");

// The text half of one read, `cr` from record_coords (postproc_core.hpp), in two steps so that a wavefront can claim the text of its 64 reads with
// one atomic add: record_text_sizes fills every field of the record but the offsets and says how many text bytes and pairs the read needs;
// record_text_write puts the bytes behind the offsets.
MAPAD_HD void record_text_sizes(const TextIndex& T, const HitRec* hits, const uint32_t* ops, const CoordRec& cr, DevRecord& out, uint32_t& text_bytes, uint32_t& n_pairs) {
    out.pos = -1; out.tid = -1; out.mapped = 0; out.reverse = 0; out.as_score = 0.0f; out.xs_score = 0.0f; out.nm = 0; out.x0 = 0; out.x1 = 0; out.has_xs = 0; out.xt = 0;
    out.text_off = 0; out.cigar_len = 0; out.md_len = 0; out.xa_len = 0; out.best_size_f = 0.0f; out.read_len = 0; out.mq_off = 0; out.mq_n = 0; out.error = cr.error;
    text_bytes = 0; n_pairs = 0;
    if (!cr.mapped || cr.error) return;
    const HitRec& best = hits[cr.best];
    const uint32_t* bops = ops + best.ops_off;
    TextSink c0{nullptr, 0, false}, m0{nullptr, 0, false}, x0{nullptr, 0, false};
    const int32_t nm = bam_fields_hd(T, bops, best.n_ops, cr.first.backward != 0, cr.first.abs, &c0, &m0);
    for (uint32_t k = 0; k < cr.n_xa; ++k) xa_entry_hd(T, cr.xa[k], hits[cr.xa[k].hit], ops, x0);
    for (uint32_t k = 0; k < cr.n_order; ++k) n_pairs += cross_check_hd(best, hits[cr.order[k]]) ? 0u : 1u;
    out.mapped = 1; out.reverse = cr.first.backward; out.tid = cr.first.tid; out.pos = (int64_t)cr.first.rel;
    out.as_score = best.score; out.nm = nm;
    out.x0 = cr.x0 > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)cr.x0;
    out.x1 = cr.x1 > 0x7FFFFFFFull ? 0x7FFFFFFF : (int32_t)cr.x1;
    out.xs_score = cr.n_order ? hits[cr.order[cr.n_order - 1]].score : 0.0f;  // :510-513
    out.has_xs = out.x1 > 0 ? 1u : 0u;                                         // :895
    out.xt = cr.x0 == 0 ? 'N' : cr.x0 == 1 ? 'U' : 'R';
    out.best_size_f = (float)cr.x0;
    uint32_t rl = 0;
    for (uint32_t i = 0; i < best.n_ops; ++i) rl += (bops[i] >> 24) != OP_DEL;
    out.read_len = rl;
    out.cigar_len = c0.n; out.md_len = m0.n; out.xa_len = x0.n; out.mq_n = n_pairs;
    text_bytes = c0.n + m0.n + x0.n;
}
MAPAD_HD void record_text_write(const TextIndex& T, const HitRec* hits, const uint32_t* ops, const CoordRec& cr, char* text_pool, float* pair_pool, const DevRecord& out) {
    if (!out.mapped || out.error) return;
    const HitRec& best = hits[cr.best];
    const uint32_t* bops = ops + best.ops_off;
    TextSink c1{text_pool + out.text_off, 0, true}, m1{text_pool + out.text_off + out.cigar_len, 0, true}, x1{text_pool + out.text_off + out.cigar_len + out.md_len, 0, true};
    bam_fields_hd(T, bops, best.n_ops, cr.first.backward != 0, cr.first.abs, &c1, &m1);
    for (uint32_t k = 0; k < cr.n_xa; ++k) xa_entry_hd(T, cr.xa[k], hits[cr.xa[k].hit], ops, x1);
    uint32_t w = 0;
    for (uint32_t k = 0; k < cr.n_order; ++k) {
        const HitRec& h = hits[cr.order[k]];
        if (cross_check_hd(best, h)) continue;
        pair_pool[2 * ((uint64_t)out.mq_off + w)] = h.score; pair_pool[2 * ((uint64_t)out.mq_off + w) + 1] = (float)h.size;
        w += 1;
    }
}

}  // namespace mapad
