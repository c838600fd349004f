// capi.cpp — flat C interface over the CPU ORACLE for ctypes (tests / smoke / bench cpu_baseline only).
// Test infrastructure, not product code: nothing under mapad_amd/ links or loads this.
#include "mapad_oracle.hpp"

#include <atomic>
#include <memory>
#include <thread>

using namespace mo;

extern "C" {

typedef struct mo_params {
    int32_t model_kind;  // 0 SimpleAncientDnaModel, 1 VindijaPwm, 2 TestDifferenceModel
    int32_t library_prep;  // 0 single_stranded, 1 double_stranded
    float five_prime_overhang, three_prime_overhang, ds_deamination_rate, ss_deamination_rate, divergence;
    int32_t ignore_base_quality;
    float deam_score, mm_score, match_score;  // TestDifferenceModel
    int32_t bound_kind;  // 0 Discrete, 1 Continuous, 2 TestBound
    float poisson_threshold, base_error_rate;  // Discrete
    float cutoff, exponent;                    // Continuous
    float threshold, repr_mm_bound;            // TestBound
    float penalty_gap_open, penalty_gap_extend;
    int32_t gap_dist_ends, max_num_gaps_open, stack_limit_abort;
    uint32_t stack_limit, edit_tree_limit;
    int32_t heap_variant;
} mo_params;

}  // extern "C"

namespace {

struct Models {
    std::unique_ptr<SequenceDifferenceModel> sdm;
    std::unique_ptr<MismatchBound> mb;
    AlignmentParameters ap;
    int heap_variant = 0;
};

Models make_models(const mo_params& p) {
    Models m;
    switch (p.model_kind) {
        case 0:
            m.sdm = std::make_unique<SimpleAncientDnaModel>(p.library_prep == 0, p.five_prime_overhang,
                                                            p.library_prep == 0 ? p.three_prime_overhang : p.five_prime_overhang,
                                                            p.ds_deamination_rate, p.ss_deamination_rate, p.divergence,
                                                            p.ignore_base_quality != 0);
            break;
        case 1: m.sdm = std::make_unique<VindijaPwm>(); break;
        default: m.sdm = std::make_unique<TestDifferenceModel>(p.deam_score, p.mm_score, p.match_score);
    }
    const float repr = m.sdm->get_representative_mismatch_penalty();
    switch (p.bound_kind) {
        case 0: m.mb = std::make_unique<Discrete>(p.poisson_threshold, p.base_error_rate, repr); break;
        case 1: m.mb = std::make_unique<Continuous>(p.cutoff, p.exponent, repr); break;
        default: m.mb = std::make_unique<TestBound>(p.threshold, p.repr_mm_bound);
    }
    m.ap.penalty_gap_open = p.penalty_gap_open;
    m.ap.penalty_gap_extend = p.penalty_gap_extend;
    m.ap.gap_dist_ends = (uint8_t)p.gap_dist_ends;
    m.ap.max_num_gaps_open = (uint8_t)p.max_num_gaps_open;
    m.ap.stack_limit_abort = p.stack_limit_abort != 0;
    if (p.stack_limit) m.ap.stack_limit = p.stack_limit;
    if (p.edit_tree_limit) m.ap.edit_tree_limit = p.edit_tree_limit;
    m.heap_variant = p.heap_variant;
    return m;
}

struct Index {
    RtFmdIndex fmd;
    std::vector<uint64_t> sa;  // full SA when built from text
    SampledSuffixArray ssa;
    FastaIdPositions idmap;
    OriginalSymbols orig;
    bool has_ssa = false;
};

struct Result {
    std::vector<HitHeap> hits;  // per read, BinaryHeap array order
    std::vector<Counters> counters;
    std::vector<std::vector<float>> d_arrays;
};

uint32_t pack_op(const EditOperation& op) { return ((uint32_t)op.kind << 24) | ((uint32_t)op.base << 16) | op.pos; }

uint32_t seed_for(uint64_t read_idx, uint32_t call) {  // deterministic stand-in for rand::rng().next_u32()
    uint64_t z = (read_idx + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)call * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)z;
}

}  // namespace

extern "C" {

void* mo_index_build(const uint8_t* text, uint64_t n, const char* alphabet_sorted, uint32_t occ_k) {
    try {
        auto idx = new Index();
        idx->fmd = build_index_from_text(std::vector<uint8_t>(text, text + n), alphabet_sorted, occ_k, &idx->sa);
        return idx;
    } catch (const std::exception& e) { std::fprintf(stderr, "mo_index_build: %s\n", e.what()); return nullptr; }
}
void* mo_index_from_bwt(const uint8_t* bwt, uint64_t n, const char* alphabet_sorted, uint32_t occ_k) {
    auto idx = new Index();
    idx->fmd = build_index_from_bwt(std::vector<uint8_t>(bwt, bwt + n), alphabet_sorted, occ_k);
    return idx;
}
void mo_index_free(void* h) { delete (Index*)h; }
uint64_t mo_index_len(void* h) { return ((Index*)h)->fmd.bwt.size(); }
void mo_index_get_bwt(void* h, uint8_t* out) { auto& b = ((Index*)h)->fmd.bwt; std::memcpy(out, b.data(), b.size()); }
int mo_index_get_sa(void* h, uint64_t* out) {
    auto& s = ((Index*)h)->sa; if (s.empty()) return -1; std::memcpy(out, s.data(), s.size() * 8); return 0;
}
void mo_index_get_less(void* h, uint64_t* out, int n) { auto& l = ((Index*)h)->fmd.less; for (int i = 0; i < n && i < (int)l.size(); ++i) out[i] = l[i]; }
uint64_t mo_index_occ(void* h, uint64_t r, int a) { return ((Index*)h)->fmd.occ(r, (uint8_t)a); }
// extension of a bi-interval: out[4][3] for c = T,G,C,A
void mo_index_extend(void* h, uint64_t lower, uint64_t lower_rev, uint64_t size, uint64_t* out) {
    RtBiInterval o[4];
    ((Index*)h)->fmd.extend_all({lower, lower_rev, size}, o);
    for (int k = 0; k < 4; ++k) { out[3 * k] = o[k].lower; out[3 * k + 1] = o[k].lower_rev; out[3 * k + 2] = o[k].size; }
}
void mo_index_add_contig(void* h, uint64_t start, uint64_t end, const char* name) { ((Index*)h)->idmap.id_position.push_back({start, end, name}); }
void mo_index_set_original_symbol(void* h, uint64_t pos, int sym) { ((Index*)h)->orig.map[pos] = (uint8_t)sym; }
int mo_index_sample_sa(void* h, uint64_t rate) {
    auto idx = (Index*)h;
    if (idx->sa.empty()) return -1;
    idx->ssa = SampledSuffixArray::sample_from(idx->sa, idx->fmd, rate);
    idx->has_ssa = true;
    return 0;
}
// sampled SA handed over from outside (bench cpu_baseline / cross-checks)
void mo_index_set_sampled_sa(void* h, const uint64_t* sample, uint64_t n_sample, uint64_t rate, const uint64_t* extra_rows,
                             const uint64_t* extra_vals, uint64_t n_extra) {
    auto idx = (Index*)h;
    idx->ssa = SampledSuffixArray();
    idx->ssa.fmd = &idx->fmd; idx->ssa.sampling_rate = rate;
    idx->ssa.sample.assign(sample, sample + n_sample);
    for (uint64_t i = 0; i < n_extra; ++i) idx->ssa.extra_rows[extra_rows[i]] = extra_vals[i];
    idx->has_ssa = true;
}
int mo_ssa_get(void* h, uint64_t row, uint64_t* out) { auto idx = (Index*)h; return idx->has_ssa && idx->ssa.get(row, *out) ? 0 : -1; }

// ---- plugin trait surface ----
float mo_sdm_get(const mo_params* p, uint64_t i, uint64_t len, int from, int to, int q) { return make_models(*p).sdm->get(i, len, (uint8_t)from, (uint8_t)to, (uint8_t)q); }
float mo_sdm_repr_mm(const mo_params* p) { return make_models(*p).sdm->get_representative_mismatch_penalty(); }
float mo_sdm_min_penalty(const mo_params* p, uint64_t i, uint64_t len, int to, int q, int only_mm) { return make_models(*p).sdm->get_min_penalty(i, len, (uint8_t)to, (uint8_t)q, only_mm != 0); }
int mo_sdm_alignment_start(const mo_params* p, uint64_t len) { return make_models(*p).sdm->find_alignment_start(len); }
int mo_mb_reject(const mo_params* p, float v, uint64_t len) { return make_models(*p).mb->reject(v, len); }
int mo_mb_reject_iterative(const mo_params* p, float v, float ref) { return make_models(*p).mb->reject_iterative(v, ref); }
float mo_mb_remaining_frac(const mo_params* p, float v, uint64_t len) { return make_models(*p).mb->remaining_frac_of_repr_mm(v, len); }
float mo_discrete_get(float poisson, float err, uint64_t len) { return Discrete(poisson, err, -1.0f).get(len); }
// bulk: fill table[len][5 to-bases ACGTN][4 from-bases ACGT] for one (len, q)
void mo_sdm_table(const mo_params* p, uint64_t len, int q, float* out) {
    auto m = make_models(*p);
    static const uint8_t TO[5] = {'A', 'C', 'G', 'T', 'N'};
    for (uint64_t i = 0; i < len; ++i)
        for (int t = 0; t < 5; ++t)
            for (int f = 0; f < 4; ++f) out[(i * 5 + t) * 4 + f] = m.sdm->get(i, len, DNA_UPPERCASE_ALPHABET[f], TO[t], (uint8_t)q);
}

// ---- D array ----
int mo_d_array(void* h, const mo_params* p, const uint8_t* seq, const uint8_t* qual, uint64_t len, int64_t split, float* out) {
    auto m = make_models(*p);
    const size_t sp = split < 0 ? (size_t)m.sdm->find_alignment_start(len) : (size_t)split;
    BiDArray d(seq, qual, len, sp, m.ap, ((Index*)h)->fmd, *m.sdm, nullptr);
    std::memcpy(out, d.d_composite.data(), len * sizeof(float));
    return 0;
}
float mo_d_array_get(void* h, const mo_params* p, const uint8_t* seq, const uint8_t* qual, uint64_t len, int64_t split, int k, int l) {
    auto m = make_models(*p);
    const size_t sp = split < 0 ? (size_t)m.sdm->find_alignment_start(len) : (size_t)split;
    BiDArray d(seq, qual, len, sp, m.ap, ((Index*)h)->fmd, *m.sdm, nullptr);
    return d.get((int16_t)k, (int16_t)l);
}

// ---- search over a batch (order-preserving parallel map, like rayon's collect) ----
void* mo_map_batch(void* h, const mo_params* p, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n,
                   int n_threads, int keep_d) {
    auto idx = (Index*)h;
    auto res = new Result();
    res->hits.resize(n); res->counters.resize(n);
    if (keep_d) res->d_arrays.resize(n);
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        auto m = make_models(*p);
        MinMaxHeap stack; stack.variant = m.heap_variant;
        Tree tree;
        while (true) {
            const uint64_t i = next.fetch_add(1);
            if (i >= n) break;
            const uint64_t o = offsets[i], len = offsets[i + 1] - o;
            res->hits[i] = k_mismatch_search(seqs + o, quals + o, len, m.ap, idx->fmd, stack, tree, *m.sdm, *m.mb, &res->counters[i],
                                             keep_d ? &res->d_arrays[i] : nullptr);
        }
    };
    if (n_threads <= 1) worker();
    else { std::vector<std::thread> ts; for (int t = 0; t < n_threads; ++t) ts.emplace_back(worker); for (auto& t : ts) t.join(); }
    return res;
}
void mo_result_free(void* r) { delete (Result*)r; }
uint64_t mo_result_total_hits(void* r) { uint64_t t = 0; for (auto& h : ((Result*)r)->hits) t += h.len(); return t; }
uint64_t mo_result_total_ops(void* r) { uint64_t t = 0; for (auto& h : ((Result*)r)->hits) for (auto& x : h.data) t += x.edit_operations.size(); return t; }
// hit_offsets[n+1]; per hit: interval[3], score, op_offsets[total_hits+1]; ops[total_ops]
void mo_result_export(void* r, uint64_t* hit_offsets, uint64_t* intervals, float* scores, uint64_t* op_offsets, uint32_t* ops) {
    auto res = (Result*)r;
    uint64_t hi = 0, oi = 0;
    hit_offsets[0] = 0; op_offsets[0] = 0;
    for (size_t i = 0; i < res->hits.size(); ++i) {
        for (auto& h : res->hits[i].data) {
            intervals[3 * hi] = h.interval.lower; intervals[3 * hi + 1] = h.interval.lower_rev; intervals[3 * hi + 2] = h.interval.size;
            scores[hi] = h.alignment_score;
            for (auto& op : h.edit_operations) ops[oi++] = pack_op(op);
            ++hi; op_offsets[hi] = oi;
        }
        hit_offsets[i + 1] = hi;
    }
}
// counters[n][6]: e_search, e_darray, n_push, n_pop, n_node, n_hits
void mo_result_counters(void* r, uint64_t* out) {
    auto res = (Result*)r;
    for (size_t i = 0; i < res->counters.size(); ++i) {
        auto& c = res->counters[i];
        out[6 * i] = c.e_search; out[6 * i + 1] = c.e_darray; out[6 * i + 2] = c.n_push; out[6 * i + 3] = c.n_pop; out[6 * i + 4] = c.n_node; out[6 * i + 5] = c.n_hits;
    }
}
int mo_result_d_array(void* r, uint64_t read, float* out) {
    auto res = (Result*)r;
    if (read >= res->d_arrays.size()) return -1;
    std::memcpy(out, res->d_arrays[read].data(), res->d_arrays[read].size() * 4);
    return 0;
}
// CIGAR / MD / NM of one hit at (strand, absolute_pos) — buffers must hold 4*ops+16 bytes
int mo_hit_bam_fields(void* h, void* r, uint64_t read, uint64_t hit, int backward, uint64_t abs_pos, int use_orig, char* cigar, char* md, int* nm) {
    auto res = (Result*)r;
    static const OriginalSymbols empty;
    const auto& hv = res->hits[read].data[hit];
    const BamFields bf = to_bam_fields(hv.edit_operations, backward ? Direction::Backward : Direction::Forward, abs_pos,
                                       use_orig ? ((Index*)h)->orig : empty);
    std::strcpy(cigar, cigar_string(bf).c_str()); std::strcpy(md, bf.md.c_str()); *nm = bf.nm;
    return 0;
}
// best hit per BinaryHeap::pop()/peek() == data[0]
// ---- post-processing to record fields; returns malloc'ed TSV (caller frees with mo_free) ----
char* mo_records_tsv(void* h, void* r, const mo_params* p, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets,
                     const uint16_t* flags, uint64_t n) {
    auto idx = (Index*)h; auto res = (Result*)r;
    auto m = make_models(*p);
    std::string out;
    for (uint64_t i = 0; i < n; ++i) {
        InRecord in;
        in.flags = flags ? flags[i] : 4;
        in.seq.assign(seqs + offsets[i], seqs + offsets[i + 1]);
        in.qual.assign(quals + offsets[i], quals + offsets[i + 1]);
        uint32_t call = 0;
        HitHeap copy = res->hits[i];
        OutRecord rec = intervals_to_record(in, std::move(copy), idx->ssa, idx->idmap, idx->orig, *m.mb, [&]() { return seed_for(i, call++); });
        char buf[256];
        std::string q; for (auto c : rec.qual) q.push_back((char)(c + 33));
        uint32_t as_bits, xs_bits; std::memcpy(&as_bits, &rec.as, 4); std::memcpy(&xs_bits, &rec.xs, 4);
        std::snprintf(buf, sizeof buf, "%u\t%d\t%lld\t%u\t", rec.flags, rec.tid, (long long)rec.pos, rec.mapq);
        out += buf; out += rec.cigar.empty() ? "*" : rec.cigar; out += "\t"; out += rec.seq; out += "\t"; out += q; out += "\t";
        if (rec.mapped) {
            std::snprintf(buf, sizeof buf, "%08x\t%d\t", as_bits, rec.nm); out += buf; out += rec.md; out += "\t";
            out += rec.xa.empty() ? "*" : rec.xa;
            std::snprintf(buf, sizeof buf, "\t%d\t%d\t", rec.x0, rec.x1); out += buf;
            if (rec.has_xs) { std::snprintf(buf, sizeof buf, "%08x", xs_bits); out += buf; } else out += "*";
            out += "\t"; out.push_back(rec.xt);
        } else out += "*\t*\t*\t*\t*\t*\t*\t*";
        out += "\n";
    }
    char* c = (char*)std::malloc(out.size() + 1);
    std::memcpy(c, out.c_str(), out.size() + 1);
    return c;
}
void mo_free(void* p) { std::free(p); }

// ---- post-processing of hits handed over from outside (round 6: the record-level audit at n > 2^32, profiles/audit_c4.py --records) ----
// intervals_to_record (src/map/mapping.rs:402-718, record.rs:269-449 restated in mapad_oracle.hpp) over hits that were found elsewhere — the product's, proven
// identical to this oracle's own search results read by read — given per read in BinaryHeap array order: hit_begin[n + 1]; per hit intervals[3], score,
// op_begin[n_hits + 1] into the packed edit operations.  Read i draws the stand-ins for rand::rng() of read first_read_index + i (seed_for), as mo_records_tsv does.
// Output: one fixed record per read + a text pool ([CIGAR][MD][XA] per mapped read, in read order), fetched with mo_hit_records_export.
struct MoRecord {
    int64_t pos; int32_t tid; uint32_t as_bits, xs_bits; int32_t nm, x0, x1; uint32_t cigar_len, md_len, xa_len;
    uint16_t flags; uint8_t mapq, mapped, reverse, has_xs, xt, pad;
    uint64_t text_off;
};
static_assert(sizeof(MoRecord) == 64, "oracle record layout (oracle/binding.py: MO_RECORD_DTYPE)");
struct HitRecords { std::vector<MoRecord> recs; std::string text; std::string error; };
void* mo_records_from_hits(void* h, const mo_params* p, const uint64_t* hit_begin, const uint64_t* intervals, const float* scores, const uint64_t* op_begin,
                           const uint32_t* ops, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, const uint16_t* flags, uint64_t n,
                           uint64_t first_read_index, int n_threads) {
    auto idx = (Index*)h;
    auto out = new HitRecords();
    out->recs.resize(n);
    const int T = std::max(1, n_threads);
    std::vector<std::string> texts((size_t)T);
    std::vector<std::string> errors((size_t)T);
    auto worker = [&](int t) {
        auto m = make_models(*p);
        const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
        std::string& text = texts[(size_t)t];
        try {
            for (uint64_t i = lo; i < hi; ++i) {
                InRecord in;
                in.flags = flags ? flags[i] : 4;
                in.seq.assign(seqs + offsets[i], seqs + offsets[i + 1]);
                in.qual.assign(quals + offsets[i], quals + offsets[i + 1]);
                HitHeap heap;
                for (uint64_t k = hit_begin[i]; k < hit_begin[i + 1]; ++k) {
                    HitInterval hv;
                    hv.interval = RtBiInterval{intervals[3 * k], intervals[3 * k + 1], intervals[3 * k + 2]};
                    hv.alignment_score = scores[k];
                    for (uint64_t o = op_begin[k]; o < op_begin[k + 1]; ++o)
                        hv.edit_operations.push_back(EditOperation{(OpKind)(ops[o] >> 24), (uint16_t)(ops[o] & 0xFFFFu), (uint8_t)((ops[o] >> 16) & 0xFFu)});
                    heap.data.push_back(std::move(hv));  // array order as given: the heap is not re-built
                }
                uint32_t call = 0;
                const OutRecord rec = intervals_to_record(in, std::move(heap), idx->ssa, idx->idmap, idx->orig, *m.mb, [&]() { return seed_for(first_read_index + i, call++); });
                MoRecord r{};
                r.pos = rec.pos; r.tid = rec.tid; r.flags = rec.flags; r.mapq = rec.mapq; r.mapped = rec.mapped; r.reverse = rec.reverse;
                if (rec.mapped) {
                    std::memcpy(&r.as_bits, &rec.as, 4); std::memcpy(&r.xs_bits, &rec.xs, 4);
                    r.nm = rec.nm; r.x0 = rec.x0; r.x1 = rec.x1; r.has_xs = rec.has_xs; r.xt = (uint8_t)rec.xt;
                    r.cigar_len = (uint32_t)rec.cigar.size(); r.md_len = (uint32_t)rec.md.size(); r.xa_len = (uint32_t)rec.xa.size();
                    r.text_off = text.size();  // relative to this thread's piece; rebased below
                    text += rec.cigar; text += rec.md; text += rec.xa;
                }
                out->recs[i] = r;
            }
        } catch (const std::exception& e) { errors[(size_t)t] = e.what(); }
    };
    if (T == 1) worker(0);
    else { std::vector<std::thread> ts; for (int t = 0; t < T; ++t) ts.emplace_back(worker, t); for (auto& t : ts) t.join(); }
    uint64_t base = 0;
    for (int t = 0; t < T; ++t) {
        const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
        for (uint64_t i = lo; i < hi; ++i) if (out->recs[i].mapped) out->recs[i].text_off += base;
        base += texts[(size_t)t].size();
        out->text += texts[(size_t)t];
        if (!errors[(size_t)t].empty()) out->error = errors[(size_t)t];
    }
    return out;
}
uint64_t mo_hit_records_text_bytes(void* r) { return ((HitRecords*)r)->text.size(); }
const char* mo_hit_records_error(void* r) { return ((HitRecords*)r)->error.c_str(); }
void mo_hit_records_export(void* r, void* recs, char* text) {
    auto hr = (HitRecords*)r;
    std::memcpy(recs, hr->recs.data(), hr->recs.size() * sizeof(MoRecord));
    std::memcpy(text, hr->text.data(), hr->text.size());
}
void mo_hit_records_free(void* r) { delete (HitRecords*)r; }

// ---- PrRange ----
int64_t mo_prrange(uint64_t start, uint64_t end, uint64_t seed, uint64_t* out, uint64_t max_out) {
    auto pr = PrRange::try_new(start, end, seed);
    if (!pr) return -1;
    uint64_t v; int64_t n = 0;
    while ((uint64_t)n < max_out && pr->next(v)) out[n++] = v;
    return n;
}
// first and last element without materialising (prrange.rs:191-199 style tests on huge ranges)
int mo_prrange_count(uint64_t start, uint64_t end, uint64_t seed, uint64_t limit, uint64_t* count, uint64_t* xor_all) {
    auto pr = PrRange::try_new(start, end, seed);
    if (!pr) return -1;
    uint64_t v, c = 0, x = 0;
    while (c < limit && pr->next(v)) { ++c; x ^= v; }
    *count = c; *xor_all = x;
    return 0;
}

// ---- slab tree semantics (backtrack_tree.rs tests) : tiny scripted interface ----
// ops: pairs (code, arg): 0 clear, 1 add_node(parent=arg) value=pos counter, 2 remove(arg); returns ids/len in out
int mo_tree_script(const int32_t* script, int n, int64_t* out) {
    Tree t; int o = 0; uint16_t ctr = 0;
    for (int i = 0; i < n; ++i) {
        const int code = script[2 * i], arg = script[2 * i + 1];
        if (code == 0) out[o++] = t.clear();
        else if (code == 1) out[o++] = t.add_node(EditOperation{OpKind::Match, ctr++, 0}, (uint32_t)arg);
        else if (code == 2) { t.remove((uint32_t)arg); out[o++] = (int64_t)t.len(); }
        else if (code == 3) { int64_t cnt = 0; t.ancestors((uint32_t)arg, [&](const EditOperation&) { ++cnt; }); out[o++] = cnt; }
        else if (code == 4) out[o++] = (int64_t)t.len();
    }
    return o;
}

}  // extern "C"
