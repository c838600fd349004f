// wire.hpp — the reference's dispatcher <-> worker messages (src/distributed/mod.rs:14-44, src/map/input_chunk_reader.rs:245-253),
// decoded and encoded by hand.
//
// Both messages are `bincode::serialize` (bincode 1.3 defaults: little endian, fixed-width integers, u64 sequence lengths, u32 enum
// variant index, u8 tag for Option / bool, usize as u64) of a struct whose first field is the total encoded size as u64
// (`Message::PROTO_LEN`, mod.rs:16-19): the receiver reads 8 bytes, then the remaining size - 8 (comm_buffers.rs:23-60).
//
//   TaskSheet   { encoded_size: u64, chunk_id: usize, records: Vec<Record>, reference_path: Option<String>,
//                 alignment_parameters: Option<AlignmentParameters> }                       (input_chunk_reader.rs:246-253)
//   Record      { sequence: Vec<u8>, base_qualities: Vec<u8>, name: Option<Vec<u8>>, bam_tags: Vec<([u8; 2], BamAuxField)>,
//                 bam_flags: u16 }                                                          (record.rs:138-145)
//   BamAuxField 18 variants, Char .. ArrayFloat                                             (record.rs:21-42)
//   AlignmentParameters { difference_model, mismatch_bound, penalty_gap_open: f32, penalty_gap_extend: f32, chunk_size: usize,
//                 gap_dist_ends: u8, max_num_gaps_open: u8, stack_limit_abort: bool }       (map/mod.rs:21-31)
//   SequenceDifferenceModelDispatch { SimpleAncientDnaModel, VindijaPwm, TestDifferenceModel }   (sequence_difference_models.rs:67-72)
//   SimpleAncientDnaModel { library_prep: LibraryPrep { SingleStranded { five, three } | DoubleStranded(f32) }, ds_deamination_rate,
//                 ss_deamination_rate, divergence: f32, use_default_base_quality: Option<f32>, cache: Vec<f32>,
//                 three_prime_flank_offset: Option<i16> }                                   (:93-114)
//   MismatchBoundDispatch { Continuous { cutoff, exponent, representative_mismatch_penalty, cache: Vec<f32> },
//                 Discrete { poisson_threshold, base_error_rate, representative_mismatch_penalty, cache },
//                 TestBound { threshold, representative_mm_bound } }                        (mismatch_bounds.rs:25-30,76-82,122-128,263-267)
//   ResultSheet { encoded_size: u64, chunk_id: usize, results: Vec<(Record, BinaryHeap<HitInterval>, Duration)> }   (mod.rs:21-26)
//   HitInterval { interval: RtBiInterval { lower, lower_rev, size: usize }, alignment_score: f32,
//                 edit_operations: EditOperationsTrack(Vec<EditOperation>) }                (map/mod.rs:34-39, fmd_index.rs:184-189)
//   EditOperation { Insertion(u16), Deletion(u16, u8), Match(u16), Mismatch(u16, u8) }      (record.rs:225-231)
//   BinaryHeap serialises as the sequence of its backing array; Duration as { secs: u64, nanos: u32 }.
//
// No reference binary can run in this build environment (no Rust toolchain), so these bytes are pinned only by the struct definitions
// above and bincode's published format: wire parity is UNPINNED until it has met a real dispatcher.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mapad_amd.h"

namespace wire {

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    void need(size_t n) const { if ((size_t)(end - p) < n) throw std::runtime_error("task message is shorter than its contents"); }
    template <class T> T get() { need(sizeof(T)); T v; std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    const uint8_t* bytes(size_t n) { need(n); const uint8_t* q = p; p += n; return q; }
    bool option() { const uint8_t t = get<uint8_t>(); if (t > 1) throw std::runtime_error("bad Option tag"); return t == 1; }
    void skip_vec(size_t elem) { const uint64_t n = get<uint64_t>(); if (n > (uint64_t)(end - p) / elem) throw std::runtime_error("bad sequence length"); p += n * elem; }
};

struct RecordView {
    const uint8_t* raw = nullptr;  // the record exactly as it arrived; echoed into the result
    size_t raw_len = 0;
    const uint8_t* seq = nullptr;
    const uint8_t* qual = nullptr;
    uint64_t len = 0, qual_len = 0;
};

struct Task {
    uint64_t chunk_id = 0;
    std::vector<RecordView> records;
    bool has_reference = false, has_params = false;
    std::string reference_path;
    mapad_params_t params{};
};

inline void skip_aux_field(Cursor& c) {
    switch (c.get<uint32_t>()) {
    case 0: case 1: case 2: c.bytes(1); break;              // Char, I8, U8
    case 3: case 4: c.bytes(2); break;                      // I16, U16
    case 5: case 6: case 7: c.bytes(4); break;              // I32, U32, Float
    case 8: c.bytes(8); break;                              // Double
    case 9: case 10: case 11: case 12: c.skip_vec(1); break;  // String, HexByteArray, ArrayI8, ArrayU8
    case 13: case 14: c.skip_vec(2); break;                 // ArrayI16, ArrayU16
    case 15: case 16: case 17: c.skip_vec(4); break;        // ArrayI32, ArrayU32, ArrayFloat
    default: throw std::runtime_error("unknown BamAuxField variant");
    }
}

inline mapad_params_t decode_params(Cursor& c) {
    mapad_params_t p;
    std::memset(&p, 0, sizeof p);
    switch (c.get<uint32_t>()) {  // SequenceDifferenceModelDispatch
    case 0: {
        p.model_kind = MAPAD_MODEL_SIMPLE_ADNA;
        const uint32_t lib = c.get<uint32_t>();
        if (lib == 0) { p.library_prep = MAPAD_LIBRARY_SINGLE_STRANDED; p.five_prime_overhang = c.get<float>(); p.three_prime_overhang = c.get<float>(); }
        else if (lib == 1) { p.library_prep = MAPAD_LIBRARY_DOUBLE_STRANDED; p.five_prime_overhang = p.three_prime_overhang = c.get<float>(); }
        else throw std::runtime_error("unknown LibraryPrep variant");
        p.ds_deamination_rate = c.get<float>(); p.ss_deamination_rate = c.get<float>();
        p.divergence = c.get<float>();  // stored divided by 3 (main.rs:452), like mapad_params_t::divergence
        p.ignore_base_quality = c.option() ? (c.get<float>(), 1) : 0;
        c.skip_vec(4);                  // cache of per-quality error probabilities: rebuilt from the parameters
        if (c.option()) c.get<int16_t>();  // three_prime_flank_offset: computed, never read (sequence_difference_models.rs:299-316)
        break;
    }
    case 2: p.model_kind = MAPAD_MODEL_TEST; p.deam_score = c.get<float>(); p.mm_score = c.get<float>(); p.match_score = c.get<float>(); break;
    case 1: {  // VindijaPwm { ppm_read_ends_symmetric_ct: [f32; 7], position_probability_ct_default, observed_substitution_probability_default } (:340-345): bincode
        // writes a fixed-size array without a length.  The model has no parameters a caller can set — VindijaPwm::new() (:384-396) is its only constructor — and the
        // library's tables are built from those constants (host_models.hpp: vindija_get); a message that carries other values is refused rather than mis-scored.
        static const float want[9] = {0.4f, 0.25f, 0.1f, 0.06f, 0.05f, 0.04f, 0.03f, 0.02f, 0.0005f};
        for (float w : want) if (c.get<float>() != w) throw std::runtime_error("VindijaPwm with values other than VindijaPwm::new()'s is not supported");
        p.model_kind = MAPAD_MODEL_VINDIJA_PWM;
        break;
    }
    default: throw std::runtime_error("unknown SequenceDifferenceModelDispatch variant");
    }
    switch (c.get<uint32_t>()) {  // MismatchBoundDispatch
    case 0: p.bound_kind = MAPAD_BOUND_CONTINUOUS; p.cutoff = c.get<float>(); p.exponent = c.get<float>(); c.get<float>(); c.skip_vec(4); break;
    case 1: p.bound_kind = MAPAD_BOUND_DISCRETE; p.poisson_threshold = c.get<float>(); p.base_error_rate = c.get<float>(); c.get<float>(); c.skip_vec(4); break;
    case 2: p.bound_kind = MAPAD_BOUND_TEST; p.threshold = c.get<float>(); p.repr_mm_bound = c.get<float>(); break;
    default: throw std::runtime_error("unknown MismatchBoundDispatch variant");
    }
    p.penalty_gap_open = c.get<float>(); p.penalty_gap_extend = c.get<float>();
    p.chunk_size = c.get<uint64_t>();
    p.gap_dist_ends = c.get<uint8_t>(); p.max_num_gaps_open = c.get<uint8_t>();
    p.stack_limit_abort = c.get<uint8_t>();
    return p;
}

// msg: the whole message, size field included
inline Task decode_task(const uint8_t* msg, size_t n) {
    Cursor c{msg, msg + n};
    if (c.get<uint64_t>() != n) throw std::runtime_error("task header does not match the message size");
    Task t;
    t.chunk_id = c.get<uint64_t>();
    const uint64_t n_rec = c.get<uint64_t>();
    if (n_rec > n / 24) throw std::runtime_error("bad record count");  // a Record is at least 24 bytes on the wire (three empty vectors), so the count is bounded by the message
    t.records.resize(n_rec);
    for (auto& r : t.records) {
        r.raw = c.p;
        r.len = c.get<uint64_t>(); r.seq = c.bytes(r.len);
        r.qual_len = c.get<uint64_t>(); r.qual = c.bytes(r.qual_len);
        if (c.option()) c.skip_vec(1);  // name
        const uint64_t n_tags = c.get<uint64_t>();
        for (uint64_t k = 0; k < n_tags; ++k) { c.bytes(2); skip_aux_field(c); }
        c.get<uint16_t>();              // bam_flags
        r.raw_len = (size_t)(c.p - r.raw);
    }
    if ((t.has_reference = c.option())) { const uint64_t l = c.get<uint64_t>(); const uint8_t* s = c.bytes(l); t.reference_path.assign((const char*)s, l); }
    if ((t.has_params = c.option())) t.params = decode_params(c);
    if (c.p != c.end) throw std::runtime_error("trailing bytes in the task message");
    return t;
}

struct Out {
    std::vector<uint8_t> b;
    template <class T> void put(T v) { const size_t o = b.size(); b.resize(o + sizeof(T)); std::memcpy(b.data() + o, &v, sizeof(T)); }
    void raw(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
};

// res == nullptr: every read comes back without hits (dry run).  read_of[i] = index of record i in `res`, -1 = not mapped (no hits).
inline std::vector<uint8_t> encode_result(const Task& t, const mapad_batch_result_t* res, const std::vector<int64_t>& read_of, double seconds_per_read) {
    Out o;
    o.put<uint64_t>(0);  // encoded_size, patched below
    o.put<uint64_t>(t.chunk_id);
    o.put<uint64_t>(t.records.size());
    const uint64_t secs = (uint64_t)seconds_per_read;
    const uint32_t nanos = (uint32_t)((seconds_per_read - (double)secs) * 1e9);
    for (size_t i = 0; i < t.records.size(); ++i) {
        o.raw(t.records[i].raw, t.records[i].raw_len);
        const int64_t r = res ? read_of[i] : -1;
        const uint64_t h0 = r >= 0 ? res->hit_begin[r] : 0, h1 = r >= 0 ? res->hit_begin[r + 1] : 0;
        o.put<uint64_t>(h1 - h0);  // BinaryHeap<HitInterval>: its backing array, in order
        for (uint64_t h = h0; h < h1; ++h) {
            const mapad_hit_t& hit = res->hits[h];
            o.put<uint64_t>(hit.lower); o.put<uint64_t>(hit.lower_rev); o.put<uint64_t>(hit.size);
            o.put<float>(hit.alignment_score);
            o.put<uint64_t>(hit.n_ops);
            for (uint32_t k = 0; k < hit.n_ops; ++k) {
                const uint32_t op = res->ops[hit.ops_offset + k], kind = op >> 24;
                o.put<uint32_t>(kind);  // Insertion, Deletion, Match, Mismatch: the packed kind is the variant index
                o.put<uint16_t>((uint16_t)(op & 0xFFFF));
                if (kind == 1 || kind == 3) o.put<uint8_t>((uint8_t)((op >> 16) & 0xFF));
            }
        }
        o.put<uint64_t>(secs); o.put<uint32_t>(nanos);  // Duration
    }
    const uint64_t total = o.b.size();
    std::memcpy(o.b.data(), &total, 8);
    return std::move(o.b);
}

}  // namespace wire
