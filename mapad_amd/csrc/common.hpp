// common.hpp — plain-old-data shared by host code and the gfx950 kernels of the mapAD hot path.
//
// Everything here is usable from both __host__ and __device__ code.  The "host" compilation of the
// per-read logic (search_core.hpp / darray_core.hpp) exists only so that the CPU test-suite can run the very
// same source under g++ (tests/emu); the product library always runs it on the GPU.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MAPAD_HD __host__ __device__ __forceinline__
#else
#define MAPAD_HD inline
#endif

// MAPAD_TOUCH(pointer, bytes, is_write): nothing in the product.  tests/emu defines it to feed a line-granular cache model with every arena and index access of
// the host build of the step (tests/emu/emu.cpp: which lines a pop asks memory for — index, nodes, heap levels, hit staging; profiles/r05/request_attribution.json).
#if !defined(MAPAD_TOUCH)
#define MAPAD_TOUCH(p, bytes, wr) ((void)0)
#endif


namespace mapad {

// ---- alphabet -------------------------------------------------------------------------------------------
// Reference ranks (src/index/indexing.rs:146-148): $=0 A=1 C=2 G=3 T=4 X=5.
// Device symbol codes (3 bit-planes): $=0 X=1 A=4 C=5 G=6 T=7 — plane 2 marks "is ACGT", planes 1..0 the base index.
constexpr int kBlockRows = 96;   // BWT rows per 64-byte block
constexpr int kBlockBytes = 64;  // 4 sub-blocks x 16 bytes {40-bit count[base w], 3 planes x 24 rows} (fmd_device.hpp)
constexpr int kBlockWords = kBlockBytes / 8;
constexpr int kSubRows = kBlockRows / 4;

MAPAD_HD int base_index(uint8_t c) {  // ASCII -> 0..3 for ACGT, 4 otherwise
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}

struct BiInterval {  // src/map/fmd_index.rs:184-219
    uint64_t lower, lower_rev, size;
};

// ---- packed edit operation (src/map/record.rs:225-237): kind<<24 | ref-base ASCII<<16 | read position ------
enum : uint32_t { OP_INS = 0, OP_DEL = 1, OP_MATCH = 2, OP_MISMATCH = 3 };
MAPAD_HD uint32_t pack_op(uint32_t kind, uint32_t pos, uint32_t base) { return (kind << 24) | (base << 16) | (pos & 0xFFFFu); }

// ---- device view of the FMD index ---------------------------------------------------------------------------
struct DevIndex {
    const uint64_t* blocks;  // n_blocks * kBlockWords u64
    uint64_t n;              // BWT length = 2*|G| + 2
    uint64_t n_blocks;
    uint64_t less[8];        // less[rank], rank 0..6
    uint64_t sentinel[2];    // rows with bwt == '$' (ascending)  (src/map/fmd_index.rs:38-47)
};

// ---- scoring / pruning parameters as the kernels see them -------------------------------------------------------
// All transcendental math (log2, powi, exp, powf) is done on the host with glibc exactly like the reference;
// the GPU only adds, compares and takes min/max of these f32 values, in the reference's association order.
enum : int32_t { BOUND_DISCRETE = 0, BOUND_CONTINUOUS = 1, BOUND_TEST = 2 };
constexpr int kMaxReadLen = 32767;  // the reference's limit: i16::MAX (src/map/record.rs:144-150)

struct DevParams {
    const float* sdm_table;      // float4 entries [A,C,G,T as `from`]: index table_base[L] + ((i*nq + q)*5 + to_class)
    const int32_t* table_base;   // per read length; -1 = not built
    const float* reject_thr;     // per L: Discrete get(L)*repr_mm | Test threshold | Continuous L^e
    int32_t nq;                  // quality levels in the table (1 when base qualities are ignored by the model)
    int32_t bound_kind;
    float cutoff;                // Continuous
    float repr_mm;               // representative mismatch penalty (reject_iterative)
    float gap_open, gap_extend;
    int32_t gap_dist_ends, max_num_gaps_open;
    int32_t start_at_end;        // SimpleAncientDnaModel: alignment starts at the 3' end (sequence_difference_models.rs:209-211)
    int32_t stack_limit_abort;
    uint32_t stack_limit, edit_tree_limit;  // src/map/mapping.rs:52-54
};

// reject(): `thr` is DevParams::reject_thr[L] of the read.  CONT = the Continuous bound (needs an IEEE division).
template <bool CONT>
MAPAD_HD bool mb_reject(float thr, float cutoff, float v) {  // mismatch_bounds.rs:85-87,131-134,269-271
    if (CONT) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __fdiv_rn(v, thr) < cutoff;
#else
        return (v / thr) < cutoff;
#endif
    }
    return v < thr;
}
MAPAD_HD bool mb_reject_iterative(const DevParams& p, float v, float ref) {  // :89-91,136-138,273-275
    if (p.bound_kind == BOUND_TEST) return false;
    return v < ref + p.repr_mm;
}

MAPAD_HD float f32_max(float a, float b) { return a > b ? a : b; }  // no NaNs on this path
MAPAD_HD float f32_min(float a, float b) { return a < b ? a : b; }
constexpr float kF32Min = -3.402823466e+38f;  // Rust f32::MIN

struct Float4 { float a, c, g, t; };
MAPAD_HD float f4_get(const Float4& f, int i) { float r = f.t; r = i == 2 ? f.g : r; r = i == 1 ? f.c : r; r = i == 0 ? f.a : r; return r; }

// `base` = DevParams::table_base[L] of the read (looked up once per read: it is a dependent load otherwise)
MAPAD_HD Float4 sdm_row_at(const DevParams& p, int32_t base, int i, int q, int to_class) {
    const int qi = p.nq == 1 ? 0 : q;
    const float* e = p.sdm_table + 4 * ((size_t)base + ((size_t)i * p.nq + qi) * 5 + to_class);
    return Float4{e[0], e[1], e[2], e[3]};
}
MAPAD_HD Float4 sdm_row(const DevParams& p, int L, int i, int q, int to_class) { return sdm_row_at(p, p.table_base[L], i, q, to_class); }
// get_min_penalty(.., only_mismatches=false)  (sequence_difference_models.rs:34-57)
MAPAD_HD float sdm_optimal(const Float4& r, int to_class) {
    if (to_class > 3) return 0.0f;
    return f32_max(f32_max(f32_max(f32_max(kF32Min, r.a), r.c), r.g), r.t);
}
// get_min_penalty(.., only_mismatches=true)
MAPAD_HD float sdm_best_mismatch(const Float4& r, int to_class) {
    float m = kF32Min;
    if (to_class != 0) m = f32_max(m, r.a);
    if (to_class != 1) m = f32_max(m, r.c);
    if (to_class != 2) m = f32_max(m, r.g);
    if (to_class != 3) m = f32_max(m, r.t);
    return m;
}

// per-read event counters (identical on the CPU oracle and the kernels — itself a parity check)
struct ReadCounters {
    uint32_t e_search, e_darray, n_push, n_pop, n_node, n_hits;
};

}  // namespace mapad
