// darray_core.hpp — the bidirectional D array (lower bound of the remaining penalty) of one read.
//
// Device equivalent of BiDArray::new / compute_part (src/map/bi_d_array.rs:24-198).  Each of the 15 offset chains is a
// plain exact-match FM search that restarts whenever the interval empties; one quad runs one chain (its 4 lanes split
// the two rank queries of every step, fmd_device.hpp), so the 15 chains of a read occupy 15 of the 16 quads of one
// wavefront.  The chain only needs the size of the interval and the bound on the side it extends, never the other
// strand's bound, so forward_ext on the swapped interval (fmd_index.rs:93-96) is a mono-directional step with the
// complemented base.
#pragma once
#include "search_core.hpp"

namespace mapad {

constexpr int kMaxOffset = 15;  // bi_d_array.rs:37

// penalty a D-array chain adds when it hits a mismatch at read position r (bi_d_array.rs:152-189)
MAPAD_HD float d_penalty(const DevParams& P, const uint8_t* seq, const uint8_t* qual, int L, int r) {
    const int to_class = base_index(seq[r]);
    const Float4 row = sdm_row(P, L, r, qual[r], to_class);
    float v = sdm_best_mismatch(row, to_class) - sdm_optimal(row, to_class);
    const int dist = r < (L - r - 1) ? r : (L - r - 1);
    if (dist >= P.gap_dist_ends) v = f32_max(v, P.gap_extend);
    return v;
}

MAPAD_HD void ext1_any(const DevIndex& ix, uint64_t lower, uint64_t size, int k, int w, uint64_t& nl, uint64_t& ns) {
#if defined(__HIP_DEVICE_COMPILE__)
    ext1_quad(ix, lower, size, k, w, nl, ns);
#else
    (void)w;
    ext1_scalar(ix, lower, size, k, nl, ns);
#endif
}

// One offset chain over one part of the read.  left_part: pattern[..split] scanned left->right with forward_ext;
// otherwise pattern[split..] scanned from the read end with backward_ext.  Writes out[0 .. part_len).
// `pen` is indexed by read position.  Returns the number of single-base extension calls (E_darray events).
MAPAD_HD uint32_t d_chain(const DevIndex& ix, const uint8_t* seq, int L, int split, bool left_part, int offset, const float* pen,
                          float* out, int w) {
    const int part_len = left_part ? split : L - split;
    for (int p = 0; p <= offset && p < part_len; ++p) out[p] = 0.0f;
    float z = 0.0f, m = kF32Min;
    uint64_t lower = 0, size = ix.n;  // init_interval
    uint32_t n_ext = 0;
    // the reference's lazy iterator is pulled part_len times, so the extension of the last base is never executed
    for (int i = offset; i + 1 < part_len; ++i) {
        const int r = left_part ? i : L - 1 - i;
        m = f32_max(m, pen[r]);
        const int b = base_index(seq[r]);
        uint64_t nl = 0, ns = 0;
        if (b < 4) ext1_any(ix, lower, size, left_part ? 3 - b : b, w, nl, ns);  // non-alphabet symbol -> empty interval (fmd_index.rs:79-85)
        n_ext += 1;
        if (ns < 1) {
            z += m;
            m = kF32Min;
            lower = 0; size = ix.n;
        } else {
            lower = nl; size = ns;
        }
        out[i + 1] = z;
    }
    return n_ext;
}

// Host-side composition (tests/emu): the whole D array of a read, 15 chains per part, min-reduced (bi_d_array.rs:40-98).
MAPAD_HD uint32_t d_array_scalar(const DevIndex& ix, const DevParams& P, const uint8_t* seq, const uint8_t* qual, int L, float* pen_buf,
                                 float* chain_buf, float* d_out) {
    const int split = P.start_at_end ? L : L / 2;
    for (int r = 0; r < L; ++r) { pen_buf[r] = d_penalty(P, seq, qual, L, r); d_out[r] = 0.0f; }
    uint32_t n_ext = 0;
    for (int part = 0; part < 2; ++part) {
        const bool left = part == 0;
        const int part_len = left ? split : L - split;
        for (int o = 0; o < kMaxOffset; ++o) {
            n_ext += d_chain(ix, seq, L, split, left, o, pen_buf, chain_buf, 0);
            for (int p = 0; p < part_len; ++p) {
                float& d = d_out[(left ? 0 : split) + p];
                d = f32_min(d, chain_buf[p]);
            }
        }
    }
    return n_ext;
}

}  // namespace mapad
