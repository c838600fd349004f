// heavy_kernel.hpp — one wavefront per read, for the reads whose search frontier has outgrown a quad's base arena.
//
// (Included by mapad_amd.hip inside its anonymous namespace; uses BatchDev, GrowPools, DeviceGrow, finalize_read, HeavyItem.)
//
// Why: k_mismatch_search (src/map/mapping.rs:1012-1383) is a serial chain of pops, and the cost of 35-100 bp reads under `-p 0.03` is heavy-tailed
// (C5 read mix: 0.5 % of the reads hold 18 % of all pops, the heaviest makes 2.3 M of them).  The reference absorbs such reads on CPU threads
// (mapping.rs:1358-1380 bounds them); a quad — four lanes, heap levels 0-5 in LDS, 16 reads of a wavefront in lockstep — takes 6.4 us per pop
// once the heap is 2^14-2^17 entries deep: 4-5 dependent arena round trips for the sift, one per pushed child, and ~1 100 serial instructions.
// Here the read has the wavefront to itself, so every decision is wavefront-uniform (scalar branches, no lockstep with other reads) and the
// 64 lanes are spent on what a pop can do side by side:
//   * heap levels 0-9 (1 023 entries) live in LDS; a sift below them fetches the next three strides — children and grandchildren of every slot
//     the hole can reach, 63 aligned pairs — in ONE load instruction and walks them from LDS;
//   * the ancestors of all <= 9 pushes of a step (consecutive tail slots share them) are fetched in one load instruction as well;
//   * lanes 0-7 answer the two rank queries, lanes 0-8 score, gate and pack one child each.
// Semantics are those of search_core.hpp's search_step (same min-max heap, slab and hit list, bit for bit): the parity tests run reads
// through both paths.
#pragma once

static_assert(MAPAD_SUBTREE_HEAP == 0, "heavy_kernel's wavefront-wide strides address the implicit heap array (heap_core.hpp: HeapLayout)");
constexpr int kHeavyTop = 1023;  // heap levels 0-9
using HeavyArena = ArenaT<true, kHeavyTop>;
using HeavyRead = ReadInT<true>;

// LDS of a heavy wavefront: [kHeavyTop + 1 heap slots][2 bytes per read position: class, quality][D array][scratch]
constexpr uint32_t kHeavyScratchBytes = 8192;
constexpr uint32_t kHeavyMaxLdsReadLen = 1024;  // longer reads keep their position data (class / quality, D array) in the arena's near area in HBM
inline __host__ __device__ uint32_t heavy_lds_bytes(uint32_t lmax) {
    const uint32_t l = lmax <= kHeavyMaxLdsReadLen ? lmax : 0;
    return (kHeavyTop + 1) * 8 + ((2 * l + 15) & ~15u) + ((4 * l + 15) & ~15u) + kHeavyScratchBytes;
}

// ---- the wavefront-cooperative step ---------------------------------------------------------------------------------------------------------
// LDS scratch of a heavy wavefront (kHeavyScratchBytes): the speculative block of a deep sift and the ancestors of the step's pushes.
//   spec: the eight levels below the hole of a sift, by LOCAL index inside the hole's subtree (root = 0, children of i = 2i + 1, 2i + 2), shifted by
//         one entry like the heap itself so that a children pair is one aligned 16-byte unit: 511 entries + the shift.
//   anc : ancestors of the tail slots n1 .. n1 + 8 (where this step's children go), one row per distance d = 1, 2, ...: row d, column
//         slot - anc_d(n1).  Consecutive slots share ancestors: <= 6 / 4 / 3 / 2 / 2 / ... distinct slots per row, 29 rows cover any heap.
constexpr int kSpecEntries = 512, kAncRows = 30;
struct HeavyScratch {
    MAPAD_LDS HeapEntry* spec;  // logical local index i at spec[i] (the array starts one entry into the 16-byte aligned block)
    MAPAD_LDS HeapEntry* anc;   // [kAncRows][8]
};
static_assert((kSpecEntries + kAncRows * 8) * sizeof(HeapEntry) <= kHeavyScratchBytes, "heavy scratch layout");

__device__ __forceinline__ uint32_t anc_slot(uint32_t p, uint32_t d) { return ((p + 1) >> d) - 1; }  // d-th ancestor of heap slot p (d <= level of p)
__device__ __forceinline__ uint32_t heap_level(uint32_t p) { return 31u - (uint32_t)__clz((int)(p + 1)); }

// One stride of a trickle-down (the `stride` of search_core.hpp's mm_trickle_down: candidates in the order child 1, child 2, grandchildren 1-4, a
// later one wins only if strictly better) with the six candidates in six lanes: lane i < 6 reads its candidate from `base` (logical index
// c1b + i for the children, g1b + i - 2 for the grandchildren; c1 / g1 are their heap slots), the best is found by a three-step DPP reduction of
// 64-bit keys (score in sortable form, then 7 - scan rank so that the earlier candidate wins a tie) over the first eight lanes.
// The hole `pos` moves to the best candidate; returns false when the sift ends at `pos`.  set(slot, entry) stores.
__device__ __forceinline__ uint32_t sortable_f32(float x) {  // unsigned order == float order (no NaNs; -0.0 is folded into +0.0 first)
    const uint32_t u = __builtin_bit_cast(uint32_t, x + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
template <bool MAX, class Ptr, class Set>
__device__ __forceinline__ bool heavy_stride(Ptr base, uint32_t c1b, uint32_t g1b, uint32_t c1, uint32_t g1, uint32_t n, uint32_t& pos, HeapEntry& elt, int lane, Set&& set) {
    const uint32_t i = (uint32_t)lane & 7u;
    const bool child = i < 2;
    const uint32_t slot = child ? c1 + i : g1 + i - 2;
    const bool valid = (i < 6) & (slot < n);
    const HeapEntry e = load_entry(base + (child ? c1b + i : g1b + i - 2));  // lanes 6, 7 (and invalid slots) read an in-range neighbour: discarded
    const uint32_t s = sortable_f32(e.score);
    // scan rank of candidate i (child 1, child 2, grandchildren 1-4): ascending index, or family order with MAPAD_HEAP_VARIANT bit 0 (child 1, its two children, child 2, its two)
    const uint32_t rank = (MAPAD_HEAP_VARIANT & 1) ? ((0x00542130u >> (4u * i)) & 7u) : i;  // family order: i = 0..5 -> 0, 3, 1, 2, 4, 5
    uint32_t k_hi = valid ? (MAX ? s : ~s) : 0u, k_lo = ((7u - rank) << 3) | i;
    auto step = [&](auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const uint32_t o_hi = dpp_quad<C>(k_hi), o_lo = dpp_quad<C>(k_lo);
        const bool take = (o_hi > k_hi) | ((o_hi == k_hi) & (o_lo > k_lo));
        k_hi = take ? o_hi : k_hi; k_lo = take ? o_lo : k_lo;
    };
    step(std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    step(std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    step(std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad of the first eight lanes
    const uint32_t win = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_lo) & 7u;  // lane 0 holds the maximum of lanes 0-7: the candidate's number rides in the key's low bits
    HeapEntry be;
    be.score = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.score), (int)win));
    be.node = (uint32_t)__builtin_amdgcn_readlane((int)e.node, (int)win);
    const uint32_t best = win < 2 ? c1 + win : g1 + win - 2;
    if (!(MAX ? (be.score > elt.score) : (be.score < elt.score))) return false;
    set(pos, be);
    pos = best;
    if (win < 2) return false;  // moved to a child: done
    const uint32_t pl = (best - 1) >> 1 == c1 ? 0u : 1u;  // the parent of a grandchild is one of the two children: lane 0 or 1
    HeapEntry pe;
    pe.score = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.score), (int)pl));
    pe.node = (uint32_t)__builtin_amdgcn_readlane((int)e.node, (int)pl);
    if (MAX ? (pe.score > elt.score) : (pe.score < elt.score)) { set(c1 + pl, elt); elt = pe; }
    return true;
}

// trickle-down of `elt` from the hole `pos` in a heap of n entries.  Levels 0-9 are walked in LDS; below them the next four strides — children
// and grandchildren of every slot the hole can reach: up to 255 aligned pairs — are fetched together (lane l = the children pairs of the local
// nodes l, l + 64, ...: four loads in flight per lane) into LDS and walked from there.  Returns the hole position at which the sift left the
// LDS-only levels (every slot written in the arena lies in that slot's subtree), or ~0 if it never did.
template <bool MAX>
__device__ __forceinline__ uint32_t heavy_trickle_down(const HeavyArena& A, const HeavyScratch& S, uint32_t n, uint32_t pos, HeapEntry elt, int lane) {
    bool going = true;
    auto set_any = [&](uint32_t i, const HeapEntry e) {
        if (i < (uint32_t)kHeavyTop) store_entry(A.top + i, e);
        else if (lane == 0) store_entry(A.heap + i, e);
    };
    for (;;) {  // strides whose children and grandchildren all live in LDS
        const uint32_t c1 = 2 * pos + 1, g1 = 2 * c1 + 1;
        if (!(going & (c1 < n) & (g1 + 3 < (uint32_t)kHeavyTop))) break;
        going = heavy_stride<MAX>(A.top, c1, g1, c1, g1, n, pos, elt, lane, set_any);
    }
    uint32_t boundary = ~0u;
    while (going && 2 * pos + 1 < n) {
        if (boundary == ~0u) boundary = pos;
        // the block: local node i (root = the hole, children of i = 2i + 1, 2i + 2) <-> heap slot ((pos + 1) << depth) - 1 + offset
        const uint32_t levels_left = heap_level(n - 1) - heap_level(pos);              // levels below the hole that hold entries
        const uint32_t n_pairs = levels_left >= 8 ? 255u : (1u << levels_left) - 1u;   // local nodes whose children can exist
        HeapPair pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l = (uint32_t)lane + 64u * q;
            pr[q] = HeapPair{};
            if (l < n_pairs) {
                const uint32_t dd = 31u - (uint32_t)__clz((int)(l + 1)), off = l + 1 - (1u << dd);
                const uint32_t c = 2 * (((pos + 1) << dd) - 1 + off) + 1;  // first child of the local node's heap slot
                if (c < n) {
                    if (c < (uint32_t)kHeavyTop) pr[q] = load_pair(A.top + c);
                    else pr[q] = load_pair(A.heap + c);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l = (uint32_t)lane + 64u * q;
            if (l < n_pairs) { store_entry(S.spec + 2 * l + 1, pr[q].a); store_entry(S.spec + 2 * l + 2, pr[q].b); }
        }
        uint32_t h = 0;  // local index of the hole
#pragma unroll 1
        for (int k = 0; k < 4 && going && 2 * pos + 1 < n; ++k) {
            const uint32_t c1 = 2 * pos + 1, g1 = 2 * c1 + 1, c1l = 2 * h + 1, g1l = 2 * c1l + 1;
            going = heavy_stride<MAX>(S.spec, c1l, g1l, c1, g1, n, pos, elt, lane, set_any);
            h = pos >= g1 ? g1l + (pos - g1) : h;
        }
    }
    set_any(pos, elt);
    return boundary;
}

enum : int { HEAVY_DONE = 0, HEAVY_CONTINUE = 1, HEAVY_GENERAL = 2 };

// -DMAPAD_HEAVY_PROF: shader cycles per section of the step, summed over the wavefront's steps and added to the batch's cursors (a diagnostic build)
#if defined(MAPAD_HEAVY_PROF)
struct HeavyProf { unsigned long long t, acc[8]; };
#define HPROF_ARG , HeavyProf& hp
#define HPROF_MARK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); hp.acc[k] += t_ - hp.t; hp.t = t_; } while (0)
#else
#define HPROF_ARG
#define HPROF_MARK(k) ((void)0)
#endif

// One iteration of the `while let Some(stack_frame) = stack.pop_max()` loop (mapping.rs:1058-1355) by a whole wavefront, for the common case: no
// child can complete the read and the slab has no free list (search_core.hpp: the fast commit loop).  Anything else returns HEAVY_GENERAL with
// nothing touched, and the caller runs the general step (search_core.hpp's search_step, one lane).
// Every lane holds the same state; lanes differ only in: the sub-block of a rank query (lane & 3), the child they build (lanes 0-8), the heap
// entries they fetch for the block of a deep sift and for the ancestors of the pushes.
template <bool CONT, class RD>
__device__ __forceinline__ int heavy_step(const DevIndex& ix, const DevParams& P, const RD& rd, const HeavyArena& A, SearchState& st, const int lane, const HeavyScratch& S HPROF_ARG) {
    const uint32_t n = st.heap_len;
    const int L = rd.L;
    uint32_t top_idx;
    const HeapEntry top = mm_find_max(A, n, top_idx);
    const Node top_node = A.nodes[top.node];
    const uint32_t n1 = n - 1;
    const HeapEntry last = hp_get(A, n1);
    const Frame f = unpack_frame(top_node);
    if (!(f.len + 1 < L && st.tree_next == st.tree_entries)) return HEAVY_GENERAL;
    HPROF_MARK(0);
    st.c_pop += 1;
    const int alignment_start = alignment_start_of(P, L);
    const float open_ext = P.gap_open + P.gap_extend;
    const float f_score = top.score;
    const bool forward = f.start <= L - f.start - f.len;  // :1077-1097
    const int j = forward ? f.start + f.len : f.start - 1;
    const int d_k = forward ? f.start : f.start - 1, d_l = forward ? f.start + f.len : f.start + f.len - 1;
    const int to_class = rd.qc[2 * j];
    const Float4 row = sdm_row_at(P, rd.table, j, rd.qc[2 * j + 1], to_class);
    const uint32_t gap_side = forward ? f.gap_f : f.gap_b;
    const float insertion_score = (gap_side == GAP_INS ? P.gap_extend : open_ext) + f_score;  // :1127-1136,1165-1174
    const float deletion_score = (gap_side == GAP_DEL ? P.gap_extend : open_ext) + f_score;
    const uint32_t num_gaps_open = gap_side == GAP_CLOSED ? f.ngaps + 1 : f.ngaps;             // :1148-1152
    const float lower_bound = d_get(rd.d, L, alignment_start, d_k, d_l);                       // :1195
    if (st.n_hits > 0 && mb_reject_iterative(P, f_score + lower_bound, st.best_score)) { st.heap_len = n1; return HEAVY_DONE; }  // :1201-1208

    HPROF_MARK(1);
    // rank queries of the extension (:1245): lane & 3 = sub-block, as in a quad; every quad of the wavefront computes the same four extensions
    const int w = lane & 3;
    const uint64_t x_lower = forward ? f.lower_rev : f.lower, x_lower_rev = forward ? f.lower : f.lower_rev;
    const ExtLoads ext_loads = ext4_quad_issue(ix, x_lower, f.size, w);

    // ancestors of the tail slots n1 .. n1 + 8, fetched before the sift below stores anything (checked against the sift's subtree afterwards)
    uint32_t a_row, a_col;
    if (lane < 6) { a_row = 1; a_col = (uint32_t)lane; }
    else if (lane < 10) { a_row = 2; a_col = (uint32_t)lane - 6; }
    else if (lane < 13) { a_row = 3; a_col = (uint32_t)lane - 10; }
    else { a_row = 4 + (((uint32_t)lane - 13) >> 1); a_col = ((uint32_t)lane - 13) & 1; }
    const uint32_t lvl_hi = heap_level(n1 + 8);
    const uint32_t a_slot = a_row <= heap_level(n1) ? anc_slot(n1, a_row) + a_col : ~0u;  // rows beyond the root do not exist
    const bool a_valid = a_row <= lvl_hi && a_slot != ~0u && a_slot >= (uint32_t)kHeavyTop && a_slot <= anc_slot(n1 + 8, a_row);
    HeapEntry a_val = HeapEntry{0.0f, 0u};
    if (a_valid) a_val = load_entry(A.heap + a_slot);

    // pop_max of the crate, second half: the last entry takes the place of the maximum and trickles down
    st.heap_len = n1;
    uint32_t boundary = ~0u;
    HPROF_MARK(2);
    if (top_idx < n1) boundary = heavy_trickle_down<true>(A, S, n1, top_idx, last, lane);
    HPROF_MARK(3);
    if (boundary != ~0u) {  // did the sift write into a subtree that holds ancestors of the tail slots?  (rare: one path in 2^level)
        const uint32_t lb = heap_level(boundary);
        bool conflict = false;
        for (uint32_t dl = 0; dl < 2; ++dl) {  // the tail slots may straddle a level
            const uint32_t lt = heap_level(n1) + dl;
            if (lt > lvl_hi || lt < lb) continue;
            const uint32_t d = lt - lb;
            const uint32_t first_tail = dl == 0 ? n1 : (1u << lt) - 1, last_tail = (dl == 0 && heap_level(n1 + 8) != lt) ? (1u << (lt + 1)) - 2 : n1 + 8;
            if (d == 0) continue;  // the boundary is on the tail's own level: no descendants there
            conflict |= boundary >= anc_slot(first_tail, d) && boundary <= anc_slot(last_tail, d);
        }
        if (conflict) { a_val = HeapEntry{0.0f, 0u}; if (a_valid) a_val = load_entry(A.heap + a_slot); }
    }
    if (a_valid) store_entry(S.anc + a_row * 8 + a_col, a_val);

    HPROF_MARK(4);
    ExtLane x;
    ext4_quad_lane_finish(ix, ext_loads, x_lower, x_lower_rev, f.size, w, rd.lane_less, x);
    st.c_esearch += 1;
    HPROF_MARK(5);

    // children: lanes 0-3 the match / mismatch child of base w, lanes 4-7 the deletion child of base w, lane 8 the insertion child.
    // commit order (:1213-1339): Ins; then for k = T, G, C, A: Del(k), M/MM(k)  ->  t = 0; 1 + 2i, 2 + 2i with i = 3 - k
    const int ins_dist = j < (L - j - 1) ? j : (L - j - 1);
    const int dist5 = forward ? j : j + 1, dist3 = L - dist5;
    const int del_dist = dist5 < dist3 ? dist5 : dist3;
    const bool ins_ok = !mb_reject<CONT>(rd.thr, P.cutoff, insertion_score + lower_bound) & (ins_dist >= P.gap_dist_ends);  // :1214-1216
    const bool del_ok = !mb_reject<CONT>(rd.thr, P.cutoff, deletion_score + lower_bound) & (del_dist >= P.gap_dist_ends);   // :1279-1281
    const float optimal = sdm_optimal(row, to_class);
    const int cb = forward ? 3 - w : w;  // symbol in read orientation
    const float mm_score = f4_get(row, cb) - optimal + f_score;  // :1138-1145
    const bool has = ((x.nonempty >> w) & 1u) != 0;
    const bool is_mm = lane < 4, is_del = lane >= 4 && lane < 8, is_ins = lane == 8;
    const bool gaps_ok = (int)num_gaps_open <= P.max_num_gaps_open;
    float my_score = is_mm ? mm_score : is_del ? deletion_score : insertion_score;
    bool gate = is_mm ? (has & !mb_reject<CONT>(rd.thr, P.cutoff, mm_score + lower_bound)) : is_del ? (has & del_ok & gaps_ok) : is_ins ? (ins_ok & gaps_ok) : false;
    if (st.n_hits > 0 && P.bound_kind != BOUND_TEST) gate = gate & !(my_score < st.best_score + P.repr_mm);  // mb_reject_iterative (static inside the step: no child completes the read)
    const uint64_t b = __ballot(gate);
    const uint32_t m_mm = (uint32_t)b & 0xFu, m_del = (uint32_t)(b >> 4) & 0xFu, m_ins = (uint32_t)(b >> 8) & 1u;
    uint32_t cand = m_ins;
#pragma unroll
    for (int i = 0; i < 4; ++i) { cand |= ((m_del >> (3 - i)) & 1u) << (1 + 2 * i); cand |= ((m_mm >> (3 - i)) & 1u) << (2 + 2 * i); }
    const uint32_t my_t = is_mm ? 8u - 2u * (uint32_t)w : is_del ? 7u - 2u * (uint32_t)w : 0u;
    // node (= frame payload) of this lane's child
    const int child_start = forward ? f.start : f.start - 1;
    const uint32_t id0 = st.tree_next;  // == tree_entries: the slab grows at its end
    if (gate) {
        Frame c;
        uint32_t op;
        const uint32_t c_ascii = (0x54474341u >> (8 * cb)) & 0xFFu;  // "ACGT"[cb]
        if (is_ins) {
            c.lower = f.lower; c.lower_rev = f.lower_rev; c.size = f.size; c.start = child_start; c.len = f.len + 1;
            c.gap_f = forward ? (uint32_t)GAP_INS : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_INS; c.ngaps = num_gaps_open;
            op = pack_op(OP_INS, (uint32_t)j, 0);
        } else {
            c.size = x.size; c.lower = forward ? x.lower_rev : x.lower; c.lower_rev = forward ? x.lower : x.lower_rev;  // swapped back for forward extension (:1256)
            if (is_del) {
                c.start = f.start; c.len = f.len;
                c.gap_f = forward ? (uint32_t)GAP_DEL : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_DEL; c.ngaps = num_gaps_open;
                op = pack_op(OP_DEL, (uint32_t)j, c_ascii);
            } else {
                c.start = child_start; c.len = f.len + 1;
                c.gap_f = forward ? (uint32_t)GAP_CLOSED : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_CLOSED; c.ngaps = f.ngaps;
                op = (cb == to_class) ? pack_op(OP_MATCH, (uint32_t)j, 0) : pack_op(OP_MISMATCH, (uint32_t)j, c_ascii);
            }
        }
        A.nodes[id0 + (uint32_t)__popc(cand & ((1u << my_t) - 1u))] = pack_node(op, top.node, c);
    }
    HPROF_MARK(6);
    // pushes, in commit order; the ancestors come from LDS (heap levels 0-9, or the table filled above, kept current by the pushes themselves)
    uint32_t pushed = 0;
    for (uint32_t rest = cand; rest != 0; rest &= rest - 1) {
        const uint32_t t = (uint32_t)__ffs((int)rest) - 1u;
        const uint32_t src = t == 0 ? 8u : (t & 1u) ? (15u - t) >> 1 : (8u - t) >> 1;  // the lane that built child t
        HeapEntry elt;
        elt.score = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_score), (int)src));
        elt.node = id0 + pushed;
        const uint32_t pos0 = n1 + pushed;
        auto rd_anc = [&](uint32_t d) -> HeapEntry {  // d-th ancestor of pos0
            const uint32_t s = anc_slot(pos0, d);
            if (s < (uint32_t)kHeavyTop) return load_entry(A.top + s);
            return load_entry(S.anc + d * 8 + (s - anc_slot(n1, d)));
        };
        auto wr = [&](uint32_t d, const HeapEntry e) {  // the slot d levels above pos0 (0: pos0 itself)
            const uint32_t s = d ? anc_slot(pos0, d) : pos0;
            if (s < (uint32_t)kHeavyTop) { store_entry(A.top + s, e); return; }
            if (d) store_entry(S.anc + d * 8 + (s - anc_slot(n1, d)), e);
            if (lane == 0) store_entry(A.heap + s, e);
        };
        // bubble-up of the crate (search_core.hpp: mm_bubble_up): all comparisons strict
        const uint32_t lvl = heap_level(pos0);
        const bool min_level = (lvl & 1u) == 0;
        uint32_t d = 0;
        bool greater = !min_level;  // which grandparent chain the element follows: on a max level it climbs while greater
        if (pos0 > 0) {
            const HeapEntry e1 = rd_anc(1);
            if (min_level ? (elt.score > e1.score) : (elt.score < e1.score)) { wr(0, e1); d = 1; greater = min_level; }
        }
        while (d + 2 <= lvl) {
            const HeapEntry g = rd_anc(d + 2);
            if (!(greater ? (elt.score > g.score) : (elt.score < g.score))) break;
            wr(d, g);
            d += 2;
        }
        wr(d, elt);
        pushed += 1;
    }
    st.tree_next = id0 + pushed; st.tree_entries = id0 + pushed; st.tree_len += pushed;
    st.heap_len = n1 + pushed;
    st.c_node += pushed; st.c_push += pushed;
    HPROF_MARK(7);
    return HEAVY_CONTINUE;
}

// a full-limit arena of the last stage: one per wavefront, owner words like the size-class pools (shared by all XCDs)
__device__ __forceinline__ uint32_t acquire_slot(const ArenaPool& ap) {
    uint32_t slot = 0;
    if ((threadIdx.x & 63) == 0) {
        uint32_t i = blockIdx.x % ap.n_sets, since = 0;
        for (;;) {
            if (atomicCAS(&ap.set_owner[i], 0u, 1u) == 0u) break;
            if (++i == ap.n_sets) i = 0;
            if (++since == ap.n_sets) { since = 0; __builtin_amdgcn_s_sleep(64); }
        }
        slot = i;
    }
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    __threadfence();
    return slot;
}
__device__ __forceinline__ void release_slot(const ArenaPool& ap, uint32_t slot) {
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) atomicExch(&ap.set_owner[slot], 0u);
}

// MODE 0: continue the reads the quad stages suspended (growable through the size classes, never gives up).
// MODE 1: the reads no size class could hold, from scratch, in arenas with the reference's full limits.
// RDL: the read's position data in LDS; false (full-limit stage of batches with reads beyond 1 024 bp): in the near area of the wavefront's arena.
template <bool CONT, int MODE, bool RDL>
__global__ void __launch_bounds__(64) heavy_kernel(DevIndex ix, DevParams P, BatchDev B, ArenaPool AP, const GrowPools* GP, uint32_t lmax, int tier) {
    static_assert(RDL || MODE == 1, "suspended reads are continued with their position data in LDS");
    const int lane = threadIdx.x & 63;
    // MODE 0: `tier` = the quad stage whose suspended reads this launch continues (its own list and work counter)
    const uint32_t n_items = MODE == 0 ? min(B.cursors[CUR_HEAVY_N + tier], B.heavy_cap) : B.cursors[CUR_OVF + 2 * (tier - 1)];
    uint32_t* work = MODE == 0 ? &B.cursors[CUR_HEAVY_WORK + tier] : &B.cursors[CUR_WORK + 2 * tier];
    if (*(volatile uint32_t*)work >= n_items) return;  // nothing (left) to do: the usual case
    extern __shared__ __attribute__((aligned(16))) uint8_t heavy_lds[];
    MAPAD_LDS uint8_t* lds = (MAPAD_LDS uint8_t*)heavy_lds;
    MAPAD_LDS HeapEntry* top = (MAPAD_LDS HeapEntry*)lds + 1;
    const uint32_t lds_l = RDL ? lmax : 0;
    MAPAD_LDS uint8_t* lds_qc = lds + (kHeavyTop + 1) * sizeof(HeapEntry);
    MAPAD_LDS float* lds_d = (MAPAD_LDS float*)(lds_qc + ((2 * lds_l + 15) & ~15u));
    HeavyScratch S;
    S.spec = (MAPAD_LDS HeapEntry*)((MAPAD_LDS uint8_t*)lds_d + ((4 * lds_l + 15) & ~15u)) + 1;
    S.anc = S.spec - 1 + kSpecEntries;
    const bool use_fast = GP->heavy_fast != 0;  // MAPAD_HEAVY_FAST=0: every step by the general single-lane code (a debugging aid)
    uint32_t slot = 0;
    if (MODE == 1) slot = acquire_slot(AP);
    const uint32_t* items = B.overflow_list + (size_t)(tier > 0 ? tier - 1 : 0) * B.n_reads;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(work, 1u);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= n_items) break;
        HeavyArena A;
        SearchState st;
        uint32_t read;
        bool foreign = false;
        if (MODE == 0) {
            const HeavyItem it = B.heavy[(size_t)tier * B.heavy_cap + item];
            read = it.read; st = it.st;
            const uint32_t cls = (it.grown >> kGrownShift) - 1, idx = it.grown & ((1u << kGrownShift) - 1);
            uint8_t* b = GP->base[cls] + (uint64_t)idx * GP->stride[cls];
            A.heap = (MAPAD_GLOBAL HeapEntry*)(b) + 1;
            A.nodes = (MAPAD_GLOBAL Node*)(b + GP->off_nodes[cls]);
            A.hits = (MAPAD_GLOBAL HitRec*)(b + GP->off_hits[cls]);
            A.hit_ops = (MAPAD_GLOBAL uint32_t*)(b + GP->off_hit_ops[cls]);
            A.scratch = (MAPAD_GLOBAL uint16_t*)(b + GP->off_scratch[cls]);
            A.heap_cap = GP->heap_cap[cls]; A.node_cap = GP->node_cap[cls]; A.hit_ops_cap = GP->hit_ops_cap;
            A.grown = it.grown;
            const uint32_t n_cls = GP->count[cls];
            foreign = n_cls >= kPartitionMin && idx / (n_cls / 8) != xcc_id();
        } else {
            read = items[item] - 1u;
            if (lane == 0) B.overflow_list[(size_t)(tier > 0 ? tier - 1 : 0) * B.n_reads + item] = 0u;  // the list is left as it was found: zeros (search_kernel)
            const ArenaT<false> a = carve<false>(AP, slot);
            A.heap = a.heap; A.nodes = a.nodes; A.hits = a.hits; A.hit_ops = a.hit_ops; A.scratch = a.scratch;
            A.heap_cap = a.heap_cap; A.node_cap = a.node_cap; A.hit_ops_cap = a.hit_ops_cap; A.grown = 0;
        }
        A.top = top;
        const uint64_t off = B.offsets[read];
        using NearBytes = typename near_ptr<uint8_t, RDL>::type;
        using NearFloats = typename near_ptr<float, RDL>::type;
        NearBytes near_qc;
        NearFloats near_d;
        if constexpr (RDL) { near_qc = lds_qc; near_d = lds_d; }
        else {  // the arena's near area: [kTop + 1 heap slots (unused here)][2 bytes per position][D array], as search_kernel lays it out
            uint8_t* nb = AP.base + (uint64_t)slot * AP.stride + AP.off_near + (kTop + 1) * sizeof(HeapEntry);
            near_qc = nb; near_d = (float*)(nb + ((2 * lmax + 15) & ~15u));
        }
        ReadInT<RDL> rd{near_qc, near_d, 0, 0.0f, 0};
        rd.lane_less = (lane & 3) == 0 ? ix.less[1] : (lane & 3) == 1 ? ix.less[2] : (lane & 3) == 2 ? ix.less[3] : ix.less[4];
        rd.L = (int)(B.offsets[read + 1] - off);
        rd.thr = P.reject_thr[rd.L];
        rd.table = P.table_base[rd.L];
        read_setup(B.seqs + off, B.quals + off, B.d_arrays + off, rd.L, near_qc, near_d, lane, 64);
        if (MODE == 0) {  // heap levels 0-9 into LDS (the quad left its 63 near entries in the arena's first slots)
            const uint32_t n_top = st.heap_len < (uint32_t)kHeavyTop ? st.heap_len : (uint32_t)kHeavyTop;
            for (uint32_t i = lane; i < n_top; i += 64) store_entry(top + i, load_entry(A.heap + i));
        } else if (lane == 0) {
            SearchState tmp;
            search_init(ix.n, alignment_start_of(P, rd.L), rd, A, tmp);
            st = tmp;
        }
        if (MODE == 1) {  // the state lives in every lane
            uint32_t* sw = (uint32_t*)&st;
#pragma unroll
            for (int k = 0; k < (int)(sizeof(SearchState) / 4); ++k) sw[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sw[k]);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const DeviceGrow<64, true, kHeavyTop, true> grow{GP, &B.cursors[CUR_GROWN], blockIdx.x, lane, false, foreign};
        uint32_t pops0 = st.c_pop;
        bool cont = true;
#if defined(MAPAD_HEAVY_PROF)
        HeavyProf hp;
        for (auto& a : hp.acc) a = 0;
        hp.t = __builtin_amdgcn_s_memtime();
#endif
        while (cont) {
            if (MODE == 0 && (st.tree_len + kStepNodes > A.node_cap || st.heap_len + kStepNodes > A.heap_cap)) {  // wavefront-wide migration into the next size class
                const int g = grow(A, st);
                if (g == GROW_WAIT) {
                    // Every suitable arena is taken.  Its holders are wavefronts that will finish — or suspended reads no wavefront has picked up yet,
                    // which cannot move while all wavefronts wait: after a long wait the read goes to the full-limit stage (restarted there).
                    A.wait = 0;
                    if (++A.n_waits > GP->max_waits * 256u + 4096u) { st.status = ST_ARENA_OVERFLOW; break; }
                    __builtin_amdgcn_s_sleep(127);
                    continue;
                }
                if (g == GROW_NEVER || st.tree_len + kStepNodes > A.node_cap || st.heap_len + kStepNodes > A.heap_cap) { st.status = ST_ARENA_OVERFLOW; break; }
                __builtin_amdgcn_s_waitcnt(0);
            }
            if (st.heap_len == 0 || st.status != ST_OK) break;
#if defined(MAPAD_HEAVY_PROF)
            const int r = use_fast ? heavy_step<CONT>(ix, P, rd, A, st, lane, S, hp) : (int)HEAVY_GENERAL;
#else
            const int r = use_fast ? heavy_step<CONT>(ix, P, rd, A, st, lane, S) : (int)HEAVY_GENERAL;
#endif
            if (r == HEAVY_DONE) break;
            uint32_t c = 1;
            bool one_lane = false;
            if (r == HEAVY_GENERAL) {  // a child may complete the read, or the slab has a free list: the general step, by one lane
                if (lane == 0) c = search_step<1, CONT, true>(ix, P, rd, A, st, 0, NoGrow()) ? 1u : 0u;
                one_lane = true;
            } else if (MAPAD_UNLIKELY((st.heap_len > P.stack_limit) | (st.tree_len > P.edit_tree_limit))) {  // :1358-1380
                if (P.stack_limit_abort) { st.status = ST_LIMIT_ABORT; break; }
                if (lane == 0) {
                    const int64_t a = (int64_t)st.heap_len - (int64_t)P.stack_limit, b = (int64_t)st.tree_len - (int64_t)P.edit_tree_limit;
                    SearchState tmp = st;
                    evict_worst(A, tmp, a > b ? a : b);
                    st = tmp;
                    c = st.heap_len > 0 ? 1u : 0u;
                }
                one_lane = true;
            }
            if (one_lane) {  // the state lives in every lane
                __builtin_amdgcn_s_waitcnt(0);
                uint32_t* sw = (uint32_t*)&st;
#pragma unroll
                for (int k = 0; k < (int)(sizeof(SearchState) / 4); ++k) sw[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sw[k]);
                cont = __builtin_amdgcn_readfirstlane((int)c) != 0;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd((unsigned long long*)(B.cursors + CUR_HEAVY_POPS), (unsigned long long)(st.c_pop - pops0));
#if defined(MAPAD_HEAVY_PROF)
        if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)(B.cursors + CUR_HPROF) + k, hp.acc[k]);
#endif
        finalize_read<64>(B, rd, A, st, read, lane, MODE == 0 ? 1 : tier);  // MODE 0: a read that no class can hold goes on to the full-limit stage (list 1)
        if (MODE == 0) release_grown<64>(GP, A.grown, lane, grow.foreign);
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (MODE == 1) release_slot(AP, slot);
}
