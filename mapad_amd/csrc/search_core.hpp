// search_core.hpp — one read's best-first backtracking search, as run by one quad of a gfx950 wavefront.
//
// Device equivalent of k_mismatch_search + check_and_push_stack_frame (src/map/mapping.rs:932-1383) with
//   * the search frontier = min-max heap (min-max-heap crate semantics) of 8-byte {score, node} entries,
//   * the edit tree = slab (slab 0.4 semantics: LIFO key reuse) whose 40-byte nodes also carry the frame payload
//     (a frame and the tree node created for it are 1:1, mapping.rs:969-971, so the node id doubles as the frame id),
//   * the hit list = Rust std BinaryHeap array order,
// all living in a per-quad arena in HBM.  The four lanes of the quad execute this control flow redundantly on
// quad-uniform values; they split the work only inside ext4_any() (rank queries, fmd_device.hpp).
// The same source compiles for the host (tests/emu) so that the CPU test-suite can run it against the oracle.
#pragma once
#include "fmd_device.hpp"

namespace mapad {

enum : uint8_t { GAP_INS = 0, GAP_DEL = 1, GAP_CLOSED = 2 };  // src/map/mod.rs:93-98

struct HeapEntry {
    float score;
    uint32_t node;
};

struct Node {  // 40 bytes
    uint32_t op;      // packed edit operation that created this frame
    uint32_t parent;  // parent node id (vacant slot: next free key)
    uint64_t lower, lower_rev, size;
    int16_t start, len;
    uint8_t gap_f, gap_b, ngaps, occupied;
};
static_assert(sizeof(Node) == 40, "node + frame payload is 40 bytes");

struct HitRec {  // 40 bytes; the public hit record (include/mapad_amd.h: mapad_hit_t)
    uint64_t lower, lower_rev, size;
    float score;
    uint32_t n_ops;
    uint32_t ops_off;  // into the per-read staging area, later into the global ops pool
    uint32_t pad;
};
static_assert(sizeof(HitRec) == 40, "hit record is 40 bytes");

constexpr int kMaxHits = 20;  // a pop adds <= 9 hits and the search returns once more than 9 exist (mapping.rs:1348)

enum : uint32_t { ST_OK = 0, ST_ARENA_OVERFLOW = 1, ST_LIMIT_ABORT = 2 };

struct Arena {
    HeapEntry* heap;
    Node* nodes;
    HitRec* hits;       // kMaxHits
    uint32_t* hit_ops;  // staging for the hits' edit tracks
    uint16_t* scratch;  // 2 * (Lmax + 1) u16 for the bucket sort of extract_edit_operations
    uint32_t heap_cap, node_cap, hit_ops_cap;
};

struct ReadIn {
    const uint8_t* seq;
    const uint8_t* qual;
    const float* d;  // D array of this read (darray_core.hpp)
    int L;
};

struct SearchState {
    uint32_t heap_len;
    uint32_t tree_entries, tree_next, tree_len;  // slab: backing length, free-list head, occupied count
    uint32_t n_hits, hit_ops_used;
    uint32_t status;
    ReadCounters ctr;
};

MAPAD_HD void ext4_any(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, int w, Ext4& out) {
#if defined(__HIP_DEVICE_COMPILE__)
    ext4_quad(ix, lower, lower_rev, size, w, out);
#else
    (void)w;
    ext4_scalar(ix, lower, lower_rev, size, out);
#endif
}

// ---- min-max heap (index 0 = min; even levels are min levels) ----------------------------------------------------
MAPAD_HD bool mm_is_min_level(uint32_t pos) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__clz((int)(pos + 1)) & 1) == 1;
#else
    return (__builtin_clz(pos + 1) & 1) == 1;
#endif
}

MAPAD_HD void mm_bubble_up(HeapEntry* v, uint32_t pos) {
    const HeapEntry elt = v[pos];
    bool greater;  // which grandparent chain to follow
    if (pos == 0) return;
    {
        const uint32_t parent = (pos - 1) >> 1;
        const HeapEntry pe = v[parent];
        if (mm_is_min_level(pos)) {
            if (elt.score > pe.score) { v[pos] = pe; pos = parent; greater = true; } else greater = false;
        } else {
            if (elt.score < pe.score) { v[pos] = pe; pos = parent; greater = false; } else greater = true;
        }
    }
    while (pos > 2) {
        const uint32_t gp = (pos - 3) >> 2;
        const HeapEntry ge = v[gp];
        const bool go = greater ? (elt.score > ge.score) : (elt.score < ge.score);
        if (!go) break;
        v[pos] = ge;
        pos = gp;
    }
    v[pos] = elt;
}

// candidates scanned in ascending index order (child1, child2, grandchildren); a later one wins only if strictly better
template <bool MAX>
MAPAD_HD void mm_trickle_down(HeapEntry* v, uint32_t n, uint32_t pos) {
    HeapEntry elt = v[pos];
    while (2 * pos + 1 < n) {
        const uint32_t c1 = 2 * pos + 1;
        uint32_t best = c1;
        HeapEntry be = v[c1];
        bool grandchild = false;
        {
            const uint32_t idx[5] = {c1 + 1, 2 * c1 + 1, 2 * c1 + 2, 2 * c1 + 3, 2 * c1 + 4};
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (idx[t] < n) {
                    const HeapEntry e = v[idx[t]];
                    if (MAX ? (e.score > be.score) : (e.score < be.score)) { best = idx[t]; be = e; grandchild = t > 0; }
                }
            }
        }
        if (!(MAX ? (be.score > elt.score) : (be.score < elt.score))) break;
        v[pos] = be;
        pos = best;
        if (!grandchild) break;
        const uint32_t parent = (pos - 1) >> 1;
        const HeapEntry pe = v[parent];
        if (MAX ? (pe.score > elt.score) : (pe.score < elt.score)) { v[parent] = elt; elt = pe; }
    }
    v[pos] = elt;
}

MAPAD_HD HeapEntry mm_pop_max(HeapEntry* v, uint32_t& n) {
    uint32_t idx;
    if (n == 1) idx = 0;
    else if (n == 2) idx = 1;
    else idx = (v[1].score > v[2].score) ? 1 : 2;
    const HeapEntry last = v[n - 1];
    n -= 1;
    if (idx < n) {
        const HeapEntry item = v[idx];
        v[idx] = last;
        mm_trickle_down<true>(v, n, idx);
        return item;
    }
    return last;
}
MAPAD_HD HeapEntry mm_pop_min(HeapEntry* v, uint32_t& n) {
    const HeapEntry last = v[n - 1];
    n -= 1;
    if (n > 0) {
        const HeapEntry item = v[0];
        v[0] = last;
        mm_trickle_down<false>(v, n, 0);
        return item;
    }
    return last;
}

// ---- slab tree ----------------------------------------------------------------------------------------------------
MAPAD_HD uint32_t tree_insert(Node* nodes, SearchState& st, const Node& nd) {
    const uint32_t key = st.tree_next;
    if (key == st.tree_entries) { st.tree_entries += 1; st.tree_next = key + 1; }
    else st.tree_next = nodes[key].parent;  // vacant slot stores the next free key
    nodes[key] = nd;
    st.tree_len += 1;
    return key;
}
MAPAD_HD void tree_remove(Node* nodes, SearchState& st, uint32_t key) {  // backtrack_tree.rs:50-54
    if (key == 0) return;
    Node nd = nodes[key];
    nd.occupied = 0;
    nd.parent = st.tree_next;
    nodes[key] = nd;
    st.tree_next = key;
    st.tree_len -= 1;
}

// ---- extract_edit_operations (src/map/record.rs:465-500) -------------------------------------------------------------
// Walk leaf -> root (root excluded, stop at a vacant slot), bucket by read position ascending; a bucket keeps walk order
// if its position is left of the alignment start, else it is reversed.  Counting sort over positions 0..L.
MAPAD_HD uint32_t extract_ops(const Node* nodes, uint32_t end_node, int alignment_start, int L, uint16_t* scratch, uint32_t* out,
                              uint32_t out_cap) {
    uint16_t* cnt = scratch;          // [L + 1]
    uint16_t* fill = scratch + L + 1;  // [L + 1]
    for (int i = 0; i <= L; ++i) { cnt[i] = 0; fill[i] = 0; }
    uint32_t m = 0;
    for (uint32_t s = end_node; s != 0;) {
        const Node nd = nodes[s];
        if (!nd.occupied) break;
        cnt[nd.op & 0xFFFFu] += 1;
        m += 1;
        s = nd.parent;
    }
    if (m > out_cap) return 0xFFFFFFFFu;
    // exclusive prefix sum in place (cnt[p] -> offset of bucket p; bucket size kept in fill as negative space below)
    uint32_t acc = 0;
    for (int i = 0; i <= L; ++i) { const uint32_t c = cnt[i]; cnt[i] = (uint16_t)acc; fill[i] = (uint16_t)c; acc += c; }
    // second walk: place.  For reversed buckets fill from the back.
    for (uint32_t s = end_node; s != 0;) {
        const Node nd = nodes[s];
        if (!nd.occupied) break;
        const uint32_t p = nd.op & 0xFFFFu;
        uint32_t slot;
        if ((int)p < alignment_start) { slot = cnt[p]; cnt[p] += 1; }
        else { fill[p] -= 1; slot = (uint32_t)cnt[p] + fill[p]; }
        out[slot] = nd.op;
        s = nd.parent;
    }
    return m;
}

// ---- hit list: Rust BinaryHeap::push (sift_up moves while strictly greater than the parent) -----------------------------------
MAPAD_HD void hits_push(HitRec* hits, uint32_t& n, const HitRec& h) {
    uint32_t pos = n;
    n += 1;
    while (pos > 0) {
        const uint32_t parent = (pos - 1) >> 1;
        const HitRec pe = hits[parent];
        if (h.score <= pe.score) break;
        hits[pos] = pe;
        pos = parent;
    }
    hits[pos] = h;
}

// ---- D array access (src/map/bi_d_array.rs:200-224) ------------------------------------------------------------------------
MAPAD_HD float d_get(const float* d, int L, int split, int backward_index, int forward_index) {
    float d_rev = 0.0f, d_fwd = 0.0f;
    if (backward_index >= 0 && backward_index < L) d_rev = d[backward_index];
    const int sub = 1 + forward_index;
    if (L >= sub) {
        const int idx = (L - sub) + split;
        if (idx < L) d_fwd = d[idx];
    }
    return d_rev + d_fwd;
}

struct ChildFrame {
    uint64_t lower, lower_rev, size;
    int16_t start, len;
    uint8_t gap_f, gap_b, ngaps;
    float score;
};

// check_and_push_stack_frame (mapping.rs:932-987)
MAPAD_HD void check_and_push(const DevParams& P, const ReadIn& rd, const Arena& A, SearchState& st, int alignment_start,
                             const ChildFrame& c, uint32_t parent_node, uint32_t op) {
    if (st.status != ST_OK) return;
    if (st.n_hits > 0 && mb_reject_iterative(P, c.score, A.hits[0].score)) return;
    if ((int)c.ngaps > P.max_num_gaps_open) return;
    if (st.tree_next == st.tree_entries && st.tree_entries >= A.node_cap) { st.status = ST_ARENA_OVERFLOW; return; }
    Node nd;
    nd.op = op; nd.parent = parent_node;
    nd.lower = c.lower; nd.lower_rev = c.lower_rev; nd.size = c.size;
    nd.start = c.start; nd.len = c.len; nd.gap_f = c.gap_f; nd.gap_b = c.gap_b; nd.ngaps = c.ngaps; nd.occupied = 1;
    const uint32_t id = tree_insert(A.nodes, st, nd);
    st.ctr.n_node += 1;
    if ((int)c.len == rd.L) {
        if (st.n_hits >= (uint32_t)kMaxHits) { st.status = ST_ARENA_OVERFLOW; return; }
        HitRec h;
        h.lower = c.lower; h.lower_rev = c.lower_rev; h.size = c.size; h.score = c.score; h.pad = 0;
        h.ops_off = st.hit_ops_used;
        const uint32_t m = extract_ops(A.nodes, id, alignment_start, rd.L, A.scratch, A.hit_ops + st.hit_ops_used, A.hit_ops_cap - st.hit_ops_used);
        if (m == 0xFFFFFFFFu) { st.status = ST_ARENA_OVERFLOW; return; }
        h.n_ops = m;
        st.hit_ops_used += m;
        hits_push(A.hits, st.n_hits, h);
        st.ctr.n_hits += 1;
        return;
    }
    if (st.heap_len >= A.heap_cap) { st.status = ST_ARENA_OVERFLOW; return; }
    A.heap[st.heap_len] = HeapEntry{c.score, id};
    st.heap_len += 1;
    mm_bubble_up(A.heap, st.heap_len - 1);
    st.ctr.n_push += 1;
}

// k_mismatch_search (mapping.rs:1012-1383) after the D array has been computed, split into init / step so that a
// persistent quad can fetch its next read as soon as the current one finishes.  `w` = lane index inside the quad.
MAPAD_HD int alignment_start_of(const DevParams& P, int L) { return P.start_at_end ? L : (L / 2); }  // find_alignment_start

MAPAD_HD void search_init(const DevIndex& ix, const DevParams& P, const ReadIn& rd, const Arena& A, SearchState& st) {
    st.heap_len = 0; st.tree_entries = 0; st.tree_next = 0; st.tree_len = 0; st.n_hits = 0; st.hit_ops_used = 0; st.status = ST_OK;
    st.ctr.e_search = 0; st.ctr.n_push = 0; st.ctr.n_pop = 0; st.ctr.n_node = 0; st.ctr.n_hits = 0;
    Node root;  // Tree::clear(): id 0
    root.op = pack_op(OP_MATCH, 0, 0); root.parent = 0;
    root.lower = 0; root.lower_rev = 0; root.size = ix.n;  // init_interval
    root.start = (int16_t)alignment_start_of(P, rd.L); root.len = 0; root.gap_f = GAP_CLOSED; root.gap_b = GAP_CLOSED; root.ngaps = 0; root.occupied = 1;
    tree_insert(A.nodes, st, root);
    A.heap[0] = HeapEntry{0.0f, 0u};
    st.heap_len = 1;
    st.ctr.n_push += 1;
}

// One iteration of the `while let Some(stack_frame) = stack.pop_max()` loop.  Returns false when the search is over.
MAPAD_HD bool search_step(const DevIndex& ix, const DevParams& P, const ReadIn& rd, const Arena& A, SearchState& st, int w) {
    if (st.heap_len == 0 || st.status != ST_OK) return false;
    const int L = rd.L;
    const int alignment_start = alignment_start_of(P, L);
    const float open_ext = P.gap_open + P.gap_extend;
    const HeapEntry top = mm_pop_max(A.heap, st.heap_len);
    st.ctr.n_pop += 1;
    const Node f = A.nodes[top.node];
    const float f_score = top.score;
    int j, d_k, d_l;
    bool forward;
    if ((int)f.start <= L - (int)f.start - (int)f.len) {  // :1077-1097
        j = f.start + f.len; forward = true; d_k = f.start; d_l = f.start + f.len;
    } else {
        j = f.start - 1; forward = false; d_k = f.start - 1; d_l = f.start + f.len - 1;
    }
    const int to_class = base_index(rd.seq[j]);
    const Float4 row = sdm_row(P, L, j, rd.qual[j], to_class);
    const float optimal = sdm_optimal(row, to_class);
    const uint8_t gap_side = forward ? f.gap_f : f.gap_b;
    const float insertion_score = (gap_side == GAP_INS ? P.gap_extend : open_ext) + f_score;  // :1127-1136,1165-1174
    const float deletion_score = (gap_side == GAP_DEL ? P.gap_extend : open_ext) + f_score;
    const uint8_t num_gaps_open = gap_side == GAP_CLOSED ? (uint8_t)(f.ngaps + 1) : f.ngaps;     // :1148-1152
    const float lower_bound = d_get(rd.d, L, alignment_start, d_k, d_l);                        // :1195
    if (st.n_hits > 0 && mb_reject_iterative(P, f_score + lower_bound, A.hits[0].score)) return false;  // :1201-1208

    const int16_t child_start = forward ? f.start : (int16_t)(f.start - 1);
    // Insertion in read (:1213-1242)
    {
        const int dist = j < (L - j - 1) ? j : (L - j - 1);
        if (!mb_reject(P, insertion_score + lower_bound, L) && dist >= P.gap_dist_ends) {
            ChildFrame c;
            c.lower = f.lower; c.lower_rev = f.lower_rev; c.size = f.size;
            c.start = child_start; c.len = (int16_t)(f.len + 1);
            c.gap_f = forward ? (uint8_t)GAP_INS : f.gap_f; c.gap_b = forward ? f.gap_b : (uint8_t)GAP_INS;
            c.ngaps = num_gaps_open; c.score = insertion_score;
            check_and_push(P, rd, A, st, alignment_start, c, top.node, pack_op(OP_INS, (uint32_t)j, 0));
        }
    }
    // Extension (:1245-1339); forward extension works on the swapped interval
    Ext4 e;
    if (forward) ext4_any(ix, f.lower_rev, f.lower, f.size, w, e);
    else ext4_any(ix, f.lower, f.lower_rev, f.size, w, e);
    st.ctr.e_search += 1;
    const int dist5 = forward ? j : j + 1;
    const int dist3 = L - dist5;
    const int del_dist = dist5 < dist3 ? dist5 : dist3;
    const bool del_ok = !mb_reject(P, deletion_score + lower_bound, L) && del_dist >= P.gap_dist_ends;
    for (int k = 3; k >= 0; --k) {  // iterator order T, G, C, A
        if (e.size[k] < 1) continue;
        // symbol in read orientation: backward = base k, forward = its complement (3 - k); interval swapped back
        const int cb = forward ? 3 - k : k;
        const uint32_t c_ascii = cb == 0 ? 'A' : cb == 1 ? 'C' : cb == 2 ? 'G' : 'T';
        const uint64_t lo = forward ? e.lower_rev[k] : e.lower[k];
        const uint64_t lr = forward ? e.lower[k] : e.lower_rev[k];
        if (del_ok) {  // Deletion in read (:1265-1302)
            ChildFrame c;
            c.lower = lo; c.lower_rev = lr; c.size = e.size[k];
            c.start = f.start; c.len = f.len;
            c.gap_f = forward ? (uint8_t)GAP_DEL : f.gap_f; c.gap_b = forward ? f.gap_b : (uint8_t)GAP_DEL;
            c.ngaps = num_gaps_open; c.score = deletion_score;
            check_and_push(P, rd, A, st, alignment_start, c, top.node, pack_op(OP_DEL, (uint32_t)j, c_ascii));
        }
        // Match / mismatch (:1307-1338).  mm_scores: get(from, pattern[j]) - optimal + score, `from` = reference base
        const float mm = f4_get(row, cb) - optimal + f_score;
        if (!mb_reject(P, mm + lower_bound, L)) {
            ChildFrame c;
            c.lower = lo; c.lower_rev = lr; c.size = e.size[k];
            c.start = child_start; c.len = (int16_t)(f.len + 1);
            c.gap_f = forward ? (uint8_t)GAP_CLOSED : f.gap_f; c.gap_b = forward ? f.gap_b : (uint8_t)GAP_CLOSED;
            c.ngaps = f.ngaps; c.score = mm;
            const uint32_t op = (c_ascii == rd.seq[j]) ? pack_op(OP_MATCH, (uint32_t)j, 0) : pack_op(OP_MISMATCH, (uint32_t)j, c_ascii);
            check_and_push(P, rd, A, st, alignment_start, c, top.node, op);
        }
    }
    if (st.status != ST_OK) return false;
    // :1348-1355
    if (st.n_hits > 9 || (st.n_hits > 0 && A.hits[0].size > 1)) return false;
    // :1358-1380
    if (st.heap_len > P.stack_limit || st.tree_len > P.edit_tree_limit) {
        if (P.stack_limit_abort) { st.status = ST_LIMIT_ABORT; return false; }
        const int64_t a = (int64_t)st.heap_len - (int64_t)P.stack_limit;
        const int64_t b = (int64_t)st.tree_len - (int64_t)P.edit_tree_limit;
        const int64_t cnt = a > b ? a : b;
        for (int64_t i = 0; i < cnt; ++i) {
            if (st.heap_len == 0) break;
            const HeapEntry m = mm_pop_min(A.heap, st.heap_len);
            tree_remove(A.nodes, st, m.node);
        }
    }
    return st.heap_len > 0;
}

MAPAD_HD void search_read(const DevIndex& ix, const DevParams& P, const ReadIn& rd, const Arena& A, SearchState& st, int w) {
    search_init(ix, P, rd, A, st);
    while (search_step(ix, P, rd, A, st, w)) {}
}

}  // namespace mapad
