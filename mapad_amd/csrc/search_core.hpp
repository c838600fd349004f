// search_core.hpp — one read's best-first backtracking search, as run by one quad of a gfx950 wavefront.
//
// Device equivalent of k_mismatch_search + check_and_push_stack_frame (src/map/mapping.rs:932-1383) with
//   * the search frontier = min-max heap (min-max-heap crate semantics) of 8-byte {score, node} entries,
//   * the edit tree = slab (slab 0.4 semantics: LIFO key reuse) whose 32-byte nodes also carry the frame payload
//     (a frame and the tree node created for it are 1:1, mapping.rs:969-971, so the node id doubles as the frame id),
//   * the hit list = Rust std BinaryHeap array order,
// living in the read slot's arena in HBM, except for the top 31 heap entries, the read's position data and its D array, which are
// "near" data (LDS on the device).  One pop costs: heap sift + one 32-byte node + one 16-byte score-table row + two 64-byte index
// blocks.  The four lanes of the quad execute the control flow on quad-uniform values; they split the work in the rank queries
// (fmd_device.hpp: lane w counts sub-block w and keeps the extension by base w) and in building the children of a frame (lane w packs
// the deletion and match/mismatch nodes of base w; search_step / commit_child).
// The same source compiles for the host (tests/emu) so that the CPU test-suite can run it against the oracle.
#pragma once
#include "heap_core.hpp"

namespace mapad {

struct SearchState {
    uint32_t c_esearch, c_push, c_pop, c_node, c_hits;  // event counters of the read (identical on the oracle: itself a parity check)
    uint32_t heap_len;
    uint32_t tree_entries, tree_next, tree_len;  // slab: backing length, free-list head, occupied count
    uint32_t n_hits, hit_ops_used;
    uint32_t status;
    float best_score;     // hits[0] (BinaryHeap::peek) kept in registers
    uint64_t best_size;
};

// LPR = lanes per read: 4 = the quad splits every rank query (one coalesced 64-byte block per query);
//                      1 = every lane owns a read and answers its own rank queries (more reads per instruction issued).
template <int LPR>
MAPAD_HD void ext4_any(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, int w, Ext4& out) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (LPR == 4) ext4_quad(ix, lower, lower_rev, size, w, out);
    else ext4_scalar(ix, lower, lower_rev, size, out);
#else
    (void)w;
    ext4_scalar(ix, lower, lower_rev, size, out);
#endif
}

// ---- slab tree ----------------------------------------------------------------------------------------------------
template <class NP>
MAPAD_HD uint32_t tree_insert(NP nodes, SearchState& st, const Node& nd) {
    const uint32_t key = st.tree_next;
    if (key == st.tree_entries) { st.tree_entries += 1; st.tree_next = key + 1; }
    else { MAPAD_TOUCH(&nodes[key], 8, false); st.tree_next = node_parent(nodes[key]); }  // vacant slot stores the next free key
    MAPAD_TOUCH(&nodes[key], sizeof(Node), true);
    nodes[key] = nd;
    st.tree_len += 1;
    return key;
}
// tree_insert without the store: the caller writes the node (one lane of the quad owns it, search_step)
// The free list is only ever non-empty after an overflow eviction (mapping.rs:1371-1379).  On the device its load is wrapped in an asm block
// that waits for itself: left to the compiler, the load's destination register is shared with the common path's `key + 1`, and the wait-count
// pass then guards that register with a full `s_waitcnt vmcnt(0)` on EVERY allocation — a drain of all stores in flight per pushed child
// (measured: 29 % of the wave time of a step sat in front of this wait).
template <class NP>
MAPAD_HD uint32_t free_list_next(NP nodes, uint32_t key) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    const MAPAD_GLOBAL Node* p = (const MAPAD_GLOBAL Node*)(nodes + key);
    asm volatile("global_load_dword %0, %1, off offset:4\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
#else
    MAPAD_TOUCH(&nodes[key], 8, false);
    return node_parent(nodes[key]);
#endif
}
template <class NP>
MAPAD_HD uint32_t tree_alloc(NP nodes, SearchState& st) {
    const uint32_t key = st.tree_next;
    if (MAPAD_UNLIKELY(key != st.tree_entries)) st.tree_next = free_list_next(nodes, key);
    else { st.tree_entries += 1; st.tree_next = key + 1; }
    st.tree_len += 1;
    return key;
}
template <class NP>
MAPAD_HD void tree_remove(NP nodes, SearchState& st, uint32_t key) {  // backtrack_tree.rs:50-54
    if (key == 0) return;
    MAPAD_TOUCH(&nodes[key], sizeof(Node), true);
    Node nd = nodes[key];
    nd.w2 &= ~(1ull << 52);
    nd.w0 = (nd.w0 & 0xFFFFFFFFull) | ((uint64_t)st.tree_next << 32);
    nodes[key] = nd;
    st.tree_next = key;
    st.tree_len -= 1;
}

// ---- extract_edit_operations (src/map/record.rs:465-500) -------------------------------------------------------------
// Walk leaf -> root (root excluded, stop at a vacant slot), bucket by read position ascending; a bucket keeps walk order
// if its position is left of the alignment start, else it is reversed.  Counting sort over positions 0..L.
template <class NP, class SP, class OP>
MAPAD_RARE uint32_t extract_ops_general(NP nodes, uint32_t end_node, int alignment_start, int L, SP scratch, OP out,
                                      uint32_t out_cap) {
    SP cnt = scratch;           // [L + 1]
    SP fill = scratch + L + 1;  // [L + 1]
    for (int i = 0; i <= L; ++i) { cnt[i] = 0; fill[i] = 0; }
    uint32_t m = 0;
    for (uint32_t s = end_node; s != 0;) {
        MAPAD_TOUCH(&nodes[s], sizeof(Node), false);
        const Node nd = nodes[s];
        if (!node_occupied(nd)) break;
        cnt[node_op(nd) & 0xFFFFu] += 1;
        m += 1;
        s = node_parent(nd);
    }
    if (m > out_cap) return 0xFFFFFFFFu;
    uint32_t acc = 0;  // exclusive prefix sum: cnt[p] -> first slot of bucket p, fill[p] -> bucket size
    for (int i = 0; i <= L; ++i) { const uint32_t c = cnt[i]; cnt[i] = (uint16_t)acc; fill[i] = (uint16_t)c; acc += c; }
    for (uint32_t s = end_node; s != 0;) {  // second walk: place; reversed buckets fill from the back
        const Node nd = nodes[s];
        if (!node_occupied(nd)) break;
        const uint32_t op = node_op(nd), p = op & 0xFFFFu;
        uint32_t slot;
        if ((int)p < alignment_start) { slot = cnt[p]; cnt[p] += 1; }
        else { fill[p] -= 1; slot = (uint32_t)cnt[p] + fill[p]; }
        out[slot] = op;
        s = node_parent(nd);
    }
    return m;
}
#if !defined(MAPAD_QUAD_WALK)
#define MAPAD_QUAD_WALK 1  // (0 for A/B runs)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// The dense walk by a quad (round 6).  A leaf-to-root walk is a chain of ~L dependent loads — 0.6 us each on the loaded chip, ~30 us per hit, during which the other
// 15 read slots of the wavefront idle (section profile: 3.4 % of the wave time at C4, 4.6 % at C2).  But a search that runs straight down a matching stretch pops the
// child it has just pushed, and that child's children take the slab's next keys: along such stretches parent = child - 1 ... child - 9 (dense slab: keys only grow,
// a parent's key is below its child's).  So lane w of the quad loads the words of nodes s - w, s - w - 4, s - w - 8, s - w - 12 — one trip for the 16 nodes below
// s —, and the walk hops through registers (a quad broadcast per hop) until it leaves that window.  Same writes, same order, same `in_order` test as the loop above.
template <class NP, class OP>
__device__ __forceinline__ uint32_t extract_ops_quad(NP nodes, uint32_t end_node, int alignment_start, OP out, uint32_t out_cap, int w, bool& in_order) {
    uint32_t m = 0, prev = 0, s = end_node;
    in_order = true;
    while (s != 0) {
        const uint32_t top = s, b = s - (uint32_t)w;  // (b is only used where s >= w + 4 i)
        uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (top >= (uint32_t)w) w0 = nodes[b].w0;
        if (top >= (uint32_t)w + 4u) w1 = nodes[b - 4u].w0;
        if (top >= (uint32_t)w + 8u) w2 = nodes[b - 8u].w0;
        if (top >= (uint32_t)w + 12u) w3 = nodes[b - 12u].w0;
        do {
            const uint32_t d = top - s;  // 0 .. 15, quad-uniform
            const uint32_t i = d >> 2;
            const uint64_t mine = i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : w3;
            const uint64_t x = quad_pick64(mine, (int)(d & 3u));
            const uint32_t op = (uint32_t)x, p = op & 0xFFFFu;
            in_order = in_order && p >= prev && (int)p < alignment_start;
            prev = p;
            if (m < out_cap) out[m] = op;
            m += 1;
            s = (uint32_t)(x >> 32);
        } while (s != 0 && top - s < 16u);
    }
    return m;
}
#endif

// The walk is a chain of dependent loads (one arena round trip per edit operation) during which the other read slots of the wavefront
// idle, so it is done once where that is enough: with the production model the alignment starts at the 3' end (every position is left of
// the alignment start) and the search only moves backward, so positions do not decrease from leaf to root and the bucket order of
// record.rs:491-496 IS the walk order.  The walk checks exactly that while it writes; anything else (the bidirectional test models)
// falls back to the counting sort.  `dense`: the slab has no vacant entries, so the occupancy word of a node need not be loaded.
template <class NP, class SP, class OP>
MAPAD_RARE uint32_t extract_ops(NP nodes, uint32_t end_node, int alignment_start, int L, SP scratch, OP out, uint32_t out_cap, bool dense, int quad_lane = -1) {
#if defined(__HIP_DEVICE_COMPILE__) && MAPAD_QUAD_WALK
    if (dense && quad_lane >= 0) {
        bool in_order;
        const uint32_t m = extract_ops_quad(nodes, end_node, alignment_start, out, out_cap, quad_lane, in_order);
        if (m > out_cap) return 0xFFFFFFFFu;
        if (in_order) return m;
        dense = false;  // (the bidirectional test models: the counting sort below)
    }
#endif
    (void)quad_lane;
    if (dense) {
        uint32_t m = 0, prev = 0;
        bool in_order = true;
        for (uint32_t s = end_node; s != 0;) {
            MAPAD_TOUCH(&nodes[s], 8, false);
            const uint64_t w0 = nodes[s].w0;
            const uint32_t op = (uint32_t)w0, p = op & 0xFFFFu;
            in_order = in_order && p >= prev && (int)p < alignment_start;
            prev = p;
            if (m < out_cap) { MAPAD_TOUCH(&out[m], 4, true); out[m] = op; }
            m += 1;
            s = (uint32_t)(w0 >> 32);
        }
        if (m > out_cap) return 0xFFFFFFFFu;
        if (in_order) return m;
    }
    return extract_ops_general(nodes, end_node, alignment_start, L, scratch, out, out_cap);
}

// ---- hit list: Rust BinaryHeap::push (sift_up moves while strictly greater than the parent) -----------------------------------
template <class HP>
MAPAD_HD void hits_push(HP hits, uint32_t& n, const HitRec& h) {
    uint32_t pos = n;
    n += 1;
    while (pos > 0) {
        const uint32_t parent = (pos - 1) >> 1;
        MAPAD_TOUCH(&hits[parent], sizeof(HitRec), false);
        const HitRec pe = hits[parent];
        if (h.score <= pe.score) break;
        MAPAD_TOUCH(&hits[pos], sizeof(HitRec), true);
        hits[pos] = pe;
        pos = parent;
    }
    MAPAD_TOUCH(&hits[pos], sizeof(HitRec), true);
    hits[pos] = h;
}

// ---- D array access (src/map/bi_d_array.rs:200-224) ------------------------------------------------------------------------
template <class DPtr>
MAPAD_HD float d_get(DPtr d, int L, int split, int backward_index, int forward_index) {
    // both reads unconditionally from clamped indices (L >= 1: index 0 exists), the range checks as selects: no branches around two LDS reads
    const bool rev_ok = (backward_index >= 0) & (backward_index < L);
    const int sub = 1 + forward_index, idx = (L - sub) + split;
    const bool fwd_ok = (L >= sub) & (idx < L) & (idx >= 0);
    const float v_rev = d[rev_ok ? backward_index : 0], v_fwd = d[fwd_ok ? idx : 0];
    const float d_rev = rev_ok ? v_rev : 0.0f, d_fwd = fwd_ok ? v_fwd : 0.0f;
    return d_rev + d_fwd;
}

// Fills the quad's per-read position data; lane w of LPR lanes handles positions w, w + LPR, ...
template <class QcPtr, class DPtr>
MAPAD_HD void read_setup(const uint8_t* seq, const uint8_t* qual, const float* d_in, int L, QcPtr qc, DPtr d, int w, int lpr) {
    // eight positions per lane at a time: their 24 loads are issued before the first store, so a 50 bp read costs two waits for memory, not 13
    constexpr int kChunk = 8;
    for (int base = w; base < L; base += lpr * kChunk) {
        uint8_t s[kChunk], q[kChunk];
        float dv[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int j = base + i * lpr;
            const int jj = j < L ? j : L - 1;  // clamped: the loads stay unconditional
            s[i] = seq[jj]; q[i] = qual[jj]; dv[i] = d_in[jj];
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int j = base + i * lpr;
            if (j < L) { qc[2 * j] = (uint8_t)base_index(s[i]); qc[2 * j + 1] = q[i]; d[j] = dv[i]; }
        }
    }
}

MAPAD_HD int alignment_start_of(const DevParams& P, int L) { return P.start_at_end ? L : (L / 2); }  // find_alignment_start

// The `len == pattern.len()` branch of check_and_push_stack_frame (mapping.rs:973-984): a finished alignment becomes a hit.
template <bool NLR, bool NL, int TOP>
MAPAD_RARE void record_hit(const ReadInT<NLR> rd, const ArenaT<NL, TOP> A, SearchState& st, int alignment_start, uint64_t lower, uint64_t lower_rev, uint64_t size,
                           float score, uint32_t id, int quad_lane = -1) {
    if (st.n_hits >= (uint32_t)kMaxHits) { st.status = ST_ARENA_OVERFLOW; return; }
    HitRec h;
    h.lower = lower; h.lower_rev = lower_rev; h.size = size; h.score = score; h.pad = 0;
    h.ops_off = st.hit_ops_used;
    const uint32_t m = extract_ops(A.nodes, id, alignment_start, rd.L, A.scratch, A.hit_ops + st.hit_ops_used, A.hit_ops_cap - st.hit_ops_used, st.tree_len == st.tree_entries, quad_lane);
    if (m == 0xFFFFFFFFu) { st.status = ST_ARENA_OVERFLOW; return; }
    h.n_ops = m;
    st.hit_ops_used += m;
    hits_push(A.hits, st.n_hits, h);
    if (st.n_hits == 1 || score > st.best_score) { st.best_score = score; st.best_size = size; }  // new BinaryHeap root
    st.c_hits += 1;
}

// Before a step starts, an arena that cannot take the step's worst case (9 new nodes, 8 more frames) asks `grow` for a bigger one:
// grow(A, st) migrates heap and nodes into a larger arena and updates A (device: arenas from size-class pools, mapad_amd.hip).
// It returns GROW_OK, GROW_WAIT (a suitable arena exists but none is free right now: the step is retried later, nothing has been
// touched) or GROW_NEVER (the read goes to the full-limit pass).
enum : int { GROW_OK = 1, GROW_WAIT = 0, GROW_NEVER = -1 };
constexpr uint32_t kStepNodes = 9;
struct NoGrow {
    template <class AR> MAPAD_HD int operator()(AR&, const SearchState&) const { return GROW_NEVER; }
};

// check_and_push_stack_frame (mapping.rs:932-987) for a child whose tree node `nd` is already packed.  On the device the quad
// builds the <= 9 children of a frame lane-parallel (search_step); `store` says whether this lane owns the child and writes its node,
// `owner` is the owning lane (the frame of a finished alignment is fetched from it).  Everything else is quad-uniform.
template <int LPR, bool NLR, bool NL, int TOP>
MAPAD_HD void commit_child(const DevParams& P, const ReadInT<NLR>& rd, ArenaT<NL, TOP>& A, SearchState& st, int alignment_start, float score, uint32_t ngaps, int len,
                           const Node& nd, bool store, int owner) {
    if (st.n_hits > 0 && mb_reject_iterative(P, score, st.best_score)) return;
    if ((int)ngaps > P.max_num_gaps_open) return;
    if (MAPAD_UNLIKELY(st.tree_next == st.tree_entries && st.tree_entries >= A.node_cap)) { st.status = ST_ARENA_OVERFLOW; return; }  // cannot happen (search_step)
    const uint32_t id = tree_alloc(A.nodes, st);
    // the loads of a push go out before the stores of its node: the wait for them then leaves the stores in flight (vmcnt counts in order)
    const bool pushes = len != rd.L;
    const uint32_t pos = st.heap_len;
    Ancestors an{};
    if (pushes) an = load_ancestors(A, pos);
    if (store) { MAPAD_TOUCH(&A.nodes[id], sizeof(Node), true); A.nodes[id] = nd; }
    st.c_node += 1;
    if (MAPAD_UNLIKELY(len == rd.L)) {  // rare: the state takes a round trip through memory only here, so it can live in registers otherwise
        Node u = nd;
#if defined(__HIP_DEVICE_COMPILE__)
        if (LPR == 4) { u.w1 = quad_pick64(nd.w1, owner); u.w2 = quad_pick64(nd.w2, owner); u.w3 = quad_pick64(nd.w3, owner); }
        if (LPR == 2) { u.w1 = pair_pick64(nd.w1, owner); u.w2 = pair_pick64(nd.w2, owner); u.w3 = pair_pick64(nd.w3, owner); }
#endif
        (void)owner;
        const Frame c = unpack_frame(u);
        SearchState tmp = st;
        MAPAD_MARK(PROF_COMMIT);
        int quad_lane = -1;  // quads walk the hit's path with a window of 16 nodes per trip (extract_ops_quad)
#if defined(__HIP_DEVICE_COMPILE__)
        if (LPR == 4) quad_lane = (int)(threadIdx.x & 3u);
#endif
        record_hit(rd, A, tmp, alignment_start, c.lower, c.lower_rev, c.size, score, id, quad_lane);
        drain_memory();
        MAPAD_MARK(PROF_HIT);
        st = tmp;
        return;
    }
    if (MAPAD_UNLIKELY(st.heap_len >= A.heap_cap)) { st.status = ST_ARENA_OVERFLOW; return; }  // cannot happen (search_step)
    st.heap_len += 1;
    MAPAD_MARK(PROF_C_PRE);
    mm_bubble_up(A, pos, HeapEntry{score, id}, an);  // the ancestors were read from memory, which every earlier store of this step has reached
    st.c_push += 1;
}

// Overflow recovery (mapping.rs:1371-1379): evict the worst frames and free their tree nodes.
template <bool NL, int TOP>
MAPAD_RARE void evict_worst(const ArenaT<NL, TOP> A, SearchState& st, int64_t cnt) {
    for (int64_t i = 0; i < cnt; ++i) {
        if (st.heap_len == 0) break;
        const HeapEntry m = mm_pop_min(A, st.heap_len);
        tree_remove(A.nodes, st, m.node);
    }
}

// k_mismatch_search (mapping.rs:1012-1383) after the D array has been computed, split into init / step so that a
// persistent quad can fetch its next read as soon as the current one finishes.  `w` = lane index inside the quad.
template <bool NLR, bool NL, int TOP>
MAPAD_RARE void search_init(uint64_t n_text, int alignment_start, const ReadInT<NLR> rd, const ArenaT<NL, TOP> A, SearchState& st) {
    st.c_esearch = 0; st.c_push = 0; st.c_pop = 0; st.c_node = 0; st.c_hits = 0;
    st.heap_len = 0; st.tree_entries = 0; st.tree_next = 0; st.tree_len = 0; st.n_hits = 0; st.hit_ops_used = 0; st.status = ST_OK;
    st.best_score = 0.0f; st.best_size = 0;
    Frame root;  // Tree::clear() -> id 0; root frame (mapping.rs:1045-1054)
    root.lower = 0; root.lower_rev = 0; root.size = n_text;  // init_interval
    root.start = alignment_start; root.len = 0; root.gap_f = GAP_CLOSED; root.gap_b = GAP_CLOSED; root.ngaps = 0;
    tree_insert(A.nodes, st, pack_node(pack_op(OP_MATCH, 0, 0), 0, root));
    A.top[0] = HeapEntry{0.0f, 0u};
    st.heap_len = 1;
    st.c_push += 1;
}

// One iteration of the `while let Some(stack_frame) = stack.pop_max()` loop.  Returns false when the search is over.
// PC: payload cache.  The frame payload (w1..w3 of the node) of the entries in heap slots 1 and 2 — one of which is the next pop_max — is kept next to the
// quad (A.pc, LDS on the device), keyed by node id, so that a pop does not start with a dependent trip to the arena for its node.  It is filled (a) when a pop's
// sift puts a new entry into the slot it emptied: that frame's node is fetched behind the rank-query loads of the same step and stored at the end of the step,
// (b) from registers when a child of this step bubbles up into slot 1 or 2.  Everything else (tiny heaps, the general commit loop, evictions) just misses:
// a payload is only used if its key equals the node id the heap holds, a live node's payload never changes, and evictions — after which ids are reused — clear
// the cache.  What leaves the step's dependent chain is one HBM round trip in six.
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
static unsigned long long g_pc_stats[5];
static unsigned long long g_par_stats[4];  // commits with >= 2 movers in a round of four, their movers, the rounds of bubble-ups they take with the parallel prefix
static unsigned long long g_commit_stats[8];  // steps with children, children, movers, steps without a mover, steps with >= 3 children, ... of those without a mover, movers in those
#endif
template <bool NL, int TOP>
MAPAD_HD void pc_store(const ArenaT<NL, TOP>& A, uint32_t s, uint32_t id, uint64_t w1, uint64_t w2, uint64_t w3) {
    A.pc[4 * s] = (1ull << 32) | id; A.pc[4 * s + 1] = w1; A.pc[4 * s + 2] = w2; A.pc[4 * s + 3] = w3;
}
template <bool NL, int TOP>
MAPAD_HD void pc_clear(const ArenaT<NL, TOP>& A) { A.pc[0] = 0; A.pc[4] = 0; }

// MAPAD_PAR_COMMIT (device quads): the children of a frame are pushed by the quad's four lanes side by side instead of one after the other.  A push that stays
// where it is appended (mm_push_stays: 70 % of them with the no-damage model, where ties are the rule) writes its own slot and nothing else, and it stays a stayer
// whatever its siblings do: a sibling that moves up along the max levels only ever RAISES the entries of max-level slots, one that moves along the min levels only
// lowers min-level entries, and "stays" means not above the max-level ancestor and not below the min-level one.  So every lane loads the ancestors of its own child
// (slot heap_len + rank), the stayers are stored at once, and only the movers go one after the other, in commit order, each by the lane that holds it — the first
// with the ancestors it already has (stayers changed none of them), later ones with fresh loads.  One memory round trip per frame where the sequential loop made one
// per child; with 16 read slots in lockstep that loop ran 4 trips in 73 % of the wavefront steps (section profile, C4: 47 % of the wave time) although two thirds of
// the frames have one child.  Same heap, entry for entry, as the sequential pushes (children cannot be each other's ancestors once the heap has 16 entries).
#if !defined(MAPAD_PAR_COMMIT)
#define MAPAD_PAR_COMMIT 1
#endif
// MAPAD_SPEC_SIFT=1: the pop's sift fetches two strides per trip to the arena (heap_core.hpp: mm_trickle_down<.., SPEC>).  Quads with near data in LDS only.
// (Round 6, measured and removed — commit c63a3b5..: profiles/r06/ab_quad_movers.txt.  The movers of a commit round decided by the whole quad from registers — static DPP
//  broadcasts of the mover's element and ancestors, every lane patching its own ancestors with what the mover wrote — instead of each by its own lane behind a reload from
//  memory: no trip to memory between movers, parity-green, and 7-8 % SLOWER on C3 and C2.  Like round 4's forwarding (-3 %): the movers' cost is their instructions, and
//  four unrolled decisions per round are more instructions than the reloads they replace.)
#if !defined(MAPAD_LAZY_GAP_NODES)
#define MAPAD_LAZY_GAP_NODES 0  // (measured +-0: profiles/r06/ab_lazy_gap_nodes.txt — the compiler sinks the packing into the predicated stores by itself)
#endif
#if !defined(MAPAD_SPEC_SIFT)
#define MAPAD_SPEC_SIFT 1
#endif
// NodePrefetch (device quads, MAPAD_NODE_PREFETCH): the node of the frame the NEXT step will pop, requested at the end of this step.  Between the end of a step and the
// node load of the next lie the step's tail, the kernel loop's hand-over and grow checks and the next step's prologue — a tenth of a wave's time (section profile: loop
// head + step tail + read setup), during which nothing is in flight for this read; the frontier does not change in between (an arena migration copies the nodes
// unchanged), so the next step's `find max` names the same node and takes the payload from these registers.  A new read starts with `ok = false`.
// Built and parity-green in round 6, and NEUTRAL (profiles/r06/ab_node_prefetch.txt: C3 2.64 / 2.64 M reads/s, C4 3.10 / 3.10 M, C2 6.08 / 6.13 M with / without): like
// the payload cache and the early ancestors of round 4, taking a trip out of the step's chain does not shorten the step.  Off by default (MAPAD_NODE_PREFETCH=1).
struct NodePrefetch {
    static constexpr bool on = true;
    uint64_t w1 = 0, w2 = 0, w3 = 0;
    uint32_t id = 0;
    bool ok = false;
};
struct NoPrefetch { static constexpr bool on = false; };
template <int LPR, bool CONT, bool NL, bool PC = false, class Grow = NoGrow, int TOP = kTop, bool NLR = NL, class PFT = NoPrefetch>
MAPAD_HD bool search_step_pf(const DevIndex& ix, const DevParams& P, const ReadInT<NLR>& rd, ArenaT<NL, TOP>& A, SearchState& st, int w, const Grow& grow, PFT& pf);
template <int LPR, bool CONT, bool NL, bool PC = false, class Grow = NoGrow, int TOP = kTop, bool NLR = NL>
MAPAD_HD bool search_step(const DevIndex& ix, const DevParams& P, const ReadInT<NLR>& rd, ArenaT<NL, TOP>& A, SearchState& st, int w, const Grow& grow) {
    NoPrefetch np;
    return search_step_pf<LPR, CONT, NL, PC>(ix, P, rd, A, st, w, grow, np);
}
template <int LPR, bool CONT, bool NL, bool PC, class Grow, int TOP, bool NLR, class PFT>
MAPAD_HD bool search_step_pf(const DevIndex& ix, const DevParams& P, const ReadInT<NLR>& rd, ArenaT<NL, TOP>& A, SearchState& st, int w, const Grow& grow, PFT& pf) {
    if (st.heap_len == 0 || st.status != ST_OK) return false;
    if (MAPAD_UNLIKELY(st.tree_len + kStepNodes > A.node_cap || st.heap_len + kStepNodes > A.heap_cap)) {
        MAPAD_MARK(PROF_LOOP);
        const int g = grow(A, st);
        drain_memory();
        MAPAD_MARK(PROF_GROW);
        if (g == GROW_WAIT) return true;
        if (g == GROW_NEVER || st.tree_len + kStepNodes > A.node_cap || st.heap_len + kStepNodes > A.heap_cap) { st.status = ST_ARENA_OVERFLOW; return false; }
    }
    const int L = rd.L;
    const int alignment_start = alignment_start_of(P, L);
    const float open_ext = P.gap_open + P.gap_extend;
    uint32_t top_idx;
#if defined(MAPAD_PROFILE_SECTIONS) && defined(__HIP_DEVICE_COMPILE__)
    if (w == 0) atomicAdd(&g_prof_hist[31 - __clz((int)st.heap_len)], 1u);
    const uint32_t prof_nodes0 = st.tree_len;
#endif
    // Order of a step, chosen for the memory round trips it takes (each of them ~1-2 us when the chip is loaded):
    //   1. {popped frame's node, the heap's last entry}      2. {score row, four index loads} issued as soon as the frame is known, and behind
    //   them the repair of the heap (its round trips through the arena levels run while the index loads are in flight) and the window of
    //   ancestors for this step's pushes      3. nothing: counts, gates, children and pushes work on what has arrived.
    const HeapEntry top = mm_find_max(A, st.heap_len, top_idx);
    Node top_node;
    if constexpr (PC) {
        const uint32_t cs = top_idx == 2 ? 1u : 0u;
        const uint64_t key = A.pc[4 * cs], c1 = A.pc[4 * cs + 1], c2 = A.pc[4 * cs + 2], c3 = A.pc[4 * cs + 3];
        const bool hit = (top_idx != 0) & (key == ((1ull << 32) | top.node));
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
        g_pc_stats[hit ? 0 : top_idx == 0 ? 2 : 1] += 1;  // hits, misses, pops of slot 0 (heaps of one entry)
#endif
        uint64_t g1 = 0, g2 = 0, g3 = 0;
        if (MAPAD_UNLIKELY(!hit)) {  // a miss takes the trip to the arena and waits for it here, so that the hit path carries no wait at all (drain_memory)
            MAPAD_TOUCH(&A.nodes[top.node], sizeof(Node), false);
            const Node g = A.nodes[top.node];
            g1 = g.w1; g2 = g.w2; g3 = g.w3;
            drain_memory();
        }
        top_node.w0 = 0; top_node.w1 = hit ? c1 : g1; top_node.w2 = hit ? c2 : g2; top_node.w3 = hit ? c3 : g3;
    } else if constexpr (PFT::on) {
        const bool hit = pf.ok & (pf.id == top.node);
        uint64_t g1 = 0, g2 = 0, g3 = 0;
        if (MAPAD_UNLIKELY(!hit)) {  // the first step of a read: the usual trip, waited for here so that the common path carries no wait of its own (drain_memory)
            const Node g = A.nodes[top.node];
            g1 = g.w1; g2 = g.w2; g3 = g.w3;
            drain_memory();
        }
        top_node.w0 = 0; top_node.w1 = hit ? pf.w1 : g1; top_node.w2 = hit ? pf.w2 : g2; top_node.w3 = hit ? pf.w3 : g3;
    } else {
        MAPAD_TOUCH(&A.nodes[top.node], sizeof(Node), false);
        top_node = A.nodes[top.node];
    }
    HeapEntry last;
    if constexpr (!PC) {  // with the node: both trips are needed before anything else can start
        last = hp_get(A, st.heap_len - 1);
    }
    st.c_pop += 1;
    const Frame f = unpack_frame(top_node);
    const float f_score = top.score;
    const bool forward = f.start <= L - f.start - f.len;  // :1077-1097
    const int j = forward ? f.start + f.len : f.start - 1;
    const int d_k = forward ? f.start : f.start - 1, d_l = forward ? f.start + f.len : f.start + f.len - 1;
    const int to_class = rd.qc[2 * j];
    Float4 row;  // shared score table: hot in L1/L2; consumed after the rank queries
    row = sdm_row_at(P, rd.table, j, rd.qc[2 * j + 1], to_class);
    const uint32_t gap_side = forward ? f.gap_f : f.gap_b;
    const float insertion_score = (gap_side == GAP_INS ? P.gap_extend : open_ext) + f_score;  // :1127-1136,1165-1174
    const float deletion_score = (gap_side == GAP_DEL ? P.gap_extend : open_ext) + f_score;
    const uint32_t num_gaps_open = gap_side == GAP_CLOSED ? f.ngaps + 1 : f.ngaps;             // :1148-1152
    const float lower_bound = d_get(rd.d, L, alignment_start, d_k, d_l);                       // :1195
    if (st.n_hits > 0 && mb_reject_iterative(P, f_score + lower_bound, st.best_score)) { st.heap_len -= 1; return false; }  // :1201-1208 (the frame was popped; the search is over)
    MAPAD_MARK(PROF_NODE);
    if constexpr (PC) {   // The heap's last entry, loaded behind the stop rule above so that every path that issues the load also reaches the point where it counts as used
        // (consume_here).  Near read (clamped) for every slot, arena load predicated: as one two-armed branch the arms share their destination registers, and
        // the wait-count pass then puts a full drain in front of the near arm (any wavefront with a slot whose heap is still small).
        const uint32_t li = st.heap_len - 1;
        const bool l_near = li < (uint32_t)TOP;
        const HeapEntry ln = load_entry(A.top + (l_near ? li : 0u));
        HeapEntry lg = HeapEntry{0.0f, 0u};
        if (!l_near) lg = load_entry(A.heap + HeapLayout<TOP>::slot(li));
        last.score = l_near ? ln.score : lg.score; last.node = l_near ? ln.node : lg.node;
    }

    // Extension (:1245); forward extension works on the swapped interval.  Device quads: lane w keeps the extension by base w only.
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool kLaneKids = LPR == 4 || LPR == 2;  // the lanes of a read's group each build the children of their own base(s)
    constexpr int kBases = LPR == 2 ? 2 : 1;           // bases per lane: lane w owns bases kBases * w .. kBases * w + kBases - 1
#else
    constexpr bool kLaneKids = false;
    constexpr int kBases = 1;
#endif
    const uint64_t x_lower = forward ? f.lower_rev : f.lower, x_lower_rev = forward ? f.lower : f.lower_rev;
#if defined(__HIP_DEVICE_COMPILE__)
    ExtLoads ext_loads{};
    ExtLoads2 ext_loads2{};
    if constexpr (LPR == 4) ext_loads = ext4_quad_issue(ix, x_lower, f.size, w);  // one call site: forward and backward quads of a wavefront share the round trip
    if constexpr (LPR == 2) ext_loads2 = ext4_pair_issue(ix, x_lower, f.size, w);
#endif
    // pop_max of the crate, second half: the last entry takes the place of the maximum and trickles down
    st.heap_len -= 1;
    uint64_t pf1 = 0, pf2 = 0, pf3 = 0;  // PC: payload of the frame the sift moves into the emptied slot, fetched behind the rank-query loads
    uint32_t pf_id = 0;
    bool pf_valid = false;
    if constexpr (PC) {
        auto fetch = [&](uint32_t id) {
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
            g_pc_stats[3] += 1;  // nodes fetched ahead
#endif
            MAPAD_TOUCH(&A.nodes[id], sizeof(Node), false);
            const Node g = A.nodes[id]; pf1 = g.w1; pf2 = g.w2; pf3 = g.w3; pf_id = id; pf_valid = true;
        };
        if (top_idx < st.heap_len) mm_trickle_down<true>(A, st.heap_len, top_idx, last, fetch);
    } else {
#if defined(__HIP_DEVICE_COMPILE__) && MAPAD_SPEC_SIFT
        if constexpr (LPR == 4 && NL) { if (top_idx < st.heap_len) mm_trickle_down<true, NL, TOP, NoOccupantHook, true>(A, st.heap_len, top_idx, last, NoOccupantHook(), w); }
        else
#endif
        if (top_idx < st.heap_len) mm_trickle_down<true>(A, st.heap_len, top_idx, last);
    }
    MAPAD_MARK(PROF_POP);
    Ext4 e;
    uint64_t my_lower[kBases] = {}, my_lower_rev[kBases] = {}, my_size[kBases] = {};  // kLaneKids: extension by this lane's base(s)
    uint32_t nonempty;
    if constexpr (kLaneKids) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (LPR == 4) {
            ExtLane x;
            ext4_quad_lane_finish(ix, ext_loads, x_lower, x_lower_rev, f.size, w, rd.lane_less, x);
            my_lower[0] = x.lower; my_lower_rev[0] = x.lower_rev; my_size[0] = x.size; nonempty = x.nonempty;
        } else {
            ExtLane2 x;
            ext4_pair_lane_finish(ix, ext_loads2, x_lower, x_lower_rev, f.size, w, rd.lane_less, rd.lane_less1, x);
#pragma unroll
            for (int b = 0; b < kBases; ++b) { my_lower[b] = x.lower[b]; my_lower_rev[b] = x.lower_rev[b]; my_size[b] = x.size[b]; }
            nonempty = x.nonempty;
        }
#endif
    } else {
        ext4_any<LPR>(ix, x_lower, x_lower_rev, f.size, w, e);
        nonempty = (e.size[0] >= 1 ? 1u : 0u) | (e.size[1] >= 1 ? 2u : 0u) | (e.size[2] >= 1 ? 4u : 0u) | (e.size[3] >= 1 ? 8u : 0u);
    }
    st.c_esearch += 1;
    if constexpr (PC) { consume_here(last.score); consume_here(last.node); }  // older than the rank-query loads just waited for; a step whose pop needs no sift would leave it "pending" (consume_here)
    MAPAD_MARK(PROF_EXT);

    // Static gates of the <= 9 children in commit order: Ins; then for k = T,G,C,A: Del(k), Match/Mismatch(k).
    // bit 0 = Ins, bit 1+2i = Del, bit 2+2i = M/MM with i = 0..3 <-> k = 3..0.
    const int ins_dist = j < (L - j - 1) ? j : (L - j - 1);
    const int dist5 = forward ? j : j + 1, dist3 = L - dist5;
    const int del_dist = dist5 < dist3 ? dist5 : dist3;
    const bool ins_ok = !mb_reject<CONT>(rd.thr, P.cutoff, insertion_score + lower_bound) & (ins_dist >= P.gap_dist_ends);  // :1214-1216
    const bool del_ok = !mb_reject<CONT>(rd.thr, P.cutoff, deletion_score + lower_bound) & (del_dist >= P.gap_dist_ends);   // :1279-1281
    uint32_t cand = ins_ok ? 1u : 0u;
    const float optimal = sdm_optimal(row, to_class);
    float mm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 3 - i;
        const int cb = forward ? 3 - k : k;  // symbol in read orientation: backward = base k, forward = its complement
        mm[i] = f4_get(row, cb) - optimal + f_score;  // get - optimal + score, left to right (:1138-1145)
        const uint32_t has = (nonempty >> k) & 1u;
        cand |= (has & (uint32_t)del_ok) << (1 + 2 * i);
        cand |= (has & (uint32_t)!mb_reject<CONT>(rd.thr, P.cutoff, mm[i] + lower_bound)) << (2 + 2 * i);  // :1308
    }
    const int child_start = forward ? f.start : f.start - 1;
    // tree node (= frame payload) of child t; (xl, xr, xs) = extension of the frame's interval by base k (unused for t = 0)
    auto make_child = [&](int t, int k, uint64_t xl, uint64_t xr, uint64_t xs) -> Node {
        Frame c;
        uint32_t op;
        if (t == 0) {  // Insertion in read (:1213-1242)
            c.lower = f.lower; c.lower_rev = f.lower_rev; c.size = f.size;
            c.start = child_start; c.len = f.len + 1;
            c.gap_f = forward ? (uint32_t)GAP_INS : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_INS;
            c.ngaps = num_gaps_open;
            op = pack_op(OP_INS, (uint32_t)j, 0);
        } else {
            const int cb = forward ? 3 - k : k;
            const uint32_t c_ascii = (0x54474341u >> (8 * cb)) & 0xFFu;  // "ACGT"[cb] (a chain of conditionals comes out as nested branches)
            c.size = xs;
            c.lower = forward ? xr : xl;  // swapped back for forward extension (:1256)
            c.lower_rev = forward ? xl : xr;
            if (t & 1) {  // Deletion in read (:1265-1302)
                c.start = f.start; c.len = f.len;
                c.gap_f = forward ? (uint32_t)GAP_DEL : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_DEL;
                c.ngaps = num_gaps_open;
                op = pack_op(OP_DEL, (uint32_t)j, c_ascii);
            } else {  // Match / mismatch (:1307-1338)
                c.start = child_start; c.len = f.len + 1;
                c.gap_f = forward ? (uint32_t)GAP_CLOSED : f.gap_f; c.gap_b = forward ? f.gap_b : (uint32_t)GAP_CLOSED;
                c.ngaps = f.ngaps;
                op = (cb == to_class) ? pack_op(OP_MATCH, (uint32_t)j, 0) : pack_op(OP_MISMATCH, (uint32_t)j, c_ascii);
            }
        }
        return pack_node(op, top.node, c);
    };
    Node nd_ins{}, nd_del[kBases] = {}, nd_mm[kBases] = {};
    if constexpr (kLaneKids) {  // every lane packs the children of its base(s) once; the commit loop below only selects and stores
        // Round 6 (MAPAD_LAZY_GAP_NODES): the insertion and deletion children only where a gate let one through (`cand` is quad-uniform and only loses bits from here
        // on).  At -p 0.03 that is 0.2 % of the pops — a few per cent of the wavefront steps —, and their ~30 pack instructions used to run in every step.
        const bool gap_kids = !MAPAD_LAZY_GAP_NODES || (cand & 0xABu) != 0u;  // bit 0 = Ins, bits 1, 3, 5, 7 = Del
        if (gap_kids) nd_ins = make_child(0, 0, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < kBases; ++b) {
            const int k = kBases * w + b;
            if (gap_kids) nd_del[b] = make_child(1 + 2 * (3 - k), k, my_lower[b], my_lower_rev[b], my_size[b]);
            nd_mm[b] = make_child(2 + 2 * (3 - k), k, my_lower[b], my_lower_rev[b], my_size[b]);
        }
    }
    MAPAD_MARK(PROF_GATES);
#if defined(MAPAD_PROFILE_SECTIONS) && defined(__HIP_DEVICE_COMPILE__)
    {   // trips of this wave step = max candidates over the active quads
        uint32_t m = (uint32_t)__popc(cand);
        for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&g_prof_hist[36 + min(m, 11u)], 1u);
    }
#endif
    // Fast path of the commit loop: when no child can complete the read (len + 1 < L) and the slab has no free list, nothing a sibling does
    // changes the gates of the next one — reject_iterative compares against a best hit that cannot change within the step, the gap limit is
    // static — so both fold into the candidate mask and the loop body is: slab key, arena loads of the push, node store, bubble-up.  (The
    // general loop below pays ~350 instructions per child for the checks, the selects around them and the hit path; children are 2.4 loop
    // trips per wavefront step and were half of a step's instructions.)
    if (f.len + 1 < L && st.tree_next == st.tree_entries) {
        if ((int)num_gaps_open > P.max_num_gaps_open) cand &= 0x154u;  // Ins and Del children open or extend a gap; M/MM children keep the frame's count
        if (st.n_hits > 0 && P.bound_kind != BOUND_TEST) {             // mb_reject_iterative (commit_child)
            const float lim = st.best_score + P.repr_mm;
            if (insertion_score < lim) cand &= ~1u;
            if (deletion_score < lim) cand &= ~0xAAu;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (mm[i] < lim) cand &= ~(4u << (2 * i));
        }
        const uint32_t cand0 = cand, id0 = st.tree_next;  // == tree_entries: the slab grows at its end, child t gets key id0 + (children before t)
#if defined(__HIP_DEVICE_COMPILE__)
        // The nodes of all children in three store groups (a node is only read when its frame is popped, at the earliest in the next step): lane w stores the
        // match / mismatch and the deletion child of base w, lane 0 the insertion child.
        auto store_child_nodes = [&]() {
            if constexpr (kLaneKids) {
#pragma unroll
                for (int b = 0; b < kBases; ++b) {
                    const uint32_t t_mm = 2u + 2u * (3u - (uint32_t)(kBases * w + b)), t_del = t_mm - 1u;
                    if ((cand0 >> t_mm) & 1u) A.nodes[id0 + (uint32_t)__popc(cand0 & ((1u << t_mm) - 1u))] = nd_mm[b];
                    if ((cand0 >> t_del) & 1u) A.nodes[id0 + (uint32_t)__popc(cand0 & ((1u << t_del) - 1u))] = nd_del[b];
                }
                if ((cand0 & 1u) && w == 0) A.nodes[id0] = nd_ins;
            }
        };
        if constexpr (MAPAD_PAR_COMMIT != 0 && LPR == 4) {
            if (st.heap_len >= 16u) {
                const uint32_t n0 = st.heap_len, kids = (uint32_t)__popc(cand0);
                uint32_t rest = cand0;
                for (uint32_t base = 0; base < kids; base += 4) {  // four children per round (a frame has at most nine)
                    // commit-order rank base + w -> which child (bit of cand0)
                    int t = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int tq = rest ? __ffs((int)rest) - 1 : 0; rest &= rest - 1; t = q == w ? tq : t; }
                    const uint32_t r = base + (uint32_t)w;
                    const bool act = r < kids;
                    const bool is_ins = t == 0, is_del = (t & 1) != 0;
                    const int i = is_ins ? 0 : (t - 1) >> 1;
                    float score = mm[3];
                    score = i == 2 ? mm[2] : score; score = i == 1 ? mm[1] : score; score = i == 0 ? mm[0] : score;
                    score = is_del ? deletion_score : score; score = is_ins ? insertion_score : score;
                    const uint32_t pos = n0 + (act ? r : 0u);
                    const HeapEntry elt{score, id0 + r};
                    MAPAD_MARK(PROF_C_PRE);  // (section profile) gates .. here: instructions only
                    const Ancestors an = load_ancestors(A, pos);
                    const bool stays = mm_push_stays(pos, elt, an);
#if defined(MAPAD_PROFILE_SECTIONS)
                    if (__ballot(stays) != 0x0123456789ABCDEFull) MAPAD_MARK(PROF_C_ANC);  // ... the one trip of the commit: the ancestors have arrived
#endif
                    if (act & stays) hp_set(A, pos, elt);
                    // the movers of this round, in commit order, each by its own lane
                    uint32_t movers = (uint32_t)(__ballot(act & !stays) >> (threadIdx.x & 60u)) & 15u;  // this quad's four lanes
                    bool first = true;  // the first mover's ancestors are what the heap holds (stayers changed none of them); whoever moves behind it reloads
                    while (movers) {
                        const int m = __ffs((int)movers) - 1;
                        movers &= movers - 1;
                        if (w == m) {
                            Ancestors my_an = an;
                            if (MAPAD_UNLIKELY(!first)) { my_an = load_ancestors(A, pos); drain_memory(); }
                            mm_bubble_up(A, pos, elt, my_an);
                        }
                        first = false;
                    }
                }
                st.tree_next = id0 + kids; st.tree_entries = id0 + kids; st.tree_len += kids; st.heap_len = n0 + kids;
                st.c_node += kids; st.c_push += kids;
                cand = 0;
            }
        }
#elif defined(MAPAD_PAR_COMMIT_EMU)
        // the same scheme on the host, lane by lane (tests/emu: the CPU suite checks the argument above on every read it maps)
        if (!PC && st.heap_len >= 16u) {
            const uint32_t n0 = st.heap_len, kids = (uint32_t)__builtin_popcount(cand0);
            uint32_t rest = cand0;
            for (uint32_t base = 0; base < kids; base += 4) {
                HeapEntry elt4[4]; Ancestors an4[4]; uint32_t pos4[4], movers = 0;
                for (int wv = 0; wv < 4 && base + wv < kids; ++wv) {
                    const int t = __builtin_ctz(rest); rest &= rest - 1;
                    const bool is_ins = t == 0, is_del = (t & 1) != 0;
                    const int i = is_ins ? 0 : (t - 1) >> 1, k = 3 - i;
                    const float score = is_ins ? insertion_score : is_del ? deletion_score : mm[i];
                    pos4[wv] = n0 + base + wv; elt4[wv] = HeapEntry{score, id0 + base + (uint32_t)wv};
                    an4[wv] = load_ancestors(A, pos4[wv]);
                    const uint64_t xl = e.lower[k], xr = e.lower_rev[k], xs = e.size[k];
                    MAPAD_TOUCH(&A.nodes[elt4[wv].node], sizeof(Node), true);
                    A.nodes[elt4[wv].node] = make_child(t, k, xl, xr, xs);
                }
                for (int wv = 0; wv < 4 && base + wv < kids; ++wv) {  // all decisions against the heap as it was, stayers stored
                    if (mm_push_stays(pos4[wv], elt4[wv], an4[wv])) hp_set(A, pos4[wv], elt4[wv]); else movers |= 1u << wv;
                }
#if defined(MAPAD_PC_STATS)
                {   // How many of these rounds of bubble-ups could run side by side?  (round 5: the movers of a round go one after the other, each behind the first with a
                    // reload — at C3 that is 40 % of the wave time.)  Count the longest prefix of movers, in commit order, none of which depends on an earlier one's writes
                    // (heap_core.hpp: mm_mover_plan / mm_mover_depends): measured 5 % fewer rounds at C3 (2.9 movers per multi-mover round, nearly all dependent: the four
                    // leaves share two grandparents and the movers of the damage model climb to them) — not built.
                    MoverPlan plan4[4];
                    int n_par = 0, n_mov = 0;
                    bool open = true;
                    for (int wv = 0; wv < 4; ++wv) if ((movers >> wv) & 1u) {
                        plan4[wv] = mm_mover_plan(pos4[wv], elt4[wv], an4[wv]);
                        bool dep = false;
                        for (int u = 0; u < wv; ++u) if ((movers >> u) & 1u) dep |= mm_mover_depends(pos4[wv], plan4[wv].climbs, plan4[u]);
                        open = open && !dep;
                        n_par += open ? 1 : 0;
                        n_mov += 1;
                    }
                    if (n_mov >= 2) { g_par_stats[0] += 1; g_par_stats[1] += n_mov; g_par_stats[2] += 1 + (n_mov - n_par); }
                }
#endif
                bool first = true;
                for (int wv = 0; wv < 4; ++wv) if ((movers >> wv) & 1u) {
                    if (!first) an4[wv] = load_ancestors(A, pos4[wv]);
                    mm_bubble_up(A, pos4[wv], elt4[wv], an4[wv]);
                    first = false;
                }
            }
            st.tree_next = id0 + kids; st.tree_entries = id0 + kids; st.tree_len += kids; st.heap_len = n0 + kids;
            st.c_node += kids; st.c_push += kids;
            cand = 0;
        }
#endif
        uint32_t land_t = 0xFFu, land_s = 0;  // PC: the last child of this step that ended up in heap slot 1 or 2 (land_s = slot - 1)
        Node land_nd{};                       //     ... and its node where the children are not built lane-parallel
        (void)land_nd;
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
        int stat_k = 0, stat_movers = 0;
#endif
        while (cand != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int t = __ffs((int)cand) - 1;
#else
            const int t = __builtin_ctz(cand);
#endif
            cand &= cand - 1;
            const bool is_ins = t == 0, is_del = (t & 1) != 0;
            const int i = is_ins ? 0 : (t - 1) >> 1, k = 3 - i;
            float score = mm[3];  // a chain of plain selects (nested conditionals on floats come out as nested branches)
            score = i == 2 ? mm[2] : score; score = i == 1 ? mm[1] : score; score = i == 0 ? mm[0] : score;
            score = is_del ? deletion_score : score; score = is_ins ? insertion_score : score;
            const uint32_t id = st.tree_next;
            st.tree_next = id + 1; st.tree_entries = id + 1; st.tree_len += 1;
            const uint32_t pos = st.heap_len;
            st.heap_len = pos + 1;
            const Ancestors an = load_ancestors(A, pos);
            Node made{};
            (void)made;
            if constexpr (!kLaneKids) {
                const uint64_t xl = k == 0 ? e.lower[0] : k == 1 ? e.lower[1] : k == 2 ? e.lower[2] : e.lower[3];
                const uint64_t xr = k == 0 ? e.lower_rev[0] : k == 1 ? e.lower_rev[1] : k == 2 ? e.lower_rev[2] : e.lower_rev[3];
                const uint64_t xs = k == 0 ? e.size[0] : k == 1 ? e.size[1] : k == 2 ? e.size[2] : e.size[3];
                made = make_child(t, k, xl, xr, xs);
                MAPAD_TOUCH(&A.nodes[id], sizeof(Node), true);
                A.nodes[id] = made;
            }
            const uint32_t fin = mm_bubble_up(A, pos, HeapEntry{score, id}, an);
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
            stat_k += 1; stat_movers += fin != pos;
#endif
            if constexpr (PC) {
                const bool lands = (fin - 1u) < 2u;
                land_t = lands ? (uint32_t)t : land_t; land_s = lands ? fin - 1u : land_s;
                if constexpr (!kLaneKids) { if (lands) land_nd = made; }
            }
            st.c_node += 1; st.c_push += 1;
        }
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
        if (stat_k) { g_commit_stats[0] += 1; g_commit_stats[1] += stat_k; g_commit_stats[2] += stat_movers; g_commit_stats[3] += stat_movers == 0; g_commit_stats[4] += stat_k >= 3; g_commit_stats[5] += (stat_k >= 3) & (stat_movers == 0);
                      g_commit_stats[6] += (stat_k >= 3) ? stat_movers : 0; }
#endif
        if constexpr (PC) {
            // the frame the sift moved into the emptied slot first (its loads have long arrived: they are older than the ancestors the pushes waited for) ...
            if (pf_valid) pc_store(A, top_idx - 1u, pf_id, pf1, pf2, pf3);
            pf_valid = false;
            // ... then the child that bubbled into slot 1 or 2, if any (it may have displaced that frame): the lane that built the child holds its payload
            if (land_t != 0xFFu) {
#if defined(MAPAD_PC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
                g_pc_stats[4] += 1;  // steps in which a child ended up in slot 1 or 2
#endif
                const uint32_t lid = id0 + (uint32_t)popc32(cand0 & ((1u << land_t) - 1u));
                pc_store(A, land_s, lid, land_nd.w1, land_nd.w2, land_nd.w3);
            }
        }
        if constexpr (kLaneKids) {
            // Behind the sequential loop: inside it the stores sat between a push's loads and its wait, which then had to cover them as well (round 2).
#if defined(__HIP_DEVICE_COMPILE__)
            store_child_nodes();
#endif
        }
    }
    while (cand != 0 && st.status == ST_OK) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int t = __ffs((int)cand) - 1;
#else
        const int t = __builtin_ctz(cand);
#endif
        cand &= cand - 1;
        const bool is_ins = t == 0, is_del = (t & 1) != 0;
        const int i = is_ins ? 0 : (t - 1) >> 1, k = 3 - i;
        const float score = is_ins ? insertion_score : is_del ? deletion_score : (i == 0 ? mm[0] : i == 1 ? mm[1] : i == 2 ? mm[2] : mm[3]);
        const uint32_t ngaps = (is_ins || is_del) ? num_gaps_open : f.ngaps;
        const int len = is_del ? f.len : f.len + 1;
        if constexpr (kLaneKids) {
            const int b = k & (kBases - 1);  // which of the owner's bases
            Node sel_del = nd_del[0], sel_mm = nd_mm[0];
            if constexpr (kBases == 2) {  // word-wise selects (pick_node: a conditional on whole structs would pin them in scratch memory)
                const bool second = b != 0;
                sel_del = pick_node(second, false, nd_del[kBases - 1], nd_del[0], nd_del[0]);
                sel_mm = pick_node(second, false, nd_mm[kBases - 1], nd_mm[0], nd_mm[0]);
            }
            const Node nd = pick_node(is_ins, is_del, nd_ins, sel_del, sel_mm);
            const int owner = is_ins ? 0 : k / kBases;
            commit_child<LPR>(P, rd, A, st, alignment_start, score, ngaps, len, nd, w == owner, owner);
        } else {
            const uint64_t xl = k == 0 ? e.lower[0] : k == 1 ? e.lower[1] : k == 2 ? e.lower[2] : e.lower[3];
            const uint64_t xr = k == 0 ? e.lower_rev[0] : k == 1 ? e.lower_rev[1] : k == 2 ? e.lower_rev[2] : e.lower_rev[3];
            const uint64_t xs = k == 0 ? e.size[0] : k == 1 ? e.size[1] : k == 2 ? e.size[2] : e.size[3];
            commit_child<LPR>(P, rd, A, st, alignment_start, score, ngaps, len, make_child(t, k, xl, xr, xs), true, 0);
        }
    }
    if constexpr (PC) { if (MAPAD_UNLIKELY(pf_valid)) pc_store(A, top_idx - 1u, pf_id, pf1, pf2, pf3); }  // the step went through the general loop only (whose pushes are not tracked: they miss)
    MAPAD_MARK(PROF_COMMIT);
#if defined(MAPAD_PROFILE_SECTIONS) && defined(__HIP_DEVICE_COMPILE__)
    if (w == 0) atomicAdd(&g_prof_hist[24 + min(st.tree_len - prof_nodes0, 11u)], 1u);
#endif
    if (MAPAD_UNLIKELY(st.status != ST_OK)) return false;
    // :1348-1355
    if (MAPAD_UNLIKELY((st.n_hits > 9) | ((st.n_hits > 0) & (st.best_size > 1)))) return false;
    // :1358-1380
    if (MAPAD_UNLIKELY((st.heap_len > P.stack_limit) | (st.tree_len > P.edit_tree_limit))) {
        if (P.stack_limit_abort) { st.status = ST_LIMIT_ABORT; return false; }
        const int64_t a = (int64_t)st.heap_len - (int64_t)P.stack_limit;
        const int64_t b = (int64_t)st.tree_len - (int64_t)P.edit_tree_limit;
        SearchState tmp = st;
#if !defined(__HIP_DEVICE_COMPILE__)
        if (g_host_prefetch.next_pop && tmp.heap_len > 8) {
            // The next step pops slot 1 or 2 (evictions touch them only when the entry that takes the root's place beats a max-level one: rare) and starts with two
            // dependent misses: the frame's node — the payload cache is cleared below — and then its index blocks.  Asked for here, the node travels during the
            // first eviction's sift and the index blocks during the others.
            __builtin_prefetch(&A.nodes[A.top[1].node]); __builtin_prefetch(&A.nodes[A.top[2].node]);
            evict_worst(A, tmp, 1);
            uint32_t ni;
            const HeapEntry nt = mm_find_max(A, tmp.heap_len, ni);
            const Frame nf = unpack_frame(A.nodes[nt.node]);
            const uint64_t nx = (nf.start <= L - nf.start - nf.len) ? nf.lower_rev : nf.lower;
            __builtin_prefetch(block_ptr(ix, nx ? nx - 1 : 0)); __builtin_prefetch(block_ptr(ix, nx + nf.size - 1));
            evict_worst(A, tmp, (a > b ? a : b) - 1);
        } else
#endif
        evict_worst(A, tmp, a > b ? a : b);
        if constexpr (PC) pc_clear(A);  // slots 1 and 2 may hold other entries now, and the freed node ids will be reused
        drain_memory();
        st = tmp;
    }
    if constexpr (PFT::on) {  // the next step's node (NodePrefetch): the loads stay in flight across the kernel's loop head
        pf.ok = false;
        if (st.heap_len > 0) {
            uint32_t ni;
            const HeapEntry nt = mm_find_max(A, st.heap_len, ni);
            const Node g = A.nodes[nt.node];
            pf.w1 = g.w1; pf.w2 = g.w2; pf.w3 = g.w3; pf.id = nt.node; pf.ok = true;
        }
    }
    return st.heap_len > 0;
}

// one read from start to end by one thread (host builds: tests/emu, the host tail of host_tail.hpp); A.pc set = with the payload cache
template <class Grow = NoGrow>
MAPAD_HD void search_read(const DevIndex& ix, const DevParams& P, const ReadIn& rd, Arena& A, SearchState& st, int w, const Grow& grow = Grow()) {
    search_init(ix.n, alignment_start_of(P, rd.L), rd, A, st);
    if (A.pc) {
        pc_clear(A);
        if (P.bound_kind == BOUND_CONTINUOUS) { while (search_step<1, true, false, true>(ix, P, rd, A, st, w, grow)) {} }
        else { while (search_step<1, false, false, true>(ix, P, rd, A, st, w, grow)) {} }
    } else if (P.bound_kind == BOUND_CONTINUOUS) { while (search_step<1, true, false>(ix, P, rd, A, st, w, grow)) {} }
    else { while (search_step<1, false, false>(ix, P, rd, A, st, w, grow)) {} }
}

}  // namespace mapad
