// heap_selftest.cpp — TEST-ONLY: the product's min-max heap (mapad_amd/csrc/heap_core.hpp, host build, near levels + arena levels in the build's physical
// layout: HeapLayout — subtree-contiguous 64-byte blocks by default, the implicit array with -DMAPAD_SUBTREE_HEAP=0) against the oracle's MinMaxHeap
// (oracle/mapad_oracle.hpp: a plain std::vector) under random pushes, pop_max and pop_min, slot by slot.  The layout is a logical -> physical slot change and must
// leave the heap, entry for entry, what it was; the mapping tests only reach the levels their reads' heaps grow to, this test reaches 2^20 entries, pop_min sifts
// through arena levels and every level boundary.  Built by tests/emu_util.py; never loaded by the product.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../oracle/mapad_oracle.hpp"
#include "../../mapad_amd/csrc/heap_core.hpp"

using namespace mapad;

namespace {
template <int TOP>
uint64_t layout_properties(uint32_t n_max) {
    using HL = HeapLayout<TOP>;
    uint64_t bad = 0;
    std::vector<uint8_t> used((size_t)HL::phys_end(n_max) + 64, 0);
    uint32_t prev_end = HL::phys_end(0);
#if MAPAD_SUBTREE_HEAP
    bad += prev_end < (uint32_t)TOP + 1;  // the shadow of the near levels stays in front
#endif
    for (uint32_t i = TOP; i < n_max; ++i) {
        const uint32_t off = HL::slot(i), phys = off + 1;  // physical entry counted from the allocation's start
        const uint32_t end = HL::phys_end(i + 1);
        bad += phys <= (uint32_t)TOP;       // never inside the shadow
        bad += phys >= end;                 // covered by what a migration copies
        bad += end < prev_end;              // monotone
        prev_end = end;
        if (phys < used.size()) { bad += used[phys]; used[phys] = 1; } else bad += 1;  // injective
        if (i & 1) {                        // first of a sibling pair: 16-byte aligned, its sibling next to it
            bad += (phys & 1u) != 0;
            bad += HL::slot(i + 1) != off + 1;
        }
        if (i % 4 == 3) {                   // first of four grandchildren of (i - 3) / 4
            const uint32_t gp = (i - 3) / 4;
            const bool gp_max_level = !mm_is_min_level(gp);
#if MAPAD_SUBTREE_HEAP
            // a max-level entry's children and grandchildren share one 64-byte block; a min-level entry's grandchildren are the first pairs of two adjacent blocks
            if (gp_max_level) { bad += HL::slot(i + 2) != off + 2; if (2 * gp + 1 >= (uint32_t)TOP) bad += HL::slot(2 * gp + 1) != off - 2; bad += ((phys - 2) & 7u) != 0; }
            else bad += HL::slot(i + 2) != off + 8;
#else
            (void)gp_max_level;
            bad += HL::slot(i + 2) != off + 2;
#endif
        }
    }
    return bad;
}
}  // namespace

extern "C" {

// structural properties of HeapLayout<63> (quads, host tail) and HeapLayout<31> (pairs) for logical slots below n_max; returns the number of violations (must be 0)
uint64_t heap_layout_properties(uint32_t n_max) { return layout_properties<63>(n_max) + layout_properties<31>(n_max); }

// `ops` random operations on a heap that grows to about max_n entries and shrinks again; scores from `levels` distinct values (few = ties everywhere).
// After every operation the popped entry must equal the oracle's, and every `check_every` operations (and at the end) all slots are compared.
// Returns the number of mismatches (must be 0).
uint64_t heap_random_ops_selftest(uint64_t seed, uint32_t ops, uint32_t max_n, uint32_t levels, uint32_t check_every) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1, bad = 0;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::vector<HeapEntry> top(kTop + 9), heap((size_t)std::max<uint32_t>(2 * max_n, HeapLayout<kTop>::phys_end(max_n + 16)) + 256);
    Arena A;
    A.top = top.data() + 1; A.heap = heap.data() + 1;
    mo::MinMaxHeap ref;
    ref.variant = MAPAD_HEAP_VARIANT;
    uint32_t n = 0, next_id = 1;
    auto compare_all = [&]() {
        if (ref.v.size() != n) { bad += 1; return; }
        for (uint32_t i = 0; i < n; ++i) { const HeapEntry e = hp_get(A, i); bad += !(e.score == ref.v[i].alignment_score && e.node == ref.v[i].edit_node_id); }
    };
    for (uint32_t t = 0; t < ops; ++t) {
        // grow in the first half, shrink in the second; pop_min is the rarer pop, as in the search (evictions)
        const bool growing = t < ops / 2;
        const uint64_t r = rnd() % 100;
        const bool push = n == 0 || (n < max_n && (growing ? r < 70 : r < 30));
        if (push) {
            const float score = -(float)(rnd() % levels) * 0.75f;
            mo::Frame f; f.alignment_score = score; f.edit_node_id = next_id;
            ref.push(f);
            mm_bubble_up(A, n, HeapEntry{score, next_id});
            n += 1; next_id += 1;
        } else if (r % 4 == 0) {
            mo::Frame f; ref.pop_min(f);
            const HeapEntry e = mm_pop_min(A, n);
            bad += !(e.score == f.alignment_score && e.node == f.edit_node_id);
        } else {
            mo::Frame f; ref.pop_max(f);
            uint32_t idx;
            const HeapEntry e = mm_find_max(A, n, idx);
            mm_remove_at<true>(A, n, idx);
            bad += !(e.score == f.alignment_score && e.node == f.edit_node_id);
        }
        if (check_every && t % check_every == 0) compare_all();
    }
    compare_all();
    return bad;
}

}  // extern "C"
