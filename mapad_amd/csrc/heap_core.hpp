// heap_core.hpp — what a read slot keeps in memory (32-byte tree nodes that double as frames, 8-byte frontier-heap entries, the arena that holds them) and the
// search frontier itself: a min-max heap with the semantics of the `min-max-heap` crate the reference uses (call sites src/map/mapping.rs:147,986,1058,1376).
// Split out of search_core.hpp in round 5; compiled for the device (quads, heavy wavefronts) and for the host (tests/emu, host_tail.hpp) alike.
#pragma once
#include <type_traits>

#include "fmd_device.hpp"

// MAPAD_HEAP_VARIANT — the two details of the crate's tie handling that the reference's own tests cannot pin (the crate's source is not in /root/reference;
// oracle/mapad_oracle.hpp: MinMaxHeap::variant carries the same four readings, all of which pass the 43 known-answer tests):
//   bit 0 = 0: a trickle-down stride scans its candidates in ascending index order — child 1, child 2, then the four grandchildren (default)
//   bit 0 = 1: ... in family order — child 1, its two children, child 2, its two children
//   bit 1 = 0: pop_max takes slot 2 when slots 1 and 2 tie (v[1] > v[2] ? 1 : 2; default)      bit 1 = 1: slot 1 (v[1] >= v[2])
// A later candidate wins only if strictly better in every reading.  The build constant selects the reading for every kernel and for the host tail at once
// (mapad_amd/build.py: build(heap_variant=V) -> libmapad_amd.hvV.so); pinning the crate later is a one-flag change.  Exposure measured in
// profiles/heap_variant_exposure.json: hit intervals and scores identical across readings, edit tracks (CIGAR / MD) differ for 0.2-0.4 % of indel reads.
#if !defined(MAPAD_HEAP_VARIANT)
#define MAPAD_HEAP_VARIANT 0
#endif
static_assert(MAPAD_HEAP_VARIANT >= 0 && MAPAD_HEAP_VARIANT <= 3, "MAPAD_HEAP_VARIANT is a two-bit mask");

// Rare paths (a hit is found, limit recovery, read set-up).  Out-of-line variants were measured on MI355X (C2): they shrink the
// kernel by 30 % of its instructions but cost 5-10 % run time and +25 % memory traffic (call-site spills through scratch),
// so they are inlined by default; -DMAPAD_OUTLINE_RARE builds the out-of-line variant.
#define MAPAD_UNLIKELY(x) __builtin_expect(!!(x), 0)
#if defined(__HIPCC__) && defined(MAPAD_OUTLINE_RARE)
#define MAPAD_RARE __host__ __device__ __attribute__((noinline))
#elif defined(__HIPCC__)
#define MAPAD_RARE __host__ __device__ __forceinline__
#else
#define MAPAD_RARE inline
#endif


// -DMAPAD_PROFILE_SECTIONS: wave time and lane time per section of the search loop (s_memtime deltas accumulated in LDS, dumped by the kernel).
// A diagnostic build: the marks cost a few percent and the numbers are relative.
#if defined(MAPAD_PROFILE_SECTIONS) && defined(__HIPCC__)
enum { PROF_POP = 0, PROF_NODE = 1, PROF_EXT = 2, PROF_GATES = 3, PROF_COMMIT = 4, PROF_TAIL = 5, PROF_SETUP = 6, PROF_FINALIZE = 7, PROF_GROW = 8, PROF_HIT = 9, PROF_LOOP = 10, PROF_C_PRE = 11, PROF_C_LOAD = 12, PROF_C_ANC = 13, PROF_N = 14 };
__shared__ unsigned long long g_prof_lds[2 * PROF_N + 2];  // [k] wave cycles, [PROF_N + k] lane cycles, [2 PROF_N] last stamp
__shared__ unsigned int g_prof_hist[64];  // [0..23] log2(heap_len) at pop, [24..35] children committed by a pop, [36..47] commit-loop trips of a wave step, [48..63] trickle levels
__device__ __forceinline__ void prof_mark(int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned long long act = __ballot(1);
    if ((int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0)) == 0) {  // first active lane
        const unsigned long long dt = t - g_prof_lds[2 * PROF_N];
        g_prof_lds[2 * PROF_N] = t;
        g_prof_lds[k] += dt;
        g_prof_lds[PROF_N + k] += dt * (unsigned long long)__popcll(act);
    }
#else
    (void)k;
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define MAPAD_MARK(k) prof_mark(k)
#else
#define MAPAD_MARK(k) ((void)0)
#endif
#else
#define MAPAD_MARK(k) ((void)0)
#endif

namespace mapad {

// The compiler's wait-count pass is not path sensitive: a load or store issued on a RARE path of the search loop counts as "possibly still in
// flight" at the join, and the common path then carries a full `s_waitcnt vmcnt(0)` — a drain of every store in flight — in front of the
// next instruction that touches one of the registers involved.  Rare paths therefore end with an explicit wait of their own, which the pass
// does model: behind it nothing is pending and the common path keeps only the waits it needs.
// Marks a loaded value as used here: the wait-count pass then places the wait for it at this point — chosen where younger loads are waited for anyway, so that it
// costs nothing — instead of carrying "possibly still in flight" around the loop into a full drain in front of the next write of the same register.
template <class T> MAPAD_HD void consume_here(const T& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#else
    (void)v;
#endif
}
MAPAD_HD void drain_memory() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
#endif
}

enum : uint32_t { GAP_INS = 0, GAP_DEL = 1, GAP_CLOSED = 2 };  // src/map/mod.rs:93-98

struct HeapEntry {
    float score;
    uint32_t node;
};

// 32-byte tree node + frame payload (MismatchSearchStackFrame, src/map/mod.rs:105-137, minus the score that lives in the heap)
//   w0 = op | parent << 32          (vacant slot: parent = next free key)
//   w1 = lower (40 bit) | start << 40 (15 bit: reads up to i16::MAX, record.rs:144-150)
//   w2 = lower_rev (40 bit) | gap_f << 40 | gap_b << 42 | ngaps << 44 (8 bit) | occupied << 52
//   w3 = size (40 bit) | len << 40 (15 bit)
struct alignas(16) Node {
    uint64_t w0, w1, w2, w3;
};
static_assert(sizeof(Node) == 32, "node + frame payload is 32 bytes");
constexpr uint64_t kMask40 = (1ull << 40) - 1;

struct Frame {
    uint64_t lower, lower_rev, size;
    int32_t start, len;
    uint32_t gap_f, gap_b, ngaps;
};
MAPAD_HD Node pack_node(uint32_t op, uint32_t parent, const Frame& f) {
    Node n;
    n.w0 = (uint64_t)op | ((uint64_t)parent << 32);
    n.w1 = f.lower | ((uint64_t)(uint32_t)f.start << 40);
    n.w2 = f.lower_rev | ((uint64_t)f.gap_f << 40) | ((uint64_t)f.gap_b << 42) | ((uint64_t)f.ngaps << 44) | (1ull << 52);
    n.w3 = f.size | ((uint64_t)(uint32_t)f.len << 40);
    return n;
}
MAPAD_HD Frame unpack_frame(const Node& n) {
    Frame f;
    f.lower = n.w1 & kMask40; f.start = (int32_t)(n.w1 >> 40);
    f.lower_rev = n.w2 & kMask40; f.gap_f = (uint32_t)(n.w2 >> 40) & 3; f.gap_b = (uint32_t)(n.w2 >> 42) & 3; f.ngaps = (uint32_t)(n.w2 >> 44) & 0xFF;
    f.size = n.w3 & kMask40; f.len = (int32_t)(n.w3 >> 40);
    return f;
}
// word-wise selects: a conditional expression on whole structs selects an ADDRESS and copies from it, which pins all three nodes in scratch memory
MAPAD_HD Node pick_node(bool first, bool second, const Node& a, const Node& b, const Node& c) {
    Node n;
    n.w0 = first ? a.w0 : second ? b.w0 : c.w0; n.w1 = first ? a.w1 : second ? b.w1 : c.w1;
    n.w2 = first ? a.w2 : second ? b.w2 : c.w2; n.w3 = first ? a.w3 : second ? b.w3 : c.w3;
    return n;
}
MAPAD_HD uint32_t node_op(const Node& n) { return (uint32_t)n.w0; }
MAPAD_HD uint32_t node_parent(const Node& n) { return (uint32_t)(n.w0 >> 32); }
MAPAD_HD bool node_occupied(const Node& n) { return (n.w2 >> 52) & 1; }

struct HitRec {  // 40 bytes; the public hit record (include/mapad_amd.h: mapad_hit_t)
    uint64_t lower, lower_rev, size;
    float score;
    uint32_t n_ops;
    uint32_t ops_off;  // into the per-read staging area, later into the global ops pool
    uint32_t pad;
};
static_assert(sizeof(HitRec) == 40, "hit record is 40 bytes");

// Per-read position data kept next to the quad (LDS on the device; MAPAD_MAX_LDS_READ_LEN and shorter reads):
//   qc[2j] = read-base class (0..3 = ACGT, 4 otherwise), qc[2j+1] = Phred quality  -> one shared score-table row per pop
//   d[j]   = BiDArray::d_composite[j] (written by darray_kernel)
#if !defined(MAPAD_KTOP)
#define MAPAD_KTOP 63
#endif
constexpr int kTop = MAPAD_KTOP;  // logical heap slots 0..kTop-1 (63: levels 0-5) live in the `top` array (LDS on the device); must be 2^k - 1

constexpr int kMaxHits = 20;  // a pop adds <= 9 hits and the search returns once more than 9 exist (mapping.rs:1348)

enum : uint32_t { ST_OK = 0, ST_ARENA_OVERFLOW = 1, ST_LIMIT_ABORT = 2 };

// "Near" data of a read slot (heap top, position data) is addressed through LDS-typed pointers on the device when NL is set, so the
// compiler emits ds_* instructions for it and global_* for the arena (a pointer that may be either would force flat_* accesses,
// which occupy the texture addresser and make every wait a full vmcnt(0)+lgkmcnt(0) wait).  NL = false: plain pointers (host build,
// very long reads whose near data stays in the HBM arena).
// The arena pointers of a grown read come out of a descriptor in memory, so their address space must be spelled out as well: left
// generic they turn every heap and node access into a flat_* instruction, which counts against both vmcnt and lgkmcnt and so
// serialises the LDS and the HBM halves of every sift.
#if defined(__HIP_DEVICE_COMPILE__)
#define MAPAD_LDS __attribute__((address_space(3)))
#define MAPAD_GLOBAL __attribute__((address_space(1)))
#else
#define MAPAD_LDS
#define MAPAD_GLOBAL
#endif
template <class T, bool NL> struct near_ptr { using type = T*; };
template <class T> struct near_ptr<T, true> { using type = MAPAD_LDS T*; };

// TOP = logical heap slots kept in the near array (2^k - 1): 63 for a quad's read slot, 1023 for a read that has a wavefront to itself (heavy_core.hpp)
template <bool NL, int TOP = kTop>
struct ArenaT {
    static constexpr int kTopN = TOP;
    typename near_ptr<HeapEntry, NL>::type top;  // logical heap slots [0, TOP), shifted by one entry like `heap`
    MAPAD_GLOBAL HeapEntry* heap;    // logical heap slots [TOP, ..) are used from here
    MAPAD_GLOBAL Node* nodes;
    MAPAD_GLOBAL HitRec* hits;       // kMaxHits
    MAPAD_GLOBAL uint32_t* hit_ops;  // staging for the hits' edit tracks
    MAPAD_GLOBAL uint16_t* scratch;  // 2 * (Lmax + 1) u16 for the bucket sort of extract_edit_operations
    typename near_ptr<uint64_t, NL>::type pc = nullptr;  // payload cache of heap slots 1 and 2 (search_step<.., PC = true>): 2 x {1 << 32 | node id, w1, w2, w3}, near data
    uint32_t heap_cap, node_cap, hit_ops_cap;
    uint32_t grown = 0;  // 0: heap/nodes are the slot's base arena; else (class + 1) << 27 | arena index (mapad_amd.hip: DeviceGrow)
    uint32_t wait = 0;   // steps to sit out before asking the pools again
    uint32_t n_waits = 0;  // fruitless requests of the current read
};

using Arena = ArenaT<false>;

template <bool NL>
struct ReadInT {
    typename near_ptr<const uint8_t, NL>::type qc;  // 2 bytes per position: base class, quality
    typename near_ptr<const float, NL>::type d;     // D array
    int L;
    float thr;          // DevParams::reject_thr[L]
    int32_t table;      // DevParams::table_base[L]
    uint64_t lane_less = 0;  // device quads: Less of the base this lane extends by (DevIndex::less[w + 1]), picked once per kernel; pairs: of base 2w
    uint64_t lane_less1 = 0; // device pairs (lanes-per-read 2): Less of this lane's second base, 2w + 1
};
using ReadIn = ReadInT<false>;

struct alignas(16) HeapPair { HeapEntry a, b; };
MAPAD_HD HeapPair load_pair(const HeapEntry* p) {  // p is 16-byte aligned
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    HeapPair r;
    r.a.score = __uint_as_float(q.x); r.a.node = q.y; r.b.score = __uint_as_float(q.z); r.b.node = q.w;
    return r;
#else
    HeapPair r;
    MAPAD_TOUCH(p, sizeof r, false);
    std::memcpy(&r, p, sizeof r);
    return r;
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ HeapPair load_pair(const MAPAD_GLOBAL HeapEntry* p) {
    const uint4 q = *(const MAPAD_GLOBAL uint4*)p;
    HeapPair r;
    r.a.score = __uint_as_float(q.x); r.a.node = q.y; r.b.score = __uint_as_float(q.z); r.b.node = q.w;
    return r;
}
__device__ __forceinline__ HeapPair load_pair(const MAPAD_LDS HeapEntry* p) {
    const uint4 q = *(const MAPAD_LDS uint4*)p;
    HeapPair r;
    r.a.score = __uint_as_float(q.x); r.a.node = q.y; r.b.score = __uint_as_float(q.z); r.b.node = q.w;
    return r;
}
#endif

// ---- where a logical heap slot of the arena levels lives (round 6) ---------------------------------------------------------------------------------------
// MAPAD_SUBTREE_HEAP=0 (default): the implicit array, logical slot i at A.heap + i.
// MAPAD_SUBTREE_HEAP=1 (libmapad_amd.sub.so, mapad_amd/build.py): the arena's heap levels in subtree-contiguous 64-byte blocks.  Every entry p of an ODD level (a
// max level: the levels a pop_max's sift steps through) owns one block of eight 8-byte slots: its two children in slots 0-1, its four grandchildren in slots 2-5 —
// exactly what one stride of the sift reads — so a stride is ONE 128-byte line instead of two (children pair + grandchildren quad of the implicit array lie in
// different lines from level 8 on), and a push's leaf shares its block with its parent (odd leaf level) or its parent and grandparent share theirs (even).
// Entries of even levels live in their parent's block, entries of odd levels in their grandparent's; blocks are numbered level by level (key levels K0,
// K0 + 2, ...; K0 = the last odd level whose grandchildren are arena levels).  The first TOP + 1 physical entries stay what they were: the unused shadow of the
// near levels (a hand-over writes the near levels there).  Purely a logical -> physical slot change: the heap, entry for entry, is the same min-max heap
// (tests/emu/heap_selftest.cpp; the GPU parity suite runs on both builds).
// Measured, same box (profiles/r06/ab_layout.txt, ab_pmc_layout.txt): the blocks take 15-21 % of the READ requests behind the L2 away (C3 7.79 -> 6.12 per pop,
// C2 5.42 -> 4.58; writes unchanged) — and the kernel is 4-6 % SLOWER (C3 2.60 -> 2.47 M reads/s, C2 6.04 -> 5.75 M, C4 3.10 -> 2.97 M).  MAPAD_SUBTREE_HEAP=2
// separates the two effects: the blocks' address arithmetic computed and kept alive, the implicit array addressed — C3 2.42 M, C2 5.78 M: the arithmetic (a dozen
// dependent VALU instructions in front of every arena access of the sift and the pushes, +28 % static VALU in the kernel) costs 4-7 %, and a fifth fewer read
// requests buy back 0-2 %.  The step is a chain of dependent round trips whose length, not the request rate behind the L2, sets its time; the implicit array stays.
#if !defined(MAPAD_SUBTREE_HEAP)
#define MAPAD_SUBTREE_HEAP 0
#endif
#if !defined(MAPAD_UNIFORM_SIFT_STORES)
#define MAPAD_UNIFORM_SIFT_STORES 0  // (mm_trickle_down: arena strides store to the arena without hp_set's per-lane near / arena branch; measured +-0 on C2 / C3 / C4, profiles/r06/ab_lds_pad_and_uniform_stores.txt: off)
#endif
#if !defined(MAPAD_MAX_SIFT_ARENA_ONLY)
#define MAPAD_MAX_SIFT_ARENA_ONLY 1  // (mm_trickle_down; 0 for A/B runs)
#endif
#if defined(MAPAD_HEAVY_KERNEL) && MAPAD_SUBTREE_HEAP != 0
#error "heavy_kernel's wavefront-wide strides address the implicit heap array"
#endif
template <int TOP>
struct HeapLayout {
    static constexpr int kT = TOP >= 1023 ? 10 : TOP >= 511 ? 9 : TOP >= 255 ? 8 : TOP >= 127 ? 7 : TOP >= 63 ? 6 : TOP >= 31 ? 5 : TOP >= 15 ? 4 : 3;  // first arena level
    static_assert(TOP == (1 << kT) - 1, "the near levels are whole levels");
    static constexpr int kK0 = (kT & 1) ? kT - 2 : kT - 1;       // first key level (odd)
    static constexpr uint32_t kC0 = ((1u << kK0) + 1u) / 3u;
    // logical slot i >= TOP -> offset from A.heap (which points one entry into the 16-byte aligned allocation)
    static MAPAD_HD uint32_t slot(uint32_t i) {
#if MAPAD_SUBTREE_HEAP
        const uint32_t x = i + 1u;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t level = 31u - (uint32_t)__clz((int)x);
#else
        const uint32_t level = 31u - (uint32_t)__builtin_clz(x);
#endif
        const bool even = (level & 1u) == 0u;
        const uint32_t owner = even ? x >> 1 : x >> 2;                  // 1-based index of the odd-level entry whose block holds i
        const uint32_t s = even ? (x & 1u) : 2u + (x & 3u);
        const uint32_t m = (even ? 1u << level : 1u << (level - 1u));  // 2^(owner's level + 1), a power of four: (m - 1) / 3 = (m - 1) & 0x5555...
        const uint32_t block = owner - ((m - 1u) & 0x55555555u) - kC0;
#if MAPAD_SUBTREE_HEAP == 2
        // experiment (profiles/r06/ab_layout.txt): the subtree layout's address arithmetic is computed and kept alive, the implicit array is addressed — separates what the
        // layout's instructions cost from what its memory behaviour gives
        uint32_t keep = (uint32_t)TOP + 8u * block + s;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(keep));
#endif
        return i + (keep & 0u);
#endif
        return (uint32_t)TOP + 8u * block + s;
#else
        return i;
#endif
    }
    // physical entries (counted from the allocation's start, A.heap - 1) that hold logical slots [0, n), the shadow of the near levels included; monotone in n
    static MAPAD_HD uint32_t phys_end(uint32_t n) {
#if MAPAD_SUBTREE_HEAP
        if (n <= (uint32_t)TOP) return (uint32_t)TOP + 1u;
        const uint32_t i = n - 1u;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t level = 31u - (uint32_t)__clz((int)n);
#else
        const uint32_t level = 31u - (uint32_t)__builtin_clz(n);
#endif
        if (level & 1u) return (uint32_t)TOP + 1u + 8u * (((1u << level) - (1u << kK0)) / 3u);  // the even level above is full: every block of key level `level - 2` is in use
        return ((slot(i) + 1u) & ~7u) + 8u;                                                       // blocks of key level `level - 1` up to the parent's (blocks start at physical multiples of 8)
#else
        return n + 1u;
#endif
    }
};

// heap slot i of this read: the top levels sit in the near array (LDS), the rest in the HBM arena
// Entries are moved field by field: copying the struct would bind the source to a reference in the generic address space, and after
// the near/arena branches are merged the access would stay a flat_* instruction (vmcnt and lgkmcnt, no overlap with anything).
template <class P> MAPAD_HD HeapEntry load_entry(P p) { HeapEntry e; MAPAD_TOUCH(&*p, sizeof e, false); e.score = p->score; e.node = p->node; return e; }
template <class P> MAPAD_HD void store_entry(P p, const HeapEntry e) { MAPAD_TOUCH(&*p, sizeof e, true); p->score = e.score; p->node = e.node; }
template <bool NL, int TOP> MAPAD_HD HeapEntry hp_get(const ArenaT<NL, TOP>& A, uint32_t i) {
    if (i < (uint32_t)TOP) return load_entry(A.top + i);
    return load_entry(A.heap + HeapLayout<TOP>::slot(i));
}
template <bool NL, int TOP> MAPAD_HD void hp_set(const ArenaT<NL, TOP>& A, uint32_t i, const HeapEntry e) {
    if (i < (uint32_t)TOP) store_entry(A.top + i, e);
    else store_entry(A.heap + HeapLayout<TOP>::slot(i), e);
}

// ---- min-max heap (index 0 = min; even levels are min levels) ----------------------------------------------------
MAPAD_HD bool mm_is_min_level(uint32_t pos) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__clz((int)(pos + 1)) & 1) == 1;
#else
    return (__builtin_clz(pos + 1) & 1) == 1;
#endif
}

// "a beats b" on a min level means a > b, on a max level a < b.  Flipping the sign bit of both operands reverses the order of two floats
// (scores are never NaN), so one compare serves both cases: the two-sided form `(min & (a > b)) | (!min & (a < b))` came out as two compares,
// two materialised booleans, a select and a third compare.
MAPAD_HD float flip_sign(float x, uint32_t mask) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) ^ mask); }

// The first two compares of a bubble-up (parent, then the grandparent of wherever the element sits after the first compare) decide
// 98 % of all pushes (measured, C2/C3); their three possible slots are known from `pos` alone, so they are loaded together and the
// dependent chain of a push is one memory round trip instead of two.
struct Ancestors { HeapEntry e1, e2, e3; };  // parent, grandparent of pos, grandparent of the parent (slot 0 where there is none)
template <bool NL, int TOP>
MAPAD_HD Ancestors load_ancestors(const ArenaT<NL, TOP>& A, uint32_t pos) {
    const uint32_t i1 = pos > 0 ? (pos - 1) >> 1 : 0, i2 = pos > 2 ? (pos - 3) >> 2 : 0, i3 = i1 > 2 ? (i1 - 3) >> 2 : 0;
    Ancestors a;  // i3 <= i2 <= i1
    // The read slots of a wavefront are at different heap sizes: a four-way branch on where the three entries live runs its cases one after
    // the other, each with its own wait for memory.  Here every slot reads the near array (index clamped into it) and only the arena loads
    // are predicated, back to back, so that a wavefront waits for memory once.
    const uint32_t k1 = i1 < (uint32_t)TOP ? i1 : 0, k2 = i2 < (uint32_t)TOP ? i2 : 0, k3 = i3 < (uint32_t)TOP ? i3 : 0;
    const HeapEntry n1 = load_entry(A.top + k1), n2 = load_entry(A.top + k2), n3 = load_entry(A.top + k3);
    HeapEntry g1 = HeapEntry{0.0f, 0u}, g2 = HeapEntry{0.0f, 0u}, g3 = HeapEntry{0.0f, 0u};
    if (i1 >= (uint32_t)TOP) g1 = load_entry(A.heap + HeapLayout<TOP>::slot(i1));
    if (i2 >= (uint32_t)TOP) g2 = load_entry(A.heap + HeapLayout<TOP>::slot(i2));
    if (i3 >= (uint32_t)TOP) g3 = load_entry(A.heap + HeapLayout<TOP>::slot(i3));
    a.e1 = i1 < (uint32_t)TOP ? n1 : g1; a.e2 = i2 < (uint32_t)TOP ? n2 : g2; a.e3 = i3 < (uint32_t)TOP ? n3 : g3;
    return a;
}
template <bool NL, int TOP>
MAPAD_HD uint32_t mm_bubble_up(const ArenaT<NL, TOP>& A, uint32_t pos, const HeapEntry elt, const Ancestors& an) {  // elt is the new element, destined for slot pos; returns the slot it ends up in
    // Both compares are evaluated unconditionally (slots that do not exist compare as "stay"): the three entries are then consumed on the main
    // path, where the compiler places the one wait for them, and the outcome is a store of elt plus at most two displaced entries.
    const uint32_t i1 = pos > 0 ? (pos - 1) >> 1 : 0;   // parent
    const uint32_t i2 = pos > 2 ? (pos - 3) >> 2 : 0;   // grandparent of pos
    const uint32_t i3 = i1 > 2 ? (i1 - 3) >> 2 : 0;     // grandparent of the parent
    const HeapEntry e1 = an.e1, e2 = an.e2, e3 = an.e3;
    const bool min_level = mm_is_min_level(pos);
    // (bitwise on purpose: short-circuit forms of these predicates come out as nested branches on the device)
    const uint32_t flip1 = min_level ? 0u : 0x80000000u;
    const bool moved = (pos > 0) & (flip_sign(elt.score, flip1) > flip_sign(e1.score, flip1));
    const bool greater = min_level == moved;            // which grandparent chain to follow
    const uint32_t flip2 = greater ? 0u : 0x80000000u;
    const float elt_key = flip_sign(elt.score, flip2);  // along the chain: climbs while its key is greater
#if defined(MAPAD_PROFILE_SECTIONS) && defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(moved) != 0xFFFFFFFFFFFFFFFFull || e2.score != e3.score) MAPAD_MARK(PROF_C_LOAD);  // forces the wait for the three entries before the mark
#endif
    const uint32_t pos1 = moved ? i1 : pos;
    const HeapEntry ge = moved ? e3 : e2;
    const uint32_t gp = moved ? i3 : i2;
    const bool moved2 = (pos1 > 2) & (elt_key > flip_sign(ge.score, flip2));
    if (moved) hp_set(A, pos, e1);
    if (moved2) {
        hp_set(A, pos1, ge);
        pos = gp;
        if (MAPAD_UNLIKELY(pos > 2)) {  // 2 % of the pushes climb further
            while (pos > 2) {
                const uint32_t g2 = (pos - 3) >> 2;
                const HeapEntry g = hp_get(A, g2);
                if (!(elt_key > flip_sign(g.score, flip2))) break;
                hp_set(A, pos, g);
                pos = g2;
            }
            drain_memory();
        }
    } else pos = pos1;
    hp_set(A, pos, elt);
    return pos;
}
template <bool NL, int TOP>
MAPAD_HD uint32_t mm_bubble_up(const ArenaT<NL, TOP>& A, uint32_t pos, const HeapEntry elt) { return mm_bubble_up(A, pos, elt, load_ancestors(A, pos)); }
// What a bubble-up of `elt` from the leaf `pos` will write ABOVE the leaf, judged from its three ancestors alone (the first two compares of mm_bubble_up): the
// parent's slot if it swaps with the parent (w1), the grandparent's slot of wherever it then sits if it climbs there (w2), and whether it may climb on from there
// (`climbs`: mm_bubble_up's rare third and further levels, which read and write slots four and more levels above the leaf).  kNoSlot where it writes nothing.
// The lane-parallel commit lets several movers of a frame bubble up side by side when no later one's ancestors are among an earlier one's writes.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
struct MoverPlan { uint32_t w1, w2; bool climbs; };
MAPAD_HD MoverPlan mm_mover_plan(uint32_t pos, const HeapEntry elt, const Ancestors& an) {
    const uint32_t i1 = pos > 0 ? (pos - 1) >> 1 : 0, i2 = pos > 2 ? (pos - 3) >> 2 : 0, i3 = i1 > 2 ? (i1 - 3) >> 2 : 0;
    const bool min_level = mm_is_min_level(pos);
    const uint32_t flip1 = min_level ? 0u : 0x80000000u;
    const bool moved = (pos > 0) & (flip_sign(elt.score, flip1) > flip_sign(an.e1.score, flip1));
    const bool greater = min_level == moved;
    const uint32_t flip2 = greater ? 0u : 0x80000000u;
    const uint32_t pos1 = moved ? i1 : pos, gp = moved ? i3 : i2;
    const float ge = moved ? an.e3.score : an.e2.score;
    const bool moved2 = (pos1 > 2) & (flip_sign(elt.score, flip2) > flip_sign(ge, flip2));
    MoverPlan m;
    m.w1 = moved ? i1 : kNoSlot; m.w2 = moved2 ? gp : kNoSlot; m.climbs = moved2 & (gp > 2);
    return m;
}
// Does a mover whose ancestors sit in slots (i1, i2, i3 of `pos`) depend on what an EARLIER mover with plan `a` writes?  (Both may climb on: their further
// levels can meet.)  A mover's own writes above its leaf are among its ancestors' slots, so two movers that do not depend on each other write disjoint slots.
MAPAD_HD bool mm_mover_depends(uint32_t pos, bool climbs, const MoverPlan& a) {
    const uint32_t i1 = pos > 0 ? (pos - 1) >> 1 : 0, i2 = pos > 2 ? (pos - 3) >> 2 : 0, i3 = i1 > 2 ? (i1 - 3) >> 2 : 0;
    const bool hit1 = (a.w1 != kNoSlot) & ((a.w1 == i1) | (a.w1 == i2) | (a.w1 == i3));
    const bool hit2 = (a.w2 != kNoSlot) & ((a.w2 == i1) | (a.w2 == i2) | (a.w2 == i3));
    return hit1 | hit2 | (climbs & a.climbs);
}
// Would mm_bubble_up leave `elt` in slot pos (neither of its first two compares moves it)?  Such a push stores one entry and touches no other slot.
MAPAD_HD bool mm_push_stays(uint32_t pos, const HeapEntry elt, const Ancestors& an) {
    const bool min_level = mm_is_min_level(pos);
    const uint32_t flip1 = min_level ? 0u : 0x80000000u;
    const bool moved = (pos > 0) & (flip_sign(elt.score, flip1) > flip_sign(an.e1.score, flip1));
    const uint32_t flip2 = min_level ? 0x80000000u : 0u;  // not moved: the grandparent chain of the other kind of level
    const bool moved2 = (pos > 2) & (flip_sign(elt.score, flip2) > flip_sign(an.e2.score, flip2));
    return !moved & !moved2;
}

// The near levels (and, with MAPAD_SUBTREE_HEAP=0, the arena levels) are stored shifted by one entry (logical index i lives in physical slot i + 1; `v` points
// at logical 0), so the two children of a node (logical 2p+1, 2p+2) form one 16-byte aligned pair and its four grandchildren (4p+3 .. 4p+6) one
// 32-byte aligned group: a trickle-down level is three 16-byte loads instead of six 8-byte ones.  The arena levels' subtree blocks (HeapLayout) keep those
// pairs aligned too: children in slots 0-1, grandchildren in slots 2-5 of a 64-byte block.
// candidates scanned in ascending index order (child1, child2, grandchildren) — or in family order, MAPAD_HEAP_VARIANT bit 0 —; a later one wins only if strictly better.
// `elt` is the element being placed, starting at the hole `pos`.  Entries at or beyond n are stale memory: they are loaded
// (the arena has slack) but neutralised by an index test.
// A sift that starts at slot 1 or 2 (every pop_max of a heap with more than two entries) takes its first two strides through levels 1-5, which
// with kTop = 63 lie in the near array entirely: those strides run without the near / arena selection of the general stride.
// `occupant(node)`: for a sift that starts in slots 0-2, called once, as soon as it is known which entry ends up in the slot the sift started from (after
// the first stride, or at once if that slot has no children): search_step's payload cache fetches that frame while the rest of the sift and the rank
// queries are in flight.
struct NoOccupantHook { MAPAD_HD void operator()(uint32_t) const {} };
#if !defined(__HIP_DEVICE_COMPILE__)
// Host builds (the host tail of host_tail.hpp, tests/emu): what a step asks memory for ahead of time.  A read at the reference's limits walks a 16 MB heap and a
// 320 MB slab at random; its step is a chain of cache misses — the pop's sift, then one pop_min sift per evicted frame.  Results never depend on these requests.
// Measured on the round-4 GPU box (EPYC 9575F, 16 threads, 16 reads at the limits on the 3 Gbp index; profiles/r04/host_tail_ab.txt): no prefetch 0.400 us per pop,
// sift lookahead alone 0.413, with the next pop's node and index blocks 0.372; a second stride of sift lookahead (32 + 64 more entries per stride) 0.378: not kept.
// MAPAD_TAIL_PREFETCH=<sift lookahead 0|1><next pop 0|1> overrides (default "11").
struct HostPrefetch {
    int sift_lookahead = 1;  // a deep sift requests the next stride's candidates (8 + 16 entries) while the current stride is decided
    bool next_pop = true;    // before / between the evictions of a step: the node of the frame the next step will pop, then its two index blocks
};
inline HostPrefetch g_host_prefetch;
#endif

// SPEC (device quads, MAPAD_SPEC_SIFT): two strides per trip to the arena.  A sift through a deep heap is a chain of dependent trips — at C4 three or four of
// them per pop for the read slots with 2^12 ... 2^14 frames, each as long as the slowest of the wavefront's 16 read slots takes (an HBM miss) — and is the
// longest serial piece of a step; the rank-query loads (one such trip) hide behind it, not the other way round.  With the demand loads of a stride (the hole's
// children and grandchildren, the same addresses in all four lanes) lane w also asks for the children and grandchildren of grandchild w: whichever grandchild
// the hole moves to, the next stride's six candidates are then in that lane's registers (handed round by ds_bpermute), and the stride after next starts its
// trip one trip earlier.  The speculative lines are the ones the next stride would have asked for anyway (8 + 16 contiguous entries), plus their siblings.
template <bool MAX, bool NL, int TOP, class Hook = NoOccupantHook, bool SPEC = false>
MAPAD_HD void mm_trickle_down(const ArenaT<NL, TOP>& A, uint32_t n, uint32_t pos, HeapEntry elt, Hook&& occupant = Hook(), int w = 0) {
    bool going = true;
    uint32_t placed_node = elt.node;  // node of the entry that was stored into the slot a stride started from (elt itself if the stride stored nothing there)
    const uint32_t start = pos;
    // one stride: the hole moves to the best child or grandchild; false = the sift ends at `pos`
    auto stride = [&](const HeapPair& c, const HeapPair& ga, const HeapPair& gb, uint32_t c1, uint32_t g1, auto&& set) -> bool {
        uint32_t best = c1;
        HeapEntry be = c.a;
        auto consider = [&](uint32_t idx, const HeapEntry cand) {  // a later candidate wins only if strictly better; plain selects
            const bool take = (idx < n) & (MAX ? (cand.score > be.score) : (cand.score < be.score));
            best = take ? idx : best; be.score = take ? cand.score : be.score; be.node = take ? cand.node : be.node;
        };
        if constexpr ((MAPAD_HEAP_VARIANT & 1) == 0) { consider(c1 + 1, c.b); consider(g1, ga.a); consider(g1 + 1, ga.b); consider(g1 + 2, gb.a); consider(g1 + 3, gb.b); }
        else { consider(g1, ga.a); consider(g1 + 1, ga.b); consider(c1 + 1, c.b); consider(g1 + 2, gb.a); consider(g1 + 3, gb.b); }  // family order
        if (!(MAX ? (be.score > elt.score) : (be.score < elt.score))) return false;
        set(pos, be);
        placed_node = pos == start ? be.node : placed_node;
        pos = best;
        if (best < g1) return false;  // moved to a child: done
        const uint32_t parent = (pos - 1) >> 1;
        HeapEntry pe;  // the parent of a grandchild is one of the two children just loaded
        pe.score = parent == c1 ? c.a.score : c.b.score; pe.node = parent == c1 ? c.a.node : c.b.node;
        if (MAX ? (pe.score > elt.score) : (pe.score < elt.score)) { set(parent, elt); elt = pe; }
        return true;
    };
    auto set_near = [&](uint32_t i, const HeapEntry e) { store_entry(A.top + i, e); };
    auto set_any = [&](uint32_t i, const HeapEntry e) { hp_set(A, i, e); };
    auto set_arena = [&](uint32_t i, const HeapEntry e) { store_entry(A.heap + HeapLayout<TOP>::slot(i), e); };
    // (MAX sifts from slot 1 or 2, even number of near levels: see the loads below) the hole of the general loop's FIRST stride sits on the last near level, its
    // children and everything any later stride stores to are arena slots: only that first stride needs the per-lane near / arena choice of hp_set
    constexpr bool kArenaOnly = MAX && MAPAD_MAX_SIFT_ARENA_ONLY && TOP >= 31 && (HeapLayout<TOP>::kT % 2) == 0 && MAPAD_SUBTREE_HEAP == 0;
    bool first_general = true;
    if constexpr (TOP >= 31) {
        constexpr int kNearStrides = TOP >= 1023 ? 4 : TOP >= 255 ? 3 : TOP >= 63 ? 2 : 1;  // strides that stay inside the near levels when the sift starts at slot 1 or 2
#pragma unroll
        for (int k = 0; k < kNearStrides; ++k) {
            const uint32_t c1 = 2 * pos + 1, g1 = 2 * c1 + 1;
            if (!(going & (c1 < n) & (g1 + 3 < (uint32_t)TOP))) break;  // levels below 5, or a sift that started deeper: the general loop
            going = stride(load_pair(A.top + c1), load_pair(A.top + g1), load_pair(A.top + g1 + 2), c1, g1, set_near);
            if (k == 0) occupant(placed_node);
        }
        if (2 * start + 1 >= n) occupant(placed_node);  // no children: elt stays where the sift started
    } else static_assert(std::is_same<typename std::decay<Hook>::type, NoOccupantHook>::value, "the occupant hook needs the first stride in the near array");
    while (going && 2 * pos + 1 < n) {
        const uint32_t c1 = 2 * pos + 1, g1 = 2 * c1 + 1;
#if !defined(__HIP_DEVICE_COMPILE__)
        // host build (the host tail's deep heaps: 2 M entries, 16 MB): whichever grandchild the hole moves to, the next stride looks at slots 2 g + 1 ... 2 g + 2 and
        // 4 g + 3 ... 4 g + 6 for g in [g1, g1 + 3] — two short contiguous runs; asked for now, they arrive while this stride is decided (a sift is otherwise a
        // chain of ten cache misses, each waiting for the one before)
        if (g1 >= (uint32_t)TOP && g_host_prefetch.sift_lookahead >= 1) {
#if MAPAD_SUBTREE_HEAP == 1
            // subtree blocks: a pop_max sift (hole on an odd level) next looks at the blocks of g1 .. g1 + 3 — four adjacent 64-byte blocks; a pop_min sift (even
            // level) at the children of g1 .. g1 + 3, which share the two blocks this stride reads anyway, and at their grandchildren: eight adjacent blocks
            const HeapEntry* nb = A.heap + HeapLayout<TOP>::slot(MAX ? 2 * g1 + 1 : 4 * g1 + 3);
            __builtin_prefetch(nb); __builtin_prefetch(nb + 8); __builtin_prefetch(nb + 16); __builtin_prefetch(nb + 24);
            if (!MAX) { __builtin_prefetch(nb + 32); __builtin_prefetch(nb + 40); __builtin_prefetch(nb + 48); __builtin_prefetch(nb + 56); }
#else
            const HeapEntry* nc = A.heap + (2 * g1 + 1);  // 8 entries
            const HeapEntry* ng = A.heap + (4 * g1 + 3);  // 16 entries
            __builtin_prefetch(nc); __builtin_prefetch(nc + 7);
            __builtin_prefetch(ng); __builtin_prefetch(ng + 8); __builtin_prefetch(ng + 15);
#endif
        }
#endif
        HeapPair c, ga, gb;  // a level is entirely near or entirely in the arena
        // A pop_max sift starts at slot 1 or 2 and every stride takes it two levels down: with an even number of near levels (TOP = 63: levels 0-5) the strides above
        // have used up the near levels exactly, and every stride of this loop reads arena levels only — no near reads, no selects (round 6: 3 ds_read_b128, their
        // clamped addresses and 12 v_cndmask per arena stride were spent on values that were never taken).
        if constexpr (kArenaOnly) {
            c = load_pair(A.heap + c1); ga = load_pair(A.heap + g1); gb = load_pair(A.heap + g1 + 2);
        } else
        {   // near reads for every slot (clamped), arena loads predicated and back to back: one wait per level for the whole wavefront
            const bool c_near = c1 < (uint32_t)TOP, g_near = g1 + 3 < (uint32_t)TOP;
            const uint32_t kc = c_near ? c1 : 1u, kg = g_near ? g1 : 3u;  // clamped indices keep the 16-byte alignment of a pair (odd logical index)
            const HeapPair nc = load_pair(A.top + kc), nga = load_pair(A.top + kg), ngb = load_pair(A.top + kg + 2);
            HeapPair hc = HeapPair{}, hga = HeapPair{}, hgb = HeapPair{};
#if MAPAD_SUBTREE_HEAP == 2
            const uint32_t sa = HeapLayout<TOP>::slot(g1), sc = MAX ? c1 : HeapLayout<TOP>::slot(c1), sb = sa + 2u;
            if (!c_near) hc = load_pair(A.heap + sc);
            if (!g_near) { hga = load_pair(A.heap + sa); hgb = load_pair(A.heap + sb); }
#elif MAPAD_SUBTREE_HEAP
            // pop_max sift (hole on an odd level): the six candidates are the hole's own block.  pop_min sift (even level): the children sit in their grandparent's
            // block, the grandchildren in slots 0-1 of the two children's blocks, which are adjacent; a block whose first entry is beyond the heap is not loaded
            // (its arena may end before it — the implicit array reads such slots and ignores them).
            const uint32_t sa = HeapLayout<TOP>::slot(g1), sc = MAX ? sa - 2u : HeapLayout<TOP>::slot(c1), sb = MAX ? sa + 2u : sa + 8u;
            if (!c_near) hc = load_pair(A.heap + sc);
            if (!g_near) { if (MAX || g1 < n) hga = load_pair(A.heap + sa); if (MAX || g1 + 2 < n) hgb = load_pair(A.heap + sb); }
#else
            if (!c_near) hc = load_pair(A.heap + c1);
            if (!g_near) { hga = load_pair(A.heap + g1); hgb = load_pair(A.heap + g1 + 2); }
#endif
            c = c_near ? nc : hc; ga = g_near ? nga : hga; gb = g_near ? ngb : hgb;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SPEC) {
            // grandchild w of the hole: its children (one pair) and grandchildren (two pairs), if they exist and live in the arena; nothing this stride stores
            // (the hole's slot, one of its children) is among them
            const bool spec = (g1 >= (uint32_t)TOP) & (g1 < n);  // quad-uniform
            const uint32_t G = g1 + (uint32_t)w, sc1 = 2 * G + 1, sg1 = 2 * sc1 + 1;
            HeapPair sc = HeapPair{}, sga = HeapPair{}, sgb = HeapPair{};
            const uint32_t ss = HeapLayout<TOP>::slot(sc1);  // subtree blocks: grandchild w's own block (the four lanes' blocks are adjacent: two lines)
#if MAPAD_SUBTREE_HEAP == 1
            static_assert(MAX, "the speculative stride is a pop_max sift's");
            if (spec & (sc1 < n)) sc = load_pair(A.heap + ss);
            if (spec & (sg1 < n)) { sga = load_pair(A.heap + ss + 2); sgb = load_pair(A.heap + ss + 4); }
#else
            if (spec & (sc1 < n)) sc = load_pair(A.heap + ss);
            if (spec & (sg1 < n)) { sga = load_pair(A.heap + sg1); sgb = load_pair(A.heap + sg1 + 2); }
#endif
            if (kArenaOnly && MAPAD_UNIFORM_SIFT_STORES && !first_general) going = stride(c, ga, gb, c1, g1, set_arena);
            else going = stride(c, ga, gb, c1, g1, set_any);
            first_general = false;
            if (spec & going & (2 * pos + 1 < n)) {  // the hole went to grandchild pos = g1 + k: lane k holds the next stride's candidates
                const int src = (int)((threadIdx.x & ~3u) + (pos - g1)) << 2;
                auto take = [&](const HeapPair& p) {
                    HeapPair r;
                    r.a.score = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(p.a.score))); r.a.node = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)p.a.node);
                    r.b.score = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(p.b.score))); r.b.node = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)p.b.node);
                    return r;
                };
                const HeapPair c2 = take(sc), ga2 = take(sga), gb2 = take(sgb);
                const uint32_t c1b = 2 * pos + 1;
                if (kArenaOnly && MAPAD_UNIFORM_SIFT_STORES) going = stride(c2, ga2, gb2, c1b, 2 * c1b + 1, set_arena);
                else going = stride(c2, ga2, gb2, c1b, 2 * c1b + 1, set_any);
            }
            continue;
        }
#endif
        if (kArenaOnly && MAPAD_UNIFORM_SIFT_STORES && !first_general) going = stride(c, ga, gb, c1, g1, set_arena);
        else going = stride(c, ga, gb, c1, g1, set_any);
        first_general = false;
    }
    hp_set(A, pos, elt);
}

// pop_max of the crate in two steps so that the caller can start loading the popped frame's node before the sift's stores:
// mm_find_max() says which slot holds the maximum (slot 2 wins a tie between slots 1 and 2 unless MAPAD_HEAP_VARIANT bit 1 is set), mm_remove_at() removes it.
template <bool NL, int TOP>
MAPAD_HD HeapEntry mm_find_max(const ArenaT<NL, TOP>& A, uint32_t n, uint32_t& idx) {
    const HeapPair p = load_pair(A.top + 1);  // logical slots 1 and 2 (stale if n < 3, handled below)
    const HeapEntry first = A.top[0];
    // selects, not branches: n >= 3 -> the larger of slots 1 and 2 (slot 2 on a tie); n == 2 -> slot 1; n == 1 -> slot 0
    const bool a_wins = (MAPAD_HEAP_VARIANT & 2) ? (p.a.score >= p.b.score) : (p.a.score > p.b.score);  // who takes a tie: MAPAD_HEAP_VARIANT bit 1
    const bool use_a = (n == 2) | ((n >= 3) & a_wins);
    const bool use_b = (n >= 3) & !a_wins;
    idx = use_a ? 1u : use_b ? 2u : 0u;
    HeapEntry r = first;
    r.score = use_a ? p.a.score : use_b ? p.b.score : r.score;
    r.node = use_a ? p.a.node : use_b ? p.b.node : r.node;
    return r;
}
template <bool MAX, bool NL, int TOP>
MAPAD_HD void mm_remove_at(const ArenaT<NL, TOP>& A, uint32_t& n, uint32_t idx) {
    const HeapEntry last = hp_get(A, n - 1);
    n -= 1;
    if (idx < n) mm_trickle_down<MAX>(A, n, idx, last);
}
template <bool NL, int TOP>
MAPAD_HD HeapEntry mm_pop_min(const ArenaT<NL, TOP>& A, uint32_t& n) {
    const HeapEntry item = A.top[0];
    mm_remove_at<false>(A, n, 0);
    return item;
}

}  // namespace mapad
