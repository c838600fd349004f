// fmd_device.hpp — rank queries and FMD extension on the 64-byte-block device layout.
//
// Replaces, on the device: FmdExtIterator (src/map/fmd_index.rs:109-182) + Occ::get_small_k / Less / BWT of the
// rust-bio fork (byte BWT + u64 checkpoints every 128 rows).  occ(r, c) is a mathematically unique number, so the
// layout is free: here one 64-byte block answers occ(r, A|C|G|T) for 96 rows (until late round 5: 128 bytes per 256 rows; either is one
// request to the L2 and one 128-byte line behind it — the smaller block halves the bytes between L1 and L2 and the registers a query holds: DESIGN.md section 4).
//
//   block b (rows 96b .. 96b+95), 8 x u64:
//     sub-block w = words [2w, 2w+1], rows 96b+24w .. +23:
//       word 0 = count of base w in rows [0, 96b) (bits 0..39) | plane0 << 40 (24 bits)
//       word 1 = plane1 (bits 0..23) | plane2 << 24 (bits 24..47); bits 48..63 zero
//   symbol codes: $=0 X=1 A=4 C=5 G=6 T=7  -> "is base k" = plane2 & (plane1 == k>>1) & (plane0 == k&1)
//
// One quad (4 adjacent lanes) serves one rank query: lane w loads sub-block w (16 contiguous bytes, so the quad
// reads the whole 64-byte block coalesced), popcounts its 24 rows for all four bases into the four bytes of one word, and a
// 2-step DPP butterfly inside the quad sums the partial counts.  Lane w ends up with occ(r, base w).
#pragma once
#include "common.hpp"

namespace mapad {

MAPAD_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

MAPAD_HD int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

constexpr uint64_t kCountMask = (1ull << 40) - 1;

// block and row inside the block of BWT row r (r < 2^51): r / 96 without a 64-bit division — with x = r / 32 = 65536 xh + xl and 65536 = 3 * 21845 + 1,
// x / 3 = 21845 xh + (xh + xl) / 3 exactly, and x % 3 = (xh + xl) % 3.
struct BlockPos {
    uint64_t b;
    int r_in;  // 0..95
};
MAPAD_HD BlockPos block_pos(uint64_t r) {
    const uint64_t x = r >> 5;
    const uint32_t xh = (uint32_t)(x >> 16), y = xh + (uint32_t)(x & 0xFFFFu);
    const uint32_t q = y / 3u;
    BlockPos p;
    p.b = (uint64_t)xh * 21845u + q;
    p.r_in = (int)((uint32_t)r & 31u) + 32 * (int)(y - 3u * q);
    return p;
}
MAPAD_HD const uint64_t* block_ptr(const DevIndex& ix, uint64_t r) { return ix.blocks + block_pos(r).b * kBlockWords; }

// rows of sub-block w that are <= r_in (r_in = row index inside the block), as a bit mask over its 24 rows
MAPAD_HD uint32_t sub_mask(int w, int r_in) {
    int t = r_in + 1 - kSubRows * w;  // rows of this sub-block that count: <= 0 none, >= 24 all
    t = t < 0 ? 0 : (t > kSubRows ? kSubRows : t);
    return (1u << t) - 1u;
}
// the planes of a sub-block, out of its two words (plane 2 carries bits 24.. of word 1 above it: always used under a sub_mask)
MAPAD_HD uint32_t sub_p0(uint64_t w0) { return (uint32_t)(w0 >> 40); }
MAPAD_HD uint32_t sub_p1(uint64_t w1) { return (uint32_t)w1; }
MAPAD_HD uint32_t sub_p2(uint64_t w1) { return (uint32_t)(w1 >> 24); }
MAPAD_HD uint64_t pack_sub0(uint64_t count, uint32_t p0) { return (count & kCountMask) | ((uint64_t)p0 << 40); }
MAPAD_HD uint64_t pack_sub1(uint32_t p1, uint32_t p2) { return (uint64_t)p1 | ((uint64_t)p2 << 24); }
// rows of the sub-block holding base k (0..3), under mask m
MAPAD_HD uint32_t sub_is_base(uint64_t w0, uint64_t w1, int k, uint32_t m) {
    const uint32_t inv1 = (k & 2) ? 0u : ~0u, inv0 = (k & 1) ? 0u : ~0u;
    return sub_p2(w1) & (sub_p1(w1) ^ inv1) & (sub_p0(w0) ^ inv0) & m;
}
MAPAD_HD uint32_t sub_is_x(uint64_t w0, uint64_t w1, uint32_t m) { return sub_p0(w0) & ~sub_p1(w1) & ~sub_p2(w1) & m; }  // 'X' (code 1): plane 0 only

// ---- scalar reference of the same layout (host emulation + device single-lane paths such as SA walks) -------------
// occurrences of base k (0..3 = ACGT) in bwt[0..=r]
MAPAD_HD uint64_t occ_scalar(const DevIndex& ix, uint64_t r, int k) {
    const BlockPos bp = block_pos(r);
    const uint64_t* blk = ix.blocks + bp.b * kBlockWords;
    uint64_t c = blk[2 * k] & kCountMask;
    for (int w = 0; w * kSubRows <= bp.r_in; ++w) c += (uint64_t)popc32(sub_is_base(blk[2 * w], blk[2 * w + 1], k, sub_mask(w, bp.r_in)));
    return c;
}
// occurrences of 'X' in bwt[0..=r]; x_counts[b] = 'X' rows in [0, 96b) (null: the text has none)
MAPAD_HD uint64_t occ_x_scalar(const DevIndex& ix, const uint64_t* x_counts, uint64_t r) {
    const BlockPos bp = block_pos(r);
    const uint64_t* blk = ix.blocks + bp.b * kBlockWords;
    uint64_t c = x_counts ? x_counts[bp.b] : 0;
    for (int w = 0; w * kSubRows <= bp.r_in; ++w) c += (uint64_t)popc32(sub_is_x(blk[2 * w], blk[2 * w + 1], sub_mask(w, bp.r_in)));
    return c;
}
// device symbol code of the row `bit` (0..23) of a sub-block
MAPAD_HD int sub_code(uint64_t w0, uint64_t w1, int bit) {
    return (int)((sub_p0(w0) >> bit) & 1u) | (int)(((sub_p1(w1) >> bit) & 1u) << 1) | (int)(((sub_p2(w1) >> bit) & 1u) << 2);
}
// device symbol code of bwt[r] (0 '$', 1 'X', 4..7 ACGT)
MAPAD_HD int bwt_code(const DevIndex& ix, uint64_t r) {
    const BlockPos bp = block_pos(r);
    const int w = bp.r_in / kSubRows;
    const uint64_t* sb = ix.blocks + bp.b * kBlockWords + 2 * w;
    return sub_code(sb[0], sb[1], bp.r_in - kSubRows * w);
}
// number of '$' rows in [0, pos]  -> sentinel_occ() of fmd_index.rs:140-146 is "count of sentinel rows <= pos"
MAPAD_HD uint64_t sentinel_le(const DevIndex& ix, uint64_t pos) {
    return (uint64_t)(pos >= ix.sentinel[0]) + (uint64_t)(pos >= ix.sentinel[1]);
}

// Result of extending a bi-interval by all four bases (iterator order of the reference is T,G,C,A; here indexed by base 0..3).
struct Ext4 {
    uint64_t lower[4], size[4], lower_rev[4];
};

// FmdExtIterator::new + 4 x extend_once_internal (fmd_index.rs:137-181) for input (lower, lower_rev, size), size >= 1.
MAPAD_HD void finish_ext4(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, const uint64_t occ_lo[4],
                          const uint64_t occ_hi[4], Ext4& out) {
    const uint64_t hi = lower + size - 1;
    const uint64_t o_s = lower == 0 ? 0 : sentinel_le(ix, lower - 1);
    uint64_t s = sentinel_le(ix, hi) - o_s;  // '$' rows inside the interval
    uint64_t l = lower_rev;
    for (int k = 3; k >= 0; --k) {  // T, G, C, A
        l += s;
        s = occ_hi[k] - occ_lo[k];
        out.lower[k] = ix.less[k + 1] + occ_lo[k];
        out.lower_rev[k] = l;
        out.size[k] = s;
    }
}

// the four base counts of a sub-block's rows under mask m, one per byte (A in bits 0..7, ... T in bits 24..31; each <= 24, a block's sum <= 96)
MAPAD_HD uint32_t sub_counts4(uint64_t w0, uint64_t w1, uint32_t m) {
    const uint32_t p0 = sub_p0(w0), p1 = sub_p1(w1), p2 = sub_p2(w1) & m;
    const uint32_t hi1 = p2 & p1, lo1 = p2 & ~p1;  // {G,T} / {A,C}
    return (uint32_t)popc32(lo1 & ~p0) | ((uint32_t)popc32(lo1 & p0) << 8) | ((uint32_t)popc32(hi1 & ~p0) << 16) | ((uint32_t)popc32(hi1 & p0) << 24);
}

// occ(r, A|C|G|T) by ONE lane: the whole 64-byte block, all four sub-blocks, branch-free.
MAPAD_HD void occ4_lane(const DevIndex& ix, uint64_t r, uint64_t out[4]) {
    const BlockPos bp = block_pos(r);
    const uint64_t* blk = ix.blocks + bp.b * kBlockWords;
    MAPAD_TOUCH(blk, kBlockBytes, false);
    uint32_t c4 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) c4 += sub_counts4(blk[2 * w], blk[2 * w + 1], sub_mask(w, bp.r_in));
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (blk[2 * k] & kCountMask) + ((c4 >> (8 * k)) & 0xFFu);
}

MAPAD_HD void ext4_scalar(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, Ext4& out) {
    uint64_t lo[4] = {0, 0, 0, 0}, hi[4];
    if (lower != 0) occ4_lane(ix, lower - 1, lo);
    occ4_lane(ix, lower + size - 1, hi);
    finish_ext4(ix, lower, lower_rev, size, lo, hi, out);
}

#if defined(__HIPCC__)
// ---- quad-cooperative versions (gfx950) ----------------------------------------------------------------------------
// DPP quad_perm controls: [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, broadcast lane k = k*0x55.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int K>
__device__ __forceinline__ uint64_t quad_bcast64(uint64_t v) {
    const uint32_t lo = dpp_quad<K * 0x55>((uint32_t)v), hi = dpp_quad<K * 0x55>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t quad_sum32(uint32_t v) {
    v += dpp_quad<0xB1>(v);
    v += dpp_quad<0x4E>(v);
    return v;
}

// lane w (= threadIdx & 3) returns occ(r, base w) for the quad-uniform row r.  Split into the loads and the arithmetic so that a
// caller with two queries has all four loads in flight before the first popcount (the DPP moves are scheduling barriers for the
// compiler, it does not hoist the second query's loads over them by itself).
struct OccLoads {
    ulonglong2 v;  // the lane's sub-block
    int r_in;      // row inside the block
};
__device__ __forceinline__ OccLoads quad_occ_issue(const DevIndex& ix, uint64_t r, int w) {
    const BlockPos bp = block_pos(r);
    OccLoads l;
    l.v = *reinterpret_cast<const ulonglong2*>(ix.blocks + bp.b * kBlockWords + 2 * w);
    l.r_in = bp.r_in;
    return l;
}
__device__ __forceinline__ uint64_t quad_occ_finish(const OccLoads& l, uint64_t, int w) {
    const uint32_t c4 = quad_sum32(sub_counts4(l.v.x, l.v.y, sub_mask(w, l.r_in)));
    return (l.v.x & kCountMask) + ((c4 >> (8 * w)) & 0xFFu);
}
__device__ __forceinline__ uint64_t quad_occ(const DevIndex& ix, uint64_t r, int w) { return quad_occ_finish(quad_occ_issue(ix, r, w), r, w); }

// All four lanes of the quad return the complete Ext4 of the (quad-uniform) input interval.
__device__ __forceinline__ void ext4_quad(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, int w, Ext4& out) {
    // both rank queries unconditionally, so that their four loads are in flight together (lower == 0: the result is discarded)
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    const OccLoads l_lo = quad_occ_issue(ix, r_lo, w), l_hi = quad_occ_issue(ix, r_hi, w);
    const uint64_t occ_lo = quad_occ_finish(l_lo, r_lo, w);
    const uint64_t my_hi = quad_occ_finish(l_hi, r_hi, w);
    const uint64_t my_lo = lower == 0 ? 0 : occ_lo;
    uint64_t lo[4], hi[4];
    lo[0] = quad_bcast64<0>(my_lo); lo[1] = quad_bcast64<1>(my_lo); lo[2] = quad_bcast64<2>(my_lo); lo[3] = quad_bcast64<3>(my_lo);
    hi[0] = quad_bcast64<0>(my_hi); hi[1] = quad_bcast64<1>(my_hi); hi[2] = quad_bcast64<2>(my_hi); hi[3] = quad_bcast64<3>(my_hi);
    finish_ext4(ix, lower, lower_rev, size, lo, hi, out);
}

// Lane w of the quad returns the extension by base w only (plus the quad-uniform mask of non-empty extensions): the search builds
// the children of a frame lane-parallel, so nobody needs all twelve values (search_core.hpp: search_step).
struct ExtLane {
    uint64_t lower, lower_rev, size;
    uint32_t nonempty;  // bit k: extension by base k has size >= 1
};
__device__ __forceinline__ uint64_t quad_pick64(uint64_t v, int k) {  // value of lane k (k quad-uniform, dynamic)
    const uint64_t a = quad_bcast64<0>(v), b = quad_bcast64<1>(v), c = quad_bcast64<2>(v), d = quad_bcast64<3>(v);
    return k == 0 ? a : k == 1 ? b : k == 2 ? c : d;
}
__device__ __forceinline__ uint32_t quad_pick32(uint32_t v, int k) {  // value of lane k (k quad-uniform, dynamic)
    const uint32_t a = dpp_quad<0 * 0x55>(v), b = dpp_quad<1 * 0x55>(v), c = dpp_quad<2 * 0x55>(v), d = dpp_quad<3 * 0x55>(v);
    return k == 0 ? a : k == 1 ? b : k == 2 ? c : d;
}
// The same in two halves, so that a caller can put other memory traffic (the repair of the heap) between issuing the four loads and using them.
struct ExtLoads { OccLoads lo, hi; };
__device__ __forceinline__ ExtLoads ext4_quad_issue(const DevIndex& ix, uint64_t lower, uint64_t size, int w) {
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    ExtLoads l;
    l.lo = quad_occ_issue(ix, r_lo, w);
    l.hi = quad_occ_issue(ix, r_hi, w);
    return l;
}
// lane_less = ix.less[w + 1], picked once per kernel by the caller: as a select over the kernel arguments inside the step it came out as nested
// branches, each arm re-reading the eight spilled scalar registers of `less` (45 v_readlane per step); `above` is written as masked adds for the
// same reason.
__device__ __forceinline__ void ext4_quad_lane_finish(const DevIndex& ix, const ExtLoads& l, uint64_t lower, uint64_t lower_rev, uint64_t size, int w, uint64_t lane_less,
                                                      ExtLane& out) {
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    const uint64_t occ_lo = quad_occ_finish(l.lo, r_lo, w);
    const uint64_t my_hi = quad_occ_finish(l.hi, r_hi, w);
    const uint64_t my_lo = lower == 0 ? 0 : occ_lo;
    const uint64_t my_size = my_hi - my_lo;
    const uint64_t s0 = quad_bcast64<0>(my_size), s1 = quad_bcast64<1>(my_size), s2 = quad_bcast64<2>(my_size), s3 = quad_bcast64<3>(my_size);
    const uint64_t o_s = lower == 0 ? 0 : sentinel_le(ix, lower - 1);
    const uint64_t sent = sentinel_le(ix, lower + size - 1) - o_s;  // '$' rows inside the interval
    const uint64_t above = (w < 3 ? s3 : 0ull) + (w < 2 ? s2 : 0ull) + (w < 1 ? s1 : 0ull);  // fmd_index.rs:137-181 iterates T, G, C, A
    out.lower = lane_less + my_lo;
    out.lower_rev = lower_rev + sent + above;
    out.size = my_size;
    out.nonempty = (s0 >= 1 ? 1u : 0u) | (s1 >= 1 ? 2u : 0u) | (s2 >= 1 ? 4u : 0u) | (s3 >= 1 ? 8u : 0u);
}
__device__ __forceinline__ void ext4_quad_lane(const DevIndex& ix, uint64_t lower, uint64_t lower_rev, uint64_t size, int w, ExtLane& out) {
    // both rank queries unconditionally, so that their four loads are in flight together (lower == 0: the result is discarded)
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    const OccLoads l_lo = quad_occ_issue(ix, r_lo, w), l_hi = quad_occ_issue(ix, r_hi, w);
    const uint64_t occ_lo = quad_occ_finish(l_lo, r_lo, w);
    const uint64_t my_hi = quad_occ_finish(l_hi, r_hi, w);
    const uint64_t my_lo = lower == 0 ? 0 : occ_lo;
    const uint64_t my_size = my_hi - my_lo;
    const uint64_t s0 = quad_bcast64<0>(my_size), s1 = quad_bcast64<1>(my_size), s2 = quad_bcast64<2>(my_size), s3 = quad_bcast64<3>(my_size);
    const uint64_t o_s = lower == 0 ? 0 : sentinel_le(ix, lower - 1);
    const uint64_t sent = sentinel_le(ix, lower + size - 1) - o_s;  // '$' rows inside the interval
    // fmd_index.rs:137-181 iterates T, G, C, A: lower_rev of base k = lower_rev + '$' rows + sizes of the bases above k
    const uint64_t above = w == 3 ? 0 : w == 2 ? s3 : w == 1 ? s3 + s2 : s3 + s2 + s1;
    const uint64_t less = w == 0 ? ix.less[1] : w == 1 ? ix.less[2] : w == 2 ? ix.less[3] : ix.less[4];
    out.lower = less + my_lo;
    out.lower_rev = lower_rev + sent + above;
    out.size = my_size;
    out.nonempty = (s0 >= 1 ? 1u : 0u) | (s1 >= 1 ? 2u : 0u) | (s2 >= 1 ? 4u : 0u) | (s3 >= 1 ? 8u : 0u);
}

// ---- pair-cooperative versions: two adjacent lanes per read (lanes-per-read 2) ----------------------------------------------------------------
// Lane h (= lane & 1) loads sub-blocks 2h and 2h + 1 of a block — 32 contiguous bytes, the pair reads the 64-byte block — and ends up with
// occ(r, base 2h) and occ(r, base 2h + 1): the count words of its two sub-blocks are the running counts of exactly these two bases.  Twice the
// loads and popcounts of a quad lane per instruction stream, but a wavefront then serves 32 reads instead of 16.
struct OccLoads2 {
    ulonglong2 a, b;
    int r_in;
};
__device__ __forceinline__ OccLoads2 pair_occ_issue(const DevIndex& ix, uint64_t r, int h) {
    const BlockPos bp = block_pos(r);
    const uint64_t* sb = ix.blocks + bp.b * kBlockWords + 4 * h;
    OccLoads2 l;
    l.a = *reinterpret_cast<const ulonglong2*>(sb);
    l.b = *reinterpret_cast<const ulonglong2*>(sb + 2);
    l.r_in = bp.r_in;
    return l;
}
__device__ __forceinline__ void pair_occ_finish(const OccLoads2& l, uint64_t, int h, uint64_t& o0, uint64_t& o1) {
    uint32_t c4 = sub_counts4(l.a.x, l.a.y, sub_mask(2 * h, l.r_in)) + sub_counts4(l.b.x, l.b.y, sub_mask(2 * h + 1, l.r_in));
    c4 += dpp_quad<0xB1>(c4);  // the other lane of the pair
    const uint32_t pr = h ? (c4 >> 16) : c4;
    o0 = (l.a.x & kCountMask) + (pr & 0xFFu);
    o1 = (l.b.x & kCountMask) + ((pr >> 8) & 0xFFu);
}
__device__ __forceinline__ uint64_t pair_other64(uint64_t v) {  // the value the other lane of the pair holds
    const uint32_t lo = dpp_quad<0xB1>((uint32_t)v), hi = dpp_quad<0xB1>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t pair_pick64(uint64_t v, int k) {  // value of pair lane k (k pair-uniform, dynamic)
    const uint32_t lo0 = dpp_quad<0xA0>((uint32_t)v), hi0 = dpp_quad<0xA0>((uint32_t)(v >> 32));  // quad_perm [0,0,2,2]
    const uint32_t lo1 = dpp_quad<0xF5>((uint32_t)v), hi1 = dpp_quad<0xF5>((uint32_t)(v >> 32));  // quad_perm [1,1,3,3]
    return k == 0 ? (((uint64_t)hi0 << 32) | lo0) : (((uint64_t)hi1 << 32) | lo1);
}
struct ExtLoads2 { OccLoads2 lo, hi; };
__device__ __forceinline__ ExtLoads2 ext4_pair_issue(const DevIndex& ix, uint64_t lower, uint64_t size, int h) {
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    ExtLoads2 l;
    l.lo = pair_occ_issue(ix, r_lo, h);
    l.hi = pair_occ_issue(ix, r_hi, h);
    return l;
}
// Lane h of the pair returns the extensions by bases 2h and 2h + 1 (plus the pair-uniform mask of non-empty extensions).
struct ExtLane2 {
    uint64_t lower[2], lower_rev[2], size[2];
    uint32_t nonempty;
};
__device__ __forceinline__ void ext4_pair_lane_finish(const DevIndex& ix, const ExtLoads2& l, uint64_t lower, uint64_t lower_rev, uint64_t size, int h, uint64_t less0, uint64_t less1,
                                                      ExtLane2& out) {
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    uint64_t lo0, lo1, hi0, hi1;
    pair_occ_finish(l.lo, r_lo, h, lo0, lo1);
    pair_occ_finish(l.hi, r_hi, h, hi0, hi1);
    if (lower == 0) { lo0 = 0; lo1 = 0; }
    const uint64_t m0 = hi0 - lo0, m1 = hi1 - lo1;             // sizes of this lane's two bases
    const uint64_t o0 = pair_other64(m0), o1 = pair_other64(m1);  // ... and of the other lane's
    const uint64_t s0 = h ? o0 : m0, s1 = h ? o1 : m1, s2 = h ? m0 : o0, s3 = h ? m1 : o1;
    const uint64_t o_s = lower == 0 ? 0 : sentinel_le(ix, lower - 1);
    const uint64_t sent = sentinel_le(ix, lower + size - 1) - o_s;  // '$' rows inside the interval
    // fmd_index.rs:137-181 iterates T, G, C, A: lower_rev of base k = lower_rev + '$' rows + sizes of the bases above k
    const uint64_t above_hi = h ? 0ull : s3 + s2;          // above base 2h + 1: (h = 0: bases 2, 3; h = 1: none)
    const uint64_t above1 = above_hi;                       // base 2h + 1
    const uint64_t above0 = above_hi + m1;                  // base 2h: additionally this lane's upper base
    out.lower[0] = less0 + lo0; out.lower[1] = less1 + lo1;
    out.lower_rev[0] = lower_rev + sent + above0; out.lower_rev[1] = lower_rev + sent + above1;
    out.size[0] = m0; out.size[1] = m1;
    out.nonempty = (s0 >= 1 ? 1u : 0u) | (s1 >= 1 ? 2u : 0u) | (s2 >= 1 ? 4u : 0u) | (s3 >= 1 ? 8u : 0u);
}

// Single-base step for the D-array chains: new (lower, size) of the quad-uniform interval extended by base k (0..3, quad-uniform).
// Lane w counts base k in its own sub-block only (one popcount instead of four) and every lane reads the block's count word of base k.
__device__ __forceinline__ void ext1_quad(const DevIndex& ix, uint64_t lower, uint64_t size, int k, int w, uint64_t& new_lower, uint64_t& new_size) {
    const uint64_t r_lo = lower == 0 ? 0 : lower - 1, r_hi = lower + size - 1;
    const BlockPos p_lo = block_pos(r_lo), p_hi = block_pos(r_hi);
    const uint64_t* b_lo = ix.blocks + p_lo.b * kBlockWords;
    const uint64_t* b_hi = ix.blocks + p_hi.b * kBlockWords;
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(b_lo + 2 * w), h = *reinterpret_cast<const ulonglong2*>(b_hi + 2 * w);
    const uint64_t c_lo = b_lo[2 * k] & kCountMask, c_hi = b_hi[2 * k] & kCountMask;
    const uint32_t n_lo = (uint32_t)popc32(sub_is_base(a.x, a.y, k, sub_mask(w, p_lo.r_in)));
    const uint32_t n_hi = (uint32_t)popc32(sub_is_base(h.x, h.y, k, sub_mask(w, p_hi.r_in)));
    const uint32_t both = quad_sum32(n_lo | (n_hi << 16));  // each count <= 96
    const uint64_t lo = lower == 0 ? 0 : c_lo + (both & 0xFFFFu);
    const uint64_t hi = c_hi + (both >> 16);
    new_lower = (k == 0 ? ix.less[1] : k == 1 ? ix.less[2] : k == 2 ? ix.less[3] : ix.less[4]) + lo;
    new_size = hi - lo;
}
#endif

MAPAD_HD void ext1_scalar(const DevIndex& ix, uint64_t lower, uint64_t size, int k, uint64_t& new_lower, uint64_t& new_size) {
    const uint64_t lo = lower == 0 ? 0 : occ_scalar(ix, lower - 1, k);
    const uint64_t hi = occ_scalar(ix, lower + size - 1, k);
    new_lower = ix.less[k + 1] + lo;
    new_size = hi - lo;
}

}  // namespace mapad
