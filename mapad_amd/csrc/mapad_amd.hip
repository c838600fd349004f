// mapad_amd.hip — gfx950 kernels, device context and the C ABI (include/mapad_amd.h) of the mapAD read-mapping hot path.
//
// Two kernels per batch of reads (the data-parallel boundary of run_inner, src/map/mapping.rs:153-156):
//   darray_kernel : BiDArray::new for every read — one wavefront per read, one quad per offset chain (darray_core.hpp)
//   search_kernel : k_mismatch_search — persistent wavefronts, one quad per read, reads pulled from an atomic work
//                   counter (search_core.hpp); reads whose state outgrows the small per-quad arena are re-run by the
//                   same kernel from a second, large-arena pool that holds the reference's full limits
//                   (STACK_LIMIT / EDIT_TREE_LIMIT, mapping.rs:52-54).
// Nothing here falls back to the CPU: without a gfx950 device every mapping entry point returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mapad_amd.h"
#include "darray_core.hpp"
#include "host_index.hpp"
#include "host_index_io.hpp"
#include "host_models.hpp"
#include "host_postproc.hpp"
#include "postproc_core.hpp"
#include "text_core.hpp"
#include "search_core.hpp"
#include "host_tail.hpp"

using namespace mapad;

namespace mapad { namespace gpuidx { void suffix_products(const uint8_t* t_host, host::Index& ix, int device, bool verbose); } }  // index_gpu.hip

// ======================================================================================================================
// device side
// ======================================================================================================================
namespace {

enum : uint32_t { ST_POOL_OVERFLOW = 4, ST_NO_TABLE = 8, ST_TAIL = 16 /* handed to the host tail (host_tail.hpp); replaced by the read's final status when its result comes back */ };
// Launches over a batch (all on the batch's stream):
//   Q0  search_kernel, every read: one quad per read, a per-slot base arena that GROWS on demand (size classes and idle sets of base arenas, below).  A quad that
//       has run out of reads takes over reads of this launch that gave up waiting for an arena (the restart list).  With the host tail on (the default) a read
//       leaves for a host thread when it passes the pop budget (or an eighth of it while a host worker is idle), queues for a scarce arena class while the host has
//       room, or fits no growable arena — each rule only while the host's backlog is short (search_kernel: give_to_host).
//       (MAPAD_HEAVY=1 only: a read that has outgrown its base arena is suspended (HeavyItem) and H0 — heavy_kernel, one WAVEFRONT per read — continues it.)
//   Q1  search_kernel again over what is left of the restart list (normally nothing: the launch exits at once); these reads wait for arenas as long as it takes.
//   F   heavy_kernel from scratch, with arenas that hold the reference's full limits (STACK_LIMIT / EDIT_TREE_LIMIT, mapping.rs:52-54), for the reads that no
//       growable arena could hold — only when the host tail is off or its ring full; otherwise those reads went to the host from Q0 / Q1.
// Pop budget of a read on the GPU before a host thread takes it over (host_tail.hpp); MAPAD_TAIL_POPS / mapad_ctx_set_tail_pops override, 0 = the host tail is off.
// 2^20 pops are ~6 s of one quad's time: no 50 bp read of C1-C4 gets near it (heaviest of 10 M C4 reads: 76 272 pops), the heavy tail of the 35-100 bp mix does.
// Round 5, 1 M reads of the C5 mix on 3 Gbp at the real limits, one GPU + 16 CPUs (profiles/r05/c5_3gbp_1m_budgets.txt): 2^19 -> 51-57 s (the 16 CPUs busy
// throughout), 2^20 -> 41.5 s (GPU 32 s, host 29 s of work), 2^21 -> 55 s (the GPU's own tail: 12 s per such read).  Round 4 needed 2^17 on a large index (125 s)
// because reads that queued for an arena of a scarce class made no pops and never reached a larger budget; they now ask for the host when they queue
// (DeviceGrow::acquire), so one budget serves every index size.
#if !defined(MAPAD_DEFAULT_TAIL_POPS)
#define MAPAD_DEFAULT_TAIL_POPS (1u << 20)
#endif
// ... and the pop count from which a read leaves while a host worker is IDLE (MAPAD_TAIL_POPS_IDLE; 0 = only the budget counts).  Round 6, C5 mix on 3 Gbp, 14 workers,
// same box, default / 2^17 (profiles/r06/c5_idle_tier.txt): 200 K reads, four interleaved repetitions: search launch 11.2-12.5 s -> 8.5-8.7 s, last host read done
// 12.6-14.8 s -> 8.5-10.5 s after the first hand-over, call 14.1-20.8 s -> 14.5-16.5 s (the workers' share of the pops 5.7 % -> 10 %; reads queuing for a dry arena class
// 290-330 -> 55-60); 1 M reads: launch 36.7 -> 32.1 s, call 40.9 -> 39.7 s (there the workers are busy 61 % of the time without this tier, 93 % with it); 2^16 and 2^18
// are no better.  Results identical (sha256).
#if !defined(MAPAD_DEFAULT_TAIL_POPS_IDLE)
#define MAPAD_DEFAULT_TAIL_POPS_IDLE (1u << 17)
#endif
constexpr int kTiers = 2;   // arena pools: growable base arenas, full-limit arenas
constexpr int kStages = 3;  // hand-over lists: Q0 -> Q1 -> F
constexpr int kClasses = 10;  // grown arenas: 2x steps above the base arena (16 Ki nodes -> 32 Ki ... ), the last one with the full limits
constexpr int kKeyBins = kMaxReadLen + 2;
// cursors (u32 words): global bump allocators and work counters; per stage t: CUR_WORK + 2t = next work item, CUR_OVF + 2t = reads stage t
// handed on.  The two pool cursors are 64-bit (words 0-1 and 2-3): a batch may ask for more than 2^32 op words, which must show up as a pool
// overflow, not wrap around.
enum { CUR_HITS = 0, CUR_OPS = 2, CUR_POOL_OVF = 4, CUR_ERR = 5, CUR_WORK = 6, CUR_OVF = 7, CUR_GROWN = 6 + 2 * kStages, CUR_HEAVY_N = CUR_GROWN + 4 /* per quad stage */, CUR_HEAVY_WORK = CUR_GROWN + 6 /* per heavy stage */,
       CUR_HEAVY_POPS = CUR_GROWN + 8 /* 64-bit */, CUR_HPROF = CUR_GROWN + 10 /* 8 x 64-bit, -DMAPAD_HEAVY_PROF */, CUR_TAIL = CUR_GROWN + 26 /* reads handed to the host tail */, CUR_TAIL_DRY = CUR_GROWN + 27 /* ... of them because an arena class was dry */,
       CUR_TAIL_F = CUR_GROWN + 28 /* ... instead of going to the full-limit stage */, CUR_TAIL_STATE = CUR_GROWN + 29 /* ... handed over with their state */,
       CUR_TAIL_IDLE = CUR_GROWN + 30 /* ... below the pop budget, because host workers were idle */, CUR_COUNT = CUR_GROWN + 31 };
inline uint64_t cur64(const uint32_t* cur, int k) { return (uint64_t)cur[k] | ((uint64_t)cur[k + 1] << 32); }

struct BatchDev {
    const uint8_t* seqs;
    const uint8_t* quals;
    const uint64_t* offsets;
    uint32_t n_reads;
    float* d_arrays;   // BiDArray::d_composite of every read, same offsets as the reads (written by darray_kernel)
    ReadCounters* counters;
    uint32_t* status;
    uint32_t* hit_count;
    uint32_t* hit_first;
    HitRec* hits_pool;
    uint32_t* ops_pool;
    uint32_t hits_cap, ops_cap;
    uint32_t* cursors;
    uint32_t* overflow_list;  // [kStages][n_reads]: read ids (+1) stage t could not finish
    uint32_t* sort_key;       // [n_reads] cost class of a read (zero positions of its D array), written by darray_kernel
    uint32_t* key_hist;       // [n_chunks][kKeyBins] reads per (chunk, cost class), then the running scatter cursors
    uint32_t order_shift;     // log2 of the chunk size: reads are ordered inside chunks of 2^order_shift consecutive reads
    uint32_t* order;          // [n_reads] read ids, most expensive class first (nullptr: in input order)
    unsigned long long* prof; // -DMAPAD_PROFILE_SECTIONS builds: section cycle sums (else nullptr)
    struct HeavyItem* heavy;  // [2][heavy_cap] reads suspended by the two quad stages, continued by heavy_kernel (one wavefront each)
    uint32_t heavy_cap;
    // host tail (host_tail.hpp): a read that has made tail_pops pops is written into a record of this page-locked ring and finished by a host thread
    uint8_t* tail_ring;
    uint32_t tail_stride, tail_cap, tail_lmax;
    uint32_t tail_pops;       // 0xFFFFFFFF: never
    // ... or that needs an arena of class >= tail_min_class while all of them are taken (DeviceGrow::acquire), as long as the host has fewer than tail_backlog_max
    // reads waiting or running (*tail_ctl, host-coherent, kept current by the launch's dispatcher thread); kClasses: never
    const uint32_t* tail_ctl;
    uint32_t tail_backlog_max, tail_min_class;
    // ... and (round 6) a read past the pop budget leaves only while the host has fewer than tail_backlog_budget reads waiting or running; refused, it goes on on the GPU
    // and asks again whenever its excess over the budget has doubled
    uint32_t tail_backlog_budget;
    uint32_t tail_continue;   // 1: a read in a grown arena is handed over WITH its state (host_tail.hpp: TailState), the arena stays the read's until a host thread has copied it
    uint32_t tail_gen;        // what the kernel writes into a record's `ready` word: this launch's number in its ring (never 0) — words left by earlier launches do not match it
    // ... and a read that has made tail_pops_idle pops (<= tail_pops; == tail_pops: this tier is off) leaves while the host has fewer than tail_backlog_idle reads waiting
    // or running, i.e. while a worker is idle: a worker's pop is 15 x a quad's, and a launch lasts as long as its slowest read
    uint32_t tail_pops_idle, tail_backlog_idle;
};

// A read that has outgrown its base arena, as the quad stage leaves it: everything else (heap, nodes, hit staging) is in the grown arena.
struct HeavyItem {
    uint32_t read, grown;  // grown: (class + 1) << kGrownShift | arena index
    SearchState st;
};
static_assert(sizeof(HeavyItem) == 72, "heavy item layout");

struct ArenaPool {
    uint8_t* base;
    uint64_t stride;
    uint64_t off_nodes, off_hits, off_hit_ops, off_scratch, off_near;  // off_near: HBM stand-in for the LDS-resident data (long reads)
    uint32_t heap_cap, node_cap, hit_ops_cap;
    // Base arenas are handed out in sets of one wavefront's read slots; a wavefront claims a set when it starts and gives it back when it
    // exits, so the launches of all batches in flight share one pool no bigger than the chip's resident wavefronts (set_owner: 0 = free).
    uint32_t* set_owner;
    uint32_t n_sets;
};

// Size-class pools of grown arenas (heap + nodes only; hit staging stays in the slot's base arena).  A read that outgrows its
// arena acquires a free arena of the next class (CAS on its owner word), migrates its heap and nodes, and gives the arena back when
// the read is finished.  The descriptor lives in HBM (only the rare grow path reads it).
// Entry kClasses of every array is the SET class (round 5): a free set of base arenas — the 16 read slots' worth of HBM a wavefront claims when it starts and
// gives back when it exits (ArenaPool::set_owner) — taken whole as ONE grown arena (16 x 2.7 MB = 1 M nodes on a 3 Gbp index).  Once a launch's bulk is done its
// wavefronts have exited and thousands of sets lie idle, exactly when the launch's heavy reads queue for the few hundred arenas of the big classes (C5 mix on
// 3 Gbp, round 4: 128 arenas of 1 M nodes for 3 000 reads).  Same owner words, same per-XCD partition, same hand-off fences as the classes proper.
struct GrowPools {
    uint8_t* base[kClasses + 1];
    uint32_t* owner[kClasses + 1];  // [count] 0 = free
    uint64_t stride[kClasses + 1], off_nodes[kClasses + 1];
    uint64_t off_hits[kClasses + 1], off_hit_ops[kClasses + 1], off_scratch[kClasses + 1];  // hit staging of a suspended (heavy) read: it leaves its slot's base arena behind
    uint32_t heap_cap[kClasses + 1], node_cap[kClasses + 1], count[kClasses + 1];
    uint32_t set_next_class;  // the smallest class proper that is bigger than a set: where a read that outgrows a set arena goes
    uint32_t hit_ops_cap;
    uint32_t max_waits;  // fruitless requests (x 64 steps sat out each) after which a read of the first stages gives up and is restarted later
    uint32_t heavy_min_class;  // a read that grows into this class or beyond is handed to heavy_kernel (kClasses: never)
    uint32_t wide_copy_nodes;  // a quad's arena migration is copied by the whole wavefront from this many nodes on (DeviceGrow::wide)
    uint32_t heavy_fast;  // heavy_kernel: wavefront-cooperative steps (0: every step by the general single-lane code)
    uint32_t heavy_max_pending;  // ... unless this many reads of the launch are suspended already (each holds a grown arena)
};
constexpr uint32_t kGrownShift = 27;

template <bool NL, int TOP = kTop>
__device__ __forceinline__ ArenaT<NL, TOP> carve(const ArenaPool& ap, uint32_t slot) {
    uint8_t* b = ap.base + (uint64_t)slot * ap.stride;
    ArenaT<NL, TOP> a;
    a.heap = (MAPAD_GLOBAL HeapEntry*)(b) + 1;  // logical slot 0 = physical slot 1 (aligned child pairs, search_core.hpp)
    a.nodes = (MAPAD_GLOBAL Node*)(b + ap.off_nodes);
    a.hits = (MAPAD_GLOBAL HitRec*)(b + ap.off_hits);
    a.hit_ops = (MAPAD_GLOBAL uint32_t*)(b + ap.off_hit_ops);
    a.scratch = (MAPAD_GLOBAL uint16_t*)(b + ap.off_scratch);
    a.top = nullptr;  // set by the kernel (LDS or the arena's `near` area)
    a.heap_cap = ap.heap_cap; a.node_cap = ap.node_cap; a.hit_ops_cap = ap.hit_ops_cap;
    return a;
}

#define MAPAD_SLIM_EARLY __launch_bounds__(64)  // see MAPAD_SLIM below
// ---- D arrays: one wavefront per read, quad q = offset chain q ------------------------------------------------------------
__global__ void MAPAD_SLIM_EARLY darray_kernel(DevIndex ix, DevParams P, BatchDev B, int lmax, float* long_scratch) {
    extern __shared__ float lds[];
    // penalties and the 15 chains of a read: in LDS, or — batches with reads beyond ~750 bp, whose 16 x lmax floats do not fit — in the block's piece of a global buffer
    float* pen = long_scratch ? long_scratch + (size_t)blockIdx.x * 16 * lmax : lds;  // [lmax]
    float* chains = pen + lmax;                                                            // [15][lmax]
    __shared__ uint32_t n_ext_total;
    const int lane = threadIdx.x & 63, quad = lane >> 2, w = lane & 3;
    for (uint32_t read = blockIdx.x; read < B.n_reads; read += gridDim.x) {
        const uint64_t off = B.offsets[read];
        const int L = (int)(B.offsets[read + 1] - off);
        const uint8_t* seq = B.seqs + off;
        const uint8_t* qual = B.quals + off;
        float* dout = B.d_arrays + off;
        if (L > lmax || P.table_base[L] < 0) {  // fail loudly: the host did not prepare this read length
            if (lane == 0) {
                B.status[read] = ST_NO_TABLE; atomicOr(&B.cursors[CUR_ERR], ST_NO_TABLE); B.counters[read].e_darray = 0;
                if (B.order) { B.sort_key[read] = 0; atomicAdd(&B.key_hist[(size_t)(read >> B.order_shift) * kKeyBins], 1u); }
            }
            continue;
        }
        const int split = P.start_at_end ? L : L / 2;
        uint32_t zeros = 0;  // positions where the D array proves nothing: the search runs unpruned there
        if (lane == 0) n_ext_total = 0;
        for (int r = lane; r < L; r += 64) pen[r] = d_penalty(P, seq, qual, L, r);
        __syncthreads();
        for (int part = 0; part < 2; ++part) {
            const bool left = part == 0;
            const int part_len = left ? split : L - split;
            if (part_len == 0) continue;
            if (quad < kMaxOffset) {
                const uint32_t n_ext = d_chain(ix, seq, L, split, left, quad, pen, chains + quad * lmax, w);
                if (w == 0) atomicAdd(&n_ext_total, n_ext);
            }
            __syncthreads();
            for (int p = lane; p < part_len; p += 64) {  // fold(0.0, f32::min) over the 15 chains (bi_d_array.rs:56-65)
                float acc = 0.0f;
#pragma unroll
                for (int o = 0; o < kMaxOffset; ++o) acc = f32_min(acc, chains[o * lmax + p]);
                dout[(left ? 0 : split) + p] = acc;
                zeros += (uint32_t)__popcll(__ballot(acc == 0.0f));
            }
            __syncthreads();
        }
        if (lane == 0) {
            B.counters[read].e_darray = n_ext_total;
            if (B.order) { B.sort_key[read] = zeros; atomicAdd(&B.key_hist[(size_t)(read >> B.order_shift) * kKeyBins + zeros], 1u); }
        }
        __syncthreads();
    }
}

// ---- schedule: most expensive cost class first ----------------------------------------------------------------------------
// A read's search cost is heavy-tailed (C2: mean 860 pops, maximum > 30000) and one read is a serial chain of dependent memory
// accesses, so a heavy read that starts late sets the finish time of the whole batch.  The reads whose D array leaves the search
// unpruned the longest (many zero positions) hold all of the heavy ones; starting those first overlaps their tail with the bulk.
// The order only changes when a read is processed, never its result.
// hist[chunk][k] -> first position of class k of that chunk: chunks in input order, classes descending inside a chunk.  A chunk
// (2^20 reads by default, the scale of the reference's --batch_size) bounds how many of the expensive reads start together: sorting
// a 10 M-read batch as a whole put 150 000 arena-hungry reads in front of everything else and the size-class pools ran dry.
__global__ void order_scan_kernel(uint32_t* hist, uint32_t n_chunks, uint32_t n_bins) {  // n_bins: the classes that can occur (zero positions <= the batch's longest read)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t acc = 0;
    for (uint32_t ch = 0; ch < n_chunks; ++ch) {
        uint32_t* h = hist + (size_t)ch * kKeyBins;
        for (int k = (int)n_bins - 1; k >= 0; --k) { const uint32_t c = h[k]; h[k] = acc; acc += c; }
    }
}
// The kernels around the search (D arrays, ordering, collect, records) must find room BESIDE the persistent search wavefronts of other batches.
// Twelve of those per CU take all of its LDS and leave 56 VGPRs per SIMD (the compiler does not go below 64), so a search launch asks for a little
// more LDS than it needs: eleven fit per CU, which leaves 13 KB of LDS and the registers of one wavefront on one SIMD of every CU.  The kernels
// around it are one-wavefront blocks without static LDS (wave-level ballots / shuffles instead of block-wide LDS scans); as 1024-thread blocks
// with 8-16 KB of LDS they waited for search wavefronts to retire (round 2: 243 ms instead of 0.02 ms for this kernel, 30-45 ms instead of 0.5 ms
// for the records kernel).
#define MAPAD_SLIM __launch_bounds__(64)
__global__ void MAPAD_SLIM order_scatter_kernel(BatchDev B) {
    // ranks inside the wavefront by ballots, one global atomic per (wavefront, class that occurs in it)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 64u + lane;
    const bool act = i < B.n_reads;
    const uint32_t key = act ? B.sort_key[i] : 0xFFFFFFFFu;
    uint32_t* hist = B.key_hist + (size_t)((blockIdx.x * 64u) >> B.order_shift) * kKeyBins;  // a block never straddles chunks (chunk size >= 1024)
    uint64_t todo = __ballot(act);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const bool mine = act && key == k;
        const uint64_t m = __ballot(mine);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&hist[k], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (mine) B.order[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
        todo &= ~m;
    }
}

// ---- SA locate: SampledSuffixArray::get (src/index/mod.rs:160-187) on the device, one quad per row ------------------------
// Walk LF steps (pos = less[c] + occ(pos - 1, c), c = bwt[pos]) until a sampled row (pos % 32 == 0) or a '$' row (extra_rows).
// Every step reads the 64-byte block of `pos`: lane w of the quad holds sub-block w, the lane that owns the row extracts the
// symbol, all four count it up to row pos - 1.  <= 31 dependent steps per row: latency-bound, hidden by 16 rows per wavefront.
struct LocateDev {
    const uint64_t* rows;
    uint64_t* out;
    uint64_t n_rows;
    const uint64_t* sa_sample;
    const uint64_t* x_counts;  // per block: 'X' symbols before it; nullptr if the text has none
    uint64_t extra_row[2], extra_val[2];  // the two '$' rows of the BWT and their suffix-array values
    uint32_t sa_shift;         // log2(sampling rate)
    unsigned long long* steps; // LF steps taken (event counter for the roofline)
};
__device__ __forceinline__ uint32_t quad_or32(uint32_t v) { v |= dpp_quad<0xB1>(v); v |= dpp_quad<0x4E>(v); return v; }

__global__ void __launch_bounds__(64) locate_kernel(DevIndex ix, LocateDev Q) {
    const int lane = threadIdx.x & 63, w = lane & 3;
    const uint64_t q = (uint64_t)blockIdx.x * 16 + (lane >> 2);
    const bool active = q < Q.n_rows;
    uint64_t pos = active ? Q.rows[q] : 0, offset = 0, result = ~0ull;
    bool done = !active || pos >= ix.n;
    uint32_t n_steps = 0;
    const uint64_t rate_mask = (1ull << Q.sa_shift) - 1;
    while (!__all(done)) {
        if (!done) {
            if ((pos & rate_mask) == 0) { result = Q.sa_sample[pos >> Q.sa_shift] + offset; done = true; }
            else {
                const BlockPos bp = block_pos(pos);
                const uint64_t* blk = ix.blocks + bp.b * kBlockWords;
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(blk + 2 * w);
                const int sub = bp.r_in / kSubRows, bit = bp.r_in - kSubRows * sub;
                const uint32_t code = quad_or32(w == sub ? (uint32_t)sub_code(v.x, v.y, bit) : 0u);  // bwt[pos]: 0 '$', 1 'X', 4..7 ACGT
                if (code == 0) { result = (pos == Q.extra_row[0] ? Q.extra_val[0] : Q.extra_val[1]) + offset; done = true; }
                else {
                    // occ(pos - 1, c) from the block of pos: when pos is the block's first row the mask is empty and the block's count words are the answer
                    const uint32_t m = sub_mask(w, bp.r_in - 1);
                    uint32_t sel;
                    uint64_t base_count, less;
                    if (code >= 4) {
                        const int k = (int)code - 4;
                        sel = sub_is_base(v.x, v.y, k, m);
                        base_count = blk[2 * k] & kCountMask;
                        less = k == 0 ? ix.less[1] : k == 1 ? ix.less[2] : k == 2 ? ix.less[3] : ix.less[4];
                    } else {  // 'X' (rank 5)
                        sel = sub_is_x(v.x, v.y, m);
                        base_count = Q.x_counts ? Q.x_counts[bp.b] : 0;
                        less = ix.less[5];
                    }
                    pos = less + base_count + quad_sum32((uint32_t)popc32(sel));
                    offset += 1;
                    n_steps += 1;
                }
            }
        }
    }
    if (active && w == 0) { Q.out[q] = result; if (Q.steps) atomicAdd(Q.steps, (unsigned long long)n_steps); }
}

// ---- hits -> coordinates: the index-bound half of intervals_to_bam, one thread per read (postproc_core.hpp) -----------------
__global__ void MAPAD_SLIM_EARLY records_kernel(PostIndex Q, const uint64_t* __restrict__ hit_begin, const HitRec* __restrict__ hits, const uint32_t* __restrict__ ops,
                                                        uint64_t n_reads, uint64_t seed, CoordRec* __restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t b = hit_begin[r];
    record_coords(Q, hits + b, (uint32_t)(hit_begin[r + 1] - b), ops, seed, r, out[r]);
}

// ---- hits -> record text: the text half of intervals_to_bam, one thread per read (text_core.hpp) -----------------------------------------------
// CIGAR / MD / XA bytes go into one pool per batch, the (score, size) pairs of the mapping quality into another; a wavefront claims the space of its
// 64 reads with one atomic add per pool (sizes first, then the bytes).
struct TextDev {
    TextIndex T;
    const uint64_t* hit_begin; const HitRec* hits; const uint32_t* ops; const CoordRec* coords;
    uint64_t n_reads;
    char* text; float* pairs;
    unsigned long long* cursors;  // [0] text bytes, [1] pairs claimed so far (beyond the capacities: the host grows the pools and runs the kernel again)
    uint64_t text_cap, pair_cap;
    DevRecord* out;
};
__global__ void __launch_bounds__(64) text_kernel(TextDev Q) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * 64 + lane;
    const bool act = r < Q.n_reads;
    DevRecord rec{};
    uint32_t bytes = 0, n_pairs = 0;
    const HitRec* hits = act ? Q.hits + Q.hit_begin[r] : Q.hits;
    if (act) record_text_sizes(Q.T, hits, Q.ops, Q.coords[r], rec, bytes, n_pairs);
    unsigned long long sb = bytes, sp = n_pairs;  // inclusive scans over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long ub = __shfl_up(sb, d), up = __shfl_up(sp, d);
        if (lane >= (uint32_t)d) { sb += ub; sp += up; }
    }
    const unsigned long long tot_b = __shfl(sb, 63), tot_p = __shfl(sp, 63);
    unsigned long long base_b = 0, base_p = 0;
    if (lane == 0) { base_b = tot_b ? atomicAdd(&Q.cursors[0], tot_b) : 0; base_p = tot_p ? atomicAdd(&Q.cursors[1], tot_p) : 0; }
    base_b = __shfl(base_b, 0); base_p = __shfl(base_p, 0);
    const bool fits = base_b + tot_b <= Q.text_cap && base_p + tot_p <= Q.pair_cap && base_b + tot_b <= 0xFFFFFFFFull && base_p + tot_p <= 0xFFFFFFFFull;
    if (!act) return;
    if (!fits) { if (rec.mapped) rec.error = 2; Q.out[r] = rec; return; }
    rec.text_off = (uint32_t)(base_b + sb - bytes); rec.mq_off = (uint32_t)(base_p + sp - n_pairs);
    record_text_write(Q.T, hits, Q.ops, Q.coords[r], Q.text, Q.pairs, rec);
    Q.out[r] = rec;
}

// ---- search: persistent quads -----------------------------------------------------------------------------------------
#if defined(MAPAD_OUTLINE_RARE)
#define MAPAD_FINALIZE_ATTR __attribute__((noinline))
#else
#define MAPAD_FINALIZE_ATTR __forceinline__
#endif

template <int LPR>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {  // value of the group's first lane (64: the wavefront's first active lane)
    if constexpr (LPR == 64) return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    else if constexpr (LPR == 2) return dpp_quad<0xA0>(v);  // quad_perm [0,0,2,2]: the pair's first lane
    else return LPR == 4 ? dpp_quad<0>(v) : v;
}

template <int LPR, bool NLR, bool NL, int TOP>
__device__ MAPAD_FINALIZE_ATTR void finalize_read(const BatchDev B, const ReadInT<NLR> rd, const ArenaT<NL, TOP> A, const SearchState st, uint32_t read, int w, int tier) {
    if (st.status == ST_ARENA_OVERFLOW && tier + 1 < kStages) {  // hand the read to the next stage
        if (w == 0) {
            const uint32_t k = atomicAdd(&B.cursors[CUR_OVF + 2 * tier], 1u);
            atomicExch(&B.overflow_list[(size_t)tier * B.n_reads + k], read + 1u);  // consumed by the next launch — or, tier 0, by a quad of this one that has run out of reads (search_kernel)
        }
        return;
    }
    const uint32_t n = st.status == ST_ARENA_OVERFLOW ? 0u : st.n_hits, n_ops = st.status == ST_ARENA_OVERFLOW ? 0u : st.hit_ops_used;
    uint64_t hbase = 0, obase = 0;
    if (w == 0) {
        hbase = atomicAdd((unsigned long long*)(B.cursors + CUR_HITS), (unsigned long long)n);
        obase = atomicAdd((unsigned long long*)(B.cursors + CUR_OPS), (unsigned long long)n_ops);
    }
    hbase = (uint64_t)group_bcast<LPR>((uint32_t)hbase) | ((uint64_t)group_bcast<LPR>((uint32_t)(hbase >> 32)) << 32);
    obase = (uint64_t)group_bcast<LPR>((uint32_t)obase) | ((uint64_t)group_bcast<LPR>((uint32_t)(obase >> 32)) << 32);
    uint32_t status = st.status;
    if (hbase + n > B.hits_cap || obase + n_ops > B.ops_cap) {  // caps are < 2^32, so everything stored below fits 32 bits
        status |= ST_POOL_OVERFLOW;
        if (w == 0) atomicOr(&B.cursors[CUR_POOL_OVF], 1u);
    } else {
        for (uint32_t i = w; i < n; i += LPR) { HitRec h = A.hits[i]; h.ops_off += (uint32_t)obase; B.hits_pool[hbase + i] = h; }
        for (uint32_t i = w; i < n_ops; i += LPR) B.ops_pool[obase + i] = A.hit_ops[i];
    }
    if (w == 0) {
        B.hit_count[read] = (status & ST_POOL_OVERFLOW) ? 0u : n;
        B.hit_first[read] = (uint32_t)hbase;
        B.status[read] = status;
        ReadCounters* c = B.counters + read;
        c->e_search = st.c_esearch; c->n_push = st.c_push; c->n_pop = st.c_pop; c->n_node = st.c_node; c->n_hits = st.c_hits;
        if (status == ST_ARENA_OVERFLOW) atomicOr(&B.cursors[CUR_ERR], ST_ARENA_OVERFLOW);  // cannot happen: the last stage holds the reference's limits
    }
}

// grow(A, st): called by check_and_push when the heap or the node slab is full (search_core.hpp).  All lanes of the read's lane
// group call it together with identical arguments.  Arenas change owners inside a launch and the L2s of the eight XCDs are not
// coherent with each other for ordinary stores, hence the agent-scope fences around acquire and release.
// Pools with at least kPartitionMin arenas are split into eight parts, one per XCD (HW_REG_XCC_ID): an arena of such a pool is only
// ever touched through one L2, so handing it to another read needs no L2 write-back (`buffer_wbl2` flushes every dirty line of the
// XCD, measured 4 % of C3's run time), only the completion of the old owner's stores and an L1 invalidate on the new owner's CU.
// Smaller pools (the big, rare classes) are shared by all XCDs and use full agent-scope fences.
constexpr uint32_t kPartitionMin = 64;
__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // HW_REG_XCC_ID[3:0]

// A wavefront's set of base arenas (ArenaPool::set_owner).  Pools of at least kPartitionMin sets are split by XCD like the size classes
// above: a set is then only ever used through one L2.  The probe starts at a hash of the workgroup id; a full pool (more wavefronts
// resident than sets) makes the wavefront wait for an exit — the owners never wait for a newcomer, so this cannot deadlock.
__device__ __forceinline__ uint32_t acquire_set(const ArenaPool& ap) {
    uint32_t set = 0;
    const bool part = ap.n_sets >= kPartitionMin;
    if ((threadIdx.x & 63) == 0) {
        const uint32_t m = part ? ap.n_sets / 8 : ap.n_sets, lo = part ? xcc_id() * m : 0;
        uint32_t i = (uint32_t)(((uint64_t)blockIdx.x * 2654435761u) % m), since = 0;
        for (;;) {
            if (atomicCAS(&ap.set_owner[lo + i], 0u, 1u) == 0u) break;
            if (++i == m) i = 0;
            if (++since == m) { since = 0; __builtin_amdgcn_s_sleep(64); }
        }
        set = lo + i;
    }
    set = (uint32_t)__builtin_amdgcn_readfirstlane((int)set);
    if (part) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // stale lines of this CU's L1 from an earlier owner on this CU
    else __threadfence();
    return set;
}
__device__ __forceinline__ void release_set(const ArenaPool& ap, uint32_t set) {
    if (ap.n_sets < kPartitionMin) __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the old owner's stores have reached the L2 before the next owner starts (see release_grown)
    if ((threadIdx.x & 63) == 0) atomicExch(&ap.set_owner[set], 0u);
}

// `foreign`: this wavefront is not on the XCD whose part of a partitioned pool the arena belongs to (a heavy wavefront that continues a read a
// quad on another XCD suspended): its stores sit in another L2 and must be written back like those to a shared pool's arena.
template <int LPR>
__device__ __forceinline__ void release_grown(const GrowPools* gp, uint32_t grown, int w, bool foreign = false) {
    const uint32_t cls = (grown >> kGrownShift) - 1;
    // A partitioned pool's arena stays behind one L2, so no write-back is needed, but the old owner's stores must have COMPLETED (reached
    // that L2) before the owner word is cleared: a workgroup-scope fence emits nothing on gfx950, hence the explicit wait.
    // Shared pools: every access has completed and is written back before another XCD may take the arena; the explicit wait after the
    // fence keeps the compiler from dropping the one that orders the owner word behind the write-back.
    if (foreign || gp->count[cls] < kPartitionMin) __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w == 0) atomicExch(&gp->owner[cls][grown & ((1u << kGrownShift) - 1)], 0u);
}

template <int LPR>
__device__ __forceinline__ void copy_units(MAPAD_GLOBAL uint4* dst, const MAPAD_GLOBAL uint4* src, uint32_t begin, uint32_t end, int w) {
    uint32_t i = begin + w;
    for (; i + 3 * LPR < end; i += 4 * LPR) {
        const uint4 a = src[i], b = src[i + LPR], c = src[i + 2 * LPR], d = src[i + 3 * LPR];
        dst[i] = a; dst[i + LPR] = b; dst[i + 2 * LPR] = c; dst[i + 3 * LPR] = d;
    }
    for (; i < end; i += LPR) dst[i] = src[i];
}

constexpr uint32_t kRestarted = 0x40000000u;  // in ArenaT::n_waits: this read gave up waiting for an arena once already (taken from the restart list): it waits as long as it takes now
constexpr uint32_t kTailAskEvery = 0x3FFFu;  // a read past its hand-over threshold asks every 16 384 pops (mask)
constexpr uint32_t kAskTail = 0x80000000u;  // in ArenaT::n_waits: the read's last request found its arena class dry and the class is one the host tail takes reads of
template <int LPR, bool NL, int TOP = kTop, bool HITS = false>
struct DeviceGrow {
    const GrowPools* gp;
    uint32_t* grown_counter;
    uint32_t slot;
    int w;
    bool may_give_up;
    mutable bool foreign = false;  // HITS: the arena the read is in now was taken by a wavefront on another XCD (release_grown)
    uint32_t tail_min_class = kClasses;  // a read that finds every arena of the class it needs taken, class >= this, asks for the host tail (kAskTail in A.n_waits; search_kernel decides)
    // takes an arena of a size class that holds the read: GROW_OK (cls, idx), GROW_WAIT (all suitable arenas are busy: sit out), GROW_NEVER (no class can hold it)
    __device__ __forceinline__ int acquire(ArenaT<NL, TOP>& A, const SearchState& st, uint32_t& cls, uint32_t& idx) const {
        if (A.wait) { A.wait -= 1; return GROW_WAIT; }
        const uint32_t cur = A.grown >> kGrownShift;  // class + 1 of the arena the read is in (0: its slot's base arena)
        const uint32_t first = cur == (uint32_t)kClasses + 1 ? gp->set_next_class : cur;  // first class to try; a dry class falls through to the next
        idx = ~0u;
        bool exists = false;
        uint32_t need = (uint32_t)kClasses;  // the smallest class that can hold the read
        auto probe = [&](uint32_t k) {  // an arena of class k, if one of this XCD's part is free
            const uint32_t n = gp->count[k];
            if (w == 0) {
                const bool part = n >= kPartitionMin;
                const uint32_t m = part ? n / 8 : n, lo = part ? xcc_id() * m : 0;  // this XCD's part of the pool
                uint32_t i = (uint32_t)(((uint64_t)(slot + st.tree_len) * 2654435761u) % m);
                const uint32_t tries = m < 64u ? m : 64u;
                uint32_t* own = gp->owner[k] + lo;
                for (uint32_t t = 0; t < tries; ++t) {
                    if (atomicCAS(&own[i], 0u, 1u) == 0u) { idx = lo + i; break; }
                    if (++i == m) i = 0;
                }
                if (idx == ~0u) atomicOr(grown_counter + 1, 1u << k);  // debugging aid: classes that ran dry
            }
            idx = group_bcast<LPR>(idx);
        };
        auto holds = [&](uint32_t k) { return gp->count[k] != 0 && gp->node_cap[k] >= st.tree_len + kStepNodes && gp->heap_cap[k] >= st.heap_len + kStepNodes; };
        for (cls = first; cls < (uint32_t)kClasses; ++cls) {
            if (!holds(cls)) continue;
            need = exists ? need : cls;
            exists = true;
            probe(cls);
            if (idx != ~0u) break;
        }
        if (!HITS && idx == ~0u && cur != (uint32_t)kClasses + 1 && holds(kClasses) && gp->node_cap[kClasses] > A.node_cap) {  // an idle set of base arenas as one arena
            exists = true;
            cls = kClasses;
            probe(cls);
        }
        if (idx == ~0u) {
            if (!exists) return GROW_NEVER;  // no class can hold this read: it goes to the full-limit pass
            // only reads that are still small give up (cheap to restart, and it is the many mid-size reads that clog the big pools);
            // a read that already fills a large arena keeps waiting for the few larger ones.  With the heavy path every waiting read may give up:
            // the arenas it waits for can be held by SUSPENDED reads, which only move again in the next launch (heavy_kernel.hpp).
            if (may_give_up && !(A.n_waits & kRestarted) && (first < 4 || gp->heavy_min_class < (uint32_t)kClasses) && (++A.n_waits & ~(kAskTail | kRestarted)) > gp->max_waits) return GROW_NEVER;
            // The big classes are few and held for seconds; a read that queues for one makes no pops and so never reaches the pop budget (round 4: 1 M reads of the
            // C5 mix on 3 Gbp handed reads over until second 372 while the host threads idled).  It asks for the host instead — granted while the host keeps up.
            if (tail_min_class < (uint32_t)kClasses && need >= tail_min_class) A.n_waits |= kAskTail;
            A.wait = 64;                     // every suitable arena is taken: its owners finish and give it back
            if (w == 0) atomicAdd(grown_counter + 2, 1u);
            return GROW_WAIT;
        }
        return GROW_OK;
    }
    // the read's arena is arena idx of class cls from now on (its contents have been copied)
    __device__ __forceinline__ void adopt(ArenaT<NL, TOP>& A, uint32_t cls, uint32_t idx) const {
        uint8_t* b = gp->base[cls] + (uint64_t)idx * gp->stride[cls];
        if (A.grown) release_grown<LPR>(gp, A.grown, w, foreign);
        A.heap = (MAPAD_GLOBAL HeapEntry*)(b) + 1; A.nodes = (MAPAD_GLOBAL Node*)(b + gp->off_nodes[cls]);
        A.heap_cap = gp->heap_cap[cls]; A.node_cap = gp->node_cap[cls];
        A.grown = ((cls + 1) << kGrownShift) | idx;
        foreign = false;  // taken from this XCD's part of the pool
        if (w == 0) atomicAdd(grown_counter, 1u);
    }
    __device__ __forceinline__ int operator()(ArenaT<NL, TOP>& A, const SearchState& st) const {
        uint32_t cls, idx;
        const int rc = acquire(A, st, cls, idx);
        if (rc != GROW_OK) return rc;
        if (gp->count[cls] >= kPartitionMin) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale lines of this CU's L1
        else __threadfence();
        uint8_t* b = gp->base[cls] + (uint64_t)idx * gp->stride[cls];
        MAPAD_GLOBAL HeapEntry* nheap = (MAPAD_GLOBAL HeapEntry*)(b) + 1;
        MAPAD_GLOBAL Node* nnodes = (MAPAD_GLOBAL Node*)(b + gp->off_nodes[cls]);
        // migrate heap slots [TOP, heap_len) (the top of the heap lives in the near array) and nodes [0, tree_entries)
        // (16-byte units, four loads in flight per lane: the other reads of the wavefront wait for this copy)
        {
            const MAPAD_GLOBAL uint4* hs = (const MAPAD_GLOBAL uint4*)(A.heap - 1);  // physical slot 0 is 16-byte aligned
            MAPAD_GLOBAL uint4* hd = (MAPAD_GLOBAL uint4*)(nheap - 1);
            const uint32_t h_end = (HeapLayout<TOP>::phys_end(st.heap_len) + 1) >> 1;  // pairs covering the physical slots of logical [TOP, heap_len) (heap_core.hpp: HeapLayout)
            copy_units<LPR>(hd, hs, (TOP + 1) >> 1, h_end, w);
            copy_units<LPR>((MAPAD_GLOBAL uint4*)nnodes, (const MAPAD_GLOBAL uint4*)A.nodes, 0, 2 * st.tree_entries, w);
        }
        if constexpr (HITS) {  // the hit staging travels with the read
            MAPAD_GLOBAL HitRec* nhits = (MAPAD_GLOBAL HitRec*)(b + gp->off_hits[cls]);
            MAPAD_GLOBAL uint32_t* nops = (MAPAD_GLOBAL uint32_t*)(b + gp->off_hit_ops[cls]);
            for (uint32_t i = w; i < st.n_hits * 10u; i += LPR) ((MAPAD_GLOBAL uint32_t*)nhits)[i] = ((const MAPAD_GLOBAL uint32_t*)A.hits)[i];
            for (uint32_t i = w; i < st.hit_ops_used; i += LPR) nops[i] = A.hit_ops[i];
            A.hits = nhits; A.hit_ops = nops; A.scratch = (MAPAD_GLOBAL uint16_t*)(b + gp->off_scratch[cls]);
        }
        adopt(A, cls, idx);
        return GROW_OK;
    }
    // The same at a wavefront-uniform point of the kernel, for migrations worth it (`need`: this quad's arena is full and holds >= GrowPools::wide_copy_nodes nodes): the quads
    // that need an arena are served one after the other — the quad takes it, ALL 64 lanes copy (4 KB per round trip instead of the quad's 256 bytes: a 0.64 MB base
    // arena moves in ~0.3 ms instead of ~5 ms, during which the other 15 reads of the wavefront wait either way).  A quad whose request fails (pools busy) is left to
    // the step's own call of operator(), which sits out.  Quads only (LPR == 4, no hit staging).
    __device__ __forceinline__ void wide(ArenaT<NL, TOP>& A, const SearchState& st, int lane, bool need) const {
        static_assert(!HITS, "wide migrations are for the quad kernel");
        unsigned long long todo = __ballot(need && w == 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const bool mine = (lane >> 2) == (leader >> 2);
            uint32_t cls = ~0u, idx = ~0u;
            int rc = GROW_WAIT;
            if (mine) rc = acquire(A, st, cls, idx);
            rc = __builtin_amdgcn_readlane(rc, leader);
            if (rc != GROW_OK) continue;
            cls = (uint32_t)__builtin_amdgcn_readlane((int)cls, leader); idx = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
            if (gp->count[cls] >= kPartitionMin) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            else __threadfence();
            auto bcast64 = [&](uint64_t v) -> uint64_t {
                return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), leader) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, leader);
            };
            const MAPAD_GLOBAL uint4* hs = (const MAPAD_GLOBAL uint4*)bcast64((uint64_t)(A.heap - 1));
            const MAPAD_GLOBAL uint4* ns = (const MAPAD_GLOBAL uint4*)bcast64((uint64_t)A.nodes);
            const uint32_t heap_len = (uint32_t)__builtin_amdgcn_readlane((int)st.heap_len, leader), entries = (uint32_t)__builtin_amdgcn_readlane((int)st.tree_entries, leader);
            uint8_t* b = gp->base[cls] + (uint64_t)idx * gp->stride[cls];
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));  // (opaque: see hand_to_host)
            copy_units<64>((MAPAD_GLOBAL uint4*)b, hs, (TOP + 1) >> 1, (HeapLayout<TOP>::phys_end(heap_len) + 1) >> 1, lane_o);
            copy_units<64>((MAPAD_GLOBAL uint4*)(b + gp->off_nodes[cls]), ns, 0, 2 * entries, lane_o);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the copy has landed before the quad works in the new arena (and before its old one changes owners)
            if (mine) adopt(A, cls, idx);
        }
    }
};

// A quad hands its read to heavy_kernel: heap top (near array) -> logical slots [0, kTop) of the grown arena's heap, hit staging (base arena) ->
// the grown arena's hit area, search state -> a HeavyItem.  The item is read by a later launch on the same stream.
template <int LPR, bool NL>
__device__ __forceinline__ void suspend_heavy(const BatchDev& B, const GrowPools* gp, const ArenaT<NL>& A, const SearchState& st, uint32_t read, int w, int tier) {
    const uint32_t cls = (A.grown >> kGrownShift) - 1, idx = A.grown & ((1u << kGrownShift) - 1);
    uint8_t* b = gp->base[cls] + (uint64_t)idx * gp->stride[cls];
    const uint32_t n_top = st.heap_len < (uint32_t)kTop ? st.heap_len : (uint32_t)kTop;
    for (uint32_t i = w; i < n_top; i += LPR) store_entry(A.heap + i, load_entry(A.top + i));
    MAPAD_GLOBAL uint32_t* nhits = (MAPAD_GLOBAL uint32_t*)(b + gp->off_hits[cls]);
    MAPAD_GLOBAL uint32_t* nops = (MAPAD_GLOBAL uint32_t*)(b + gp->off_hit_ops[cls]);
    for (uint32_t i = w; i < st.n_hits * 10u; i += LPR) nhits[i] = ((const MAPAD_GLOBAL uint32_t*)A.hits)[i];
    for (uint32_t i = w; i < st.hit_ops_used; i += LPR) nops[i] = A.hit_ops[i];
    if (w == 0) {
        const uint32_t k = atomicAdd(&B.cursors[CUR_HEAVY_N + tier], 1u);
        if (k < B.heavy_cap) { HeavyItem it; it.read = read; it.grown = A.grown; it.st = st; B.heavy[(size_t)tier * B.heavy_cap + k] = it; }
        else atomicOr(&B.cursors[CUR_ERR], ST_ARENA_OVERFLOW);  // cannot happen: a suspended read holds a grown arena, and the list has room for all of them
    }
}

// A quad gives its read to the host tail (host_tail.hpp): position data and D array -> record k of the batch's ring in host-coherent page-locked memory
// (hipHostMallocCoherent), visible to the host while the kernel runs.  Every word goes out as a system-scope store — written through to the host, never
// left dirty in an L2 — so that the order "payload, then `ready`" needs only the completion of the payload's stores (vmcnt), not the write-back of the XCD's
// whole L2 that a system-scope release fence costs (round 4 used one per hand-over, and ordinary page-locked memory, whose visibility during a launch is not
// guaranteed: ADVICE r4).  tests/test_gpu_tail.py checks that records arrive while the launch is still running (tail.seen_live).
template <class T>
__device__ __forceinline__ void store_through(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <int LPR, bool NL, class AR>
__device__ __forceinline__ void hand_to_host(const BatchDev& B, const ReadInT<NL>& rd, const AR& A, const SearchState& st, uint32_t read, uint32_t k, int w, bool with_state) {
    uint8_t* rec = B.tail_ring + (size_t)k * B.tail_stride;
    uint32_t* rq = (uint32_t*)(rec + 16);
    uint32_t* rdd = (uint32_t*)(rec + 16 + ((2u * B.tail_lmax + 15u) & ~15u));
    const typename near_ptr<const uint32_t, NL>::type qc4 = (typename near_ptr<const uint32_t, NL>::type)rd.qc;  // 16-byte aligned (near layout)
    const uint32_t nq = (2u * (uint32_t)rd.L + 3u) / 4u;
    for (uint32_t i = w; i < nq; i += LPR) store_through(&rq[i], (uint32_t)qc4[i]);
    for (uint32_t i = w; i < (uint32_t)rd.L; i += LPR) store_through(&rdd[i], __float_as_uint(rd.d[i]));
    host::TailRecord* h = (host::TailRecord*)rec;
    if (w == 0) { store_through(&h->read, read); store_through(&h->L, (uint32_t)rd.L); store_through(&h->pops, st.c_pop); }
    {   // the search itself, if the read leaves its grown arena to the host (TailState): heap levels 0-5 out of the near array into the arena's unused slots,
        // the state into the record; the arena's lines must then leave this XCD's L2 — a copy engine reads them, not a CU (the one place a hand-over pays for a
        // write-back of the L2's dirty lines)
        uint32_t* ts = (uint32_t*)(rec + host::tail_state_offset(B.tail_lmax));
        if (with_state) {
            const uint32_t n_top = st.heap_len < (uint32_t)AR::kTopN ? st.heap_len : (uint32_t)AR::kTopN;
            uint32_t i0 = (uint32_t)w;
            asm volatile("" : "+v"(i0));  // (opaque: lane-derived loop bounds of this rare path were hoisted to the kernel's start and then spilled to scratch)
#pragma unroll 1
            for (uint32_t i = i0; i < n_top; i += LPR) store_entry(A.heap + i, load_entry(A.top + i));
            if (w == 0) {  // (field by field: taking the state's address would put it — the hottest registers of the loop — into scratch memory)
                store_through(&ts[0], A.grown);
                store_through(&ts[4], st.c_esearch); store_through(&ts[5], st.c_push); store_through(&ts[6], st.c_pop); store_through(&ts[7], st.c_node); store_through(&ts[8], st.c_hits);
                store_through(&ts[9], st.heap_len); store_through(&ts[10], st.tree_entries); store_through(&ts[11], st.tree_next); store_through(&ts[12], st.tree_len);
                store_through(&ts[13], st.n_hits); store_through(&ts[14], st.hit_ops_used); store_through(&ts[15], st.status); store_through(&ts[16], __float_as_uint(st.best_score));
                store_through(&ts[17], 0u); store_through(&ts[18], (uint32_t)st.best_size); store_through(&ts[19], (uint32_t)(st.best_size >> 32));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: write back
        } else if (w == 0) store_through(&ts[0], 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's payload stores have completed ...
    __builtin_amdgcn_wave_barrier();
    if (w == 0) {
        store_through(&h->ready, B.tail_gen);  // ... before the word the host polls
        B.status[read] = ST_TAIL; B.hit_count[read] = 0; B.hit_first[read] = 0;
    }
}

// PASS 0: every read, growable arenas (PASS 2: the same code for the retry launches).  PASS 1: the reads that no size class could
// hold, arenas with the reference's full limits.
// NL: the near data (heap top, position data) of every read slot is in LDS and addressed with ds_* instructions; otherwise it
// lives in the slot's HBM arena (reads longer than kMaxLdsReadLen, lanes-per-read 1).
// MAPAD_OPAQUE_LANE=1: the lane's index inside its group, the stage number and the uniform conditions on the parameters are made opaque once per trip of the search
// loop, so that the compiler recomputes `w == 0`, `nq == 1`, ... (one v_cmp / s_cmp) instead of hoisting them as 64-bit masks and spilling those into VGPR lanes:
// SGPR spills of the shipped kernel 34 -> 29, VGPRs 168 -> 156 — and 1.6 % SLOWER on all three workloads, same box (profiles/r06/ab_opaque_lane.txt: C3 2.59 -> 2.55 M
// reads/s, C2 6.05 -> 5.94 M, C4 3.10 -> 3.05 M): a v_readlane of a spilled mask is cheaper than the compare that rebuilds it.  Off.
#if !defined(MAPAD_OPAQUE_LANE)
#define MAPAD_OPAQUE_LANE 0
#endif
#if !defined(MAPAD_NODE_PREFETCH)
#define MAPAD_NODE_PREFETCH 0  // (search_core.hpp: NodePrefetch — measured +-0 on C3 / C4, -0.8 % on C2, same box: profiles/r06/ab_node_prefetch.txt; off)
#endif
#if !defined(MAPAD_MIN_WAVES)
#define MAPAD_MIN_WAVES 3  // 12 wavefronts per CU: 168 VGPRs (no SGPR spills into VGPR lanes) and 853 B of LDS per read slot (heap levels 0-5)
#endif
// The batch and arena descriptors are only needed when a read starts, ends or changes arenas, but as by-value kernel arguments they stay live
// in scalar registers across the whole search loop (~60 of them: the loop spilled 75 scalar registers into VGPR lanes, v_writelane / v_readlane).
// Inside the loop they are therefore re-read from the kernel-argument segment where they are used (a scalar load from constant memory; the empty
// asm keeps the optimiser from hoisting the loads back out of the rare paths).
#if !defined(MAPAD_KERNARG_RELOAD)
#define MAPAD_KERNARG_RELOAD 1
#endif
constexpr size_t kArgAlign(size_t off, size_t a) { return (off + a - 1) / a * a; }
constexpr size_t kArgOffP = kArgAlign(sizeof(DevIndex), alignof(DevParams));
constexpr size_t kArgOffB = kArgAlign(kArgOffP + sizeof(DevParams), alignof(BatchDev));
constexpr size_t kArgOffAP = kArgAlign(kArgOffB + sizeof(BatchDev), alignof(ArenaPool));
template <class T>
__device__ __forceinline__ T kernarg_reload(size_t off, const T& by_value) {
#if MAPAD_KERNARG_RELOAD && defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const __attribute__((address_space(4))) T*)(p + off);
#else
    return by_value;
#endif
}

// HEAVY: the kernel hands reads that outgrow their base arena to heavy_kernel (MAPAD_HEAVY=1); compiled out otherwise (the hand-over code in the
// loop costs the common path registers and waits: C2 -15 % with it in, measured).
// Lanes per read 2 (MAPAD_LANES_PER_READ=2): a pair of lanes per read, 32 reads per wavefront — the quad-uniform part of a step (heap, gates, commit loop: 80 % of
// its instructions) then serves twice the reads per instruction issued, each lane answers the rank queries for two bases and packs their children.  Its read slots
// keep heap levels 0-4 near (MAPAD_KTOP2 = 31: 576 B per slot at 50 bp), so that eight such blocks fit a CU's LDS.
#if !defined(MAPAD_KTOP2)
#define MAPAD_KTOP2 31
#endif
template <int LPR> struct top_of { static constexpr int value = LPR == 2 ? MAPAD_KTOP2 : kTop; };
template <int LPR, bool CONT, int PASS, bool NL, bool HEAVY>
__global__ void __launch_bounds__(64, (LPR == 1 && NL) ? 1 : LPR == 2 ? 2 : MAPAD_MIN_WAVES) search_kernel(DevIndex ix, DevParams P, BatchDev B0, ArenaPool AP0, const GrowPools* GP, uint32_t near_stride, uint32_t near_lmax, int stage) {
    const int lane = threadIdx.x & 63, w0 = lane & (LPR - 1);
    const int tier0 = stage;
    const uint32_t n_items = tier0 == 0 ? B0.n_reads : B0.cursors[CUR_OVF + 2 * (tier0 - 1)];
    if (n_items == 0) return;  // retry / full-limit stages normally have nothing to do
    uint32_t* const cursors = B0.cursors;
    const uint32_t set = acquire_set(AP0);
    const uint32_t slot = set * (64 / LPR) + (lane / LPR);
    constexpr int TOPK = top_of<LPR>::value;
    static_assert(!HEAVY || LPR == 4, "reads are handed to heavy wavefronts by quads only");
    ArenaT<NL, TOPK> A = carve<NL, TOPK>(AP0, slot);
    // near data of this read slot: [kTop + 1 heap slots][2*lmax bytes class/quality][lmax floats D]
    extern __shared__ __attribute__((aligned(16))) uint8_t near_lds[];
    using NearBytes = typename near_ptr<uint8_t, NL>::type;
    NearBytes near;
    if constexpr (NL) near = (NearBytes)near_lds + (size_t)(lane / LPR) * near_stride;
    else near = AP0.base + (uint64_t)slot * AP0.stride + AP0.off_near;
    A.top = (typename near_ptr<HeapEntry, NL>::type)near + 1;
    const NearBytes near_qc = near + (TOPK + 1) * sizeof(HeapEntry);
    const typename near_ptr<float, NL>::type near_d = (typename near_ptr<float, NL>::type)(near_qc + ((2 * near_lmax + 15) & ~15u));
    uint32_t* work = &cursors[CUR_WORK + 2 * tier0];
    // reads of the first stages give up after kMaxWaits fruitless waits for an arena and are restarted by the next stage, when the
    // pools are quiet; the last growable stage waits as long as it takes
    const DeviceGrow<LPR, NL, TOPK> grow{GP, &cursors[CUR_GROWN], slot, w0, stage + 2 < kStages || HEAVY, false, B0.tail_min_class};
    const uint32_t wide_copy_nodes = GP->wide_copy_nodes;
    const uint32_t tail_lo = B0.tail_pops_idle;  // first pop count at which a read asks for the host tail (<= B.tail_pops; the ask itself sorts out which rule applies)
    bool tail_denied = false;  // the ring was full when this read asked: it stays on the GPU
    bool drained = false;      // the launch's own list of reads is exhausted (this quad has seen its end)
#if defined(MAPAD_PROFILE_SECTIONS)
    if (lane < 2 * PROF_N + 2) g_prof_lds[lane] = 0;
    g_prof_hist[lane] = 0;
    if (lane == 0) g_prof_lds[2 * PROF_N] = __builtin_amdgcn_s_memtime();
#endif
    bool have = false, done = false;
    ReadInT<NL> rd{near_qc, near_d, 0, 0.0f, 0};
    if constexpr (LPR == 4) rd.lane_less = w0 == 0 ? ix.less[1] : w0 == 1 ? ix.less[2] : w0 == 2 ? ix.less[3] : ix.less[4];
    if constexpr (LPR == 2) { rd.lane_less = w0 == 0 ? ix.less[1] : ix.less[3]; rd.lane_less1 = w0 == 0 ? ix.less[2] : ix.less[4]; }
    SearchState st;
#if MAPAD_NODE_PREFETCH
    NodePrefetch node_pf;  // (search_core.hpp; quads with their near data in LDS)
#endif
    uint32_t read = 0;
    for (;;) {
        int w = w0, tier = tier0;  // (MAPAD_OPAQUE_LANE, above)
#if MAPAD_OPAQUE_LANE
        asm volatile("" : "+v"(w));
        asm volatile("" : "+s"(tier));
#endif
        MAPAD_MARK(PROF_LOOP);
        if (!have && !done) {
            uint32_t item = 0;
            if (w == 0 && !drained) item = atomicAdd(work, 1u);
            item = group_bcast<LPR>(item);
            uint32_t again = ~0u;  // an entry of the restart list
            if (PASS == 0 && !HEAVY && (drained | (item >= n_items))) {  // (not with reads suspended to heavy wavefronts: the arenas a waiting read needs may be held by suspended reads, which only move in the next launch — there every waiting read must be able to give up)
                // Out of reads.  The reads of this launch that gave up waiting for an arena start again in the next launch — which begins when the LAST read of this
                // one is done, although quads idle here from the moment the bulk is through (C5 mix, 1 M reads on 3 Gbp: 25 000-33 000 such reads; the second launch
                // was a quarter of the batch's time).  A quad that has run out takes them over: it claims the next entry of the restart list (the next launch's
                // work counter, by CAS, so that no entry is skipped) and starts that read from scratch, as the next launch would.  Entries appended later are the
                // next launch's, as before.
                drained = true;
                if (w == 0) {
                    const uint32_t cnt = __hip_atomic_load(&cursors[CUR_OVF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t cur = __hip_atomic_load(&cursors[CUR_WORK + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (cur < cnt) {
                        const uint32_t old = atomicCAS(&cursors[CUR_WORK + 2], cur, cur + 1u);
                        if (old == cur) { again = cur; break; }
                        cur = old;
                    }
                }
                again = group_bcast<LPR>(again);
            }
            if ((item >= n_items || drained) && again == ~0u) done = true;
            else {
                const BatchDev B = kernarg_reload(kArgOffB, B0);
                const uint32_t* items = B.overflow_list + (size_t)(tier > 0 ? tier - 1 : 0) * B.n_reads;
                if (again != ~0u) {  // (the entry was reserved before it was written: wait for the writer, a few instructions away)
                    uint32_t e = 0;
                    if (w == 0) { while ((e = atomicExch(&B.overflow_list[again], 0u)) == 0u) __builtin_amdgcn_s_sleep(8); }  // (taken and cleared for the next batch in one go)
                    read = group_bcast<LPR>(e) - 1u;
                } else {
                    read = tier == 0 ? (B.order ? B.order[item] : item) : items[item] - 1u;
                    if (tier >= 1 && w == 0) B.overflow_list[(size_t)(tier - 1) * B.n_reads + item] = 0u;  // the lists are left as they were found: zeros (a later batch of the slot with more reads lays its restart list over this range)
                }
                const uint64_t off = B.offsets[read];
                rd.L = (int)(B.offsets[read + 1] - off);
                const DevParams Pf = kernarg_reload(kArgOffP, P);  // the two table pointers are only used here
                rd.thr = Pf.reject_thr[rd.L];
                rd.table = Pf.table_base[rd.L];
                if (tier == 0 && B.status[read] == ST_NO_TABLE) {  // the D kernel already flagged it
                    if (w == 0) { B.hit_count[read] = 0; B.hit_first[read] = 0; }
                } else {
                    read_setup(B.seqs + off, B.quals + off, B.d_arrays + off, rd.L, near_qc, near_d, w, LPR);
                    SearchState tmp;
                    A.n_waits = again != ~0u ? kRestarted : 0u;
                    search_init(kernarg_reload(0, ix).n, alignment_start_of(P, rd.L), rd, A, tmp);
                    st = tmp;
#if MAPAD_NODE_PREFETCH
                    node_pf.ok = false;
#endif
                    have = true;
                    tail_denied = false;
                }
                drain_memory();  // rare path of the step loop (search_core.hpp: drain_memory) — on both arms: the loads of thr / table would otherwise count as pending at the step's first use of them
            }
        }
        MAPAD_MARK(PROF_SETUP);
        if (__all(done)) break;
        if constexpr (PASS != 1 && LPR == 4) {  // big arena migrations here, where every lane of the wavefront is active: all 64 of them copy (DeviceGrow::wide)
            const bool need = have && (st.heap_len != 0) & (st.status == ST_OK) & ((st.tree_len + kStepNodes > A.node_cap) | (st.heap_len + kStepNodes > A.heap_cap)) &
                              (st.tree_entries >= wide_copy_nodes);
            if (MAPAD_UNLIKELY(__any(need))) { grow.wide(A, st, lane, need); drain_memory(); }
        }
        if (have) {
            bool cont;
            DevParams Ps = P;  // (the same for the uniform conditions on the parameters — `nq == 1`, `bound_kind == 2`, `gap_dist_ends > 0`, ...: 64-bit masks when hoisted, one s_cmp when not)
#if MAPAD_OPAQUE_LANE
            asm volatile("" : "+s"(Ps.nq), "+s"(Ps.bound_kind), "+s"(Ps.gap_dist_ends), "+s"(Ps.max_num_gaps_open), "+s"(Ps.start_at_end), "+s"(Ps.stack_limit_abort));
#endif
#if MAPAD_NODE_PREFETCH
            if constexpr (PASS != 1 && LPR == 4 && NL) cont = search_step_pf<LPR, CONT, NL, false>(ix, Ps, rd, A, st, w, grow, node_pf);
            else
#endif
            if constexpr (PASS != 1) cont = search_step<LPR, CONT, NL>(ix, Ps, rd, A, st, w, grow);
            else cont = search_step<LPR, CONT, NL>(ix, Ps, rd, A, st, w, NoGrow());
            MAPAD_MARK(PROF_TAIL);
            // the read leaves this quad for a host thread (host_tail.hpp), which maps it from scratch: a record of the batch's ring, if one is left
            // `limit`: the hand-over happens only while the host has fewer reads than that waiting or running — the dispatcher's count (tail_ctl[0]) plus the records of
            // this launch it has not picked up yet (this launch's ring cursor, exact, less tail_ctl[1]); the ring slot is claimed by CAS under that test, so a crowd of
            // quads that ask in the same microsecond cannot overrun the limit (the host's word alone is a millisecond old).  0xFFFFFFFF: whatever the backlog.
            // Returns 0 = handed over, 1 = refused (backlog), 2 = the ring is full.
            auto give_to_host = [&](int why, uint32_t limit) -> int {
                const BatchDev B = kernarg_reload(kArgOffB, B0);
                if (B.tail_cap == 0) return 2;
                uint32_t k = 0xFFFFFFFEu;  // refused
                if (w == 0) {
                    if (limit == 0xFFFFFFFFu) k = atomicAdd(&cursors[CUR_TAIL], 1u);
                    else {
                        const uint32_t seen = __hip_atomic_load(B.tail_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const uint32_t pend = __hip_atomic_load(B.tail_ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        uint32_t cur = __hip_atomic_load(&cursors[CUR_TAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (;;) {
                            if (cur >= B.tail_cap) { k = 0xFFFFFFFFu; break; }
                            if ((uint64_t)pend + (cur > seen ? cur - seen : 0u) >= limit) break;
                            const uint32_t old = atomicCAS(&cursors[CUR_TAIL], cur, cur + 1u);
                            if (old == cur) { k = cur; break; }
                            cur = old;
                        }
                    }
                }
                k = group_bcast<LPR>(k);
                if (k >= B.tail_cap) return k == 0xFFFFFFFEu ? 1 : 2;
                // with its state if it sits in a grown arena and has found nothing yet (hit staging stays in the slot's base arena): the arena then belongs to the
                // read until a host thread has copied it and releases it (TailBatch::fetch_state) — this quad just lets go of it
                // (quads only: the host's step reads the arena's heap levels in kTop's layout — heap_core.hpp: HeapLayout —, a pair kernel's blocks start one level earlier)
                const bool with_state = PASS != 1 && TOPK == kTop && B.tail_continue != 0 && A.grown != 0 && st.n_hits == 0 && st.status == ST_OK;
                hand_to_host<LPR, NL>(B, rd, A, st, read, k, w, with_state);
                if (w == 0 && why) atomicAdd(&cursors[why], 1u);
                if (with_state) {
                    if (w == 0) atomicAdd(&cursors[CUR_TAIL_STATE], 1u);
                    const ArenaT<NL, TOPK> base = carve<NL, TOPK>(kernarg_reload(kArgOffAP, AP0), slot);
                    A.heap = base.heap; A.nodes = base.nodes; A.heap_cap = base.heap_cap; A.node_cap = base.node_cap; A.grown = 0;
                }
                return 0;
            };
            auto back_to_base = [&]() {  // give the grown arena back; the next read starts in the base arena again
                if (PASS != 1 && A.grown) {
                    release_grown<LPR>(GP, A.grown, w);
                    const ArenaT<NL, TOPK> base = carve<NL, TOPK>(kernarg_reload(kArgOffAP, AP0), slot);
                    A.heap = base.heap; A.nodes = base.nodes; A.heap_cap = base.heap_cap; A.node_cap = base.node_cap; A.grown = 0;
                }
            };
            if (!cont) {
                // A read the last growable stage cannot hold would start again in the full-limit stage — wavefront-per-read arenas of 336 MB, 5.7 us per pop, for
                // tens of millions of pops (mapping.rs:1358-1380).  With the host tail on it goes to a host thread instead (0.34 us per pop), like the reads past the
                // pop budget; only a full ring leaves it to the GPU's last stage.
                bool handed = false;
                const int tier_r = tier + ((A.n_waits & kRestarted) ? 1 : 0);  // a read taken from the restart list is a read of the next stage
                if (PASS != 1 && tier_r + 2 == kStages) { if (MAPAD_UNLIKELY(st.status == ST_ARENA_OVERFLOW)) handed = give_to_host(CUR_TAIL_F, 0xFFFFFFFFu) == 0; }
                if (!handed) finalize_read<LPR>(kernarg_reload(kArgOffB, B0), rd, A, st, read, w, tier_r);
                back_to_base();
                have = false;
                drain_memory();
                MAPAD_MARK(PROF_FINALIZE);
            } else if (MAPAD_UNLIKELY((((st.c_pop >= tail_lo) & (((st.c_pop - tail_lo) & kTailAskEvery) == 0u)) | (A.n_waits >= kAskTail)) & !tail_denied)) {
                // A host thread could take this read over and the quad its next read — while the host's workers keep up with what they have (every rule looks at the
                // backlog: a rank of eight on a 16-CPU box has two workers, and in round 5 the reads the GPU gave up queued behind them for minutes while its quads idled).
                //   past the pop budget (tail_pops): while fewer than tail_backlog_budget reads wait or run (8 per worker: a worker's pop is 15 x a quad's);
                //   past tail_pops_idle: while fewer than tail_backlog_idle do (one per worker: a worker is idle) — a launch lasts as long as its slowest read, and until
                //     the first reads reach the budget (1 M pops: 5.8 s) the workers had nothing to do;
                //   queuing for an arena class that is dry (DeviceGrow::acquire): while fewer than tail_backlog_max do (half the workers).
                // A read that is refused goes on on the GPU (thousands of quads run side by side) and asks again every kTailAskEvery + 1 pops (90 ms; an ask is two
                // reads of host memory: a pop's time).
                const BatchDev B = kernarg_reload(kArgOffB, B0);
                const bool dry = A.n_waits >= kAskTail, over = st.c_pop >= B.tail_pops, at_ask = (st.c_pop >= tail_lo) & (((st.c_pop - tail_lo) & kTailAskEvery) == 0u);
                A.n_waits &= ~kAskTail;  // (it asks again when its wait is over and the class is still dry)
                uint32_t limit = dry ? B.tail_backlog_max : 0u;
                int why = CUR_TAIL_DRY;
                if (at_ask & !over & (B.tail_backlog_idle > limit)) { limit = B.tail_backlog_idle; why = CUR_TAIL_IDLE; }
                if (at_ask & over) { limit = B.tail_backlog_budget; why = 0; }
                const int rc = give_to_host(why, limit);
                if (rc == 0) { back_to_base(); have = false; }
                else if (rc == 2) tail_denied = true;  // the ring is full: this read stays on the GPU
                drain_memory();
            } else if constexpr (HEAVY) { if (MAPAD_UNLIKELY(A.grown != 0)) {
                // The read has outgrown its base arena: it is heavy.  Its state goes into the grown arena (heap top from the near array, hit staging
                // from the base arena), the read is queued for heavy_kernel — a wavefront of its own — and this quad takes its next read.
                // (a suspended read keeps its grown arena until the heavy stage has finished it: no more of them than the pools can spare)
                if ((A.grown >> kGrownShift) - 1 >= GP->heavy_min_class && *(volatile uint32_t*)&cursors[CUR_HEAVY_N + tier] < GP->heavy_max_pending) {
                    suspend_heavy<LPR>(kernarg_reload(kArgOffB, B0), GP, A, st, read, w, tier);
                    const ArenaT<NL> base = carve<NL>(kernarg_reload(kArgOffAP, AP0), slot);
                    A.heap = base.heap; A.nodes = base.nodes; A.heap_cap = base.heap_cap; A.node_cap = base.node_cap; A.grown = 0;
                    have = false;
                }
                drain_memory();
            } }
        }
    }
    release_set(kernarg_reload(kArgOffAP, AP0), set);
#if defined(MAPAD_PROFILE_SECTIONS)
    __syncthreads();
    const BatchDev B = kernarg_reload(kArgOffB, B0);
    if (PASS == 0 && lane < 2 * PROF_N && B.prof) atomicAdd(&B.prof[lane], g_prof_lds[lane]);
    if (PASS == 0 && B.prof) atomicAdd(&B.prof[2 * PROF_N + lane], (unsigned long long)g_prof_hist[lane]);
#endif
}

// heavy_kernel (one wavefront per read) is compiled only into the -DMAPAD_HEAVY_KERNEL build (libmapad_amd.heavy.so, mapad_amd/build.py): measured in round 3 not to beat the
// quad per pop, off the default path since round 5.  The default library's full-limit stage is the quad kernel itself (PASS 1) in the full-limit arenas.
#if defined(MAPAD_HEAVY_KERNEL)
#include "heavy_kernel.hpp"
#endif

// ---- a few words from device memory into page-locked host memory, by a one-wavefront kernel ---------------------------------------------------------------
// What the host needs to read back between launches (a batch's cursors, its base count, the text kernel's pool cursors) is a hundred bytes — and a hipMemcpy of
// a hundred bytes is a copy KERNEL of the runtime whose workgroups do not fit beside a search launch: host trace of the C4 bench, round 4: 1.2-1.8 s per such
// copy, waiting for the persistent wavefronts to thin out, every second search launch 135 ms late.  This kernel is one wavefront like the other kernels around
// the search, which the launches leave room for.
// The same for clearing a batch's buffers: hipMemsetAsync is a fill kernel of the runtime with large workgroups, and in front of a batch's D-array kernel it kept the
// whole batch waiting for the running search to thin out (launch marks of the C4 bench: the next batch's preparation started 3 s late whenever it was submitted
// a few milliseconds after the search beside it had filled the chip).
__global__ void MAPAD_SLIM zero_words_kernel(uint32_t* __restrict__ p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64) p[i] = 0u;
}
__global__ void MAPAD_SLIM publish_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += 64) dst_host[i] = src[i];
    __threadfence_system();
}

// ---- results of the host tail -> the batch's per-read arrays (the hits and edit tracks themselves are copied into the pools) -----------------------
struct TailUp { uint32_t read, status, hit_count, hit_first, e_search, n_push, n_pop, n_node, n_hits; };
__global__ void MAPAD_SLIM tail_scatter_kernel(BatchDev B, const TailUp* __restrict__ up, uint32_t n) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const TailUp u = up[i];
    B.status[u.read] = u.status; B.hit_count[u.read] = u.hit_count; B.hit_first[u.read] = u.hit_first;
    ReadCounters* c = B.counters + u.read;  // e_darray stays as the D-array kernel wrote it
    c->e_search = u.e_search; c->n_push = u.n_push; c->n_pop = u.n_pop; c->n_node = u.n_node; c->n_hits = u.n_hits;
}

// ---- order-preserving collect (mapping.rs:288) on the device ------------------------------------------------------------------
// The search writes a read's hits wherever the bump cursors stood when the read finished.  These three kernels lay hits and edit
// operations out in read order (hit_begin / ops_begin = exclusive prefix sums over the reads), so that results leave the GPU — to
// the host or to rank 0 — as dense arrays that concatenate across shards.
constexpr int kScanTile = 256;  // reads per block (one wavefront), 4 per lane
struct CompactDev {
    const uint32_t* hit_count; const uint32_t* hit_first; const HitRec* pool; const uint32_t* ops_pool;
    uint64_t n_reads;
    uint64_t* hit_begin; uint64_t* ops_begin;  // [n_reads + 1]
    unsigned long long* tile_hits; unsigned long long* tile_ops;  // [n_tiles]
    HitRec* hits_out; uint32_t* ops_out;
};
__device__ __forceinline__ uint32_t read_ops(const CompactDev& Q, uint64_t r) {
    uint32_t s = 0;
    const uint32_t c = Q.hit_count[r], f = Q.hit_first[r];
    for (uint32_t k = 0; k < c; ++k) s += Q.pool[f + k].n_ops;
    return s;
}
// inclusive scan over the wavefront's lanes (two sums at once)
__device__ __forceinline__ void wave_scan2(unsigned long long& a, unsigned long long& b, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long ua = __shfl_up(a, d), ub = __shfl_up(b, d);
        if (lane >= (uint32_t)d) { a += ua; b += ub; }
    }
}
__global__ void MAPAD_SLIM compact_sums_kernel(CompactDev Q) {
    unsigned long long h = 0, o = 0;
    for (int q = 0; q < 4; ++q) {
        const uint64_t r = (uint64_t)blockIdx.x * kScanTile + q * 64 + threadIdx.x;
        if (r < Q.n_reads) { h += Q.hit_count[r]; o += read_ops(Q, r); }
    }
    for (int d = 32; d; d >>= 1) { h += __shfl_xor(h, d); o += __shfl_xor(o, d); }
    if (threadIdx.x == 0) { Q.tile_hits[blockIdx.x] = h; Q.tile_ops[blockIdx.x] = o; }
}
__global__ void MAPAD_SLIM compact_scan_tiles_kernel(CompactDev Q, uint64_t n_tiles) {  // one wavefront: exclusive scan of the tile sums
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long carry_h = 0, carry_o = 0;
    for (uint64_t base = 0; base < n_tiles; base += 64) {
        const uint64_t i = base + lane;
        const unsigned long long vh = i < n_tiles ? Q.tile_hits[i] : 0, vo = i < n_tiles ? Q.tile_ops[i] : 0;
        unsigned long long sh = vh, so = vo;
        wave_scan2(sh, so, lane);
        if (i < n_tiles) { Q.tile_hits[i] = carry_h + sh - vh; Q.tile_ops[i] = carry_o + so - vo; }
        carry_h += __shfl(sh, 63); carry_o += __shfl(so, 63);
    }
    if (lane == 0) { Q.hit_begin[Q.n_reads] = carry_h; Q.ops_begin[Q.n_reads] = carry_o; }
}
__global__ void MAPAD_SLIM compact_move_kernel(CompactDev Q) {
    // tile-local exclusive scan: lane t owns reads 4t .. 4t+3 of the tile
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r0 = (uint64_t)blockIdx.x * kScanTile + 4 * lane;
    uint32_t c[4], o[4];
    unsigned long long th = 0, to = 0;
    for (int q = 0; q < 4; ++q) {
        const uint64_t r = r0 + q;
        c[q] = r < Q.n_reads ? Q.hit_count[r] : 0; o[q] = r < Q.n_reads ? read_ops(Q, r) : 0;
        th += c[q]; to += o[q];
    }
    unsigned long long sh = th, so = to;
    wave_scan2(sh, so, lane);
    unsigned long long hb = Q.tile_hits[blockIdx.x] + sh - th, ob = Q.tile_ops[blockIdx.x] + so - to;
    for (int q = 0; q < 4; ++q) {
        const uint64_t r = r0 + q;
        if (r >= Q.n_reads) break;
        Q.hit_begin[r] = hb; Q.ops_begin[r] = ob;
        const uint32_t f = Q.hit_first[r];
        for (uint32_t k = 0; k < c[q]; ++k) {
            HitRec h = Q.pool[f + k];
            const uint32_t* src = Q.ops_pool + h.ops_off;
            for (uint32_t i = 0; i < h.n_ops; ++i) Q.ops_out[ob + i] = src[i];
            h.ops_off = (uint32_t)ob;
            Q.hits_out[hb + k] = h;
            ob += h.n_ops;
        }
        hb += c[q];
    }
}

}  // namespace

// ======================================================================================================================
// host side
// ======================================================================================================================
struct mapad_index {
    host::Index ix;
};

namespace {

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            std::fprintf(stderr, "mapad_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MAPAD_ERR_DEVICE;                                                                        \
        }                                                                                                   \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n, bool exact = false) {
        if (n <= cap) return MAPAD_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = exact ? n : n + n / 8 + 64;
        if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) { p = nullptr; return MAPAD_ERR_NOMEM; }
        cap = want;
        return MAPAD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// bytes of "near" data per read slot: heap top (32 physical slots), 2 bytes + 4 bytes per read position
// (+ 64 bytes for the payload cache of heap slots 1 and 2 where the kernel keeps one: quads with their near data in LDS)
// MAPAD_NEAR_PAD=1 (round 6): padded to an ODD number of 16-byte units.  The read slots of a wavefront run in lockstep and ask for the same slot-relative address at
// once (heap slot 1, the D array's entry j, ...): with the 50 bp stride of 832 bytes = 208 dwords = 16 (mod 32 banks) the sixteen quads of a wavefront fall onto two
// groups of four banks — an 8-way conflict on every 16-byte read of a near level —; an odd number of 16-byte units spreads them over all 32 banks.  Measured: no
// difference on C2 / C3 / C4 (profiles/r06/ab_lds_pad_and_uniform_stores.txt) — the LDS is idle 95 % of the time and its reads are not what the chain waits for.  Off.
uint32_t near_bytes(uint32_t lmax, uint32_t top = kTop) {
    uint32_t b = (top + 1) * 8 + ((2 * lmax + 15) & ~15u) + ((4 * lmax + 15) & ~15u);
    static const bool pad = [] { const char* e = std::getenv("MAPAD_NEAR_PAD"); return e && e[0] == '1'; }();
    if (pad && ((b / 16) & 1u) == 0) b += 16;
    return b;
}
constexpr uint32_t kMaxLdsReadLen = 256;  // longer reads keep their near data in the HBM arena instead of LDS

// bytes of an arena's heap area for `heap_cap` logical entries: the physical entries the layout needs (heap_core.hpp: HeapLayout — subtree blocks take a third more than
// the implicit array) + slack for the vector loads past the end.  kTop's layout serves every kernel of a build (pairs, MAPAD_KTOP2 = 31, need no more: their
// blocks start earlier).
uint64_t heap_bytes(uint32_t heap_cap) {
    return ((uint64_t)std::max(HeapLayout<kTop>::phys_end(heap_cap), HeapLayout<MAPAD_KTOP2>::phys_end(heap_cap)) + 16) * sizeof(HeapEntry);
}
ArenaPool make_pool_layout(uint32_t heap_cap, uint32_t node_cap, uint32_t hit_ops_cap, uint32_t lmax) {
    ArenaPool ap{};
    auto align = [](uint64_t x) { return (x + 127) & ~127ull; };
    ap.heap_cap = heap_cap; ap.node_cap = node_cap; ap.hit_ops_cap = hit_ops_cap;
    uint64_t o = align(heap_bytes(heap_cap));
    ap.off_nodes = o; o = align(o + (uint64_t)node_cap * sizeof(Node));
    ap.off_hits = o; o = align(o + (uint64_t)kMaxHits * sizeof(HitRec));
    ap.off_hit_ops = o; o = align(o + (uint64_t)hit_ops_cap * 4);
    ap.off_scratch = o; o = align(o + 2ull * (lmax + 1) * 2);
    ap.off_near = o; o = align(o + near_bytes(lmax));
    ap.stride = o;
    return ap;
}

}  // namespace

namespace {
// Page-locked host memory for everything that crosses PCIe (results out, staged inputs in): DMA at link speed instead of the driver's
// staged copies of pageable memory.  Pinning is slow (the pages are locked one by one), so blocks are recycled: power-of-two sizes, a few
// kept per size, process-wide (results may outlive the context they came from).
class PinnedPool {
public:
    static PinnedPool& instance() { static PinnedPool p; return p; }
    void* take(size_t bytes, size_t& got) {
        size_t cap = 1 << 16;
        while (cap < bytes) cap <<= 1;
        got = cap;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto& fl = free_[cap];
            if (!fl.empty()) { void* p = fl.back(); fl.pop_back(); return p; }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) return nullptr;
        return p;
    }
    void give(void* p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto& fl = free_[cap];
            if (fl.size() < 6) { fl.push_back(p); return; }
        }
        (void)hipHostFree(p);
    }
private:
    std::mutex mu_;
    std::map<size_t, std::vector<void*>> free_;
};
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0, cap_bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    bool resize(size_t count) {
        if (count * sizeof(T) > cap_bytes) {
            PinnedPool::instance().give(p, cap_bytes);
            p = (T*)PinnedPool::instance().take(std::max<size_t>(count * sizeof(T), 1), cap_bytes);
            if (!p) { cap_bytes = 0; n = 0; return false; }
        }
        n = count;
        return true;
    }
    T* data() { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    ~PinnedBuf() { PinnedPool::instance().give(p, cap_bytes); }
};

// Host memory the GPU and the host both see while a kernel runs (the host tail's hand-over ring and its control word): page-locked AND coherent — fine-grained,
// not cached in the GPU's L2 —, which plain hipHostMalloc memory is not unless HIP_HOST_COHERENT=1.  Not pooled: one per batch slot, grown when needed.
struct CoherentBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    CoherentBuf() = default;
    CoherentBuf(const CoherentBuf&) = delete;
    CoherentBuf& operator=(const CoherentBuf&) = delete;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = 1 << 16;
        while (want < bytes) want <<= 1;
        void* q = nullptr;
        if (hipHostMalloc(&q, want, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
        p = (uint8_t*)q; cap = want;
        return true;
    }
    ~CoherentBuf() { if (p) (void)hipHostFree(p); }
};

}  // namespace

namespace {
// clears `bytes` bytes (a multiple of 4) at p on stream st with one-wavefront blocks (zero_words_kernel)
int zero_async(hipStream_t st, void* p, size_t bytes) {
    const uint64_t n = bytes / 4;
    if (!n) return MAPAD_OK;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(64), 0, st, (uint32_t*)p, n);
    HIP_TRY(hipGetLastError());
    return MAPAD_OK;
}
// n words at d_src -> h[word_off ...] on stream st (publish_words_kernel), then waits for the stream
int read_back_words(hipStream_t st, const void* d_src, PinnedBuf<uint32_t>& h, size_t word_off, uint32_t n_words) {
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)d_src, h.data() + word_off, n_words);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return MAPAD_OK;
}
}  // namespace

// Everything one batch in flight owns: its stream, result and scratch buffers, per-read-slot base arenas.  A context keeps `depth` of
// them so that the serial tail of batch k (its few heaviest reads) runs beside the bulk of batch k + 1; the size-class pools the read
// slots grow into are shared (arenas are claimed through owner words, whichever launch asks).
constexpr int kMaxDepth = 16;
struct BatchSlot {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // around D arrays + ordering / growable search stages / full-limit stage
    hipEvent_t ev_in = nullptr;                                // the caller's stream at submission: inputs are ready behind it
    DevBuf<uint8_t> d_seqs, d_quals;
    DevBuf<uint64_t> d_offsets;
    DevBuf<float> d_darr, d_dscratch;
    DevBuf<ReadCounters> d_counters;
    DevBuf<uint32_t> d_status, d_hit_count, d_hit_first, d_ops, d_cursors, d_overflow, d_sort_key, d_key_hist, d_order;
    DevBuf<HitRec> d_hits;
    DevBuf<HeavyItem> d_heavy;
    // read-ordered results (compact_* kernels)
    DevBuf<uint64_t> d_c_hit_begin, d_c_ops_begin;
    DevBuf<unsigned long long> d_c_tiles;
    DevBuf<HitRec> d_c_hits;
    DevBuf<uint32_t> d_c_ops;
    uint64_t c_n_hits = 0, c_n_ops = 0;
    bool compacted = false;
    // the batch
    BatchDev last{};
    uint64_t last_total_bases = 0;
    uint32_t last_lmax = 0;
    uint32_t launch_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ev_valid = false, timed = true;  // timed: its event times are already in the context's history
    uint64_t gen = 0;                     // process-wide serial number of this slot's latest launch: a fetched result knows whether the slot still holds it
    // host tail of the slot's latest launch (host_tail.hpp)
    CoherentBuf tail_ring;  // [64-byte header: control word][records]
    uint32_t tail_ring_stride = 0, tail_ring_gen = 0;  // record size the ring's `ready` words were last cleared for; number of the slot's latest launch in this ring (TailRecord::ready)
    const uint8_t* tail_ring_base = nullptr;
    std::shared_ptr<std::atomic<bool>> tail_launched;  // set once the launch's end event has been recorded (TailBatch::launch_done)
    bool tail_failed = false;  // the host tail of this slot's batch could not be merged: the batch's collect keeps failing (reads handed over would otherwise come back unmapped)
    std::shared_ptr<host::TailBatch> tail;  // set while the launch's handed-over reads have not been merged into its pools
    DevBuf<uint8_t> d_tail_up;
    // Page-locked landing place of this slot's small device-to-host copies (cursors, a batch's base count).  A copy into pageable memory is staged by a copy
    // KERNEL, and beside a search launch that kernel waits for wave slots the persistent wavefronts do not give up: rocprofv3 showed one such copy of 120 bytes
    // taking 3.0 s, and every second search launch of the C4 bench starting 135 ms late behind it.  Page-locked destinations go through the SDMA engines.
    PinnedBuf<uint32_t> h_small;
    PinnedBuf<uint8_t> h_tail_stage;  // the host tail's results on their way into the pools (page-locked for the same reason)
    // record fields of the slot's batch on the device (mapad_records_device: what the multi-GPU gather sends)
    DevBuf<CoordRec> d_rec_coords;
    DevBuf<DevRecord> d_rec_out;
    DevBuf<char> d_rec_text;
    DevBuf<float> d_rec_pairs;
    uint64_t tail_info[16] = {};  // reads, pops on the GPU before the hand-over, pops on the host, host wall microseconds, threads, budget, host E_search, N_push, N_node, host thread microseconds,
                                  // [10] records seen while the launch was running, [11] reads handed over on a dry arena class, [12] ... instead of the full-limit stage, [13] smallest class that hands over

    void release() {
        d_seqs.release(); d_quals.release(); d_offsets.release(); d_darr.release(); d_dscratch.release(); d_counters.release(); d_status.release(); d_hit_count.release();
        d_hit_first.release(); d_ops.release(); d_cursors.release(); d_overflow.release(); d_sort_key.release(); d_key_hist.release(); d_order.release();
        d_hits.release(); d_heavy.release();
        d_c_hit_begin.release(); d_c_ops_begin.release(); d_c_tiles.release(); d_c_hits.release(); d_c_ops.release();
        if (tail) { host::tail_cancel(tail); tail.reset(); }
        d_tail_up.release();
        d_rec_coords.release(); d_rec_out.release(); d_rec_text.release(); d_rec_pairs.release();
        for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
        if (own_stream && stream) (void)hipStreamDestroy(stream);
        stream = nullptr; own_stream = false;
    }
};

struct mapad_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // the caller's stream: inputs are ordered behind it, results are consumed on it
    mapad_params_t params{};
    const mapad_index* index = nullptr;
    host::HostTables tables;
    bool tables_dirty = true;
    std::shared_ptr<const host::HostTables> tables_snap;  // the tables as uploaded last: what the host tail's threads read (`tables` may grow under them)
    uint32_t tail_pops = 0;                                // pop budget of a read on the GPU (0: no host tail)
    // index + tables on the device
    DevBuf<uint64_t> d_blocks;
    DevBuf<float> d_sdm, d_thr;
    DevBuf<int32_t> d_base;
    DevIndex dix{};
    DevParams dprm{};
    bool fetch_d = true;
    // batches in flight
    BatchSlot bs[kMaxDepth];
    int depth = 1, cur = 0, view = 0;  // cur: slot of the most recent batch; view: slot the result accessors read (cur unless selected otherwise)
    hipEvent_t ev_ref = nullptr;       // start of the first launch of the current timing history
    std::vector<float> history;        // 4 floats per finished launch: ms from ev_ref to its ev[0..3]
    // arenas
    ArenaPool pool[kTiers] = {};
    uint32_t slots[kTiers] = {0, 0}, arena_lmax = 0;  // read slots of a tier = its wave sets x reads per wavefront
    uint32_t resident_waves = 0;                      // wavefronts of the search kernel the chip holds at once
    uint32_t heavy_cap = 64;                          // suspended reads a batch can have at most (= grown arenas)
    uint64_t arena_reads = 0;
    DevBuf<uint8_t> d_arena[kTiers];
    DevBuf<uint32_t> d_set_owner[kTiers];
    DevBuf<uint8_t> d_class[kClasses];
    DevBuf<uint32_t> d_owner[kClasses];
    DevBuf<GrowPools> d_grow;
    GrowPools grow{};
    // SA locate
    DevBuf<uint64_t> d_sa, d_xc, d_rows, d_pos, d_contigs, d_r_begin;
    DevBuf<HitRec> d_r_hits;
    DevBuf<uint32_t> d_r_ops;
    DevBuf<CoordRec> d_r_out;
    // record text on the device (text_kernel)
    DevBuf<uint64_t> d_os_pos;
    DevBuf<uint8_t> d_os_sym, d_names;
    DevBuf<uint32_t> d_name_off;
    DevBuf<DevRecord> d_t_out;
    DevBuf<char> d_t_text;
    DevBuf<float> d_t_pairs;
    DevBuf<unsigned long long> d_t_cur;
    PinnedBuf<unsigned long long> h_t_used;  // page-locked landing place of the text kernel's two cursors (see BatchSlot::h_small)
    uint64_t n_os = 0;
    DevBuf<unsigned long long> d_steps;
    bool sa_uploaded = false;
    hipEvent_t lev[2] = {nullptr, nullptr};
    unsigned long long last_locate_steps = 0;
    uint64_t last_locate_rows = 0;
    int lpr = 4;  // lanes per read in the search kernel (MAPAD_LANES_PER_READ = 4 | 1)
    int n_cu = 256;
    int reserved_cus = 0;  // CUs the batch slots' streams leave free (create_slot_stream)
    hipStream_t tail_stream = nullptr;  // the host tail's workers copy handed-over searches out of grown arenas on it (fetch_tail_state, one at a time); drop_tail releases abandoned arenas on it
    std::mutex tail_mu;
    uint64_t counter_sums[6] = {0, 0, 0, 0, 0, 0};
    DevBuf<unsigned long long> d_prof;  // -DMAPAD_PROFILE_SECTIONS builds

    ~mapad_ctx() {
        (void)hipSetDevice(device);
        for (auto& b : bs) { if (b.stream) (void)hipStreamSynchronize(b.stream); b.release(); }  // (release() waits for the host tail's workers of the slot's batch)
        if (tail_stream) { (void)hipStreamDestroy(tail_stream); tail_stream = nullptr; }
        d_blocks.release(); d_sdm.release(); d_thr.release(); d_base.release();
        for (auto& a : d_class) a.release();
        for (auto& a : d_owner) a.release();
        for (auto& a : d_arena) a.release();
        for (auto& a : d_set_owner) a.release();
        d_grow.release();
        d_sa.release(); d_xc.release(); d_rows.release(); d_pos.release(); d_steps.release();
        d_contigs.release(); d_r_begin.release(); d_r_hits.release(); d_r_ops.release(); d_r_out.release();
        d_os_pos.release(); d_os_sym.release(); d_names.release(); d_name_off.release(); d_t_out.release(); d_t_text.release(); d_t_pairs.release(); d_t_cur.release();
        for (auto& e : lev) if (e) (void)hipEventDestroy(e);
        if (ev_ref) (void)hipEventDestroy(ev_ref);
    }
};

namespace {

int sync_all_slots(mapad_ctx* c) {
    for (auto& b : c->bs) if (b.ev_valid) HIP_TRY(hipStreamSynchronize(b.stream));
    return MAPAD_OK;
}

int merge_tail(mapad_ctx* c, BatchSlot& S, uint32_t* cur);  // (below)
// Before the arenas are laid out anew: the launches of all slots have ended (sync_all_slots), but reads they handed over WITH their state (TailState) keep their
// grown arenas until a host worker has copied them (fetch_tail_state, which reads c->grow without a lock) — and at depth >= 2 another slot's batch may still be
// uncollected.  Their host tails are finished and merged here, as the batch's collect would do it later (which then finds nothing left to merge).
int finish_tails(mapad_ctx* c) {
    for (auto& b : c->bs) {
        if (!b.tail || !b.ev_valid) continue;
        uint32_t cur[CUR_COUNT] = {0};
        if (!b.h_small.resize(CUR_COUNT + 8)) return MAPAD_ERR_NOMEM;
        int rc = read_back_words(b.stream, b.last.cursors, b.h_small, 0, CUR_COUNT);
        if (rc) return rc;
        std::memcpy(cur, b.h_small.data(), sizeof cur);
        if ((rc = merge_tail(c, b, cur))) return rc;
    }
    return MAPAD_OK;
}

int upload_tables(mapad_ctx* c) {
    if (!c->tables_dirty) return MAPAD_OK;
    int rc;
    if ((rc = sync_all_slots(c))) return rc;  // launches in flight read the old tables
    if ((rc = c->d_sdm.ensure(std::max<size_t>(c->tables.sdm.size(), 4)))) return rc;
    if ((rc = c->d_base.ensure(c->tables.table_base.size()))) return rc;
    if ((rc = c->d_thr.ensure(c->tables.reject_thr.size()))) return rc;
    if (!c->tables.sdm.empty()) HIP_TRY(hipMemcpyAsync(c->d_sdm.p, c->tables.sdm.data(), c->tables.sdm.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_base.p, c->tables.table_base.data(), c->tables.table_base.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_thr.p, c->tables.reject_thr.data(), c->tables.reject_thr.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // host vectors may be re-allocated by the next add_length
    const mapad_params_t& p = c->params;
    DevParams& d = c->dprm;
    d.sdm_table = c->d_sdm.p; d.table_base = c->d_base.p; d.reject_thr = c->d_thr.p;
    d.nq = c->tables.nq; d.bound_kind = p.bound_kind; d.cutoff = p.cutoff; d.repr_mm = c->tables.repr_mm;
    d.gap_open = p.penalty_gap_open; d.gap_extend = p.penalty_gap_extend;
    d.gap_dist_ends = p.gap_dist_ends; d.max_num_gaps_open = p.max_num_gaps_open;
    d.start_at_end = p.model_kind == MAPAD_MODEL_SIMPLE_ADNA;
    d.stack_limit_abort = p.stack_limit_abort;
    d.stack_limit = p.stack_limit ? p.stack_limit : 2000000u;
    d.edit_tree_limit = p.edit_tree_limit ? p.edit_tree_limit : 10000000u;
    c->tables_snap = c->tail_pops ? std::make_shared<const host::HostTables>(c->tables) : nullptr;
    c->tables_dirty = false;
    return MAPAD_OK;
}

// Every batch in flight has its own HIP stream.  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; a launch
// queued behind another batch's long-running kernel on the same hardware queue then waits for it (C2, 3 in flight: 4.97 -> 5.10 M reads/s with 12
// queues; 4 and 6 in flight: 4.57 / 4.53 -> 4.98 / 5.04 M).  The variable is read at the first HIP call of the process, so it is set when the library is loaded, unless the
// user has set it.
__attribute__((constructor)) void mapad_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* e = std::getenv(name);
    return e && e[0] ? (uint32_t)std::strtoul(e, nullptr, 10) : dflt;
}

// Pass 0: one base arena per read slot (MAPAD_TIER0_NODES nodes) plus size-class pools the slots grow into (x4 per class).
// Pass 1: arenas with the reference's full limits (STACK_LIMIT + 9 frames, EDIT_TREE_LIMIT + 9 nodes: mapping.rs:52-54,147-148) for
// the reads pass 0 could not finish (a size-class pool ran dry).  Semantic limits are the same everywhere.
int ensure_arenas(mapad_ctx* c, BatchSlot& S, uint32_t lmax, uint64_t n_reads) {
    int rc;
    if (c->pool[0].stride && lmax <= c->arena_lmax && n_reads <= c->arena_reads) return MAPAD_OK;  // layouts and pools stand
    if ((rc = sync_all_slots(c))) return rc;  // the layouts change: nothing may be in flight
    if ((rc = finish_tails(c))) return rc;    // ... nor may a host worker still come for a grown arena
    for (auto& a : c->d_arena) a.release();
    n_reads = std::max<uint64_t>(n_reads, c->arena_reads);  // pools never shrink; a batch cannot use more arenas than it has reads
    { const uint32_t l = env_u32("MAPAD_LANES_PER_READ", 4); c->lpr = l == 1 ? 1 : l == 2 ? 2 : 4; }
    const uint32_t lm = std::max<uint32_t>(std::max<uint32_t>(lmax, c->arena_lmax), 128);
    const uint64_t stack_cap = (uint64_t)c->dprm.stack_limit + 10, tree_cap = (uint64_t)c->dprm.edit_tree_limit + 10;
    const uint32_t hit_ops_cap = kMaxHits * (lm + 32);
    const uint32_t rpw = 64 / c->lpr;
    const uint64_t need_waves = (std::max<uint64_t>(n_reads, 1) + rpw - 1) / rpw;
    uint32_t n_sets[kTiers];
    {   // pass 0 base arenas: one set per wavefront the chip can hold (+ a third: the probe of a late wavefront stays short), for all batches in flight
        // 16 384 nodes: 7 K instead of 17 K arena migrations per million C2 reads, +2-3 % reads/s over 8192 (C2, C3), +6 % on the C5 read mix.
        // Large genomes (3 Gbp: spurious matches survive longer, 2.8 % of the 50 bp reads outgrow 16 Ki nodes against 0.75 % at 48 Mbp): 65 536 nodes —
        // 280 K -> 30 K migrations per 10 M C4 reads, +9 % reads/s (24 / 32 / 48 Ki: +1 / +5 / +5.5 %; C2 +1 %, C3 +0 %) for 172 GB of HBM, if that much is free.
        uint32_t nodes = env_u32("MAPAD_TIER0_NODES", c->dix.n >= (1ull << 31) ? 65536 : 16384);
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                while (nodes > 16384 && (uint64_t)c->n_cu * 16 * 64 / c->lpr * ((uint64_t)nodes * 32 + heap_bytes(nodes) + 16384) > free_b / 4 * 3) nodes /= 2;  // base arenas take at most three quarters of what is free
        }
        c->pool[0] = make_pool_layout((uint32_t)std::min<uint64_t>(nodes, stack_cap), (uint32_t)std::min<uint64_t>(nodes, tree_cap), hit_ops_cap, lm);
        const uint32_t cus_used = (uint32_t)(c->depth > 1 ? c->n_cu - c->reserved_cus : c->n_cu);  // (mapad_ctx_set_reserved_cus: the launches of a pipelined context leave CUs free)
        c->resident_waves = env_u32("MAPAD_TIER0_WAVES_PER_CU", c->lpr == 4 ? 4 * MAPAD_MIN_WAVES : 8) * cus_used;  // lanes-per-read 2: eight blocks of 32 read slots per CU
        const uint64_t per_xcd_full = ((uint64_t)c->resident_waves * 4 / 3 + 7) / 8;
        const uint64_t per_xcd_need = (need_waves * (uint64_t)std::min(c->depth, 4) + 7) / 8 + 4;  // small batches (tests): no more than they can use
        n_sets[0] = 8 * (uint32_t)std::max<uint64_t>(std::min(per_xcd_full, per_xcd_need), kPartitionMin / 8);
        // long reads (up to i16::MAX): hit staging, the bucket-sort scratch and the near data of a slot grow with the read length — ~90 bytes per base and slot;
        // such batches get as many wavefront sets as 64 GB hold (the wavefronts of a launch wait for a free set: acquire_set)
        const uint64_t set_bytes = (uint64_t)rpw * c->pool[0].stride, budget = 64ull << 30;
        if ((uint64_t)n_sets[0] * set_bytes > budget && lm > 1024) n_sets[0] = 8 * (uint32_t)std::max<uint64_t>(budget / set_bytes / 8, kPartitionMin / 8);
    }
    {   // last stage: full limits, one arena per heavy wavefront (owner word per arena, shared by all XCDs)
        c->pool[1] = make_pool_layout((uint32_t)stack_cap, (uint32_t)tree_cap, hit_ops_cap, lm);
#if defined(MAPAD_HEAVY_KERNEL)
        n_sets[1] = std::max<uint32_t>(env_u32("MAPAD_LAST_PASS_ARENAS", 16), 1);
#else
        n_sets[1] = std::max<uint32_t>((env_u32("MAPAD_LAST_PASS_ARENAS", 16) + 15) / 16, 1);  // sets of a quad wavefront's 16 read slots
#endif
    }
    for (int t = 0; t < kTiers; ++t) {
#if defined(MAPAD_HEAVY_KERNEL)
        c->slots[t] = t == 0 ? n_sets[t] * rpw : n_sets[t];
#else
        c->slots[t] = t == 0 ? n_sets[t] * rpw : n_sets[t] * 16;  // the full-limit stage runs quads whatever the lanes per read of the growable stages
#endif
        if ((rc = c->d_arena[t].ensure((size_t)c->slots[t] * c->pool[t].stride, true))) return rc;
        if ((rc = c->d_set_owner[t].ensure(n_sets[t]))) return rc;
        HIP_TRY(hipMemsetAsync(c->d_set_owner[t].p, 0, (size_t)n_sets[t] * 4, S.stream));
        c->pool[t].base = c->d_arena[t].p; c->pool[t].set_owner = c->d_set_owner[t].p; c->pool[t].n_sets = n_sets[t];
    }
    const uint64_t other_slots_bytes = 0;  // base arenas belong to the context, not to a batch
    // size classes: 2x steps; the last one holds the reference's full limits so that its owners never have to grow (no wait cycles)
    const uint32_t rs = (uint32_t)std::min<uint64_t>((uint64_t)c->resident_waves * rpw, need_waves * rpw);  // read slots that can be busy at once
    // The big classes are held for seconds by the few heaviest reads of every batch in flight (C5 read mix: 0.5 % of the reads need class 4
    // or more; with 1024 / 256 / 64 arenas a 1 M-read launch spent a third of its time waiting for them): HBM is there to be used.
    // (round 5: generous — the fit below trims the classes that hold the most bytes until the pools fit the HBM that is free, so the classes end up with equal shares
    // of it instead of these numbers' proportions; round 4: rs / 16, 4096, 1024, 256, 128 for classes 3-7)
    uint32_t counts[kClasses] = {rs / 2, rs / 4, rs / 8, rs / 8, 8192, 4096, 1024, 256, 64, 64};
    // ... in proportion to the reads in flight (1 in 64 may need class 4, ..., 1 in 16384 the full limits; a read at the reference's limits
    // holds a 400 MB arena for a minute, and only as many of those run at once as the last class has arenas), never fewer than 16
    const uint64_t in_flight = n_reads * (uint64_t)std::min(c->depth, 4);
    const uint64_t one_in[kClasses] = {1, 1, 1, 1, 64, 256, 1024, 4096, 8192, 16384};
    for (int k = 0; k < kClasses; ++k) counts[k] = std::max<uint32_t>(k >= 4 ? (uint32_t)std::min<uint64_t>(counts[k], in_flight / one_in[k]) : counts[k], 16);
    if (const char* e = std::getenv("MAPAD_CLASS_COUNTS")) {  // comma list, missing entries = 0
        for (int k = 0; k < kClasses; ++k) counts[k] = 0;
        int k = 0;
        for (const char* q = e; *q && k < kClasses; ++k) { counts[k] = (uint32_t)std::strtoul(q, const_cast<char**>(&q), 10); if (*q == ',') ++q; }
    }
    // reads are handed to heavy wavefronts only on request (MAPAD_HEAVY=1) and only in batches of reads up to 1 024 bp (a heavy wavefront keeps the read's position data in LDS)
#if defined(MAPAD_HEAVY_KERNEL)
    const bool heavy_possible = env_u32("MAPAD_HEAVY", 0) != 0 && lm <= 1024;
#else
    const bool heavy_possible = false;
#endif
    uint64_t nodes = std::min<uint64_t>(c->pool[0].node_cap, env_u32("MAPAD_CLASS_ANCHOR_NODES", 8192));  // the ladder 16 K, 32 K, ... does not move with the base arena; classes the base arena already covers get no arenas
    GrowPools& g = c->grow;
    for (int k = 0; k < kClasses; ++k) {
        nodes = k + 1 == kClasses ? tree_cap : nodes * 2;
        g.heap_cap[k] = (uint32_t)std::min<uint64_t>(nodes, stack_cap);
        g.node_cap[k] = (uint32_t)std::min<uint64_t>(nodes, tree_cap);
        auto align = [](uint64_t x) { return (x + 127) & ~127ull; };
        g.off_nodes[k] = align(heap_bytes(g.heap_cap[k]));
        g.off_hits[k] = align(g.off_nodes[k] + (uint64_t)g.node_cap[k] * sizeof(Node));  // hit staging of a suspended read (heavy_kernel.hpp): only if reads can be suspended
        g.off_hit_ops[k] = heavy_possible ? align(g.off_hits[k] + (uint64_t)kMaxHits * sizeof(HitRec)) : g.off_hits[k];
        g.off_scratch[k] = heavy_possible ? align(g.off_hit_ops[k] + (uint64_t)hit_ops_cap * 4) : g.off_hits[k];
        g.stride[k] = heavy_possible ? align(g.off_scratch[k] + 2ull * (lm + 1) * 2) : g.off_hits[k];
        // a class that is no bigger than the previous one (tiny semantic limits) is pointless: give it no arenas
        // ... nor is a class the base arena already covers (round 5: the test used to compare with the previous CLASS only, and a 3 Gbp context kept 18 000 arenas
        // of 32 Ki and 64 Ki nodes — 32 GB — that no read leaving a 64 Ki-node base arena can use)
        const bool useful = (g.node_cap[k] > (k ? g.node_cap[k - 1] : 0u) || g.heap_cap[k] > (k ? g.heap_cap[k - 1] : 0u)) &&
                            (g.node_cap[k] > c->pool[0].node_cap || g.heap_cap[k] > c->pool[0].heap_cap);
        g.count[k] = useful ? std::min<uint32_t>(counts[k], (1u << kGrownShift) - 1) : 0;
    }
    // With the host tail on, a read leaves the GPU at tail_pops pops (one to two nodes per pop): the classes beyond what such a read can fill would hold
    // HBM for nothing — they keep a handful of arenas (a read that does get there finds its class dry and asks for the host: DeviceGrow::acquire), and the fit
    // below gives their room to the classes in which the reads short of the budget queue (round 4, budget 2^19: 128 arenas of 1 M nodes for 3 000 reads).
    if (c->tail_pops && !std::getenv("MAPAD_CLASS_COUNTS")) {
        int k_budget = kClasses - 1;
        for (int k = 0; k < kClasses; ++k) if (g.node_cap[k] >= 2ull * c->tail_pops) { k_budget = k; break; }
        for (int k = k_budget + 1; k < kClasses; ++k) g.count[k] = std::min<uint32_t>(g.count[k], env_u32("MAPAD_BEYOND_BUDGET_ARENAS", 4));
    }
    {   // fit the pools into the HBM that is free (index, tables and base arenas are already there; keep room for the batch buffers)
        for (auto& a : c->d_class) a.release();
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // what the batch slots will ask for behind this: inputs, D arrays, per-read words, hit and edit-op pools and their read-ordered copies, record fields —
        // about 1.5 KB per read and slot (C4: 10 M reads x 2 slots = 30 GB; the round-4 fit left that much free by accident, the round-5 fit fills what it is given)
        const uint64_t slot_bytes = (uint64_t)std::min(c->depth, 4) * n_reads * 1536ull;
        const uint64_t reserve = std::min<uint64_t>(std::max<uint64_t>(16ull << 30, (8ull << 30) + slot_bytes), free_b / 3) + other_slots_bytes;
        uint64_t budget = free_b > reserve ? free_b - reserve : 0;
        // ... and no more than the reads in flight can make use of (128 KB of grown arenas per read in flight, at least 8 GB): a context for 250 000-read chunks or
        // for a test's few thousand reads must not take the whole device — a second context on the same GPU (`--devices 0,0`, two ranks on one GPU) needs its share
        budget = std::min<uint64_t>(budget, std::max<uint64_t>(8ull << 30, in_flight * (128ull << 10)));
        if (const uint32_t gb = env_u32("MAPAD_POOL_BUDGET_GB", 0)) budget = std::min<uint64_t>(budget, (uint64_t)gb << 30);
        for (;;) {  // halve the class that holds the most bytes until the pools fit: the classes end up with about the same share of HBM each, i.e. arena counts in
                    // inverse proportion to arena size — the shape of the reads' heavy tail (P(nodes > x) ~ 1 / x) — instead of all classes losing half at every turn
            uint64_t need = 0;
            int big = -1;
            for (int k = 0; k < kClasses; ++k) {
                need += (uint64_t)g.count[k] * g.stride[k];
                if (g.count[k] > 4 && (big < 0 || (uint64_t)g.count[k] * g.stride[k] > (uint64_t)g.count[big] * g.stride[big])) big = k;
            }
            if (need <= budget || big < 0) break;  // back-pressure copes with small pools; the allocation below reports a real shortage
            g.count[big] /= 2;
        }
    }
    for (int k = 0; k < kClasses; ++k) {
        if ((rc = c->d_class[k].ensure(std::max<size_t>((size_t)g.count[k] * g.stride[k], 128), true))) return rc;
        if ((rc = c->d_owner[k].ensure(std::max<size_t>(g.count[k], 1)))) return rc;
        HIP_TRY(hipMemsetAsync(c->d_owner[k].p, 0, std::max<size_t>(g.count[k], 1) * 4, S.stream));
        g.base[k] = c->d_class[k].p;
        g.owner[k] = c->d_owner[k].p;
    }
    {   // the set class (GrowPools): one wavefront's set of base arenas as a single grown arena — heap in front, nodes behind it, as in make_pool_layout
        const int k = kClasses;
        const uint64_t set_bytes = (uint64_t)rpw * c->pool[0].stride;
        auto align = [](uint64_t x) { return (x + 127) & ~127ull; };
        uint64_t nodes = (set_bytes - 4096) / (sizeof(HeapEntry) + sizeof(Node));
        nodes = std::min<uint64_t>(nodes & ~63ull, tree_cap);
        // (subtree blocks: the heap area of `nodes` entries is 4/3 of the implicit array's when its last level pair is full and up to 8/3 when it has just begun —
        //  heap_core.hpp: HeapLayout::phys_end —, so the capacity is stepped down until heap and nodes fit the set)
        while (nodes > 64 && align(heap_bytes((uint32_t)std::min<uint64_t>(nodes, stack_cap))) + nodes * sizeof(Node) > set_bytes) nodes = (nodes - std::max<uint64_t>(64, nodes / 64)) & ~63ull;
        g.heap_cap[k] = (uint32_t)std::min<uint64_t>(nodes, stack_cap);
        g.node_cap[k] = (uint32_t)nodes;
        g.off_nodes[k] = align(heap_bytes(g.heap_cap[k]));
        g.off_hits[k] = g.off_hit_ops[k] = g.off_scratch[k] = 0;
        g.stride[k] = set_bytes;
        g.base[k] = c->pool[0].base;
        g.owner[k] = c->pool[0].set_owner;
        // not with reads suspended to heavy wavefronts (their hit staging travels in the grown arena), not for sets smaller than the base arena's next class, and
        // MAPAD_SET_ARENAS=0 switches it off (A/B)
        const bool on = env_u32("MAPAD_SET_ARENAS", 1) != 0 && !heavy_possible && g.off_nodes[k] + (uint64_t)g.node_cap[k] * sizeof(Node) <= set_bytes && g.node_cap[k] >= 2 * c->pool[0].node_cap;
        g.count[k] = on ? c->pool[0].n_sets : 0;
        g.set_next_class = kClasses;
        for (int j = kClasses - 1; j >= 0; --j) if (g.node_cap[j] > g.node_cap[k] || g.heap_cap[j] > g.heap_cap[k]) g.set_next_class = (uint32_t)j;
    }
    g.max_waits = env_u32("MAPAD_MAX_WAITS", 64);
    g.hit_ops_cap = hit_ops_cap;
    g.heavy_fast = env_u32("MAPAD_HEAVY_FAST", 1);
    g.wide_copy_nodes = env_u32("MAPAD_WIDE_COPY_NODES", 2048);
    // MAPAD_HEAVY=1: a quad hands a read that grows into class MAPAD_HEAVY_MIN_CLASS or beyond to heavy_kernel.  Off by default: measured on MI355X, a lone
    // wavefront issues one instruction per 4-5 cycles whatever its type, a step is ~2 500 of them either way, and the wavefront-per-read step takes 5.7 us
    // per pop against the quad's 5.5 (heavy reads); the full-limit stage runs on heavy_kernel in any case.
    g.heavy_min_class = heavy_possible ? env_u32("MAPAD_HEAVY_MIN_CLASS", 0) : (uint32_t)kClasses;
    c->heavy_cap = 64;
    for (int k = 0; k < kClasses; ++k) c->heavy_cap += g.count[k];  // a suspended read holds a grown arena
    {   // suspended reads wait for the heavy stage with their arenas: they may take three quarters of the class most of them are in
        uint32_t first = 0;
        for (int k = (int)std::min<uint32_t>(g.heavy_min_class, kClasses - 1); k < kClasses; ++k) if (g.count[k]) { first = g.count[k]; break; }
        g.heavy_max_pending = env_u32("MAPAD_HEAVY_MAX_PENDING", std::max<uint32_t>(4, first / 4 * 3 / (uint32_t)std::min(c->depth, 4)));  // the pools are shared by the batches in flight
    }
    if ((rc = c->d_grow.ensure(1))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_grow.p, &c->grow, sizeof(GrowPools), hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    c->arena_lmax = lm;
    c->arena_reads = n_reads;
    return MAPAD_OK;
}

// time stamps of a finished launch -> the context's history (ms since ev_ref)
int record_times(mapad_ctx* c, BatchSlot& S) {
    if (S.timed || !S.ev_valid || !c->ev_ref) return MAPAD_OK;
    HIP_TRY(hipEventSynchronize(S.ev[3]));
    for (int i = 0; i < 4; ++i) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, c->ev_ref, S.ev[i])); c->history.push_back(ms); }
    S.timed = true;
    return MAPAD_OK;
}

// A batch slot's own stream.  With `reserved_cus` > 0 (mapad_ctx_set_reserved_cus; MAPAD_RESERVED_CUS) the stream carries a CU mask that leaves the last CUs of the
// device to others: a search launch fills every CU it may use with persistent wavefronts — 11 blocks per CU leave one wave slot of <= 176 VGPRs and 13 KB of LDS —
// and RCCL's transfer kernel (ncclDevKernel_Generic: 248-256 VGPRs, 37.6 KB of LDS, 256-512 threads per block) becomes resident only on a CU the search does not
// use (measured with a stand-in of that shape: profiles/r05/rccl_standin.txt).  One process per GPU with a gather beside the search (bench.py --gpus N) sets it.
int create_slot_stream(mapad_ctx* c, hipStream_t* out) {
    if (c->reserved_cus > 0 && c->reserved_cus < c->n_cu) {
        // Which CUs: workgroups are dealt to the eight XCDs in turn whatever the mask says, so the reserved CUs must be spread evenly over the XCDs — eight CUs
        // taken from ONE XCD leave it with a quarter fewer slots than its share of the persistent wavefronts, and the launch ends with a second round on that
        // XCD (measured: -12 % reads/s; profiles/r05/rccl_standin.txt).  MAPAD_RESERVED_CU_LAYOUT: "blocked" (default: bit i = CU i % 32 of XCD i / 32: the last
        // CUs of every 32-bit word) or "striped" (bit i = a CU of XCD i % 8: the last bits of the mask).
        std::vector<uint32_t> mask((size_t)(c->n_cu + 31) / 32, 0u);
        for (int cu = 0; cu < c->n_cu; ++cu) mask[(size_t)cu / 32] |= 1u << (cu % 32);
        const char* layout = std::getenv("MAPAD_RESERVED_CU_LAYOUT");
        const bool striped = layout && layout[0] == 's';
        const int n_xcd = 8, per_xcd = c->n_cu / n_xcd;
        for (int k = 0; k < c->reserved_cus; ++k) {
            const int xcd = k % n_xcd, nth = k / n_xcd;  // the nth-last CU of XCD xcd
            const int bit = striped ? (c->n_cu - 1 - k) : (xcd * per_xcd + per_xcd - 1 - nth);
            if (bit >= 0 && bit < c->n_cu) mask[(size_t)bit / 32] &= ~(1u << (bit % 32));
        }
        // (a CU-masked stream is created with default flags: it is a BLOCKING stream with respect to the NULL stream, unlike the hipStreamNonBlocking streams of the
        //  path below — a context that reserves CUs must be given a stream of its own (mapad_ctx_set_stream), or its batches serialise behind the NULL stream's work)
        HIP_TRY(hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()));
        return MAPAD_OK;
    }
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return MAPAD_OK;
}

void drop_tail(mapad_ctx* c, BatchSlot& S);  // (below, beside the host tail's other library-side pieces)
// Makes slot `k` ready for a new batch: its previous batch has finished, its stream exists and waits for the caller's stream.
int acquire_slot(mapad_ctx* c, int k) {
    BatchSlot& S = c->bs[k];
    if (S.ev_valid) { HIP_TRY(hipStreamSynchronize(S.stream)); int rc = record_times(c, S); if (rc) return rc; }
    drop_tail(c, S);  // the previous batch of this slot was never collected: its handed-over reads are dropped with it
    if (c->depth == 1) S.stream = c->stream;  // one batch at a time: everything runs on the caller's stream (own_stream stays false)
    else if (!S.stream) { const int rc_s = create_slot_stream(c, &S.stream); if (rc_s) return rc_s; S.own_stream = true; }
    if (c->depth > 1) {  // the slot's stream is non-blocking: order it behind whatever the caller has queued on its own stream (async uploads of the inputs)
        if (!S.ev_in) HIP_TRY(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(S.ev_in, c->stream));
        HIP_TRY(hipStreamWaitEvent(S.stream, S.ev_in, 0));
    }
    c->cur = k; c->view = k;
    return MAPAD_OK;
}

// log2 of the chunk inside which the reads of a batch are ordered by cost class (order_scan_kernel): 2^20, the scale of the reference's --batch_size; batches of
// several million reads (C4: 10 M) in chunks of 2^21 — measured at C4: 2^18 / 2^20 / 2^21 / 2^22 -> 2.31 / 2.43 / 2.56 / 2.24 M reads/s (bigger chunks keep the reads
// of a wavefront more alike, until so many expensive reads start together that the size-class pools run dry).
uint32_t order_shift_for(uint64_t n_reads) {
    return std::min<uint32_t>(std::max<uint32_t>(env_u32("MAPAD_ORDER_CHUNK_LOG2", n_reads >= (4u << 20) ? 21 : 20), 10), 31);
}

// Device buffers of a batch of n_reads reads / total_bases bases / reads up to lmax long in slot S (allocation only; hipMalloc may wait
// for running kernels, so a pipelined caller reserves every slot up front: mapad_ctx_reserve).
int ensure_batch_buffers(mapad_ctx* c, BatchSlot& S, uint64_t n_reads, uint64_t total_bases, uint32_t lmax, bool host_inputs) {
    int rc;
    if ((rc = upload_tables(c))) return rc;
    if ((rc = ensure_arenas(c, S, lmax, n_reads))) return rc;
    const size_t nr = std::max<uint64_t>(n_reads, 1);
    if (host_inputs) {
        if ((rc = S.d_seqs.ensure(std::max<uint64_t>(total_bases, 1)))) return rc;
        if ((rc = S.d_quals.ensure(std::max<uint64_t>(total_bases, 1)))) return rc;
        if ((rc = S.d_offsets.ensure(n_reads + 1))) return rc;
    }
    if ((rc = S.d_darr.ensure(std::max<uint64_t>(total_bases, 1)))) return rc;
    if ((rc = S.d_counters.ensure(nr))) return rc;
    if ((rc = S.d_status.ensure(nr))) return rc;
    if ((rc = S.d_hit_count.ensure(nr))) return rc;
    if ((rc = S.d_hit_first.ensure(nr))) return rc;
    {   // the restart list of the first stage must read as zeros before an entry is written (a quad of the same launch may look at a reserved entry first): cleared when
        // it is allocated, and every entry is cleared again by whoever consumes it — no clearing kernel per launch (one more kernel in front of every search cost
        // mapad-amd map 5 % at C4: its thousands of one-wavefront blocks queue for wave slots behind the searches in flight)
        const uint32_t* before = S.d_overflow.p;
        if ((rc = S.d_overflow.ensure(nr * kStages))) return rc;
        if (S.d_overflow.p != before) HIP_TRY(hipMemsetAsync(S.d_overflow.p, 0, S.d_overflow.cap * sizeof(uint32_t), S.stream));
    }
    const uint32_t order_shift = order_shift_for(nr);
    const uint32_t n_chunks = (uint32_t)((nr + (1ull << order_shift) - 1) >> order_shift);
    if (env_u32("MAPAD_ORDER", 1) != 0) {
        if ((rc = S.d_sort_key.ensure(nr))) return rc;
        if ((rc = S.d_order.ensure(nr))) return rc;
        if ((rc = S.d_key_hist.ensure((size_t)n_chunks * kKeyBins))) return rc;
    }
    if ((rc = S.d_cursors.ensure(CUR_COUNT))) return rc;
    if ((rc = S.d_heavy.ensure(2 * (size_t)c->heavy_cap))) return rc;
    // 2 hits per read on average + slack; MAPAD_HIT_POOL (test hook) starts smaller so that the retry of mapad_map_batch is exercised
    const size_t hits_cap = std::max(S.d_hits.cap, (size_t)env_u32("MAPAD_HIT_POOL", (uint32_t)std::min<size_t>(2 * nr + 1024, 0xFFFFFFFFu)));
    const size_t ops_cap = std::max(S.d_ops.cap, hits_cap * (size_t)(std::min<uint32_t>(lmax, 256) + 8));
    if ((rc = S.d_hits.ensure(hits_cap))) return rc;
    if ((rc = S.d_ops.ensure(ops_cap))) return rc;
    return MAPAD_OK;
}

// A worker of the host tail takes over a read WITH its search (host_tail.hpp: TailState): the physical heap entries that hold logical slots [0, heap_len) (heap_core.hpp:
// HeapLayout, the host's search step reads the same layout; the kernel has written levels 0-5 out of LDS into the shadow entries in front) and nodes [0, tree_entries) of the grown arena `grown` come over PCIe into the worker's arena, then the arena is
// released on the device (its owner word cleared by a one-wavefront kernel: fits beside the searches).  Called from worker threads, several at a time.  The arena is released whether or not the copies worked; false = map the read from scratch.
bool release_tail_arena(mapad_ctx* c, uint32_t grown, hipStream_t st) {
    const uint32_t cls = (grown >> kGrownShift) - 1, idx = grown & ((1u << kGrownShift) - 1);
    if (cls > (uint32_t)kClasses || idx >= c->grow.count[cls]) return false;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, c->grow.owner[cls] + idx, (uint64_t)1);
    return hipGetLastError() == hipSuccess;
}
// a stream of the calling thread's own on device `dev` (the host tail's workers live as long as the process: never destroyed), so that the copies of several
// workers travel side by side instead of queuing behind one another's synchronisation
hipStream_t thread_stream(int dev) {
    static thread_local hipStream_t streams[16] = {};
    if (dev < 0 || dev >= 16) return nullptr;
    if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) streams[dev] = nullptr;
    return streams[dev];
}
bool fetch_tail_state(mapad_ctx* c, uint32_t grown, uint32_t heap_len, uint32_t tree_entries, HeapEntry* heap_phys, Node* nodes) {
    if (hipSetDevice(c->device) != hipSuccess) return false;
    const uint32_t cls = (grown >> kGrownShift) - 1, idx = grown & ((1u << kGrownShift) - 1);
    if (cls > (uint32_t)kClasses || idx >= c->grow.count[cls]) return false;
    // All workers share the context's one copy stream, one fetch at a time.  MAPAD_TAIL_THREAD_STREAMS=1: a stream per worker, copies side by side — measured
    // slower on the same box (1 M reads of the C5 mix: 33.9-37.7 s against 32.8 s, profiles/r05/ab_tail_streams.txt): sixteen more streams share the process's few
    // hardware queues with the launches.
    std::unique_lock<std::mutex> g(c->tail_mu, std::defer_lock);
    hipStream_t st = nullptr;
    if (env_u32("MAPAD_TAIL_THREAD_STREAMS", 0) != 0) st = thread_stream(c->device);
    else {
        g.lock();
        if (!c->tail_stream && hipStreamCreateWithFlags(&c->tail_stream, hipStreamNonBlocking) != hipSuccess) return false;
        st = c->tail_stream;
    }
    if (!st) return false;
    const uint8_t* b = c->grow.base[cls] + (uint64_t)idx * c->grow.stride[cls];
    if (!heap_phys || !nodes) {  // the worker cannot take the state over (no room, no memory): the arena is released all the same, the read is mapped from scratch
        (void)release_tail_arena(c, grown, st);
        (void)hipStreamSynchronize(st);
        return false;
    }
    bool ok = heap_len < c->grow.heap_cap[cls] + 8 && tree_entries <= c->grow.node_cap[cls];
    ok = ok && hipMemcpyAsync(heap_phys, b, ((size_t)HeapLayout<kTop>::phys_end(heap_len) + 1) * sizeof(HeapEntry), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(nodes, b + c->grow.off_nodes[cls], (size_t)tree_entries * sizeof(Node), hipMemcpyDeviceToHost, st) == hipSuccess;
    const bool released = release_tail_arena(c, grown, st);
    ok = (hipStreamSynchronize(st) == hipSuccess) && ok && released;
    return ok;
}

// The batch of slot S is abandoned (its slot is reused, the pipeline depth changes, the context goes away) with its launch over: its handed-over reads are dropped,
// and the grown arenas that reads handed over WITH their state still hold — no worker has come for them — are released.
void drop_tail(mapad_ctx* c, BatchSlot& S) {
    if (!S.tail) return;
    std::shared_ptr<host::TailBatch> tb = S.tail;
    S.tail.reset();
    host::tail_cancel(tb);
    if (!tb->fetch_state || hipSetDevice(c->device) != hipSuccess) return;
    std::lock_guard<std::mutex> g(c->tail_mu);
    if (!c->tail_stream && hipStreamCreateWithFlags(&c->tail_stream, hipStreamNonBlocking) != hipSuccess) return;
    bool any = false;
    for (uint32_t k = 0; k < tb->cap; ++k) {
        const uint8_t* rec = tb->ring + (size_t)k * tb->stride;
        if (reinterpret_cast<const host::TailRecord*>(rec)->ready != tb->gen || tb->fetched[k]) continue;
        const host::TailState* ts = reinterpret_cast<const host::TailState*>(rec + host::tail_state_offset(tb->lmax));
        if (ts->grown) any = release_tail_arena(c, ts->grown, c->tail_stream) || any;
    }
    if (any) (void)hipStreamSynchronize(c->tail_stream);
}

// warm: an empty launch of the search kernels only (mapad_ctx_reserve): the first dispatch on a stream that needs scratch memory sets up the
// queue's scratch ring, which waits for kernels running on other queues — better paid before the pipeline starts.
int launch_batch(mapad_ctx* c, BatchSlot& S, const uint8_t* d_seqs, const uint8_t* d_quals, const uint64_t* d_offsets, uint64_t n_reads, uint64_t total_bases,
                 uint32_t lmax, bool warm = false) {
    int rc;
    if ((rc = ensure_batch_buffers(c, S, n_reads, total_bases, lmax, false))) return rc;
    const size_t nr = std::max<uint64_t>(n_reads, 1);
    const bool ordered = env_u32("MAPAD_ORDER", 1) != 0;
    const uint32_t order_shift = order_shift_for(nr);
    const uint32_t n_chunks = (uint32_t)((nr + (1ull << order_shift) - 1) >> order_shift);
    if (ordered && (rc = zero_async(S.stream, S.d_key_hist.p, (size_t)n_chunks * kKeyBins * 4))) return rc;
    if ((rc = zero_async(S.stream, S.d_cursors.p, CUR_COUNT * 4))) return rc;
    if ((rc = zero_async(S.stream, S.d_status.p, nr * 4))) return rc;
    BatchDev B{};
    B.seqs = d_seqs; B.quals = d_quals; B.offsets = d_offsets; B.n_reads = (uint32_t)n_reads;
    B.d_arrays = S.d_darr.p; B.counters = S.d_counters.p; B.status = S.d_status.p;
    B.hit_count = S.d_hit_count.p; B.hit_first = S.d_hit_first.p;
    B.hits_pool = S.d_hits.p; B.ops_pool = S.d_ops.p;
    B.hits_cap = (uint32_t)std::min<size_t>(S.d_hits.cap, 0xFFFFFFFFu); B.ops_cap = (uint32_t)std::min<size_t>(S.d_ops.cap, 0xFFFFFFFFu);
    B.cursors = S.d_cursors.p; B.overflow_list = S.d_overflow.p;
    B.sort_key = ordered ? S.d_sort_key.p : nullptr; B.key_hist = ordered ? S.d_key_hist.p : nullptr; B.order = ordered ? S.d_order.p : nullptr;
    B.order_shift = order_shift;
    B.heavy = S.d_heavy.p; B.heavy_cap = c->heavy_cap;
#if defined(MAPAD_PROFILE_SECTIONS)
    if ((rc = c->d_prof.ensure(2 * PROF_N + 64))) return rc;
    HIP_TRY(hipMemsetAsync(c->d_prof.p, 0, (2 * PROF_N + 64) * 8, S.stream));
    B.prof = c->d_prof.p;
#endif
    B.tail_ring = nullptr; B.tail_stride = 0; B.tail_cap = 0; B.tail_lmax = 0; B.tail_pops = 0xFFFFFFFFu;
    B.tail_ctl = nullptr; B.tail_backlog_max = 0; B.tail_min_class = (uint32_t)kClasses; B.tail_continue = 0; B.tail_gen = 1; B.tail_backlog_budget = 0;
    B.tail_pops_idle = 0xFFFFFFFFu; B.tail_backlog_idle = 0;
    drop_tail(c, S);
    for (auto& x : S.tail_info) x = 0;
    S.tail_failed = false;
    if (c->tail_pops && !warm && n_reads) {
        const uint32_t tl = std::max<uint32_t>(lmax, 1);
        // (records of long reads are big — 6 bytes per base: the ring never takes more than 1 GB of page-locked memory; reads that find it full stay on the GPU)
        const uint32_t stride = host::tail_record_stride(tl);
        const uint32_t cap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n_reads, env_u32("MAPAD_TAIL_RING", 65536)), (1ull << 30) / stride));
        constexpr size_t kRingHeader = 64;  // the control word has a cache line of its own in front of the records
        if (!S.tail_ring.ensure(kRingHeader + (size_t)cap * stride)) return MAPAD_ERR_NOMEM;
        uint8_t* ring = S.tail_ring.p + kRingHeader;
        uint32_t* ctl = reinterpret_cast<uint32_t*>(S.tail_ring.p);
        ctl[0] = host::TailWorkers::instance().pending(); ctl[1] = 0;  // reads the workers have waiting or running (all launches); records of THIS launch the dispatcher has picked up
        // `ready` words: a record is ready when its word holds THIS launch's number (BatchDev::tail_gen), so nothing is cleared between launches (round 5 cleared
        // the records the previous launch could have set, and lost track of them when a small batch came between two large ones: ADVICE r5; page-locked coherent
        // memory is slow to write from the host).  After a (re)allocation or a change of the record size — the words then sit elsewhere, and what lies there is old
        // payload — every record the buffer can hold is cleared once and the numbering starts again.
        if (S.tail_ring_stride != stride || S.tail_ring_base != S.tail_ring.p) {
            const size_t holds = (S.tail_ring.cap - kRingHeader) / stride;
            for (size_t k = 0; k < holds; ++k) reinterpret_cast<host::TailRecord*>(ring + k * stride)->ready = 0;
            S.tail_ring_stride = stride; S.tail_ring_base = S.tail_ring.p; S.tail_ring_gen = 0;
        }
        if (++S.tail_ring_gen == 0) {  // (wrapped: 2^32 launches of one slot)
            const size_t holds = (S.tail_ring.cap - kRingHeader) / stride;
            for (size_t k = 0; k < holds; ++k) reinterpret_cast<host::TailRecord*>(ring + k * stride)->ready = 0;
            S.tail_ring_gen = 1;
        }
        auto tb = std::make_shared<host::TailBatch>();
        tb->ix = c->index->ix.view();
        tb->tables = c->tables_snap;
        tb->P = c->dprm;
        tb->P.sdm_table = tb->tables->sdm.data(); tb->P.table_base = tb->tables->table_base.data(); tb->P.reject_thr = tb->tables->reject_thr.data();
        tb->ring = ring; tb->stride = stride; tb->cap = cap; tb->lmax = tl; tb->ctl = ctl; tb->gen = S.tail_ring_gen;
        {   // "has this launch ended?" for the dispatcher's count of hand-overs that arrive during the launch: the event is recorded at the end of this function
            auto launched = std::make_shared<std::atomic<bool>>(false);
            S.tail_launched = launched;
            hipEvent_t* evp = &S.ev[3];
            tb->launch_done = [launched, evp]() { return launched->load(std::memory_order_acquire) && hipEventQuery(*evp) != hipErrorNotReady; };
        }
        // continuation: a read handed over with its state (TailState) — a worker copies its heap and nodes out of the grown arena and releases the arena
        B.tail_continue = env_u32("MAPAD_TAIL_CONTINUE", 1) && c->grow.heavy_min_class >= (uint32_t)kClasses ? 1u : 0u;
        tb->fetched.assign(cap, 0);
        if (B.tail_continue) tb->fetch_state = [c](uint32_t grown, uint32_t heap_len, uint32_t tree_entries, HeapEntry* heap_phys, Node* nodes) { return fetch_tail_state(c, grown, heap_len, tree_entries, heap_phys, nodes); };  // (heap_phys == nullptr: release only)
        host::tail_start(tb);
        S.tail = tb;
        B.tail_ring = ring; B.tail_stride = stride; B.tail_cap = cap; B.tail_lmax = tl; B.tail_pops = c->tail_pops; B.tail_gen = S.tail_ring_gen;
        // Hand-over on a dry arena class (DeviceGrow::acquire): classes from MAPAD_TAIL_MIN_CLASS up (default 4: the classes whose arena counts are absolute numbers,
        // not a share of the resident read slots), while the host has fewer than MAPAD_TAIL_BACKLOG reads waiting or running (default: half the worker threads — such a
        // read is handed over only when the reads past the pop budget leave threads idle.  1 M reads of the C5 mix on 3 Gbp, 16 threads: limit 32 / 16 / 8 ->
        // 43.0 / 39.3 / 37.7 s, the host's threads the last to finish each time: profiles/r05/c5_3gbp_1m_budgets.txt).
        B.tail_ctl = ctl;
        B.tail_min_class = env_u32("MAPAD_TAIL_MIN_CLASS", 4);
        B.tail_backlog_max = env_u32("MAPAD_TAIL_BACKLOG", std::max(1u, host::TailWorkers::instance().size() / 2));
        // Past the budget: a worker finishes such a read ~15 x faster than its quad would, so the host is the better place while fewer than ~15 reads per worker wait
        // in front of it; MAPAD_TAIL_BACKLOG_BUDGET (default 8 per worker; 0xFFFFFFFF = round 5's unconditional hand-over).
        B.tail_backlog_budget = env_u32("MAPAD_TAIL_BACKLOG_BUDGET", 8u * std::max(1u, host::TailWorkers::instance().size()));
        // Below the budget, from MAPAD_TAIL_POPS_IDLE pops on (0 = this tier is off): while a worker is idle — fewer reads waiting or running than MAPAD_TAIL_BACKLOG_IDLE
        // (default: the workers).  The budget is what a loaded host should be spared (a million pops are 5.8 s of a quad and 0.4 s of a worker); but until the first
        // reads of a launch reach it the workers have nothing to do, and afterwards they idle whenever the reads past the budget are few — while the launch waits for
        // reads that are a few hundred thousand pops from their end.
        const uint32_t idle_pops = env_u32("MAPAD_TAIL_POPS_IDLE", MAPAD_DEFAULT_TAIL_POPS_IDLE);
        B.tail_pops_idle = idle_pops ? std::min(idle_pops, c->tail_pops) : c->tail_pops;
        B.tail_backlog_idle = env_u32("MAPAD_TAIL_BACKLOG_IDLE", std::max(1u, host::TailWorkers::instance().size()));
    }
    S.last = B; S.last_total_bases = total_bases; S.last_lmax = lmax; S.compacted = false;
    // A process-wide launch number, not a per-slot count: a result of a destroyed context must not pass for the batch of a new context that happens to sit at
    // the same address with the same slot and launch count (mapad_hits_to_coords_gpu's device-resident shortcut compares this number).
    static std::atomic<uint64_t> launch_serial{0};
    S.gen = launch_serial.fetch_add(1, std::memory_order_relaxed) + 1;
    if (n_reads == 0 && !warm) return MAPAD_OK;
    const uint32_t lds_lmax = std::max<uint32_t>(lmax, 1);
    size_t lds_bytes = (size_t)16 * lds_lmax * sizeof(float);
    uint32_t grid_d = (uint32_t)std::min<uint64_t>(n_reads, (uint64_t)c->n_cu * 32);
    float* long_scratch = nullptr;
    if (lds_bytes > 48 * 1024) {  // long reads: the chains live in HBM (a quarter of the usual grid: such batches are rare and the buffer is 16 x lmax floats per block)
        grid_d = std::max<uint32_t>(1, std::min<uint32_t>(grid_d, (uint32_t)c->n_cu * 8));
        if ((rc = S.d_dscratch.ensure((size_t)grid_d * 16 * lds_lmax))) return rc;
        long_scratch = S.d_dscratch.p;
        lds_bytes = 0;
    }
    for (auto& e : S.ev) if (!e) HIP_TRY(hipEventCreate(&e));
    if (!warm) {
        if (!c->ev_ref) { HIP_TRY(hipEventCreate(&c->ev_ref)); HIP_TRY(hipEventRecord(c->ev_ref, S.stream)); c->history.clear(); }
        HIP_TRY(hipEventRecord(S.ev[0], S.stream));
        hipLaunchKernelGGL(darray_kernel, dim3(grid_d), dim3(64), lds_bytes, S.stream, c->dix, c->dprm, B, (int)lds_lmax, long_scratch);
        HIP_TRY(hipGetLastError());
    }
    if (ordered && !warm) {
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(64), 0, S.stream, S.d_key_hist.p, n_chunks, std::min<uint32_t>(lmax + 2, kKeyBins));
        hipLaunchKernelGGL(order_scatter_kernel, dim3((uint32_t)((n_reads + 63) / 64)), dim3(64), 0, S.stream, B);
        HIP_TRY(hipGetLastError());
    }
    if (!warm) HIP_TRY(hipEventRecord(S.ev[1], S.stream));
    const uint32_t rpw = 64 / c->lpr;  // reads per wavefront
    // near data in LDS (16 read slots per wavefront) unless the batch has very long reads or every lane owns a read
    const uint32_t near_lmax = std::max<uint32_t>(lmax, 1);
    // lanes-per-read 1: 64 read slots per wavefront, near data in LDS while it fits the 64 KB a launch may ask for without an opt-in
    const uint32_t near_top = c->lpr == 2 ? MAPAD_KTOP2 : kTop;
    const bool near_fits = c->lpr == 4 ? near_lmax <= kMaxLdsReadLen : c->lpr == 2 ? (size_t)near_bytes(near_lmax, near_top) * 32 <= 65536 : (size_t)near_bytes(near_lmax) * 64 <= 65536;
    const uint32_t near_stride = (near_fits && env_u32("MAPAD_NEAR_LDS", 1)) ? near_bytes(near_lmax, near_top) : 0;
    // LDS per search block: the near data of its read slots — padded, when twelve blocks would fit a CU, to what only eleven fit (see order_scatter_kernel)
    size_t lds = (size_t)near_stride * rpw;
    {
        const size_t cu_lds = 160 * 1024;
        if (lds) {
            const size_t fit = cu_lds / lds;
            // at least 12 KB of a CU's LDS stay free: one block fewer than fit if the blocks would leave less (quads: 12 x 13.3 KB -> 11; pairs at 50 bp: 8 x 18.4 KB leave 12.8 KB)
            size_t per_cu = cu_lds - fit * lds < (size_t)env_u32("MAPAD_LDS_FREE_KB", 12) * 1024 ? fit - 1 : fit;
            per_cu = std::min<size_t>(per_cu, env_u32("MAPAD_SEARCH_BLOCKS_PER_CU", 32));
            if (per_cu >= 1 && per_cu < fit) lds = (cu_lds / (per_cu + 1) + 256) & ~(size_t)255;
        }
    }
    const bool cont = c->dprm.bound_kind == BOUND_CONTINUOUS;
    const bool heavy_on = c->grow.heavy_min_class < (uint32_t)kClasses;
    (void)heavy_on;
#if defined(MAPAD_HEAVY_KERNEL)
#define MAPAD_LAUNCH(L, C, P, N)                                                                                                                                   \
    if (heavy_on && L == 4) hipLaunchKernelGGL((search_kernel<L, C, P, N, (L == 4)>), dim3(grid), dim3(64), lds, S.stream, c->dix, c->dprm, B, ap, c->d_grow.p, near_stride, near_lmax, stage); \
    else hipLaunchKernelGGL((search_kernel<L, C, P, N, false>), dim3(grid), dim3(64), lds, S.stream, c->dix, c->dprm, B, ap, c->d_grow.p, near_stride, near_lmax, stage)
#else
#define MAPAD_LAUNCH(L, C, P, N) hipLaunchKernelGGL((search_kernel<L, C, P, N, false>), dim3(grid), dim3(64), lds, S.stream, c->dix, c->dprm, B, ap, c->d_grow.p, near_stride, near_lmax, stage)
#endif
#define MAPAD_LAUNCH_PASS(P)                                                                                      \
    if (c->lpr == 2 && near_stride) { if (!cont) { MAPAD_LAUNCH(2, false, P, true); } else { MAPAD_LAUNCH(2, true, P, true); } }   \
    else if (c->lpr == 2) { if (!cont) { MAPAD_LAUNCH(2, false, P, false); } else { MAPAD_LAUNCH(2, true, P, false); } }          \
    else if (c->lpr == 4 && near_stride) { if (!cont) { MAPAD_LAUNCH(4, false, P, true); } else { MAPAD_LAUNCH(4, true, P, true); } }   \
    else if (c->lpr == 4) { if (!cont) { MAPAD_LAUNCH(4, false, P, false); } else { MAPAD_LAUNCH(4, true, P, false); } }          \
    else if (near_stride) { if (!cont) { MAPAD_LAUNCH(1, false, P, true); } else { MAPAD_LAUNCH(1, true, P, true); } }              \
    else { if (!cont) { MAPAD_LAUNCH(1, false, P, false); } else { MAPAD_LAUNCH(1, true, P, false); } }
    // (re-measured with the shared base arenas, C2 / C3 reads/s at 4, 6, 8, 12 wavefronts per CU for a launch that finds another one running:
    // 4.96 / 2.00 M, 4.86 / 2.25 M, 4.89 / 2.28 M, 4.71 / 2.23 M -> 8)
    // A launch alone on the chip fills it (16 wavefronts per CU at 128 VGPRs).  While another batch is still running, a launch takes half: two
    // bulks then share the chip all the time and a launch that is down to its tail does not hold the next one back (measured, C2, per 1 M reads:
    // 16 per CU x 2 batches in flight 273 ms, 8 x 2 247 ms, 8 x 3 252 ms; C3: 8 x 2 589 ms, 8 x 3 529 ms).
    bool others_running = false;
    for (auto& o : c->bs) if (&o != &S && o.ev_valid && hipEventQuery(o.ev[3]) == hipErrorNotReady) others_running = true;
    // Big batches (C4: 10 M reads, a 4 s launch whose tail is 3 % of it) gain nothing from a second search beside the first — two launches that
    // share the chip ran 6-8 % slower than one after the other — but their D arrays and ordering (4 % of a step) still run beside the previous
    // batch's search, in the room its launch leaves on every CU: the search of such a batch waits for the searches before it.
    const bool serialize = env_u32("MAPAD_SERIALIZE_SEARCH", n_reads >= (4u << 20) ? 1 : 0) != 0;
    if (serialize && others_running && !warm) {
        for (auto& o : c->bs) if (&o != &S && o.ev_valid) HIP_TRY(hipStreamWaitEvent(S.stream, o.ev[3], 0));
        others_running = false;  // this launch will have the chip to itself
    }
    const uint32_t full_waves = c->resident_waves, shared_waves = std::max<uint32_t>(1, std::min<uint32_t>(full_waves, env_u32("MAPAD_SHARED_WAVES_PER_CU", 8) * (uint32_t)(c->depth > 1 ? c->n_cu - c->reserved_cus : c->n_cu)));
    const uint32_t grid_s = warm ? 1u : (uint32_t)std::min<uint64_t>((n_reads + rpw - 1) / rpw, others_running ? shared_waves : full_waves);
#if defined(MAPAD_HEAVY_KERNEL)
    // heavy wavefronts: as many as the chip holds (LDS: heap levels 0-9 + the read's position data); a wavefront that finds no work exits at once
    const uint32_t heavy_lds = heavy_lds_bytes(near_lmax);
    const uint32_t heavy_per_cu = std::max<uint32_t>(1, std::min<uint32_t>(env_u32("MAPAD_HEAVY_WAVES_PER_CU", 8), (160u * 1024u) / heavy_lds));
    const uint32_t grid_h = warm ? 1u : std::min<uint32_t>(c->heavy_cap, heavy_per_cu * (uint32_t)c->n_cu);
#define MAPAD_LAUNCH_HEAVY(M, R, GRID, AP, TIER)                                                                                                                      \
    if (!cont) hipLaunchKernelGGL((heavy_kernel<false, M, R>), dim3(GRID), dim3(64), heavy_lds, S.stream, c->dix, c->dprm, B, AP, c->d_grow.p, near_lmax, TIER);   \
    else hipLaunchKernelGGL((heavy_kernel<true, M, R>), dim3(GRID), dim3(64), heavy_lds, S.stream, c->dix, c->dprm, B, AP, c->d_grow.p, near_lmax, TIER);
#endif
    for (int stage = 0; stage + 1 < kStages; ++stage) {  // Q0 + H0: every read; Q1 + H1: the reads that gave up waiting (normally none: the launches exit at once)
        const uint32_t grid = grid_s;
        const ArenaPool ap = c->pool[0];
        if (stage == 0) { MAPAD_LAUNCH_PASS(0) } else { MAPAD_LAUNCH_PASS(2) }  // PASS 2 = PASS 0 under its own symbol, so that profiles keep the passes apart
#if defined(MAPAD_HEAVY_KERNEL)
        if ((heavy_on || warm) && near_lmax <= kHeavyMaxLdsReadLen) { MAPAD_LAUNCH_HEAVY(0, true, grid_h, ap, stage) }
#endif
        HIP_TRY(hipGetLastError());
    }
    if (!warm) HIP_TRY(hipEventRecord(S.ev[2], S.stream));
#if defined(MAPAD_HEAVY_KERNEL)
    {   // leftovers with the reference's full limits: heavy wavefronts from scratch
        const uint32_t grid = warm ? 1u : (uint32_t)std::min<uint64_t>(n_reads, c->slots[1]);
        const ArenaPool ap = c->pool[1];
        if (near_lmax <= kHeavyMaxLdsReadLen) { MAPAD_LAUNCH_HEAVY(1, true, grid, ap, kStages - 1) } else { MAPAD_LAUNCH_HEAVY(1, false, grid, ap, kStages - 1) }
        HIP_TRY(hipGetLastError());
    }
#undef MAPAD_LAUNCH_HEAVY
#else
    {   // leftovers with the reference's full limits (only when the host tail is off or its ring was full): quads from scratch in the full-limit arenas, no growing (PASS 1)
        const uint32_t grid = warm ? 1u : (uint32_t)std::min<uint64_t>((n_reads + 15) / 16, c->pool[1].n_sets);
        const ArenaPool ap = c->pool[1];
        const int stage = kStages - 1;
        const uint32_t near_stride = (near_lmax <= kMaxLdsReadLen && env_u32("MAPAD_NEAR_LDS", 1)) ? near_bytes(near_lmax, kTop) : 0;
        const size_t lds = (size_t)near_stride * 16;
        if (near_stride) { if (!cont) { MAPAD_LAUNCH(4, false, 1, true); } else { MAPAD_LAUNCH(4, true, 1, true); } }
        else { if (!cont) { MAPAD_LAUNCH(4, false, 1, false); } else { MAPAD_LAUNCH(4, true, 1, false); } }
        HIP_TRY(hipGetLastError());
    }
#endif
#undef MAPAD_LAUNCH_PASS
#undef MAPAD_LAUNCH
    if (warm) return MAPAD_OK;
    HIP_TRY(hipEventRecord(S.ev[3], S.stream));
    if (S.tail_launched) S.tail_launched->store(true, std::memory_order_release);
    S.ev_valid = true; S.timed = false;
    S.launch_info[0] = grid_d; S.launch_info[1] = 64; S.launch_info[2] = (uint32_t)lds_bytes;
    S.launch_info[3] = grid_s; S.launch_info[4] = 64; S.launch_info[5] = c->slots[1];
    S.launch_info[6] = c->pool[0].node_cap; S.launch_info[7] = (uint32_t)(c->pool[0].stride >> 10);
    return MAPAD_OK;
}

// The launch of slot S has ended (`cur` = its cursors): waits for the host threads that map its handed-over reads (host_tail.hpp) and puts their results where
// the kernel would have put them — hits and edit tracks at the pools' cursors, per-read counts, status and event counters by a scatter kernel.
int merge_tail(mapad_ctx* c, BatchSlot& S, uint32_t* cur) {
    std::shared_ptr<host::TailBatch> tb = S.tail;
    S.tail.reset();
    S.tail_failed = true;  // until the results are on the device: a failure below leaves reads marked ST_TAIL, and the batch's collect must go on failing (compact_last)
    const bool ok = host::tail_finish(tb, cur[CUR_TAIL]);
    if (!ok) { std::fprintf(stderr, "mapad_amd: a read of the host tail could not be mapped (out of host memory?)\n"); return MAPAD_ERR_NOMEM; }
    std::vector<host::TailResult>& res = tb->results;
    S.tail_info[0] = res.size(); S.tail_info[1] = tb->gpu_pops; S.tail_info[2] = tb->host_pops;
    S.tail_info[3] = res.empty() ? 0 : (uint64_t)(std::max(tb->t_last - tb->t_first, 0.0) * 1e6); S.tail_info[4] = host::TailWorkers::instance().size(); S.tail_info[5] = c->tail_pops;
    for (const auto& r : res) { S.tail_info[6] += r.e_search - r.gpu_e_search; S.tail_info[7] += r.n_push - r.gpu_n_push; S.tail_info[8] += r.n_node - r.gpu_n_node; }  // what the host threads did themselves
    S.tail_info[9] = (uint64_t)(tb->host_thread_s * 1e6);
    S.tail_info[10] = tb->seen_live; S.tail_info[11] = cur[CUR_TAIL_DRY]; S.tail_info[12] = cur[CUR_TAIL_F]; S.tail_info[13] = S.last.tail_min_class | ((uint64_t)cur[CUR_TAIL_IDLE] << 32);
    S.tail_info[14] = tb->continued; S.tail_info[15] = cur[CUR_TAIL_STATE];
    if (res.empty()) { S.tail_failed = false; return MAPAD_OK; }
    std::sort(res.begin(), res.end(), [](const host::TailResult& a, const host::TailResult& b) { return a.read < b.read; });
    const BatchDev& B = S.last;
    uint64_t n_hits = 0, n_ops = 0;
    for (const auto& r : res) { n_hits += r.hits.size(); n_ops += r.ops.size(); }
    const uint64_t hbase = cur64(cur, CUR_HITS), obase = cur64(cur, CUR_OPS);
    auto set64 = [&](int k, uint64_t v) { cur[k] = (uint32_t)v; cur[k + 1] = (uint32_t)(v >> 32); };
    set64(CUR_HITS, hbase + n_hits); set64(CUR_OPS, obase + n_ops);
    const bool fits = hbase + n_hits <= B.hits_cap && obase + n_ops <= B.ops_cap;
    if (!fits) cur[CUR_POOL_OVF] = 1;  // the caller re-runs the batch with larger pools (mapad_map_batch), sized by these cursors
    if (!S.h_small.resize(CUR_COUNT + 8)) return MAPAD_ERR_NOMEM;
    std::memcpy(S.h_small.data(), cur, (CUR_POOL_OVF + 1) * 4);
    HIP_TRY(hipMemcpyAsync(B.cursors, S.h_small.data(), (CUR_POOL_OVF + 1) * 4, hipMemcpyHostToDevice, S.stream));
    if (!fits) { HIP_TRY(hipStreamSynchronize(S.stream)); S.tail_failed = false; return MAPAD_OK; }  // (the pool overflow is reported by the cursors; the batch is re-run)
    // staged in page-locked memory: [TailUp x reads][HitRec x n_hits][u32 x n_ops]
    const size_t off_hits = (res.size() * sizeof(TailUp) + 63) & ~(size_t)63, off_ops = (off_hits + n_hits * sizeof(HitRec) + 63) & ~(size_t)63;
    if (!S.h_tail_stage.resize(off_ops + n_ops * 4 + 64)) return MAPAD_ERR_NOMEM;
    TailUp* up = reinterpret_cast<TailUp*>(S.h_tail_stage.data());
    HitRec* hits = reinterpret_cast<HitRec*>(S.h_tail_stage.data() + off_hits);
    uint32_t* ops = reinterpret_cast<uint32_t*>(S.h_tail_stage.data() + off_ops);
    size_t k_up = 0, k_hits = 0, k_ops = 0;
    for (const auto& r : res) {
        up[k_up++] = TailUp{r.read, r.status, (uint32_t)r.hits.size(), (uint32_t)(hbase + k_hits), r.e_search, r.n_push, r.n_pop, r.n_node, r.n_hits};
        for (HitRec h : r.hits) { h.ops_off += (uint32_t)(obase + k_ops); hits[k_hits++] = h; }  // like finalize_read: offsets into the global op pool
        std::memcpy(ops + k_ops, r.ops.data(), r.ops.size() * 4);
        k_ops += r.ops.size();
    }
    int rc;
    if ((rc = S.d_tail_up.ensure(res.size() * sizeof(TailUp)))) return rc;
    if (n_hits) HIP_TRY(hipMemcpyAsync(B.hits_pool + hbase, hits, n_hits * sizeof(HitRec), hipMemcpyHostToDevice, S.stream));
    if (n_ops) HIP_TRY(hipMemcpyAsync(B.ops_pool + obase, ops, n_ops * 4, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.d_tail_up.p, up, res.size() * sizeof(TailUp), hipMemcpyHostToDevice, S.stream));
    hipLaunchKernelGGL(tail_scatter_kernel, dim3((uint32_t)((res.size() + 63) / 64)), dim3(64), 0, S.stream, B, (const TailUp*)S.d_tail_up.p, (uint32_t)res.size());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(S.stream));
    S.tail_failed = false;
    return MAPAD_OK;
}

// Lays the last batch's hits out in read order on the device (no-op if already done).  Reports pool overflow / kernel errors like the fetch.
int compact_last(mapad_ctx* c) {
    BatchSlot& S = c->bs[c->view];
    if (S.compacted) return MAPAD_OK;
    if (S.tail_failed) { std::fprintf(stderr, "mapad_amd: the host tail of this batch failed earlier; its results are incomplete\n"); return MAPAD_ERR_NOMEM; }
    HIP_TRY(hipStreamSynchronize(S.stream));
    const BatchDev& B = S.last;
    const uint64_t n = B.n_reads;
    uint32_t cur[CUR_COUNT] = {0};
    if (n) {
        if (!S.h_small.resize(CUR_COUNT + 8)) return MAPAD_ERR_NOMEM;
        { const int rc_p = read_back_words(S.stream, B.cursors, S.h_small, 0, CUR_COUNT); if (rc_p) return rc_p; }
        std::memcpy(cur, S.h_small.data(), sizeof cur);
    }
    if (S.tail) { const int rc_t = merge_tail(c, S, cur); if (rc_t) return rc_t; }
    if (cur[CUR_ERR] & ST_NO_TABLE) { std::fprintf(stderr, "mapad_amd: a read length had no score table (call mapad_ctx_prepare_lengths)\n"); return MAPAD_ERR_INVALID; }
    if (cur[CUR_ERR] & ST_ARENA_OVERFLOW) { std::fprintf(stderr, "mapad_amd: arena overflow in the large-arena pass\n"); return MAPAD_ERR_NOMEM; }
    if (cur[CUR_POOL_OVF]) return MAPAD_ERR_NOMEM;  // mapad_map_batch retries with larger pools
    S.c_n_hits = cur64(cur, CUR_HITS); S.c_n_ops = cur64(cur, CUR_OPS);
    int rc;
    const uint64_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if ((rc = S.d_c_hit_begin.ensure(n + 1))) return rc;
    if ((rc = S.d_c_ops_begin.ensure(n + 1))) return rc;
    if ((rc = S.d_c_tiles.ensure(2 * std::max<uint64_t>(n_tiles, 1)))) return rc;
    if ((rc = S.d_c_hits.ensure(std::max<uint64_t>(S.c_n_hits, 1)))) return rc;
    if ((rc = S.d_c_ops.ensure(std::max<uint64_t>(S.c_n_ops, 1)))) return rc;
    if (n == 0) { HIP_TRY(hipMemsetAsync(S.d_c_hit_begin.p, 0, 8, S.stream)); HIP_TRY(hipMemsetAsync(S.d_c_ops_begin.p, 0, 8, S.stream)); S.compacted = true; return MAPAD_OK; }
    CompactDev Q{B.hit_count, B.hit_first, B.hits_pool, B.ops_pool, n, S.d_c_hit_begin.p, S.d_c_ops_begin.p, S.d_c_tiles.p, S.d_c_tiles.p + n_tiles, S.d_c_hits.p, S.d_c_ops.p};
    hipLaunchKernelGGL(compact_sums_kernel, dim3((uint32_t)n_tiles), dim3(64), 0, S.stream, Q);
    hipLaunchKernelGGL(compact_scan_tiles_kernel, dim3(1), dim3(64), 0, S.stream, Q, n_tiles);
    hipLaunchKernelGGL(compact_move_kernel, dim3((uint32_t)n_tiles), dim3(64), 0, S.stream, Q);
    HIP_TRY(hipGetLastError());
    S.compacted = true;
    return MAPAD_OK;
}

struct HostResult {
    mapad_batch_result_t pub{};
    const void* owner = nullptr;  // the context and batch slot whose device buffers still hold this result in read order (while `gen` matches the slot's)
    int slot = -1;
    uint64_t gen = 0;
    PinnedBuf<uint64_t> hit_begin;
    PinnedBuf<mapad_hit_t> hits;
    PinnedBuf<uint32_t> ops, status;
    PinnedBuf<mapad_read_counters_t> counters;
    PinnedBuf<float> d_arrays;
};
static_assert(sizeof(mapad_hit_t) == sizeof(HitRec), "public hit record == device hit record");
static_assert(sizeof(mapad_read_counters_t) == sizeof(ReadCounters), "counter layout");
// The results this library has handed out and not yet seen freed.  A caller may also pass a mapad_batch_result_t it has filled in itself (shards merged on
// another rank, a result read back from disk) to the post-search calls: only a result found here is a HostResult whose private half may be looked at.
struct LiveResults {
    std::mutex m;
    std::set<const mapad_batch_result_t*> s;
    static LiveResults& get() { static LiveResults* r = new LiveResults(); return *r; }  // never destroyed: results may be freed while the process exits
    void add(const mapad_batch_result_t* p) { std::lock_guard<std::mutex> g(m); s.insert(p); }
    bool remove(const mapad_batch_result_t* p) { std::lock_guard<std::mutex> g(m); return s.erase(p) != 0; }
    bool has(const mapad_batch_result_t* p) { std::lock_guard<std::mutex> g(m); return s.count(p) != 0; }
};

}  // namespace

extern "C" {

const char* mapad_version(void) { return "mapad_amd 0.1.0 (gfx950; mapAD 0.45.0 hot path)"; }

// ---- parameters / plugin surface ------------------------------------------------------------------------------------------
int mapad_params_from_cli(mapad_params_t* out, int library_prep, float five_prime_overhang, float three_prime_overhang, float ds_deamination_rate,
                          float ss_deamination_rate, float divergence, float poisson_prob, float as_cutoff, float as_cutoff_exponent, float indel_rate,
                          float gap_extension_penalty, int gap_dist_ends, int max_num_gaps_open, int ignore_base_quality, int no_search_limit_recovery,
                          uint64_t chunk_size) {
    if (!out) return MAPAD_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    out->model_kind = MAPAD_MODEL_SIMPLE_ADNA;
    out->library_prep = library_prep;
    out->five_prime_overhang = five_prime_overhang;
    out->three_prime_overhang = library_prep == MAPAD_LIBRARY_DOUBLE_STRANDED ? five_prime_overhang : three_prime_overhang;
    out->ds_deamination_rate = ds_deamination_rate; out->ss_deamination_rate = ss_deamination_rate;
    out->divergence = divergence / 3.0f;  // main.rs:452
    out->ignore_base_quality = ignore_base_quality;
    if (poisson_prob >= 0.0f) { out->bound_kind = MAPAD_BOUND_DISCRETE; out->poisson_threshold = poisson_prob; out->base_error_rate = divergence; }  // :456-462
    else { out->bound_kind = MAPAD_BOUND_CONTINUOUS; out->cutoff = as_cutoff * -1.0f; out->exponent = as_cutoff_exponent; }                             // :463-475
    const float repr = host::sdm_repr_mm(*out);
    out->penalty_gap_open = std::log2(indel_rate);             // :478-481
    out->penalty_gap_extend = gap_extension_penalty * repr;    // :482-485
    out->gap_dist_ends = gap_dist_ends; out->max_num_gaps_open = max_num_gaps_open;
    out->stack_limit_abort = no_search_limit_recovery;
    out->chunk_size = chunk_size ? chunk_size : 250000;
    return MAPAD_OK;
}
float mapad_sdm_get(const mapad_params_t* p, uint64_t i, uint64_t len, uint8_t from, uint8_t to, uint8_t q) { return host::sdm_get(*p, i, len, from, to, q); }
float mapad_sdm_representative_mismatch_penalty(const mapad_params_t* p) { return host::sdm_repr_mm(*p); }
float mapad_sdm_min_penalty(const mapad_params_t* p, uint64_t i, uint64_t len, uint8_t to, uint8_t q, int only_mm) { return host::sdm_min_penalty(*p, i, len, to, q, only_mm != 0); }
int32_t mapad_sdm_alignment_start(const mapad_params_t* p, uint64_t len) { return host::sdm_alignment_start(*p, len); }
int mapad_mb_reject(const mapad_params_t* p, float v, uint64_t len) { return host::mb_reject(*p, v, len); }
int mapad_mb_reject_iterative(const mapad_params_t* p, float v, float ref) { return host::mb_reject_iterative(*p, v, ref); }
float mapad_mb_remaining_frac_of_repr_mm(const mapad_params_t* p, float v, uint64_t len) { return host::mb_remaining_frac(*p, v, len); }

// ---- index ------------------------------------------------------------------------------------------------------------
int mapad_index_build(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n_contigs, uint64_t seed, mapad_index_t** out) {
    if (!out || !names || !seqs || !lens || n_contigs == 0) return MAPAD_ERR_INVALID;
    try {
        std::vector<std::string> nm;
        for (uint32_t i = 0; i < n_contigs; ++i) nm.emplace_back(names[i]);
        const char* fr = std::getenv("MAPAD_INDEX_FIXED_REPLACEMENT");  // test hook: pin the random IUPAC replacement
        auto idx = std::make_unique<mapad_index>();
        idx->ix = host::build_index(nm, seqs, lens, seed, fr && fr[0] ? (uint8_t)fr[0] : 0);
        *out = idx.release();
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_index_build: %s\n", e.what());
        return MAPAD_ERR_PARSE;
    }
}
int mapad_index_build_gpu(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n_contigs, uint64_t seed, int device_id, mapad_index_t** out) {
    if (!out || !names || !seqs || !lens || n_contigs == 0) return MAPAD_ERR_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) return MAPAD_ERR_NO_DEVICE;
    try {
        std::vector<std::string> nm;
        for (uint32_t i = 0; i < n_contigs; ++i) nm.emplace_back(names[i]);
        const char* fr = std::getenv("MAPAD_INDEX_FIXED_REPLACEMENT");
        const char* vb = std::getenv("MAPAD_INDEX_VERBOSE");
        auto idx = std::make_unique<mapad_index>();
        std::vector<uint8_t> t = host::prepare_text(nm, seqs, lens, seed, fr && fr[0] ? (uint8_t)fr[0] : 0, idx->ix);
        gpuidx::suffix_products(t.data(), idx->ix, device_id, vb && vb[0] == '1');
        *out = idx.release();
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::length_error& e) {  // a documented limit of the GPU sorter: mapad_index_build takes such texts
        std::fprintf(stderr, "mapad_index_build_gpu: %s\n", e.what());
        return MAPAD_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_index_build_gpu: %s\n", e.what());
        return MAPAD_ERR_PARSE;
    }
}
int mapad_index_open(const char* prefix, mapad_index_t** out) {
    if (!prefix || !out) return MAPAD_ERR_INVALID;
    auto idx = std::make_unique<mapad_index>();
    const int rc = host::load_index(prefix, idx->ix);
    if (rc != MAPAD_OK) return rc;
    *out = idx.release();
    return MAPAD_OK;
}
int mapad_index_save(const mapad_index_t* idx, const char* prefix) {
    if (!idx || !prefix) return MAPAD_ERR_INVALID;
    return host::save_index(prefix, idx->ix);
}
void mapad_index_free(mapad_index_t* idx) { delete idx; }
uint64_t mapad_index_text_len(const mapad_index_t* idx) { return idx ? idx->ix.n : 0; }
int mapad_index_copy_bwt(const mapad_index_t* idx, uint8_t* out) {
    if (!idx || !out) return MAPAD_ERR_INVALID;
    std::memcpy(out, idx->ix.bwt.data(), idx->ix.bwt.size());
    return MAPAD_OK;
}
uint32_t mapad_index_n_contigs(const mapad_index_t* idx) { return idx ? (uint32_t)idx->ix.contigs.size() : 0; }
int mapad_index_contig(const mapad_index_t* idx, uint32_t i, const char** name, uint64_t* start, uint64_t* end) {
    if (!idx || i >= idx->ix.contigs.size()) return MAPAD_ERR_INVALID;
    if (name) *name = idx->ix.contigs[i].name.c_str();
    if (start) *start = idx->ix.contigs[i].start;
    if (end) *end = idx->ix.contigs[i].end;
    return MAPAD_OK;
}
uint64_t mapad_index_sa_sample_len(const mapad_index_t* idx) { return idx ? idx->ix.sa_sample.size() : 0; }
uint64_t mapad_index_sa_extra_len(const mapad_index_t* idx) { return idx ? idx->ix.extra_rows.size() : 0; }
int mapad_index_copy_sa(const mapad_index_t* idx, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_vals) {
    if (!idx) return MAPAD_ERR_INVALID;
    if (sample) std::memcpy(sample, idx->ix.sa_sample.data(), idx->ix.sa_sample.size() * 8);
    size_t k = 0;
    for (auto& kv : idx->ix.extra_rows) { if (extra_rows) extra_rows[k] = kv.first; if (extra_vals) extra_vals[k] = kv.second; ++k; }
    return MAPAD_OK;
}

int mapad_index_device_view(const mapad_index_t* idx, const uint64_t** blocks, uint64_t* n_blocks, uint64_t less[8], uint64_t sentinel[2]) {
    if (!idx) return MAPAD_ERR_INVALID;
    if (blocks) *blocks = idx->ix.blocks.data();
    if (n_blocks) *n_blocks = idx->ix.blocks.size() / kBlockWords;
    if (less) std::memcpy(less, idx->ix.less, sizeof idx->ix.less);
    if (sentinel) { sentinel[0] = idx->ix.sentinel[0]; sentinel[1] = idx->ix.sentinel[1]; }
    return MAPAD_OK;
}
int mapad_index_sa_get_batch(const mapad_index_t* idx, const uint64_t* rows, uint64_t n, uint64_t* out) {
    if (!idx || (n && (!rows || !out))) return MAPAD_ERR_INVALID;
    try {
        for (uint64_t i = 0; i < n; ++i) if (!idx->ix.sa_get(rows[i], out[i])) out[i] = ~0ull;
        return MAPAD_OK;
    } catch (const std::exception&) { return MAPAD_ERR_PARSE; }
}
int mapad_index_sa_get(const mapad_index_t* idx, uint64_t row, uint64_t* out) {
    if (!idx || !out) return MAPAD_ERR_INVALID;
    try { return idx->ix.sa_get(row, *out) ? MAPAD_OK : MAPAD_ERR_INVALID; } catch (const std::exception&) { return MAPAD_ERR_PARSE; }
}

// ---- context ------------------------------------------------------------------------------------------------------------
int mapad_ctx_create(const mapad_index_t* idx, const mapad_params_t* params, int device_id, mapad_ctx_t** out) {
    if (!idx || !params || !out) return MAPAD_ERR_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device_id < 0 || device_id >= n_dev) {
        std::fprintf(stderr, "mapad_amd: no usable HIP device (there is no CPU fallback)\n");
        return MAPAD_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device_id) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "mapad_amd: device %d is %s, this library is built for gfx950 only\n", device_id, prop.gcnArchName);
        return MAPAD_ERR_NO_DEVICE;
    }
    if (idx->ix.n >= (1ull << 40)) { std::fprintf(stderr, "mapad_amd: text longer than 2^40 symbols is not supported by the packed frame layout\n"); return MAPAD_ERR_INVALID; }
    auto c = std::make_unique<mapad_ctx>();
    c->device = device_id; c->params = *params; c->index = idx; c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->tables = host::make_tables(*params);
    c->depth = (int)std::min<uint32_t>(std::max<uint32_t>(env_u32("MAPAD_PIPELINE_DEPTH", 1), 1), kMaxDepth);
    c->tail_pops = env_u32("MAPAD_TAIL_POPS", MAPAD_DEFAULT_TAIL_POPS);
    c->reserved_cus = (int)std::min<uint32_t>(env_u32("MAPAD_RESERVED_CUS", 0), (uint32_t)c->n_cu - 1);
    int rc;
    if ((rc = c->d_blocks.ensure(idx->ix.blocks.size()))) return rc;
    HIP_TRY(hipMemcpy(c->d_blocks.p, idx->ix.blocks.data(), idx->ix.blocks.size() * 8, hipMemcpyHostToDevice));
    c->dix = idx->ix.view();
    c->dix.blocks = c->d_blocks.p;
    *out = c.release();
    return MAPAD_OK;
}
void mapad_ctx_destroy(mapad_ctx_t* ctx) { delete ctx; }
int mapad_ctx_set_tail_pops(mapad_ctx_t* ctx, uint32_t pops) {
    if (!ctx) return MAPAD_ERR_INVALID;
    if ((pops != 0) != (ctx->tail_pops != 0)) ctx->tables_dirty = true;  // the host copy of the tables is taken with the next upload
    ctx->tail_pops = pops;
    return MAPAD_OK;
}
uint32_t mapad_tail_set_local_world(uint32_t local_world) { return host::TailWorkers::instance().set_local_world(local_world); }
int mapad_last_tail_info(mapad_ctx_t* ctx, uint64_t out[16]) {
    if (!ctx || !out) return MAPAD_ERR_INVALID;
    std::memcpy(out, ctx->bs[ctx->view].tail_info, sizeof ctx->bs[ctx->view].tail_info);
    return MAPAD_OK;
}
int mapad_ctx_set_fetch_d_arrays(mapad_ctx_t* ctx, int on) { if (!ctx) return MAPAD_ERR_INVALID; ctx->fetch_d = on != 0; return MAPAD_OK; }
int mapad_ctx_set_stream(mapad_ctx_t* ctx, void* s) { if (!ctx) return MAPAD_ERR_INVALID; ctx->stream = (hipStream_t)s; return MAPAD_OK; }

// tables for the read lengths of the coming batches (device-resident inputs: the host cannot see the lengths)
int mapad_ctx_prepare_lengths(mapad_ctx_t* ctx, const uint32_t* lens, uint32_t n) {
    if (!ctx || (!lens && n)) return MAPAD_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        if (lens[i] == 0) continue;
        if (lens[i] > MAPAD_MAX_READ_LEN) return MAPAD_ERR_READ_TOO_LONG;
        if (ctx->tables.table_base[lens[i]] < 0) { host::add_length(ctx->params, ctx->tables, (int)lens[i]); ctx->tables_dirty = true; }
    }
    return MAPAD_OK;
}

int mapad_map_batch_device(mapad_ctx_t* ctx, const void* d_seqs, const void* d_quals, const void* d_offsets, uint64_t n_reads, uint32_t max_read_len) {
    if (!ctx || (n_reads && (!d_seqs || !d_quals || !d_offsets))) return MAPAD_ERR_INVALID;
    if (max_read_len > MAPAD_MAX_READ_LEN) return MAPAD_ERR_READ_TOO_LONG;
    if (n_reads >= 0xFFFFFFF0ull) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = acquire_slot(ctx, (ctx->cur + 1) % ctx->depth))) return rc;  // batches rotate through the slots: this one runs beside the previous one's tail
    // the batch's base count, read on the slot's own (idle) stream: a copy on the caller's stream would wait for whatever that stream orders
    // itself behind (the legacy default stream: every blocking stream)
    uint64_t total = 0;
    if (n_reads) {
        BatchSlot& S = ctx->bs[ctx->cur];
        if (!S.h_small.resize(CUR_COUNT + 8)) return MAPAD_ERR_NOMEM;
        if ((rc = read_back_words(S.stream, (const uint64_t*)d_offsets + n_reads, S.h_small, CUR_COUNT, 2))) return rc;
        std::memcpy(&total, S.h_small.data() + CUR_COUNT, 8);
    }
    return launch_batch(ctx, ctx->bs[ctx->cur], (const uint8_t*)d_seqs, (const uint8_t*)d_quals, (const uint64_t*)d_offsets, n_reads, total, max_read_len);
}

int mapad_fetch_result(mapad_ctx_t* ctx, mapad_batch_result_t** out) {
    if (!ctx || !out) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = compact_last(ctx))) return rc;
    BatchSlot& S = ctx->bs[ctx->view];
    const BatchDev& B = S.last;
    const uint64_t n = B.n_reads;
    auto r = std::make_unique<HostResult>();
    uint32_t cur[CUR_COUNT] = {0};
    if (n) {
        if (!S.h_small.resize(CUR_COUNT + 8)) return MAPAD_ERR_NOMEM;
        hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, S.stream, (const uint32_t*)B.cursors, S.h_small.data(), (uint32_t)CUR_COUNT);
        HIP_TRY(hipGetLastError());
    }
    // order-preserving collect (mapping.rs:288): hits in read order, BinaryHeap array order inside a read — laid out by the device
    bool ok = r->status.resize(n) && r->counters.resize(n) && r->hit_begin.resize(n + 1) && r->hits.resize(S.c_n_hits) && r->ops.resize(S.c_n_ops);
    if (ctx->fetch_d) ok = ok && r->d_arrays.resize(S.last_total_bases);
    if (!ok) return MAPAD_ERR_NOMEM;
    r->hit_begin[0] = 0;
    static_assert(sizeof(mapad_hit_t) == sizeof(HitRec), "hit records are copied as they are");
    if (n) {
        HIP_TRY(hipMemcpyAsync(r->hit_begin.data(), S.d_c_hit_begin.p, (n + 1) * 8, hipMemcpyDeviceToHost, S.stream));
        HIP_TRY(hipMemcpyAsync(r->status.data(), B.status, n * 4, hipMemcpyDeviceToHost, S.stream));
        HIP_TRY(hipMemcpyAsync(r->counters.data(), B.counters, n * sizeof(ReadCounters), hipMemcpyDeviceToHost, S.stream));
        if (!r->hits.empty()) HIP_TRY(hipMemcpyAsync(r->hits.data(), S.d_c_hits.p, r->hits.size() * sizeof(HitRec), hipMemcpyDeviceToHost, S.stream));
        if (!r->ops.empty()) HIP_TRY(hipMemcpyAsync(r->ops.data(), S.d_c_ops.p, r->ops.size() * 4, hipMemcpyDeviceToHost, S.stream));
        if (ctx->fetch_d && S.last_total_bases) HIP_TRY(hipMemcpyAsync(r->d_arrays.data(), B.d_arrays, S.last_total_bases * 4, hipMemcpyDeviceToHost, S.stream));
    }
    HIP_TRY(hipStreamSynchronize(S.stream));
    if (n) std::memcpy(cur, S.h_small.data(), sizeof cur);
    if (r->hit_begin[n] != r->hits.size()) { std::fprintf(stderr, "mapad_amd: compacted hit count does not match the pool cursor\n"); return MAPAD_ERR_DEVICE; }
    uint64_t sums[6] = {0, 0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < n; ++i) {
        const auto& c = r->counters[i];
        sums[0] += c.e_search; sums[1] += c.e_darray; sums[2] += c.n_push; sums[3] += c.n_pop; sums[4] += c.n_node; sums[5] += c.n_hits;
    }
    std::memcpy(ctx->counter_sums, sums, sizeof sums);
    r->owner = ctx; r->slot = ctx->view; r->gen = S.gen;
    r->pub.n_reads = n; r->pub.n_hits = r->hits.size(); r->pub.n_ops = r->ops.size();
    r->pub.hit_begin = r->hit_begin.data(); r->pub.hits = r->hits.data(); r->pub.ops = r->ops.data();
    r->pub.status = r->status.data(); r->pub.counters = r->counters.data(); r->pub.d_arrays = ctx->fetch_d ? r->d_arrays.data() : nullptr;
    r->pub.n_second_pass = cur[CUR_GROWN];  // arena migrations in pass 0
    if (std::getenv("MAPAD_DEBUG")) {
        std::fprintf(stderr, "mapad_amd: arena classes (nodes x count):");
        for (int k = 0; k <= kClasses; ++k) std::fprintf(stderr, " %u x %u%s", ctx->grow.node_cap[k], ctx->grow.count[k], k == kClasses ? " (sets)" : "");
        std::fprintf(stderr, "; base arenas %u nodes x %u slots; host tail: %u reads, %u on a dry class, %u instead of the full-limit stage\n", ctx->pool[0].node_cap, ctx->slots[0], cur[CUR_TAIL], cur[CUR_TAIL_DRY], cur[CUR_TAIL_F]);
    }
    if (cur[CUR_GROWN + 1] && std::getenv("MAPAD_DEBUG")) std::fprintf(stderr, "mapad_amd: size-class pools that ran dry (bit per class): 0x%x, waits: %u, reads restarted: %u, re-run with full limits: %u\n", cur[CUR_GROWN + 1], cur[CUR_GROWN + 2], cur[CUR_OVF], cur[CUR_OVF + 2 * (kStages - 2)]);
    r->pub.n_third_pass = cur[CUR_OVF + 2 * (kStages - 2)] + cur[CUR_TAIL_F];  // reads no growable arena could hold: re-run with the reference's full limits, by the last GPU stage or (host tail on) by a host thread
    if (std::getenv("MAPAD_DEBUG")) {
        std::fprintf(stderr, "mapad_amd: heavy reads %u + %u, pops by heavy wavefronts %llu\n", cur[CUR_HEAVY_N], cur[CUR_HEAVY_N + 1], (unsigned long long)cur64(cur, CUR_HEAVY_POPS));
#if defined(MAPAD_HEAVY_PROF)
        static const char* names[8] = {"find max, node + last loaded", "unpack, scores, D", "rank + ancestor loads issued", "sift", "ancestor table", "rank finish", "children", "pushes"};
        double tot = 0;
        for (int k = 0; k < 8; ++k) tot += (double)cur64(cur, CUR_HPROF + 2 * k);
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, "[heavy prof] %-30s %6.1f %%  %8.1f cycles per pop\n", names[k], 100.0 * cur64(cur, CUR_HPROF + 2 * k) / std::max(tot, 1.0), (double)cur64(cur, CUR_HPROF + 2 * k) / std::max<double>((double)cur64(cur, CUR_HEAVY_POPS), 1.0));
#endif
    }
#if defined(MAPAD_PROFILE_SECTIONS)
    {
        unsigned long long pv[2 * PROF_N + 64];
        HIP_TRY(hipMemcpy(pv, ctx->d_prof.p, sizeof pv, hipMemcpyDeviceToHost));
        const unsigned long long* hv = pv + 2 * PROF_N;
        auto row = [&](const char* what, int lo, int n) {
            double tot = 0; for (int i = 0; i < n; ++i) tot += (double)hv[lo + i];
            std::fprintf(stderr, "[sections] %s:", what);
            for (int i = 0; i < n; ++i) std::fprintf(stderr, " %d:%.1f%%", i, 100.0 * hv[lo + i] / std::max(tot, 1.0));
            std::fprintf(stderr, "\n");
        };
        row("log2(heap_len) at pop", 0, 24); row("children committed per pop", 24, 12); row("commit-loop trips per wave step", 36, 12);
        static const char* names[PROF_N] = {"pop+sift", "node/row/D", "rank ext", "gates+kids", "commit loop", "step tail", "read setup", "finalize", "grow", "record hit", "loop head", "commit: pre", "commit: wait", "commit: anc"};
        double tw = 0, tl = 0;
        for (int k = 0; k < PROF_N; ++k) { tw += (double)pv[k]; tl += (double)pv[PROF_N + k]; }
        std::fprintf(stderr, "[sections] wave-cycles %.3e, lane-cycles %.3e (mean active lanes %.1f)\n", tw, tl, tl / std::max(tw, 1.0));
        for (int k = 0; k < PROF_N; ++k)
            std::fprintf(stderr, "[sections] %-12s wave %5.1f %%   lanes %5.1f %%   active lanes %4.1f\n", names[k], 100.0 * pv[k] / std::max(tw, 1.0), 100.0 * pv[PROF_N + k] / std::max(tl, 1.0),
                         (double)pv[PROF_N + k] / std::max<double>((double)pv[k], 1.0));
    }
#endif
    LiveResults::get().add(&r->pub);
    *out = &r.release()->pub;
    return MAPAD_OK;
}
int mapad_compact_result_device(mapad_ctx_t* ctx, void** d_hit_begin, void** d_hits, void** d_ops, uint64_t* n_hits, uint64_t* n_ops) {
    if (!ctx) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    const int rc = compact_last(ctx);
    if (rc) return rc;
    BatchSlot& S = ctx->bs[ctx->view];
    if (S.stream != ctx->stream) {  // the caller consumes the arrays on its own stream: order it behind the collect
        if (!S.ev_in) HIP_TRY(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(S.ev_in, S.stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, S.ev_in, 0));
    }
    if (d_hit_begin) *d_hit_begin = S.d_c_hit_begin.p;
    if (d_hits) *d_hits = S.d_c_hits.p;
    if (d_ops) *d_ops = S.d_c_ops.p;
    if (n_hits) *n_hits = S.c_n_hits;
    if (n_ops) *n_ops = S.c_n_ops;
    return MAPAD_OK;
}
void mapad_batch_result_free(mapad_batch_result_t* r) {
    if (r && LiveResults::get().remove(r)) delete reinterpret_cast<HostResult*>(r);  // pub is the first member; a pointer this library did not hand out (or a second free) is left alone
}

namespace {
// host reads -> a slot's device buffers (+ the score tables of their lengths); returns once the inputs have left the host buffers
int stage_host_batch(mapad_ctx_t* ctx, BatchSlot& S, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads, uint64_t& total, uint32_t& lmax) {
    total = n_reads ? offsets[n_reads] : 0;
    bool seen[MAPAD_MAX_READ_LEN + 1] = {false};
    std::vector<uint32_t> lv;
    lmax = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        const uint64_t l = offsets[i + 1] - offsets[i];
        if (l > MAPAD_MAX_READ_LEN) return MAPAD_ERR_READ_TOO_LONG;
        if (!seen[l]) { seen[l] = true; lv.push_back((uint32_t)l); }
        lmax = std::max<uint32_t>(lmax, (uint32_t)l);
    }
    int rc;
    if ((rc = mapad_ctx_prepare_lengths(ctx, lv.data(), (uint32_t)lv.size()))) return rc;
    if ((rc = S.d_seqs.ensure(std::max<uint64_t>(total, 1)))) return rc;
    if ((rc = S.d_quals.ensure(std::max<uint64_t>(total, 1)))) return rc;
    if ((rc = S.d_offsets.ensure(n_reads + 1))) return rc;
    if (n_reads) {  // pinned sources (mapad_host_alloc) go by DMA at link speed, pageable ones through the driver's staging buffers
        HIP_TRY(hipMemcpyAsync(S.d_seqs.p, seqs, total, hipMemcpyHostToDevice, S.stream));
        HIP_TRY(hipMemcpyAsync(S.d_quals.p, quals, total, hipMemcpyHostToDevice, S.stream));
        HIP_TRY(hipMemcpyAsync(S.d_offsets.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, S.stream));
        if (!S.ev_in) HIP_TRY(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(S.ev_in, S.stream));
        HIP_TRY(hipEventSynchronize(S.ev_in));
    }
    return MAPAD_OK;
}
}  // namespace

int mapad_map_batch(mapad_ctx_t* ctx, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads, mapad_batch_result_t** out) {
    if (!ctx || !out || (n_reads && (!seqs || !quals || !offsets))) return MAPAD_ERR_INVALID;
    if (n_reads >= 0xFFFFFFF0ull) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = acquire_slot(ctx, ctx->cur))) return rc;  // synchronous entry point: no rotation, one slot's buffers
    BatchSlot& S = ctx->bs[ctx->cur];
    uint64_t total = 0;
    uint32_t lmax = 0;
    if ((rc = stage_host_batch(ctx, S, seqs, quals, offsets, n_reads, total, lmax))) return rc;
    for (int attempt = 0; attempt < 6; ++attempt) {
        if ((rc = launch_batch(ctx, S, S.d_seqs.p, S.d_quals.p, S.d_offsets.p, n_reads, total, lmax))) return rc;
        rc = mapad_fetch_result(ctx, out);
        if (rc != MAPAD_ERR_NOMEM) return rc;
        // pools too small for this batch: quadruple and retry (rare: needs > 2 hits per read on average)
        uint32_t cur[CUR_COUNT];
        HIP_TRY(hipMemcpy(cur, S.d_cursors.p, sizeof cur, hipMemcpyDeviceToHost));
        if (!cur[CUR_POOL_OVF]) return rc;
        if (cur64(cur, CUR_HITS) + 1024 > 0xFFFFFFFFull || cur64(cur, CUR_OPS) + 1024 > 0xFFFFFFFFull) {
            std::fprintf(stderr, "mapad_amd: the batch produces more than 2^32 hit records or edit operations; split it\n");
            return MAPAD_ERR_INVALID;
        }
        if ((rc = S.d_hits.ensure((size_t)cur64(cur, CUR_HITS) + 1024))) return rc;
        if ((rc = S.d_ops.ensure((size_t)cur64(cur, CUR_OPS) + 1024))) return rc;
    }
    return MAPAD_ERR_NOMEM;
}

int mapad_submit_batch(mapad_ctx_t* ctx, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads) {
    if (!ctx || (n_reads && (!seqs || !quals || !offsets))) return MAPAD_ERR_INVALID;
    if (n_reads >= 0xFFFFFFF0ull) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = acquire_slot(ctx, (ctx->cur + 1) % ctx->depth))) return rc;
    BatchSlot& S = ctx->bs[ctx->cur];
    uint64_t total = 0;
    uint32_t lmax = 0;
    if ((rc = stage_host_batch(ctx, S, seqs, quals, offsets, n_reads, total, lmax))) return rc;
    return launch_batch(ctx, S, S.d_seqs.p, S.d_quals.p, S.d_offsets.p, n_reads, total, lmax);
}
// Page-locked blocks are recycled: pinning and unpinning cost milliseconds per block, and hipHostFree waits for the device — in a chunk loop
// that allocates its read buffers per chunk (mapad-amd map) this stalled the whole pipeline once per chunk.  Blocks come in power-of-two sizes;
// up to 8 GB of freed blocks are kept for reuse.
namespace {
struct HostBlockCache {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> size_of;
    size_t cached = 0;
    static size_t bucket(size_t bytes) { size_t b = 65536; while (b < bytes) b <<= 1; return b; }
    void* alloc(size_t bytes) {
        const size_t want = bucket(bytes);
        {
            std::lock_guard<std::mutex> l(mu);
            auto it = free_blocks.find(want);
            if (it != free_blocks.end()) { void* p = it->second; free_blocks.erase(it); cached -= want; return p; }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> l(mu);
        size_of[p] = want;
        return p;
    }
    void release(void* p) {
        size_t sz = 0;
        {
            std::lock_guard<std::mutex> l(mu);
            auto it = size_of.find(p);
            if (it == size_of.end()) return;  // not ours
            sz = it->second;
            if (cached + sz <= (8ull << 30)) { free_blocks.emplace(sz, p); cached += sz; return; }
            size_of.erase(it);
        }
        (void)hipHostFree(p);
    }
};
extern "C++" HostBlockCache& host_blocks() { static HostBlockCache* c = new HostBlockCache(); return *c; }  // never destroyed: the runtime may be gone at exit
}  // namespace
void* mapad_host_alloc(size_t bytes) { return host_blocks().alloc(bytes ? bytes : 1); }
void mapad_host_free(void* p) { if (p) host_blocks().release(p); }
unsigned mapad_host_cpus(void) { return host::cpu_share(); }

int mapad_device_result_ptrs(mapad_ctx_t* ctx, void** d_hit_count, void** d_hit_first, void** d_hits, void** d_ops, void** d_cursors) {
    if (!ctx) return MAPAD_ERR_INVALID;
    if (d_hit_count) *d_hit_count = ctx->bs[ctx->view].last.hit_count;
    if (d_hit_first) *d_hit_first = ctx->bs[ctx->view].last.hit_first;
    if (d_hits) *d_hits = ctx->bs[ctx->view].last.hits_pool;
    if (d_ops) *d_ops = ctx->bs[ctx->view].last.ops_pool;
    if (d_cursors) *d_cursors = ctx->bs[ctx->view].last.cursors;
    return MAPAD_OK;
}
int mapad_last_batch_counters(mapad_ctx_t* ctx, uint64_t out[6]) {
    if (!ctx || !out) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    HIP_TRY(hipStreamSynchronize(ctx->bs[ctx->view].stream));
    const uint64_t n = ctx->bs[ctx->view].last.n_reads;
    std::vector<ReadCounters> c(n);
    if (n) HIP_TRY(hipMemcpy(c.data(), ctx->bs[ctx->view].last.counters, n * sizeof(ReadCounters), hipMemcpyDeviceToHost));
    uint64_t s[6] = {0, 0, 0, 0, 0, 0};
    for (auto& x : c) { s[0] += x.e_search; s[1] += x.e_darray; s[2] += x.n_push; s[3] += x.n_pop; s[4] += x.n_node; s[5] += x.n_hits; }
    std::memcpy(out, s, sizeof s);
    return MAPAD_OK;
}
int mapad_last_kernel_ms(mapad_ctx_t* ctx, float out[3]) {
    if (!ctx || !out) return MAPAD_ERR_INVALID;
    BatchSlot& S = ctx->bs[ctx->view];
    if (!S.ev_valid) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    HIP_TRY(hipEventSynchronize(S.ev[3]));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&out[i], S.ev[i], S.ev[i + 1]));
    return MAPAD_OK;
}
int mapad_last_launch_info(mapad_ctx_t* ctx, uint32_t out[8]) {
    if (!ctx || !out) return MAPAD_ERR_INVALID;
    std::memcpy(out, ctx->bs[ctx->view].launch_info, sizeof ctx->bs[ctx->view].launch_info);
    return MAPAD_OK;
}

int mapad_ctx_set_reserved_cus(mapad_ctx_t* ctx, int n_cus) {
    if (!ctx || n_cus < 0 || n_cus >= ctx->n_cu) return MAPAD_ERR_INVALID;
    for (auto& b : ctx->bs) if (b.own_stream) return MAPAD_ERR_INVALID;  // the slots' streams exist already: set this before the first batch / mapad_ctx_reserve
    ctx->reserved_cus = n_cus;
    return MAPAD_OK;
}
int mapad_ctx_set_pipeline_depth(mapad_ctx_t* ctx, int depth) {
    if (!ctx || depth < 1 || depth > kMaxDepth) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = sync_all_slots(ctx))) return rc;
    for (auto& b : ctx->bs) { if ((rc = record_times(ctx, b))) return rc; drop_tail(ctx, b); b.release(); b.ev_valid = false; b.compacted = false; }
    ctx->depth = depth; ctx->cur = 0; ctx->view = 0;
    ctx->arena_reads = 0; ctx->arena_lmax = 0; ctx->pool[0].stride = 0;  // pools are re-sized around the base arenas of `depth` batches
    return MAPAD_OK;
}
int mapad_ctx_reserve(mapad_ctx_t* ctx, uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int host_inputs) {
    if (!ctx || max_read_len > MAPAD_MAX_READ_LEN) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = sync_all_slots(ctx))) return rc;
    for (int k = 0; k < ctx->depth; ++k) {
        BatchSlot& S = ctx->bs[k];
        if (ctx->depth == 1) S.stream = ctx->stream;
        else if (!S.stream) { const int rc_s = create_slot_stream(ctx, &S.stream); if (rc_s) return rc_s; S.own_stream = true; }
        if ((rc = ensure_batch_buffers(ctx, S, n_reads, total_bases, max_read_len, host_inputs != 0))) return rc;
        // the collect's outputs: ~1 hit per read, one op per base + indels (grown on demand if a batch needs more)
        if ((rc = S.d_c_hit_begin.ensure(n_reads + 1))) return rc;
        if ((rc = S.d_c_ops_begin.ensure(n_reads + 1))) return rc;
        if ((rc = S.d_c_tiles.ensure(2 * std::max<uint64_t>((n_reads + kScanTile - 1) / kScanTile, 1)))) return rc;
        if ((rc = S.d_c_hits.ensure(n_reads + n_reads / 4 + 1024))) return rc;
        if ((rc = S.d_c_ops.ensure(total_bases + total_bases / 4 + 1024))) return rc;
        // first use of a stream sets up its hardware queue and scratch ring, which waits for a busy GPU: do it now, with an empty launch
        if (!S.ev_in) HIP_TRY(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(S.ev_in, S.stream));
        if ((rc = launch_batch(ctx, S, nullptr, nullptr, nullptr, 0, 0, max_read_len, true))) return rc;
        HIP_TRY(hipStreamSynchronize(S.stream));
        S.compacted = false;
    }
    return MAPAD_OK;
}
int mapad_ctx_select_batch(mapad_ctx_t* ctx, int age) {
    if (!ctx || age < 0 || age >= ctx->depth) return MAPAD_ERR_INVALID;
    ctx->view = ((ctx->cur - age) % ctx->depth + ctx->depth) % ctx->depth;
    return ctx->bs[ctx->view].ev_valid ? MAPAD_OK : MAPAD_ERR_INVALID;
}
int mapad_kernel_history(mapad_ctx_t* ctx, float* out, uint32_t cap, uint32_t* n) {
    if (!ctx || !n || (cap && !out)) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = sync_all_slots(ctx))) return rc;
    // launches finish in submission order per slot; slots are visited in submission order: cur + 1, ..., cur
    for (int k = 1; k <= ctx->depth; ++k) if ((rc = record_times(ctx, ctx->bs[(ctx->cur + k) % ctx->depth]))) return rc;
    const uint32_t have = (uint32_t)(ctx->history.size() / 4);
    *n = have;
    for (uint32_t i = 0; i < have && i < cap; ++i) std::memcpy(out + 4 * i, ctx->history.data() + 4 * i, 16);
    ctx->history.clear();
    if (ctx->ev_ref) { (void)hipEventDestroy(ctx->ev_ref); ctx->ev_ref = nullptr; }
    return MAPAD_OK;
}
}  // extern "C"

// ---- SA locate on the device ---------------------------------------------------------------------------------------------
namespace {
int ensure_sa_uploaded(mapad_ctx* c) {
    if (c->sa_uploaded) return MAPAD_OK;
    const host::Index& ix = c->index->ix;
    int rc;
    if ((rc = c->d_sa.ensure(std::max<size_t>(ix.sa_sample.size(), 1), true))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_sa.p, ix.sa_sample.data(), ix.sa_sample.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (!ix.x_counts.empty()) {
        if ((rc = c->d_xc.ensure(ix.x_counts.size(), true))) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_xc.p, ix.x_counts.data(), ix.x_counts.size() * 8, hipMemcpyHostToDevice, c->stream));
    }
    std::vector<uint64_t> cs;  // contig starts, then ends
    for (auto& k : ix.contigs) cs.push_back(k.start);
    for (auto& k : ix.contigs) cs.push_back(k.end);
    if ((rc = c->d_contigs.ensure(std::max<size_t>(cs.size(), 2), true))) return rc;
    if (!cs.empty()) HIP_TRY(hipMemcpyAsync(c->d_contigs.p, cs.data(), cs.size() * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = c->d_steps.ensure(1))) return rc;
    // what the text kernel needs of the index: original symbols (ascending positions) and contig names
    std::vector<uint64_t> os_pos;
    std::vector<uint8_t> os_sym;
    for (const auto& kv : ix.original_symbols) { os_pos.push_back(kv.first); os_sym.push_back(kv.second); }
    c->n_os = os_pos.size();
    if ((rc = c->d_os_pos.ensure(std::max<size_t>(os_pos.size(), 1), true))) return rc;
    if ((rc = c->d_os_sym.ensure(std::max<size_t>(os_sym.size(), 1), true))) return rc;
    if (!os_pos.empty()) {
        HIP_TRY(hipMemcpyAsync(c->d_os_pos.p, os_pos.data(), os_pos.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_os_sym.p, os_sym.data(), os_sym.size(), hipMemcpyHostToDevice, c->stream));
    }
    std::vector<uint32_t> name_off{0};
    std::string names;
    for (const auto& k : ix.contigs) { names += k.name; name_off.push_back((uint32_t)names.size()); }
    if ((rc = c->d_names.ensure(std::max<size_t>(names.size(), 1), true))) return rc;
    if ((rc = c->d_name_off.ensure(name_off.size(), true))) return rc;
    if (!names.empty()) HIP_TRY(hipMemcpyAsync(c->d_names.p, names.data(), names.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_name_off.p, name_off.data(), name_off.size() * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = c->d_t_cur.ensure(2))) return rc;
    for (auto& e : c->lev) if (!e) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors go out of scope
    c->sa_uploaded = true;
    return MAPAD_OK;
}
int locate_rows(mapad_ctx* c, const uint64_t* rows, uint64_t n, uint64_t* out) {
    if (hipSetDevice(c->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    const host::Index& ix = c->index->ix;
    uint32_t shift = 0;
    while ((1ull << shift) < ix.sa_rate) ++shift;
    if ((1ull << shift) != ix.sa_rate || shift < 1 || shift > 8 || ix.extra_rows.size() > 2) return MAPAD_ERR_INVALID;  // the kernel's assumptions
    int rc;
    if ((rc = ensure_sa_uploaded(c))) return rc;
    c->last_locate_rows = n; c->last_locate_steps = 0;
    if (n == 0) return MAPAD_OK;
    if ((rc = c->d_rows.ensure(n))) return rc;
    if ((rc = c->d_pos.ensure(n))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_rows.p, rows, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_steps.p, 0, sizeof(unsigned long long), c->stream));
    LocateDev q{};
    q.rows = c->d_rows.p; q.out = c->d_pos.p; q.n_rows = n; q.sa_sample = c->d_sa.p; q.x_counts = ix.x_counts.empty() ? nullptr : c->d_xc.p;
    q.sa_shift = shift; q.steps = c->d_steps.p;
    int k = 0;
    for (const auto& kv : ix.extra_rows) { q.extra_row[k] = kv.first; q.extra_val[k] = kv.second; ++k; }
    for (; k < 2; ++k) { q.extra_row[k] = ~0ull; q.extra_val[k] = 0; }
    HIP_TRY(hipEventRecord(c->lev[0], c->stream));
    hipLaunchKernelGGL(locate_kernel, dim3((uint32_t)((n + 15) / 16)), dim3(64), 0, c->stream, c->dix, q);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->lev[1], c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->d_pos.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&c->last_locate_steps, c->d_steps.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MAPAD_OK;
}
}  // namespace

extern "C" {
int mapad_sa_locate(mapad_ctx_t* ctx, const uint64_t* rows, uint64_t n, uint64_t* out) {
    if (!ctx || (n && (!rows || !out))) return MAPAD_ERR_INVALID;
    return locate_rows(ctx, rows, n, out);
}
int mapad_last_locate_info(mapad_ctx_t* ctx, float* kernel_ms, uint64_t* rows, uint64_t* lf_steps) {
    if (!ctx || !kernel_ms || !rows || !lf_steps) return MAPAD_ERR_INVALID;
    *rows = ctx->last_locate_rows; *lf_steps = ctx->last_locate_steps; *kernel_ms = 0.0f;
    if (ctx->last_locate_rows && ctx->lev[0]) HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->lev[0], ctx->lev[1]));
    return MAPAD_OK;
}
// the device half of intervals_to_bam for one batch result: one CoordRec per read (records_kernel), in host memory
struct mapad_coords {
    std::vector<CoordRec> v;     // MAPAD_RECORDS_TEXT=host: the coordinates; strings and MAPQ from host::records_from_coords
    bool device_text = false;    // default: the text kernel's products; flags and MAPQ from host::records_from_device_text
    std::vector<DevRecord> recs;
    std::vector<char> text;
    std::vector<float> pairs;
    uint64_t n = 0;
};
// The post-search kernels over read-ordered hits on the device: records_kernel (coordinates) and, `device_text`, text_kernel (CIGAR / MD / XA bytes and the pairs
// of the mapping quality) into the given buffers; used[0] = text bytes, used[1] = pairs.  Synchronises rstream.
struct RecordBufs {
    DevBuf<CoordRec>& coords;
    DevBuf<DevRecord>& out;
    DevBuf<char>& text;
    DevBuf<float>& pairs;
};
static int run_record_kernels(mapad_ctx_t* ctx, const uint64_t* d_begin, const HitRec* d_hits, const uint32_t* d_ops, uint64_t n, uint64_t seed, hipStream_t rstream, RecordBufs bufs,
                              bool device_text, unsigned long long used[2]) {
    const host::Index& ix = ctx->index->ix;
    uint32_t shift = 0;
    while ((1ull << shift) < ix.sa_rate) ++shift;
    if ((1ull << shift) != ix.sa_rate || shift < 1 || shift > 8 || ix.extra_rows.size() > 2) return MAPAD_ERR_INVALID;  // the kernels' assumptions
    int rc;
    used[0] = used[1] = 0;
    if ((rc = bufs.coords.ensure(n))) return rc;
    PostIndex Q{};
    Q.ix = ctx->dix; Q.sa_sample = ctx->d_sa.p; Q.x_counts = ix.x_counts.empty() ? nullptr : ctx->d_xc.p; Q.sa_shift = shift;
    int k = 0;
    for (const auto& kv : ix.extra_rows) { Q.extra_row[k] = kv.first; Q.extra_val[k] = kv.second; ++k; }
    for (; k < 2; ++k) { Q.extra_row[k] = ~0ull; Q.extra_val[k] = 0; }
    Q.n_contigs = (uint32_t)ix.contigs.size(); Q.contig_start = ctx->d_contigs.p; Q.contig_end = ctx->d_contigs.p + ix.contigs.size();
    HIP_TRY(hipEventRecord(ctx->lev[0], rstream));
    hipLaunchKernelGGL(records_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, rstream, Q, d_begin, d_hits, d_ops, n, seed, bufs.coords.p);
    HIP_TRY(hipGetLastError());
    ctx->last_locate_rows = n; ctx->last_locate_steps = 0;
    if (!device_text) { HIP_TRY(hipEventRecord(ctx->lev[1], rstream)); return MAPAD_OK; }
    // the text half on the device: CIGAR / MD / XA bytes and the pairs of the mapping quality into two pools; what leaves the device is one 88-byte record
    // per read plus the text (typically "50M" + "50": a dozen bytes per read)
    if ((rc = bufs.out.ensure(n))) return rc;
    if ((rc = bufs.text.ensure(std::max<size_t>(bufs.text.cap, (size_t)n * 24 + (1u << 16))))) return rc;
    if ((rc = bufs.pairs.ensure(std::max<size_t>(bufs.pairs.cap, (size_t)n * 2 + 4096)))) return rc;
    for (int attempt = 0; attempt < 8; ++attempt) {
        if ((rc = zero_async(rstream, ctx->d_t_cur.p, 16))) return rc;
        TextDev TQ{};
        TQ.T.os_pos = ctx->d_os_pos.p; TQ.T.os_sym = ctx->d_os_sym.p; TQ.T.n_os = ctx->n_os; TQ.T.name_off = ctx->d_name_off.p; TQ.T.names = (const char*)ctx->d_names.p;
        TQ.hit_begin = d_begin; TQ.hits = d_hits; TQ.ops = d_ops; TQ.coords = bufs.coords.p; TQ.n_reads = n;
        TQ.text = bufs.text.p; TQ.pairs = bufs.pairs.p; TQ.cursors = ctx->d_t_cur.p; TQ.text_cap = bufs.text.cap; TQ.pair_cap = bufs.pairs.cap / 2; TQ.out = bufs.out.p;
        hipLaunchKernelGGL(text_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, rstream, TQ);
        HIP_TRY(hipGetLastError());
        if (!ctx->h_t_used.resize(2)) return MAPAD_ERR_NOMEM;
        hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, rstream, (const uint32_t*)ctx->d_t_cur.p, (uint32_t*)ctx->h_t_used.data(), 4u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(rstream));
        used[0] = ctx->h_t_used[0]; used[1] = ctx->h_t_used[1];
        if (used[0] <= TQ.text_cap && used[1] <= TQ.pair_cap) break;
        if (used[0] > 0xFFFFFFFFull || used[1] > 0xFFFFFFFFull) return MAPAD_ERR_INVALID;  // more than 4 GiB of record text in one batch
        if (attempt == 7) return MAPAD_ERR_NOMEM;
        if (used[0] > TQ.text_cap && (rc = bufs.text.ensure((size_t)used[0] + (1u << 16)))) return rc;
        if (used[1] > TQ.pair_cap && (rc = bufs.pairs.ensure((size_t)used[1] * 2 + 4096))) return rc;
    }
    HIP_TRY(hipEventRecord(ctx->lev[1], rstream));
    return MAPAD_OK;
}
static int record_coords_gpu(mapad_ctx_t* ctx, const mapad_batch_result_t* res, uint64_t seed, mapad_coords& co) {
    std::vector<CoordRec>& coords = co.v;
    co.n = res->n_reads;
    const char* tm = std::getenv("MAPAD_RECORDS_TEXT");
    co.device_text = !(tm && std::strcmp(tm, "host") == 0);
    int rc;
    if ((rc = ensure_sa_uploaded(ctx))) return rc;
    const uint64_t n = res->n_reads;
    if (!co.device_text) coords.resize(n);
    if (!n) return MAPAD_OK;
    // The hits are normally still on the device, laid out in read order by the collect of the fetch that produced `res` (the batch slot has not
    // been launched again since): the kernel reads them where they are.  Otherwise (an older result) they go back over PCIe first.
    const HostResult* hr = LiveResults::get().has(res) ? reinterpret_cast<const HostResult*>(res) : nullptr;  // a result of this library: pub is the first member
    const uint64_t* d_begin; const HitRec* d_hits; const uint32_t* d_ops;
    hipStream_t rstream = ctx->stream;
    if (hr && hr->owner == ctx && hr->slot >= 0 && hr->slot < kMaxDepth && ctx->bs[hr->slot].gen == hr->gen && ctx->bs[hr->slot].compacted && env_u32("MAPAD_RECORDS_RESIDENT", 1)) {
        const BatchSlot& RS = ctx->bs[hr->slot];
        d_begin = RS.d_c_hit_begin.p; d_hits = RS.d_c_hits.p; d_ops = RS.d_c_ops.p;
        rstream = RS.stream;  // the stream that wrote them (idle since the fetch)
    } else {
        if ((rc = ctx->d_r_begin.ensure(n + 1))) return rc;
        if ((rc = ctx->d_r_hits.ensure(std::max<uint64_t>(res->n_hits, 1)))) return rc;
        if ((rc = ctx->d_r_ops.ensure(std::max<uint64_t>(res->n_ops, 1)))) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->d_r_begin.p, res->hit_begin, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (res->n_hits) HIP_TRY(hipMemcpyAsync(ctx->d_r_hits.p, res->hits, res->n_hits * sizeof(HitRec), hipMemcpyHostToDevice, ctx->stream));
        if (res->n_ops) HIP_TRY(hipMemcpyAsync(ctx->d_r_ops.p, res->ops, res->n_ops * 4, hipMemcpyHostToDevice, ctx->stream));
        d_begin = ctx->d_r_begin.p; d_hits = ctx->d_r_hits.p; d_ops = ctx->d_r_ops.p;
    }
    unsigned long long used[2] = {0, 0};
    if ((rc = run_record_kernels(ctx, d_begin, d_hits, d_ops, n, seed, rstream, RecordBufs{ctx->d_r_out, ctx->d_t_out, ctx->d_t_text, ctx->d_t_pairs}, co.device_text, used))) return rc;
    if (!co.device_text) {
        HIP_TRY(hipMemcpyAsync(coords.data(), ctx->d_r_out.p, n * sizeof(CoordRec), hipMemcpyDeviceToHost, rstream));
        HIP_TRY(hipStreamSynchronize(rstream));
        return MAPAD_OK;
    }
    co.recs.resize(n); co.text.resize(used[0]); co.pairs.resize(2 * used[1]);
    HIP_TRY(hipMemcpyAsync(co.recs.data(), ctx->d_t_out.p, n * sizeof(DevRecord), hipMemcpyDeviceToHost, rstream));
    if (used[0]) HIP_TRY(hipMemcpyAsync(co.text.data(), ctx->d_t_text.p, used[0], hipMemcpyDeviceToHost, rstream));
    if (used[1]) HIP_TRY(hipMemcpyAsync(co.pairs.data(), ctx->d_t_pairs.p, 2 * used[1] * sizeof(float), hipMemcpyDeviceToHost, rstream));
    HIP_TRY(hipStreamSynchronize(rstream));
    return MAPAD_OK;
}
static mapad_records_t* records_from(const host::Index& ix, const mapad_params_t& prm, const mapad_batch_result_t& res, const uint16_t* in_flags, const mapad_coords& co) {
    if (co.device_text) return host::records_from_device_text(prm, res.n_reads, in_flags, co.recs.data(), co.text.data(), co.text.size(), co.pairs.data());
    return host::records_from_coords(ix, prm, res, in_flags, co.v.data());
}

int mapad_records_device(mapad_ctx_t* ctx, uint64_t seed, void** d_records, void** d_text, void** d_pairs, uint64_t* text_bytes, uint64_t* n_pairs) {
    if (!ctx || !d_records || !d_text || !d_pairs || !text_bytes || !n_pairs) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    int rc;
    if ((rc = compact_last(ctx))) return rc;
    if ((rc = ensure_sa_uploaded(ctx))) return rc;
    BatchSlot& S = ctx->bs[ctx->view];
    const uint64_t n = S.last.n_reads;
    unsigned long long used[2] = {0, 0};
    if (n && (rc = run_record_kernels(ctx, S.d_c_hit_begin.p, S.d_c_hits.p, S.d_c_ops.p, n, seed, S.stream, RecordBufs{S.d_rec_coords, S.d_rec_out, S.d_rec_text, S.d_rec_pairs}, true, used))) return rc;
    *d_records = S.d_rec_out.p; *d_text = S.d_rec_text.p; *d_pairs = S.d_rec_pairs.p; *text_bytes = used[0]; *n_pairs = used[1];
    return MAPAD_OK;
}

int mapad_hits_to_records_gpu(mapad_ctx_t* ctx, const mapad_batch_result_t* res, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets,
                              const uint16_t* in_flags, uint64_t seed, mapad_records_t** out) {
    if (!ctx || !res || !out || (res->n_reads && (!seqs || !quals || !offsets))) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    try {
        mapad_coords co;
        const int rc = record_coords_gpu(ctx, res, seed, co);
        if (rc) return rc;
        *out = records_from(ctx->index->ix, ctx->params, *res, in_flags, co);
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_hits_to_records_gpu: %s\n", e.what());
        return MAPAD_ERR_INVALID;
    }
}
int mapad_hits_to_coords_gpu(mapad_ctx_t* ctx, const mapad_batch_result_t* res, uint64_t seed, mapad_coords_t** out) {
    if (!ctx || !res || !out) return MAPAD_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return MAPAD_ERR_NO_DEVICE;
    try {
        auto c = std::make_unique<mapad_coords>();
        const int rc = record_coords_gpu(ctx, res, seed, *c);
        if (rc) return rc;
        *out = c.release();
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_hits_to_coords_gpu: %s\n", e.what());
        return MAPAD_ERR_INVALID;
    }
}
int mapad_coords_to_records(const mapad_index_t* idx, const mapad_params_t* params, const mapad_batch_result_t* res, const uint16_t* in_flags, const mapad_coords_t* coords,
                            mapad_records_t** out) {
    if (!idx || !params || !res || !coords || !out || coords->n != res->n_reads) return MAPAD_ERR_INVALID;
    try {
        *out = records_from(idx->ix, *params, *res, in_flags, *coords);
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_coords_to_records: %s\n", e.what());
        return MAPAD_ERR_INVALID;
    }
}
void mapad_coords_free(mapad_coords_t* c) { delete c; }
}  // extern "C"

extern "C" {
// ---- post-search ----------------------------------------------------------------------------------------------------------
int mapad_hits_to_records(const mapad_index_t* idx, const mapad_params_t* params, const mapad_batch_result_t* res, const uint8_t* seqs, const uint8_t* quals,
                          const uint64_t* offsets, const uint16_t* in_flags, uint64_t seed, mapad_records_t** out) {
    if (!idx || !params || !res || !out || (res->n_reads && (!seqs || !quals || !offsets))) return MAPAD_ERR_INVALID;
    try {
        *out = host::hits_to_records(idx->ix, *params, *res, seqs, quals, offsets, in_flags, seed);
        return MAPAD_OK;
    } catch (const std::bad_alloc&) { return MAPAD_ERR_NOMEM; } catch (const std::exception& e) {
        std::fprintf(stderr, "mapad_hits_to_records: %s\n", e.what());
        return MAPAD_ERR_INVALID;
    }
}
void mapad_records_free(mapad_records_t* r) { host::free_records(r); }
// host_postproc.hpp: seed_for(seed, read_idx, call) advances by one golden-ratio step per read
uint64_t mapad_records_seed_at(uint64_t seed, uint64_t first_read_index) { return seed + first_read_index * 0x9E3779B97F4A7C15ull; }

}  // extern "C"
