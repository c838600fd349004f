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

constexpr int kHeavyTop = 1023;  // heap levels 0-9
using HeavyArena = ArenaT<true, kHeavyTop>;
using HeavyRead = ReadInT<true>;

// LDS of a heavy wavefront: [kHeavyTop + 1 heap slots][2 bytes per read position: class, quality][D array][scratch]
constexpr uint32_t kHeavyScratchBytes = 4096;
inline __host__ __device__ uint32_t heavy_lds_bytes(uint32_t lmax) { return (kHeavyTop + 1) * 8 + ((2 * lmax + 15) & ~15u) + ((4 * lmax + 15) & ~15u) + kHeavyScratchBytes; }

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
template <bool CONT, int MODE>
__global__ void __launch_bounds__(64) heavy_kernel(DevIndex ix, DevParams P, BatchDev B, ArenaPool AP, const GrowPools* GP, uint32_t lmax, int tier) {
    const int lane = threadIdx.x & 63;
    // MODE 0: `tier` = the quad stage whose suspended reads this launch continues (its own list and work counter)
    const uint32_t n_items = MODE == 0 ? min(B.cursors[CUR_HEAVY_N + tier], B.heavy_cap) : B.cursors[CUR_OVF + 2 * (tier - 1)];
    uint32_t* work = MODE == 0 ? &B.cursors[CUR_HEAVY_WORK + tier] : &B.cursors[CUR_WORK + 2 * tier];
    if (*(volatile uint32_t*)work >= n_items) return;  // nothing (left) to do: the usual case
    extern __shared__ __attribute__((aligned(16))) uint8_t heavy_lds[];
    MAPAD_LDS uint8_t* lds = (MAPAD_LDS uint8_t*)heavy_lds;
    MAPAD_LDS HeapEntry* top = (MAPAD_LDS HeapEntry*)lds + 1;
    MAPAD_LDS uint8_t* near_qc = lds + (kHeavyTop + 1) * sizeof(HeapEntry);
    MAPAD_LDS float* near_d = (MAPAD_LDS float*)(near_qc + ((2 * lmax + 15) & ~15u));
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
            const ArenaT<false> a = carve<false>(AP, slot);
            A.heap = a.heap; A.nodes = a.nodes; A.hits = a.hits; A.hit_ops = a.hit_ops; A.scratch = a.scratch;
            A.heap_cap = a.heap_cap; A.node_cap = a.node_cap; A.hit_ops_cap = a.hit_ops_cap; A.grown = 0;
        }
        A.top = top;
        const uint64_t off = B.offsets[read];
        HeavyRead rd{near_qc, near_d, 0, 0.0f, 0};
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
            // one step of the search by one lane (the wavefront-cooperative step replaces this)
            uint32_t c = 0;
            if (lane == 0) c = search_step<1, CONT, true>(ix, P, rd, A, st, 0, NoGrow()) ? 1u : 0u;
            {
                uint32_t* sw = (uint32_t*)&st;
#pragma unroll
                for (int k = 0; k < (int)(sizeof(SearchState) / 4); ++k) sw[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sw[k]);
            }
            cont = __builtin_amdgcn_readfirstlane((int)c) != 0;
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd((unsigned long long*)(B.cursors + CUR_HEAVY_POPS), (unsigned long long)(st.c_pop - pops0));
        finalize_read<64>(B, rd, A, st, read, lane, MODE == 0 ? 1 : tier);  // MODE 0: a read that no class can hold goes on to the full-limit stage (list 1)
        if (MODE == 0) release_grown<64>(GP, A.grown, lane, grow.foreign);
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (MODE == 1) release_slot(AP, slot);
}
