// tail_bench.cpp — DEV/TEST-ONLY timing harness for the host tail (mapad_amd/csrc/host_tail.hpp): maps selected reads from scratch on host threads with the
// product's own search step, as the tail workers do, without a GPU in the loop.  Built by profiles/dev/tail_bench.py; never loaded by the product.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mapad_amd.h"
#include "../../mapad_amd/csrc/darray_core.hpp"
#include "../../mapad_amd/csrc/host_models.hpp"
#include "../../mapad_amd/csrc/host_tail.hpp"

using namespace mapad;

extern "C" double tail_bench(const uint64_t* blocks, uint64_t n_blocks, uint64_t n, const uint64_t* less8, const uint64_t* sentinel2, const mapad_params_t* p,
                             const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads, const uint32_t* sel, uint32_t n_sel,
                             uint32_t threads, uint64_t max_pops, uint32_t interleave, uint64_t* out_pops, uint32_t* out_status, uint64_t* out_digest, double* out_secs) {
    DevIndex ix;
    ix.blocks = blocks; ix.n = n; ix.n_blocks = n_blocks;
    for (int i = 0; i < 8; ++i) ix.less[i] = less8[i];
    ix.sentinel[0] = sentinel2[0]; ix.sentinel[1] = sentinel2[1];
    host::HostTables t = host::make_tables(*p);
    uint32_t lmax = 1;
    for (uint64_t i = 0; i < n_reads; ++i) { const uint32_t l = (uint32_t)(offsets[i + 1] - offsets[i]); lmax = std::max(lmax, l); if (l) host::add_length(*p, t, (int)l); }
    DevParams P{};
    P.sdm_table = t.sdm.data(); P.table_base = t.table_base.data(); P.reject_thr = t.reject_thr.data();
    P.nq = t.nq; P.bound_kind = p->bound_kind; P.cutoff = p->cutoff; P.repr_mm = t.repr_mm;
    P.gap_open = p->penalty_gap_open; P.gap_extend = p->penalty_gap_extend; P.gap_dist_ends = p->gap_dist_ends; P.max_num_gaps_open = p->max_num_gaps_open;
    P.start_at_end = p->model_kind == MAPAD_MODEL_SIMPLE_ADNA; P.stack_limit_abort = p->stack_limit_abort;
    P.stack_limit = p->stack_limit ? p->stack_limit : 2000000u; P.edit_tree_limit = p->edit_tree_limit ? p->edit_tree_limit : 10000000u;

    std::atomic<uint32_t> next{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    {   // per call (the harness compares settings inside one process)
        const char* e = std::getenv("MAPAD_TAIL_PREFETCH");
        g_host_prefetch = HostPrefetch{};
        if (e && e[0] >= '0' && e[0] <= '1') { g_host_prefetch.sift_lookahead = e[0] - '0'; if (e[1] == '0' || e[1] == '1') g_host_prefetch.next_pop = e[1] == '1'; }
    }
    for (uint32_t w = 0; w < threads; ++w) th.emplace_back([&, w] {
        host::tail_pin_worker(w);
        std::vector<uint8_t> qc(2 * (lmax + 1));
        std::vector<float> d(lmax + 1), dnear(lmax + 1), pen(lmax + 1), chain(lmax + 1);
        host::TailScratch sc;
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= n_sel) break;
            const uint64_t i = sel[k], off = offsets[i];
            const int L = (int)(offsets[i + 1] - off);
            d_array_scalar(ix, P, seqs + off, quals + off, L, pen.data(), chain.data(), d.data());
            read_setup(seqs + off, quals + off, d.data(), L, qc.data(), dnear.data(), 0, 1);
            if (!sc.ensure(P.stack_limit + 10, P.edit_tree_limit + 10, lmax)) { out_status[k] = 0xFFFFFFFFu; continue; }
            Arena A;
            A.top = sc.top.data() + 1; A.heap = sc.heap + 1; A.nodes = sc.nodes; A.hits = sc.hits.data(); A.hit_ops = sc.hit_ops.data(); A.scratch = sc.scratch.data();
            A.heap_cap = sc.heap_cap; A.node_cap = sc.node_cap; A.hit_ops_cap = (uint32_t)sc.hit_ops.size();
            A.pc = sc.pc;
            const ReadIn rd{qc.data(), dnear.data(), L, P.reject_thr[L], P.table_base[L]};
            SearchState st;
            const auto r0 = std::chrono::steady_clock::now();
            host::tail_search(ix, P, rd, A, st, max_pops, nullptr);
            out_secs[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
            out_pops[k] = st.c_pop; out_status[k] = st.status;
            uint64_t h = 1469598103934665603ull;  // FNV over what the batch would get back
            auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
            mix(st.status); mix(st.c_esearch); mix(st.c_push); mix(st.c_pop); mix(st.c_node); mix(st.c_hits); mix(st.n_hits);
            for (uint32_t q = 0; q < st.n_hits; ++q) { const HitRec& hr = sc.hits[q]; mix(hr.lower); mix(hr.lower_rev); mix(hr.size); uint32_t sb; std::memcpy(&sb, &hr.score, 4); mix(sb); mix(hr.n_ops); }
            for (uint32_t q = 0; q < st.hit_ops_used; ++q) mix(sc.hit_ops[q]);
            out_digest[k] = h;
        }
    });
    for (auto& x : th) x.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
