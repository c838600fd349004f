// host_tail.hpp — the host half of the heavy tail: reads that pass a pop budget on the GPU are finished by host threads.
//
// The reference absorbs its heaviest reads — those that reach STACK_LIMIT / EDIT_TREE_LIMIT and go on searching under pop_min eviction for
// millions of steps (src/map/mapping.rs:52-54,1358-1380) — on CPU threads at a fraction of a microsecond per pop.  On the GPU one read is a serial
// chain at ~5.5 us per pop whatever the shape (quad or wavefront: DESIGN.md section 4), so a few hundred such reads held a launch for minutes after the
// other million were done.  A read that has made `tail_pops` pops (default 2^19: none of the 50 bp workloads gets there) is therefore given up by its
// quad — the read's position data and D array go into a record in page-locked host memory, written by the kernel while it runs — and a host thread
// maps it FROM SCRATCH with the product's own search step (search_core.hpp compiled for the host: the same source as the kernel, scalar rank queries
// on the same 64-byte blocks; never the test oracle), beside the GPU's bulk of this and the next batches.  Results go back into the batch's device
// pools before the order-preserving collect, so everything downstream (collect, gather, post-search) is unchanged; hits, edit tracks and event counters
// are bit-identical to what the GPU stages produce for the same read (tests/test_gpu_tail.py runs both ways against the oracle).
// This is a stage of the GPU path, not a fallback: without a gfx950 device there is no context and nothing runs (mapad_ctx_create).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <queue>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <map>
#include <string>

#include "host_cpus.hpp"
#include "host_models.hpp"
#include "search_core.hpp"

namespace mapad {
namespace host {

// Header of a hand-over record (page-locked host memory, written by the quad that gives the read up; mapad_amd.hip: hand_to_host).  It is followed by
// qc[2 * L] (base class, quality per position: ReadIn::qc) at byte 16 and d[L] (the read's D array) at byte 16 + ((2 * lmax + 15) & ~15).
struct TailRecord {
    uint32_t ready;  // == the launch's number in this ring (TailBatch::gen, never 0) once the rest of the record is visible (system-scope store behind the payload's)
    uint32_t read;   // read number inside the batch
    uint32_t L;
    uint32_t pops;   // pops the GPU had made when it gave the read up
};
// Round 5: the record may carry the search itself.  A read that sits in a grown arena when it is handed over (and has found no hit yet) leaves that arena to the
// host instead of giving it back: heap top written out of LDS into the arena, the SearchState into this block at the end of the record.  A worker copies heap and
// nodes over PCIe, releases the arena and CONTINUES the search where the GPU stopped — a third of the host tail's pops used to be repeats of the GPU's
// (1 M reads of the C5 mix: 0.73 G of 1.68 G).  `grown` == 0: no state, the read is mapped from scratch as before.
struct TailState {
    uint32_t grown;     // (class + 1) << 27 | arena index (mapad_amd.hip: kGrownShift), 0 = none
    uint32_t reserved[3];
    SearchState st;
};
static_assert(sizeof(SearchState) == 64 && sizeof(TailState) == 80, "hand-over state block");
MAPAD_HD uint32_t tail_state_offset(uint32_t lmax) { return (16u + ((2u * lmax + 15u) & ~15u) + 4u * lmax + 15u) & ~15u; }
MAPAD_HD uint32_t tail_record_stride(uint32_t lmax) { return (tail_state_offset(lmax) + (uint32_t)sizeof(TailState) + 63u) & ~63u; }

struct TailResult {
    uint32_t read = 0, status = 0;
    uint32_t e_search = 0, n_push = 0, n_pop = 0, n_node = 0, n_hits = 0;
    uint32_t gpu_e_search = 0, gpu_n_push = 0, gpu_n_pop = 0, gpu_n_node = 0;  // of these, what the GPU had done when a continued read changed hands (0: mapped from scratch)
    std::vector<HitRec> hits;   // BinaryHeap array order; ops_off relative to `ops`
    std::vector<uint32_t> ops;
};

// Worker i of the process is kept inside one last-level-cache domain of the machine (on the round-4 GPU box: 16 CCDs of 8 cores and 32 MB L3 each — with 16
// workers one per CCD, so a read's 16 MB heap has an L3 to itself), chosen round-robin over the domains this process may run on; memory a worker touches first is
// then local to its socket and stays so.  Left to the scheduler, 16 threads under a CPU-time quota wander over 256 hardware threads and two sockets.
// MAPAD_TAIL_PIN=0 leaves the threads unpinned; a machine with one domain (or an unreadable topology) is left alone.
inline void tail_pin_worker(unsigned i) {
    const char* e = std::getenv("MAPAD_TAIL_PIN");
    if (e && e[0] == '0') return;
    static const std::vector<cpu_set_t> domains = [] {
        std::vector<cpu_set_t> out;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
        std::map<std::string, size_t> seen;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &allowed)) continue;
            char path[128], key[256] = {0};
            std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
            FILE* f = std::fopen(path, "r");
            if (!f) return std::vector<cpu_set_t>();
            const bool ok = std::fgets(key, sizeof key, f) != nullptr;
            std::fclose(f);
            if (!ok) return std::vector<cpu_set_t>();
            auto it = seen.find(key);
            if (it == seen.end()) { it = seen.emplace(key, out.size()).first; cpu_set_t z; CPU_ZERO(&z); out.push_back(z); }
            CPU_SET(c, &out[it->second]);
        }
        return out;
    }();
    if (domains.size() < 2) return;
    (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &domains[i % domains.size()]);
}
// MAPAD_TAIL_PREFETCH (search_core.hpp: HostPrefetch), read once
inline void tail_read_prefetch_env() {
#if !defined(__HIP_DEVICE_COMPILE__)
    static const bool once = [] {
        const char* e = std::getenv("MAPAD_TAIL_PREFETCH");
        if (e && e[0] >= '0' && e[0] <= '1') { g_host_prefetch.sift_lookahead = e[0] - '0'; if (e[1] == '0' || e[1] == '1') g_host_prefetch.next_pop = e[1] == '1'; }
        return true;
    }();
    (void)once;
#endif
}

// Worker threads shared by every context of the process (a read at the reference's limits keeps a thread busy for seconds and its arena is 336 MB, so
// there is one pool, as wide as this process's CPU share of the machine: MAPAD_TAIL_THREADS overrides).  Started with the first task.
// Tasks carry a weight — the pops the GPU had made when it gave the read up — and the heaviest waiting task is served first: the search cost of these reads is
// heavy-tailed, so the pops so far predict the pops to come, and the batch's finish time is set by its longest read.
class TailWorkers {
public:
    static TailWorkers& instance() { static TailWorkers* w = new TailWorkers(); return *w; }  // never destroyed: workers may outlive static destructors
    void submit(std::function<void()> f, uint64_t weight = 0) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (threads_.empty()) start();
            q_.emplace(std::make_pair(weight, ~seq_++), std::move(f));  // equal weights: first come, first served
            pending_.fetch_add(1, std::memory_order_relaxed);
        }
        cv_.notify_all();  // (not notify_one: the thread it wakes may be one beyond limit_, which goes back to sleep without the task)
    }
    unsigned size() {
        std::lock_guard<std::mutex> g(mu_);
        return threads_.empty() ? wanted() : limit_;
    }
    // "This process is one rank of `lw` on its node" (0: back to LOCAL_WORLD_SIZE / MAPAD_LOCAL_WORLD_SIZE): the workers that take tasks from now on are this rank's
    // part of the node's CPU share (wanted()); threads beyond it sleep, missing ones are started.  Returns the count.  (mapad_tail_set_local_world: bench.py measures
    // C5 as one rank of eight on a one-GPU box with it.)
    unsigned set_local_world(unsigned lw) {
        std::lock_guard<std::mutex> g(mu_);
        lw_override().store(lw, std::memory_order_relaxed);
        const unsigned n = wanted();
        limit_ = n;
        if (!threads_.empty()) start();  // (adds the missing threads, if any)
        cv_.notify_all();
        return n;
    }
    // tasks waiting or running: what a launch's dispatcher publishes to the kernel, which hands a read over on a dry arena class only while this is small
    uint32_t pending() const { return pending_.load(std::memory_order_relaxed); }
    // One process per GPU (bench.py --gpus N, `mapad-amd worker`): the ranks of a node share its CPUs, so each takes its part of the share and pins its
    // workers to its own cache domains (LOCAL_WORLD_SIZE / LOCAL_RANK as torchrun sets them; MAPAD_LOCAL_WORLD_SIZE / MAPAD_LOCAL_RANK override).
    static unsigned local_world() { const unsigned o = lw_override().load(std::memory_order_relaxed); return o ? o : env_uint("MAPAD_LOCAL_WORLD_SIZE", env_uint("LOCAL_WORLD_SIZE", 1)); }
    static unsigned local_rank() { return env_uint("MAPAD_LOCAL_RANK", env_uint("LOCAL_RANK", 0)); }
private:
    static std::atomic<unsigned>& lw_override() { static std::atomic<unsigned> v{0}; return v; }
    static unsigned env_uint(const char* name, unsigned dflt) {
        const char* e = std::getenv(name);
        if (!e || !e[0]) return dflt;
        const unsigned long v = std::strtoul(e, nullptr, 10);
        return v > 0 || e[0] == '0' ? (unsigned)v : dflt;
    }
    static unsigned wanted() {
        const char* e = std::getenv("MAPAD_TAIL_THREADS");
        if (e && e[0]) { const unsigned n = (unsigned)std::strtoul(e, nullptr, 10); return n ? n : 8; }
        const unsigned lw = std::max(1u, local_world());
        // this rank's part of the node's share, less an eighth: the launch's dispatcher thread, the HIP runtime's helpers and the caller run beside the workers, and a
        // cgroup pushed over its quota is throttled as a whole (16 CPUs: 14 workers gave the wall time of 16 with 6 throttled periods instead of 300, same box)
        const unsigned share = std::max(1u, (cpu_share() + lw - 1) / lw);
        return share >= 4 ? share - std::max(1u, share / 8) : share;
    }
    void start() {  // (mu_ held) threads 0 .. wanted() - 1 exist afterwards; worker i takes tasks while i < limit_
        const unsigned n = wanted(), first = local_rank() * n;  // pinning: this rank's workers take the domains behind those of the ranks before it
        tail_read_prefetch_env();
        limit_ = n;
        for (unsigned i = (unsigned)threads_.size(); i < n; ++i) {
            threads_.emplace_back([this, i, first] {
                tail_pin_worker(first + i);
                for (;;) {
                    std::function<void()> f;
                    {
                        std::unique_lock<std::mutex> l(mu_);
                        cv_.wait(l, [&] { return !q_.empty() && i < limit_; });
                        f = std::move(const_cast<Task&>(q_.top()).second);
                        q_.pop();
                    }
                    f();
                    pending_.fetch_sub(1, std::memory_order_relaxed);
                }
            });
            threads_.back().detach();
        }
    }
    using Task = std::pair<std::pair<uint64_t, uint64_t>, std::function<void()>>;
    struct Lighter { bool operator()(const Task& a, const Task& b) const { return a.first < b.first; } };
    std::mutex mu_;
    std::condition_variable cv_;
    std::priority_queue<Task, std::vector<Task>, Lighter> q_;
    uint64_t seq_ = 0;
    std::atomic<uint32_t> pending_{0};
    unsigned limit_ = 0;  // workers that take tasks (<= threads_.size())
    std::vector<std::thread> threads_;
};

// A worker thread's arena with the reference's full limits.  malloc'ed, never cleared: the search writes before it reads, and untouched pages cost nothing.
struct TailScratch {
    HeapEntry* heap = nullptr;
    Node* nodes = nullptr;
    uint32_t heap_cap = 0, node_cap = 0, lmax = 0;
    std::vector<HeapEntry> top;
    std::vector<HitRec> hits;
    std::vector<uint32_t> hit_ops;
    std::vector<uint16_t> scratch;
    uint64_t pc[8];
    bool ensure(uint32_t hc, uint32_t nc, uint32_t lm) {
        // a sift loads the grandchildren of its hole before it looks at the heap's length (search_core.hpp: mm_trickle_down): slots up to 2 * heap_len + 6 are
        // read (and ignored), so the allocation is twice the capacity — on the device those reads fall into the arena's node area
        // (2 MB alignment + MADV_HUGEPAGE: a sift over a 2 M-entry heap and the node lookups behind evictions are random accesses over 16 MB and 320 MB; with 4 KB
        //  pages nearly every one of them also misses the TLB)
        auto big = [](size_t bytes) -> void* {
            const size_t sz = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            void* p = std::aligned_alloc((size_t)2 << 20, sz);
#if defined(MADV_HUGEPAGE) && !defined(MAPAD_TAIL_NO_HUGEPAGE)
            if (p) (void)madvise(p, sz, MADV_HUGEPAGE);
#endif
            return p;
        };
        if (hc > heap_cap) { std::free(heap); heap = (HeapEntry*)big((std::max<size_t>(2 * (size_t)hc, HeapLayout<kTop>::phys_end(hc)) + 64) * sizeof(HeapEntry)); heap_cap = heap ? hc : 0; }
        if (nc > node_cap) { std::free(nodes); nodes = (Node*)big(((size_t)nc + 1) * sizeof(Node)); node_cap = nodes ? nc : 0; }
        if (!heap || !nodes) return false;
        if (lm > lmax || top.empty()) {
            lmax = std::max(lm, lmax);
            top.assign(kTop + 1 + 8, HeapEntry{});
            hits.assign(kMaxHits, HitRec{});
            hit_ops.assign((size_t)kMaxHits * (lmax + 32), 0);
            scratch.assign(2 * ((size_t)lmax + 2), 0);
        }
        return true;
    }
    ~TailScratch() { std::free(heap); std::free(nodes); }
};

// One launch's tail: shared by the launch's dispatcher thread, the workers and the context (shared_ptr: a worker may still be inside a read when the
// context moves on or goes away).
struct TailBatch {
    DevIndex ix{};          // host view of the index (blocks in host memory)
    DevParams P{};          // points into *tables
    std::shared_ptr<const HostTables> tables;
    const uint8_t* ring = nullptr;  // records in host-coherent page-locked memory
    uint32_t stride = 0, cap = 0, lmax = 0;
    uint32_t gen = 1;               // a record of this launch is ready when its `ready` word holds this number
    uint32_t* ctl = nullptr;        // two words the kernel reads before a hand-over: [0] tasks the workers have waiting or running, [1] records of this launch the dispatcher has picked up
    std::function<bool()> launch_done;  // has the launch that writes this ring ended? (set by the library; counts the hand-overs that arrive while it runs)
    uint32_t seen_live = 0;         // records the dispatcher saw while the launch was still running
    // continuation (TailState): copies heap slots [0, heap_len] (physical, shifted by one) and nodes [0, tree_entries) of the grown arena into the worker's arena and
    // releases the arena on the device; false = could not (the read is then mapped from scratch and the arena released all the same); null pointers = release
    // only (the worker has no room for the state).  Set by the library.
    std::function<bool(uint32_t grown, uint32_t heap_len, uint32_t tree_entries, HeapEntry* heap_phys, Node* nodes)> fetch_state;
    std::vector<uint8_t> fetched;   // per record: its arena has been taken care of (a cancelled batch releases the others: mapad_amd.hip)
    uint32_t continued = 0;         // reads continued from the GPU's state

    std::mutex mu;
    std::condition_variable cv;
    std::vector<TailResult> results;
    uint32_t dispatched = 0, done = 0;
    uint64_t gpu_pops = 0, host_pops = 0;
    double host_thread_s = 0.0;  // seconds workers spent inside the reads of this batch, summed: against (t_last - t_first) x threads it says whether the host or the GPU's hand-overs set the pace
    double t_first = 0.0, t_last = 0.0;  // steady-clock seconds of the first hand-over seen and of the last read finished
    bool failed = false;

    std::atomic<int64_t> final_count{-1};  // records the launch wrote, known once it has ended
    std::atomic<bool> cancel{false};
    std::thread dispatcher;

    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// The search of one read on a host thread: search_core.hpp's step (the kernel's own source, payload cache on) until the read is done, `max_pops` pops are
// made (0: no bound; the timing harness uses one) or `cancel` is raised.
inline void tail_search(const DevIndex& ix, const DevParams& P, const ReadIn& rd, Arena& A, SearchState& st, uint64_t max_pops, const std::atomic<bool>* cancel) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ix; (void)P; (void)rd; (void)A; (void)st; (void)max_pops; (void)cancel;
#else
    search_init(ix.n, alignment_start_of(P, rd.L), rd, A, st);
    pc_clear(A);
    uint32_t n = 0;
    auto check = [&]() { return (++n & 0xFFFFu) == 0 && ((cancel && cancel->load(std::memory_order_relaxed)) || (max_pops && st.c_pop >= max_pops)); };
    if (P.bound_kind == BOUND_CONTINUOUS) { while (search_step<1, true, false, true>(ix, P, rd, A, st, 0, NoGrow())) { if (check()) break; } }
    else { while (search_step<1, false, false, true>(ix, P, rd, A, st, 0, NoGrow())) { if (check()) break; } }
#endif
}

// k_mismatch_search for one handed-over read, from scratch, by this thread (search_core.hpp: the kernel's own step, payload cache included).
inline void tail_map_read(const std::shared_ptr<TailBatch>& tb, const TailRecord* rec) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)tb; (void)rec;  // (the device pass of hipcc parses host functions too; arena pointers carry device address spaces there)
#else
    thread_local TailScratch sc;
    TailResult r;
    r.read = rec->read;
    const double t_begin = TailBatch::now_s();
    const int L = (int)rec->L;
    bool ok = sc.ensure(tb->P.stack_limit + 10, tb->P.edit_tree_limit + 10, std::max<uint32_t>(tb->lmax, (uint32_t)L));
    uint64_t pops = 0;
    if (!ok && !tb->cancel.load(std::memory_order_relaxed)) {  // no arena for this worker: the read fails (tb->failed below), but a grown arena it came with goes back to its pool
        const TailState* ts0 = reinterpret_cast<const TailState*>(reinterpret_cast<const uint8_t*>(rec) + tail_state_offset(tb->lmax));
        const size_t k0 = (size_t)((reinterpret_cast<const uint8_t*>(rec) - tb->ring) / tb->stride);
        if (ts0->grown != 0 && tb->fetch_state) { (void)tb->fetch_state(ts0->grown, 0, 0, nullptr, nullptr); if (k0 < tb->fetched.size()) tb->fetched[k0] = 1; }
    }
    if (ok && !tb->cancel.load(std::memory_order_relaxed)) {
        const uint8_t* qc = reinterpret_cast<const uint8_t*>(rec) + 16;
        const float* d = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(rec) + 16 + ((2u * tb->lmax + 15u) & ~15u));
        Arena A;
        A.top = sc.top.data() + 1; A.heap = sc.heap + 1; A.nodes = sc.nodes; A.hits = sc.hits.data(); A.hit_ops = sc.hit_ops.data(); A.scratch = sc.scratch.data();
        A.heap_cap = sc.heap_cap; A.node_cap = sc.node_cap; A.hit_ops_cap = (uint32_t)sc.hit_ops.size();
        A.pc = sc.pc;
        const ReadIn rd{qc, d, L, tb->P.reject_thr[L], tb->P.table_base[L]};
        SearchState st;
        const TailState* ts = reinterpret_cast<const TailState*>(reinterpret_cast<const uint8_t*>(rec) + tail_state_offset(tb->lmax));
        bool resumed = false;
        if (ts->grown != 0 && tb->fetch_state) {
            const size_t k = (size_t)((reinterpret_cast<const uint8_t*>(rec) - tb->ring) / tb->stride);
            const SearchState s0 = ts->st;
            const bool fits = s0.n_hits == 0 && s0.heap_len + kStepNodes <= sc.heap_cap && s0.tree_entries + kStepNodes <= sc.node_cap;
            // fetch_state releases the arena whether or not the copy worked; a state this worker cannot take (null pointers) is a release only
            resumed = tb->fetch_state(ts->grown, s0.heap_len, s0.tree_entries, fits ? sc.heap : nullptr, fits ? sc.nodes : nullptr) && fits;
            if (k < tb->fetched.size()) tb->fetched[k] = 1;
            if (resumed) {
                const uint32_t n_top = s0.heap_len < (uint32_t)kTop ? s0.heap_len : (uint32_t)kTop;
                for (uint32_t i = 0; i < n_top; ++i) A.top[i] = A.heap[i];  // heap levels 0-5: the kernel had them in LDS and wrote them into the arena's unused slots
                st = s0;
                r.gpu_e_search = s0.c_esearch; r.gpu_n_push = s0.c_push; r.gpu_n_pop = s0.c_pop; r.gpu_n_node = s0.c_node;
                pc_clear(A);
#if !defined(__HIP_DEVICE_COMPILE__)
                if (tb->P.bound_kind == BOUND_CONTINUOUS) { while (!tb->cancel.load(std::memory_order_relaxed) && search_step<1, true, false, true>(tb->ix, tb->P, rd, A, st, 0, NoGrow())) {} }
                else { while (!tb->cancel.load(std::memory_order_relaxed) && search_step<1, false, false, true>(tb->ix, tb->P, rd, A, st, 0, NoGrow())) {} }
#endif
            }
        }
        if (!resumed) tail_search(tb->ix, tb->P, rd, A, st, 0, &tb->cancel);
        r.status = st.status;
        r.e_search = st.c_esearch; r.n_push = st.c_push; r.n_pop = st.c_pop; r.n_node = st.c_node; r.n_hits = st.c_hits;
        pops = st.c_pop;
        r.hits.assign(sc.hits.begin(), sc.hits.begin() + st.n_hits);
        r.ops.assign(sc.hit_ops.begin(), sc.hit_ops.begin() + st.hit_ops_used);
        ok = st.status != ST_ARENA_OVERFLOW;  // cannot happen: the arena holds the reference's limits
    }
    {
        std::lock_guard<std::mutex> g(tb->mu);
        if (!ok) tb->failed = true;
        if (const char* log = std::getenv("MAPAD_TAIL_LOG")) {  // diagnostics: one line per read the host finished (read, length, pops, start relative to the first hand-over, seconds, cpu)
            if (FILE* f = std::fopen(log, "a")) {
                std::fprintf(f, "%u %d %llu %.3f %.3f %d\n", r.read, L, (unsigned long long)(pops - r.gpu_n_pop), t_begin - tb->t_first, TailBatch::now_s() - t_begin, sched_getcpu());  // pops = the host's own
                std::fclose(f);
            }
        }
        tb->continued += r.gpu_n_pop ? 1u : 0u;
        pops -= r.gpu_n_pop;  // what the host did itself
        tb->results.push_back(std::move(r));
        tb->done += 1;
        tb->host_pops += pops;
        tb->host_thread_s += TailBatch::now_s() - t_begin;
        tb->t_last = TailBatch::now_s();
    }
    tb->cv.notify_all();
#endif
}

// The launch's dispatcher: watches the records appear (in slot order; the kernel claims slots with an atomic counter and marks a record ready behind a
// system-scope fence) and hands each to the workers, until the launch has ended and every record it wrote has been seen.
inline void tail_start(const std::shared_ptr<TailBatch>& tb) {
    tb->dispatcher = std::thread([tb] {
        uint32_t next = 0, idle = 0;
        for (;;) {
            if (tb->cancel.load(std::memory_order_relaxed)) break;
            const int64_t fin = tb->final_count.load(std::memory_order_acquire);
            if (fin >= 0 && (int64_t)next >= fin) break;
            if (next < tb->cap) {
                const TailRecord* rec = reinterpret_cast<const TailRecord*>(tb->ring + (size_t)next * tb->stride);
                if (__atomic_load_n(&rec->ready, __ATOMIC_ACQUIRE) == tb->gen) {
                    const bool live = tb->launch_done && !tb->launch_done();
                    {
                        std::lock_guard<std::mutex> g(tb->mu);
                        if (tb->dispatched == 0) tb->t_first = TailBatch::now_s();
                        tb->dispatched += 1;
                        tb->gpu_pops += rec->pops;
                        tb->seen_live += live ? 1u : 0u;
                    }
                    TailWorkers::instance().submit([tb, rec] { tail_map_read(tb, rec); }, rec->pops);
                    next += 1;
                    if (tb->ctl) {  // the count first: a record is then never in neither word (the kernel adds the records beyond ctl[1] to ctl[0])
                        __atomic_store_n(tb->ctl, TailWorkers::instance().pending(), __ATOMIC_RELEASE);
                        __atomic_store_n(tb->ctl + 1, next, __ATOMIC_RELEASE);
                    }
                    idle = 0;
                    continue;
                }
            }
            // nothing new: back off up to a millisecond (a hand-over is rare, and a read that needs one runs for seconds)
            if (tb->ctl) __atomic_store_n(tb->ctl, TailWorkers::instance().pending(), __ATOMIC_RELAXED);
            idle = idle < 10 ? idle + 1 : 10;
            std::this_thread::sleep_for(std::chrono::microseconds(100 * idle));
        }
    });
}

// The launch has ended having written `count` records: waits until all of them are mapped.  Returns false if a worker failed.
inline bool tail_finish(const std::shared_ptr<TailBatch>& tb, uint32_t count) {
    tb->final_count.store((int64_t)std::min(count, tb->cap), std::memory_order_release);
    if (tb->dispatcher.joinable()) tb->dispatcher.join();
    std::unique_lock<std::mutex> l(tb->mu);
    tb->cv.wait(l, [&] { return tb->done == tb->dispatched; });
    return !tb->failed;
}
// The batch is abandoned (its slot is reused or its context destroyed before the results were collected).
inline void tail_cancel(const std::shared_ptr<TailBatch>& tb) {
    tb->cancel.store(true, std::memory_order_relaxed);
    if (tb->dispatcher.joinable()) tb->dispatcher.join();
    std::unique_lock<std::mutex> l(tb->mu);
    tb->cv.wait(l, [&] { return tb->done == tb->dispatched; });  // the ring and the tables stay alive through the shared_ptr, but the ring's memory is the slot's
}

}  // namespace host
}  // namespace mapad
