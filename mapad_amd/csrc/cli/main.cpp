// mapad-amd — command line with the `mapad index` / `mapad map` flag surface (src/main.rs:57-302) on top of the C ABI
// (include/mapad_amd.h).  Everything a Rust `mapad` would do around the hot path is done here through the same extern "C"
// calls a Rust caller would bind: index open/build, parameters, mapad_map_batch (GPU), mapad_hits_to_records_gpu (SA locate on the GPU), BAM output.
//
//   mapad-amd [--seed N] [--devices K] index -g ref.fa
//   mapad-amd [--devices K] worker --host H [--port 3130] [--dry_run]
//   mapad-amd [--seed N] [--devices 0-7 | 0,1,...] map -r reads.{bam,cram,fastq,fastq.gz} -g ref.fa -o out.bam -l single_stranded|double_stranded
//             -p 0.03 | (-c CUTOFF [-e EXP]) -f F -t T -d D -s S [-D 0.02] -i I [-x 1.0] [--batch_size 250000] [--coalesce 1 (4 on a text of >= 2^31 rows)] [--coalesce_steady N (= --coalesce)] [--in_flight 4] [--ignore_base_quality]
//             [--gap_dist_ends 5] [--max_num_gaps_open 2] [--no_search_limit_recovery] [--force_overwrite] [-R ID]
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/mapad_amd.h"
#include "bam_io.hpp"
#include "wire.hpp"

#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

using namespace mapad::cli;

namespace {

[[noreturn]] void die(const std::string& msg) { std::fprintf(stderr, "mapad-amd: %s\n", msg.c_str()); std::exit(1); }
void check(int rc, const char* what) { if (rc != MAPAD_OK) die(std::string(what) + " failed with status " + std::to_string(rc)); }

struct Args {
    std::map<std::string, std::string> kv;
    std::vector<std::string> flags;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    std::string get(const std::string& k, const std::string& d = "") const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    float f(const std::string& k, float d) const { return has(k) ? std::strtof(kv.at(k).c_str(), nullptr) : d; }
    bool flag(const std::string& k) const { for (auto& f : flags) if (f == k) return true; return false; }
};

// FASTA -> contigs (names up to the first whitespace, like noodles' fasta record name)
void read_fasta(const std::string& path, std::vector<std::string>& names, std::vector<std::vector<uint8_t>>& seqs) {
    GzReader in(path);
    std::string line;
    while (in.getline(line)) {
        if (line.empty()) continue;
        if (line[0] == '>') {
            const size_t sp = line.find_first_of(" \t");
            names.push_back(line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1));
            seqs.emplace_back();
        } else {
            if (seqs.empty()) die("FASTA does not start with '>'");
            seqs.back().insert(seqs.back().end(), line.begin(), line.end());
        }
    }
}

// create_bam_header (src/map/mapping.rs:300-398)
std::string make_header(const std::string& src, const mapad_index_t* idx, const std::string& read_group, const std::string& cmdline) {
    std::vector<std::string> pg, rg, co;
    std::string line;
    for (size_t i = 0; i <= src.size(); ++i) {
        if (i == src.size() || src[i] == '\n') {
            if (line.rfind("@PG", 0) == 0) pg.push_back(line);
            else if (line.rfind("@RG", 0) == 0) rg.push_back(line);
            else if (line.rfind("@CO", 0) == 0) co.push_back(line);
            line.clear();
        } else line.push_back(src[i]);
    }
    std::string out = "@HD\tVN:1.6\tSO:unsorted\n";
    for (uint32_t i = 0; i < mapad_index_n_contigs(idx); ++i) {
        const char* name; uint64_t s, e;
        check(mapad_index_contig(idx, i, &name, &s, &e), "mapad_index_contig");
        out += std::string("@SQ\tSN:") + name + "\tLN:" + std::to_string(e - s + 1) + "\n";
    }
    if (!read_group.empty()) out += "@RG\tID:" + read_group + "\n";
    else for (auto& l : rg) out += l + "\n";
    auto id_of = [](const std::string& l) { const size_t p = l.find("\tID:"); if (p == std::string::npos) return std::string(); const size_t e = l.find('\t', p + 4); return l.substr(p + 4, e == std::string::npos ? std::string::npos : e - p - 4); };
    size_t same = 0;
    std::string last_id;
    for (auto& l : pg) { out += l + "\n"; const std::string id = id_of(l); if (id == "mapAD" || id.rfind("mapAD.", 0) == 0) same += 1; last_id = id; }
    const std::string my_id = same ? "mapAD." + std::to_string(same) : "mapAD";
    out += "@PG\tID:" + my_id + "\tPN:mapAD\tVN:" + mapad_version() + "\tDS:An aDNA aware short-read mapper\tCL:" + cmdline + (last_id.empty() ? "" : "\tPP:" + last_id) + "\n";
    for (auto& l : co) out += l + "\n";
    return out;
}

int cmd_index(const Args& a, uint64_t seed, int device) {
    const std::string ref = a.get("reference");
    if (ref.empty()) die("index: -g/--reference is required");
    std::vector<std::string> names;
    std::vector<std::vector<uint8_t>> seqs;
    read_fasta(ref, names, seqs);
    std::vector<const char*> np;
    std::vector<const uint8_t*> sp;
    std::vector<uint64_t> lens;
    for (size_t i = 0; i < names.size(); ++i) { np.push_back(names[i].c_str()); sp.push_back(seqs[i].data()); lens.push_back(seqs[i].size()); }
    mapad_index_t* idx = nullptr;
    // suffix sorting on the GPU when there is one (seconds for a 3 Gbp genome); the host SA-IS path builds the same files without a GPU
    int rc = a.flag("host_index") ? MAPAD_ERR_NO_DEVICE : mapad_index_build_gpu(np.data(), sp.data(), lens.data(), (uint32_t)names.size(), seed, device, &idx);
    if (rc == MAPAD_ERR_UNSUPPORTED || rc == MAPAD_ERR_NOMEM) {
        std::fprintf(stderr, "index: the GPU suffix sorter cannot take this text (%s); building on the host\n", rc == MAPAD_ERR_NOMEM ? "out of device memory" : "a prefix bucket or group beyond its limits");
        rc = MAPAD_ERR_NO_DEVICE;
    }
    if (rc == MAPAD_ERR_NO_DEVICE) rc = mapad_index_build(np.data(), sp.data(), lens.data(), (uint32_t)names.size(), seed, &idx);
    check(rc, "mapad_index_build");
    check(mapad_index_save(idx, ref.c_str()), "mapad_index_save");  // files are named <reference>.{tbw,...} (indexing.rs:110-208)
    mapad_index_free(idx);
    return 0;
}

// ---- `map`: a three-stage pipeline over chunks (run_inner's loop, src/map/mapping.rs:151-294) -------------------------------------------
//   reader  : parses the next chunk of input records into page-locked buffers
//   devices : one worker thread and one context per GPU; a chunk is cut into contiguous slices, one per device (the reference's rayon map over
//             the chunk, order-preserving, mapping.rs:153-156,288).  A worker submits its slice of chunk k + 1 before it collects chunk k, so the
//             GPU maps while the host turns the previous chunk's hits into records (SA locate on the same GPU, strings on host threads).
//   writer  : encodes and BGZF-compresses the records of a finished chunk on several threads, writes them in input order.
// Results reach the writer through page-locked host memory, each GPU over its own PCIe link; there is no GPU-to-GPU hop in a single process.
template <class T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> l(mu_);
        not_full_.wait(l, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_all();
    }
    bool pop(T& out) {  // false: closed and drained
        std::unique_lock<std::mutex> l(mu_);
        not_empty_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> l(mu_); closed_ = true; not_empty_.notify_all(); }
private:
    size_t cap_;
    std::deque<T> q_;
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    bool closed_ = false;
};

struct PinnedBytes {  // page-locked byte buffer from the library (DMA at link speed)
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    void append(const void* src, size_t k) {
        if (n + k > cap) {
            const size_t want = std::max<size_t>((n + k) * 2, 1 << 20);
            uint8_t* q = (uint8_t*)mapad_host_alloc(want);
            if (!q) die("out of page-locked host memory");
            if (n) std::memcpy(q, p, n);
            mapad_host_free(p);
            p = q; cap = want;
        }
        if (k) std::memcpy(p + n, src, k);
        n += k;
    }
    void resize_uninitialized(size_t k) {  // contents are written by the caller
        if (k > cap) { n = 0; append(nullptr, 0); const size_t want = std::max<size_t>(k * 2, 1 << 20); uint8_t* q = (uint8_t*)mapad_host_alloc(want); if (!q) die("out of page-locked host memory"); mapad_host_free(p); p = q; cap = want; }
        n = k;
    }
    ~PinnedBytes() { mapad_host_free(p); }
};

struct Slice {
    uint64_t lo = 0, hi = 0;  // reads [lo, hi) of the chunk's mappable reads
    std::vector<uint64_t> offsets;  // rebased to the slice
    mapad_batch_result_t* res = nullptr;
    mapad_coords_t* coords = nullptr;  // device half of the records (mapad_hits_to_coords_gpu), consumed by the records thread
    mapad_records_t* recs = nullptr;
};
struct Chunk {
    uint64_t no = 0;
    uint64_t first_read = 0;         // mapped reads of the run before this chunk: read k of the chunk is read first_read + k of the run (the records' seeds count in reads of the run)
    std::vector<InRecord> in;        // every input record, in input order
    std::vector<int64_t> read_of;    // per input record: its index among the mapped reads, -1 if it cannot be mapped (empty / longer than the device limit)
    PinnedBytes seqs, quals;         // mapped reads, concatenated
    std::vector<uint64_t> offsets;
    std::vector<uint16_t> flags;
    std::vector<Slice> slices;
    std::atomic<int> pending{0};
    std::chrono::steady_clock::time_point t_submit;
    float per_read_s = 0.0f;
};
using ChunkPtr = std::shared_ptr<Chunk>;

std::vector<int> parse_devices(const std::string& spec) {  // "0", "0,2,3", "0-7"
    std::vector<int> out;
    size_t i = 0;
    while (i < spec.size()) {
        size_t j = i;
        while (j < spec.size() && spec[j] != ',') ++j;
        const std::string tok = spec.substr(i, j - i);
        const size_t dash = tok.find('-');
        if (dash != std::string::npos && dash > 0) { for (int d = std::atoi(tok.substr(0, dash).c_str()); d <= std::atoi(tok.substr(dash + 1).c_str()); ++d) out.push_back(d); }
        else if (!tok.empty()) out.push_back(std::atoi(tok.c_str()));
        i = j + 1;
    }
    if (out.empty()) die("--devices: empty device list");
    return out;
}

void parallel_for(size_t n, unsigned threads, const std::function<void(size_t, size_t, unsigned)>& fn) {  // contiguous ranges
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n));
    if (threads == 1) { fn(0, n, 0); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back([&, t] { fn(n * t / threads, n * (t + 1) / threads, t); });
    for (auto& th : pool) th.join();
}

int cmd_map(const Args& a, uint64_t seed, const std::vector<int>& devices, const std::string& cmdline) {
    for (const char* req : {"reads", "reference", "output", "library", "five_prime_overhang", "ds_deamination_rate", "ss_deamination_rate", "indel_rate"})
        if (!a.has(req)) die(std::string("map: --") + req + " is required");
    if (!a.has("poisson_prob") && !a.has("as_cutoff")) die("map: either -p or -c is required");
    const std::string lib = a.get("library");
    if (lib != "single_stranded" && lib != "double_stranded") die("map: --library must be single_stranded or double_stranded");
    if (lib == "single_stranded" && !a.has("three_prime_overhang")) die("map: -t is required for single_stranded libraries");
    mapad_params_t prm;
    check(mapad_params_from_cli(&prm, lib == "single_stranded" ? MAPAD_LIBRARY_SINGLE_STRANDED : MAPAD_LIBRARY_DOUBLE_STRANDED, a.f("five_prime_overhang", 0),
                                a.f("three_prime_overhang", 0), a.f("ds_deamination_rate", 0), a.f("ss_deamination_rate", 0), a.f("divergence", 0.02f),
                                a.has("poisson_prob") ? a.f("poisson_prob", 0) : -1.0f, a.f("as_cutoff", 0), a.f("as_cutoff_exponent", 1.0f), a.f("indel_rate", 0),
                                a.f("gap_extension_penalty", 1.0f), std::atoi(a.get("gap_dist_ends", "5").c_str()), std::atoi(a.get("max_num_gaps_open", "2").c_str()),
                                a.flag("ignore_base_quality"), a.flag("no_search_limit_recovery"), std::strtoull(a.get("chunk_size", "250000").c_str(), nullptr, 10)),
          "mapad_params_from_cli");
    const auto t_start = std::chrono::steady_clock::now();
    mapad_index_t* idx = nullptr;
    check(mapad_index_open(a.get("reference").c_str(), &idx), "mapad_index_open");
    // chunks in flight per device: the serial tail of a chunk (its few heaviest reads) runs beside the bulk of the following ones.  On a text of >= 2^31
    // rows a read costs three times the pops and a chunk's tail is longer: 8 in flight map 8 % more reads/s than 4 there (3 Gbp, 8 M reads; 16 are slower again)
    // Launches per chunk.  On a text of >= 2^31 rows a read costs three times the pops, and a launch is not over before its heaviest read is (a serial chain of up to
    // ~0.5 s at 50 bp) while its bulk needs a tenth of that: every launch leaves wavefronts behind that idle on their last reads.  There --coalesce 4 (the default)
    // hands the device four chunks of --batch_size reads as ONE launch; the records do not depend on it (one seed per read of the run, below), only the XD tag's
    // granularity does (chunk wall time / reads, mapping.rs:912-918).
    const bool big_text = mapad_index_text_len(idx) >= (1ull << 31);
    const uint64_t coalesce = std::max<uint64_t>(1, std::min<uint64_t>(std::strtoull(a.get("coalesce", big_text ? "4" : "1").c_str(), nullptr, 10), 64));
    // --coalesce_steady N (round 6): launches behind the first `in_flight` ones take N chunks.  Bigger launches from the start map faster once the pipeline is full (3 Gbp,
    // same box, profiles/r06/cli_sweep_c4.txt: 3.03 M reads/s behind the first chunk with 1 M-read launches, 3.25 M with 2.5 M, 3.59 M with 5 M) but fill it slower
    // (2.1 / 5.3 / 8.2 s until the first chunk is on disk).  Starting small and growing was measured and is NOT the default: on the same 24 M reads the run that goes on
    // with 2 M-read launches is slower throughout (2.65 M reads/s behind the first chunk against 2.97 M, 11.0 s against 9.9 s; profiles/r06/cli_sweep_ramp_c4.txt) —
    // four 2 M-read batches in flight on top of the first four leave the device worker waiting longer per fetch than they save per launch.
    const uint64_t coalesce_steady = std::max<uint64_t>(coalesce, std::min<uint64_t>(std::strtoull(a.get("coalesce_steady", "0").c_str(), nullptr, 10), 64));
    const uint64_t chunk_reads_first = prm.chunk_size * coalesce, chunk_reads_max = prm.chunk_size * coalesce_steady;
    const char* in_flight_default = "4";
    const int in_flight = std::max(1, std::min(std::atoi(a.get("in_flight", in_flight_default).c_str()), 16));
    const size_t n_dev = devices.size();
    std::vector<mapad_ctx_t*> ctxs(n_dev, nullptr);
    for (size_t d = 0; d < n_dev; ++d) {  // the read-only index is replicated into every GPU's HBM
        check(mapad_ctx_create(idx, &prm, devices[d], &ctxs[d]), "mapad_ctx_create");
        check(mapad_ctx_set_fetch_d_arrays(ctxs[d], 0), "mapad_ctx_set_fetch_d_arrays");
        check(mapad_ctx_set_pipeline_depth(ctxs[d], in_flight), "mapad_ctx_set_pipeline_depth");
        const uint64_t per_dev = (chunk_reads_max + n_dev - 1) / n_dev;  // both batch slots' buffers up front (typical short reads; longer ones grow them)
        check(mapad_ctx_reserve(ctxs[d], per_dev, per_dev * 64, 128, 1), "mapad_ctx_reserve");
    }
    const double t_load = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    ReadSource src(a.get("reads"));
    const std::string rg = a.get("read_group");
    BgzfWriter out(a.get("output"), a.flag("force_overwrite"));
    {   // BAM header
        const std::string text = make_header(src.header_text(), idx, rg, cmdline);
        std::vector<uint8_t> h = {'B', 'A', 'M', 1};
        const uint32_t l_text = (uint32_t)text.size(), n_ref = mapad_index_n_contigs(idx);
        h.insert(h.end(), (const uint8_t*)&l_text, (const uint8_t*)&l_text + 4);
        h.insert(h.end(), text.begin(), text.end());
        h.insert(h.end(), (const uint8_t*)&n_ref, (const uint8_t*)&n_ref + 4);
        for (uint32_t i = 0; i < n_ref; ++i) {
            const char* name; uint64_t s, e;
            check(mapad_index_contig(idx, i, &name, &s, &e), "mapad_index_contig");
            const uint32_t l_name = (uint32_t)std::strlen(name) + 1, l_ref = (uint32_t)(e - s + 1);
            h.insert(h.end(), (const uint8_t*)&l_name, (const uint8_t*)&l_name + 4);
            h.insert(h.end(), name, name + l_name);
            h.insert(h.end(), (const uint8_t*)&l_ref, (const uint8_t*)&l_ref + 4);
        }
        out.write(h.data(), h.size());
    }
    // Reader and writer fan every chunk out to short-lived bursts of up to 32 threads (parsing; BAM encoding + BGZF deflate) and do the rest of a chunk themselves.
    // Round 5 replaced this by standing pools sized to the cgroup's CPU share (a source thread, a parse pool with ordered emit, an encode pool working on pieces of a
    // chunk, an ordered writer: reader 1.5 s and writer 0.1 s busy instead of 5.5 s each) — and measured it 15-25 % SLOWER end to end on the same box with the same
    // library (24 M reads at C4: 11.4-13.3 s against 9.7-9.8 s; profiles/r05/cli_sweep_c4.txt): the bursts finish a chunk's host work in a fifth of a second and leave
    // the CPUs to the device worker in between, the pools keep 10 threads busy all the time.  The pools are in the history (commit d19f153), not in the tree.
    const unsigned host_threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    std::vector<std::unique_ptr<BoundedQueue<ChunkPtr>>> dev_q;
    for (size_t d = 0; d < n_dev; ++d) dev_q.emplace_back(new BoundedQueue<ChunkPtr>(2));  // read ahead; `in_flight` more are on the device
    BoundedQueue<ChunkPtr> rec_q(4), done_q(4);
    std::atomic<bool> failed{false};
    std::atomic<uint64_t> us_reader{0}, us_device{0}, us_writer{0}, us_submit{0}, us_fetch{0}, us_records{0}, us_text{0};  // busy time of the three stages (the slowest one sets the throughput)
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::string fail_msg;
    std::mutex fail_mu;
    auto fail = [&](const std::string& m) { std::lock_guard<std::mutex> l(fail_mu); if (!failed.exchange(true)) fail_msg = m; };

    // ---- reader ----
    std::thread reader([&] {
        try {
            uint64_t chunk_no = 0, reads_so_far = 0;
            bool more = true;
            while (more && !failed) {
                const uint64_t t_r0 = now_us();
                auto c = std::make_shared<Chunk>();
                const uint64_t chunk_reads = chunk_no < (uint64_t)in_flight * n_dev ? chunk_reads_first : chunk_reads_max;
                c->no = chunk_no;
                c->first_read = reads_so_far;
                c->offsets.assign(1, 0);
                size_t bases = 0;
                auto admit = [&](InRecord&& r, bool copy) {
                    // Reads the device cannot take (longer than MAPAD_MAX_READ_LEN; the reference's limit is i16::MAX, src/map/record.rs:144-150) and empty
                    // reads stay in the output as unmapped records, so that input and output hold the same number of records.
                    const bool mappable = !r.seq.empty() && r.seq.size() <= MAPAD_MAX_READ_LEN;
                    if (!mappable) std::fprintf(stderr, "mapad-amd: read \"%s\" (%zu bp) is %s; written as unmapped\n", r.name.c_str(), r.seq.size(), r.seq.empty() ? "empty" : "longer than the device limit");
                    c->read_of.push_back(mappable ? (int64_t)(c->offsets.size() - 1) : -1);
                    if (mappable) {
                        if (copy) { c->seqs.append(r.seq.data(), r.seq.size()); c->quals.append(r.qual.data(), r.seq.size()); }
                        bases += r.seq.size();
                        c->offsets.push_back(bases);
                        c->flags.push_back(r.flags);
                    }
                    c->in.push_back(std::move(r));
                };
                std::vector<std::pair<uint32_t, uint32_t>> lines;
                const char* block = src.fastq_block(chunk_reads, lines);
                bool block_done = false;
                if (block && !lines.empty() && lines.size() % 4 == 0) {
                    // FASTQ fast path: the chunk's lines are cut sequentially (one memchr per line), the records are parsed by several threads
                    const size_t n_rec = lines.size() / 4;
                    std::vector<InRecord> recs(n_rec);
                    std::vector<char> good(n_rec, 0);
                    parallel_for(n_rec, host_threads, [&](size_t lo, size_t hi, unsigned) {
                        for (size_t i = lo; i < hi; ++i) good[i] = ReadSource::parse_fastq_record(block, &lines[4 * i], recs[i]);
                    });
                    bool all_good = true;
                    for (char g : good) all_good &= g != 0;
                    if (all_good) {
                        c->in.reserve(n_rec);
                        for (auto& r : recs) admit(std::move(r), false);
                        // bases of the mappable reads into the page-locked buffers, in parallel (offsets are known now)
                        c->seqs.append(nullptr, 0); c->quals.append(nullptr, 0);
                        c->seqs.resize_uninitialized(bases); c->quals.resize_uninitialized(bases);
                        parallel_for(c->in.size(), host_threads, [&](size_t lo, size_t hi, unsigned) {
                            for (size_t i = lo; i < hi; ++i) {
                                const int64_t k = c->read_of[i];
                                if (k < 0) continue;
                                const InRecord& r = c->in[i];
                                std::memcpy(c->seqs.p + c->offsets[(size_t)k], r.seq.data(), r.seq.size());
                                std::memcpy(c->quals.p + c->offsets[(size_t)k], r.qual.data(), r.seq.size());
                            }
                        });
                        more = lines.size() == 4 * chunk_reads;
                        block_done = true;
                    }
                }
                if (block && !block_done) {
                    // irregular text (blank lines, a malformed record, a truncated tail): the same lines through the tolerant sequential rules
                    size_t i = 0;
                    while (i < lines.size()) {
                        if (lines[i].second == 0) { ++i; continue; }
                        if (i + 4 > lines.size()) { if (lines.size() == 4 * chunk_reads) src.fastq_unread_from(lines[i].first); break; }  // a cut record goes back
                        InRecord r;
                        if (ReadSource::parse_fastq_record(block, &lines[i], r)) admit(std::move(r), true);
                        else std::fprintf(stderr, "Skip record due to an error: malformed FASTQ record\n");
                        i += 4;
                    }
                    more = lines.size() == 4 * chunk_reads;
                } else if (!block) {
                    InRecord r;
                    while (c->in.size() < chunk_reads && (more = src.next(r))) admit(std::move(r), true);
                }
                if (c->in.empty()) { if (more) continue; break; }  // a whole block of blank / malformed records: keep reading
                const uint64_t n_reads = c->offsets.size() - 1;
                c->slices.resize(n_dev);
                for (size_t d = 0; d < n_dev; ++d) {  // contiguous slices: concatenation in device order = input order
                    Slice& sl = c->slices[d];
                    const uint64_t base = n_reads / n_dev, extra = n_reads % n_dev;
                    sl.lo = d * base + std::min<uint64_t>(d, extra); sl.hi = sl.lo + base + (d < extra ? 1 : 0);
                    sl.offsets.resize(sl.hi - sl.lo + 1);
                    for (uint64_t i = sl.lo; i <= sl.hi; ++i) sl.offsets[i - sl.lo] = c->offsets[i] - c->offsets[sl.lo];
                }
                c->pending = (int)n_dev;
                us_reader += now_us() - t_r0;
                c->t_submit = std::chrono::steady_clock::now();
                for (size_t d = 0; d < n_dev; ++d) dev_q[d]->push(c);
                chunk_no += 1;
                reads_so_far += n_reads;
            }
        } catch (const std::exception& e) { fail(e.what()); }
        for (auto& q : dev_q) q->close();
    });

    // ---- device workers ----
    auto worker = [&](size_t d) {
        mapad_ctx_t* ctx = ctxs[d];
        auto submit = [&](const ChunkPtr& c) {
            const Slice& sl = c->slices[d];
            const uint64_t b0 = c->offsets[sl.lo];
            const uint64_t t0 = now_us();
            check(mapad_submit_batch(ctx, c->seqs.p + b0, c->quals.p + b0, sl.offsets.data(), sl.hi - sl.lo), "mapad_submit_batch");
            if (d == 0) us_submit += now_us() - t0;
        };
        auto records = [&](const ChunkPtr& c) {
            Slice& sl = c->slices[d];
            const uint64_t t0 = now_us();
            // The device half of intervals_to_bam (reported hit, coordinates, XA candidates, X0 / X1) from the hits still resident on this GPU; the
            // strings and mapping qualities are the records thread's work, so that this thread goes straight back to submitting and fetching.
            // One seed per read of the run, whichever device maps it and however the input is cut into chunks: the run's seed advanced to the slice's first read.
            check(mapad_hits_to_coords_gpu(ctx, sl.res, mapad_records_seed_at(seed, c->first_read + sl.lo), &sl.coords), "mapad_hits_to_coords_gpu");
            if (d == 0) us_records += now_us() - t0;
            if (--c->pending == 0) {
                c->per_read_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - c->t_submit).count() / (float)std::max<size_t>(c->in.size(), 1);
                rec_q.push(c);
            }
        };
        auto collect = [&](const ChunkPtr& c, int age) -> bool {  // false: the hit pools were too small for this slice
            check(mapad_ctx_select_batch(ctx, age), "mapad_ctx_select_batch");
            const uint64_t t0 = now_us();
            const int rc = mapad_fetch_result(ctx, &c->slices[d].res);
            if (d == 0) us_fetch += now_us() - t0;
            if (rc == MAPAD_ERR_NOMEM) return false;
            check(rc, "mapad_fetch_result");
            return true;
        };
        auto rerun = [&](const ChunkPtr& c) {  // synchronous path that grows the pools
            const Slice& sl = c->slices[d];
            const uint64_t b0 = c->offsets[sl.lo];
            check(mapad_map_batch(ctx, c->seqs.p + b0, c->quals.p + b0, sl.offsets.data(), sl.hi - sl.lo, &c->slices[d].res), "mapad_map_batch");
        };
        try {
            std::deque<ChunkPtr> flying;  // submitted, oldest first
            // the oldest chunk's results; rare: a hit pool was too small — finish everything in flight, then repair synchronously, in order
            auto retire_oldest = [&] {
                if (collect(flying.front(), (int)flying.size() - 1)) { records(flying.front()); flying.pop_front(); return; }
                std::vector<bool> ok(flying.size(), false);
                for (size_t i = 1; i < flying.size(); ++i) ok[i] = collect(flying[i], (int)(flying.size() - 1 - i));
                for (size_t i = 0; i < flying.size(); ++i) { if (!ok[i]) rerun(flying[i]); records(flying[i]); }
                flying.clear();
            };
            ChunkPtr c;
            while (dev_q[d]->pop(c)) {
                if (failed) continue;
                const uint64_t t_d0 = now_us();
                submit(c);  // the GPU starts on this chunk while older ones are collected and turned into records
                flying.push_back(c);
                c.reset();
                if ((int)flying.size() >= in_flight) retire_oldest();
                if (d == 0) us_device += now_us() - t_d0;
            }
            const uint64_t t_d1 = now_us();
            while (!flying.empty() && !failed) retire_oldest();
            if (d == 0) us_device += now_us() - t_d1;
        } catch (const std::exception& e) { fail(e.what()); }
        { ChunkPtr c; while (dev_q[d]->pop(c)) {} }  // after a failure: keep taking chunks so that the stage in front never blocks on a full queue (the process then exits non-zero)
    };
    std::vector<std::thread> workers;
    for (size_t d = 0; d < n_dev; ++d) workers.emplace_back(worker, d);

    // ---- records: the text half of intervals_to_bam (CIGAR / MD / XA strings, flags, mapping quality) on host threads, chunk by chunk in order ----
    std::thread recorder([&] {
        try {
            ChunkPtr c;
            while (rec_q.pop(c)) {
                if (failed) continue;
                const uint64_t t0 = now_us();
                for (auto& sl : c->slices) {
                    check(mapad_coords_to_records(idx, &prm, sl.res, c->flags.data() + sl.lo, sl.coords, &sl.recs), "mapad_coords_to_records");
                    mapad_coords_free(sl.coords); sl.coords = nullptr;
                }
                us_text += now_us() - t0;
                done_q.push(c);
            }
        } catch (const std::exception& e) { fail(e.what()); }
        { ChunkPtr c; while (rec_q.pop(c)) {} }  // see the device workers: a failed stage keeps draining its input
        done_q.close();
    });

    // ---- writer ----
    uint64_t n_total = 0, n_mapped = 0;
    uint64_t first_chunk_reads = 0, t_first_chunk_us = 0, t_last_chunk_us = 0;  // for the steady-state rate: everything behind the pipeline's fill (the first chunk's way through reader, device and writer)
    std::thread writer([&] {
        try {
            ChunkPtr c;
            while (done_q.pop(c)) {
                if (failed) continue;
                const uint64_t t_w0 = now_us();
                const size_t n = c->in.size();
                std::vector<std::vector<uint8_t>> enc(host_threads), comp(host_threads);
                std::vector<uint64_t> mapped(host_threads, 0);
                parallel_for(n, host_threads, [&](size_t lo, size_t hi, unsigned t) {
                    size_t d = 0;
                    for (size_t i = lo; i < hi; ++i) {
                        OutFields f;
                        const int64_t r = c->read_of[i];
                        if (r < 0) f.flags = (uint16_t)((c->in[i].flags & ~(0x8 | 0x20 | 0x2 | 0x100 | 0x800 | 0x10)) | 0x4);  // unmapped (mapping.rs:748-776)
                        else {
                            while ((uint64_t)r >= c->slices[d].hi) ++d;
                            const mapad_records_t* recs = c->slices[d].recs;
                            const mapad_record_t& m = recs->recs[(uint64_t)r - c->slices[d].lo];
                            f.mapped = m.mapped; f.reverse = m.reverse; f.flags = m.flags; f.tid = m.tid; f.pos = m.pos; f.mapq = m.mapq;
                            f.cigar.assign(recs->text + m.cigar_off, m.cigar_len); f.md.assign(recs->text + m.md_off, m.md_len); f.xa.assign(recs->text + m.xa_off, m.xa_len);
                            f.as = m.as_score; f.xs = m.xs_score; f.nm = m.nm; f.x0 = m.x0; f.x1 = m.x1; f.has_xs = m.has_xs; f.has_alt = m.mapped; f.xt = m.xt;
                            mapped[t] += m.mapped;
                        }
                        f.xd = c->per_read_s;  // the reference stores the wall time of each read's search (mapping.rs:912-918); here: chunk time / reads
                        encode_bam_record(c->in[i], f, rg, enc[t]);
                    }
                    BgzfWriter::compress_all(enc[t].data(), enc[t].size(), comp[t]);  // every thread deflates what it encoded; the blocks are written in order below
                });
                for (unsigned t = 0; t < host_threads; ++t) { out.write_compressed(comp[t]); n_mapped += mapped[t]; }
                n_total += n;
                for (auto& sl : c->slices) { mapad_records_free(sl.recs); mapad_batch_result_free(sl.res); sl.recs = nullptr; sl.res = nullptr; }
                t_last_chunk_us = now_us();
                if (!t_first_chunk_us) { t_first_chunk_us = t_last_chunk_us; first_chunk_reads = n; }
                us_writer += now_us() - t_w0;
            }
        } catch (const std::exception& e) { fail(e.what()); }
        { ChunkPtr c; while (done_q.pop(c)) {} }
    });

    reader.join();
    for (auto& w : workers) w.join();
    rec_q.close();
    recorder.join();
    writer.join();
    if (failed) die(fail_msg);
    out.close();
    const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::fprintf(stderr, "mapad-amd: %llu reads, %llu mapped; %zu device(s); index + contexts %.2f s, mapping %.2f s (%.0f reads/s)\n", (unsigned long long)n_total,
                 (unsigned long long)n_mapped, n_dev, t_load, t_all - t_load, (double)n_total / std::max(t_all - t_load, 1e-9));
    if (n_total > first_chunk_reads && t_last_chunk_us > t_first_chunk_us)  // the run without its fill: from the first chunk's records on disk to the last chunk's
        std::fprintf(stderr, "mapad-amd: steady state: %llu reads in %.2f s behind the first chunk (%.0f reads/s); pipeline fill %.2f s\n", (unsigned long long)(n_total - first_chunk_reads),
                     (t_last_chunk_us - t_first_chunk_us) * 1e-6, (double)(n_total - first_chunk_reads) / ((t_last_chunk_us - t_first_chunk_us) * 1e-6),
                     t_all - t_load - (t_last_chunk_us - t_first_chunk_us) * 1e-6);
    std::fprintf(stderr, "mapad-amd: stage busy time: reader %.2f s, device worker 0 %.2f s, writer %.2f s\n", us_reader.load() * 1e-6, us_device.load() * 1e-6, us_writer.load() * 1e-6);
    std::fprintf(stderr, "mapad-amd: device worker 0: submit %.2f s, fetch (incl. waiting for the GPU) %.2f s, coordinates %.2f s; records thread (strings, MAPQ) %.2f s\n",
                 us_submit.load() * 1e-6, us_fetch.load() * 1e-6, us_records.load() * 1e-6, us_text.load() * 1e-6);
    for (auto* c : ctxs) mapad_ctx_destroy(c);
    mapad_index_free(idx);
    return 0;
}

// I/O self-test hook (no GPU needed): parse every input record and write it back as an unmapped BAM record
int cmd_recode(const Args& a) {
    ReadSource src(a.get("reads"));
    BgzfWriter out(a.get("output"), true);
    const std::string text = "@HD\tVN:1.6\tSO:unsorted\n";
    std::vector<uint8_t> h = {'B', 'A', 'M', 1};
    const uint32_t l_text = (uint32_t)text.size(), n_ref = 0;
    h.insert(h.end(), (const uint8_t*)&l_text, (const uint8_t*)&l_text + 4);
    h.insert(h.end(), text.begin(), text.end());
    h.insert(h.end(), (const uint8_t*)&n_ref, (const uint8_t*)&n_ref + 4);
    out.write(h.data(), h.size());
    InRecord r;
    std::vector<uint8_t> enc;
    while (src.next(r)) {
        OutFields f;
        f.flags = (uint16_t)((r.flags & ~(0x8 | 0x20 | 0x2 | 0x100 | 0x800 | 0x10)) | 0x4);
        enc.clear();
        encode_bam_record(r, f, a.get("read_group"), enc);
        out.write(enc.data(), enc.size());
    }
    out.close();
    return 0;
}


// ---- `worker`: the reference's `mapad worker --host H --port P` (src/distributed/worker.rs:35-236) with the search on a GPU --------------------
// Connects to a dispatcher, and for every TaskSheet it receives maps the records (k_mismatch_search per record: here one batch on the
// device) and answers with a ResultSheet: each record as it arrived, its hits in BinaryHeap array order, the time spent.  The first task
// names the index and carries the alignment parameters (worker.rs:57-75).  The dispatcher hands a worker one task at a time, so there
// is nothing to pipeline on this side.  Reads the device cannot take (empty, longer than MAPAD_MAX_READ_LEN) come back without hits.
// --dry_run answers every task with hit-less results without touching a GPU or an index (framing / codec check).
bool read_exact(int fd, uint8_t* p, size_t n) {  // false: the peer closed the connection before the first byte
    size_t got = 0;
    while (got < n) {
        const ssize_t k = ::read(fd, p + got, n - got);
        if (k == 0) { if (got == 0) return false; die("worker: connection closed inside a message"); }
        if (k < 0) { if (errno == EINTR) continue; die(std::string("worker: read: ") + std::strerror(errno)); }
        got += (size_t)k;
    }
    return true;
}

int cmd_worker(const Args& a, const std::vector<int>& devices) {
    const std::string host = a.get("host"), port = a.get("port", "3130");
    if (host.empty()) die("worker: --host is required");
    const bool dry = a.flag("dry_run");
    addrinfo hints{}, *ai = nullptr;
    hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host.c_str(), port.c_str(), &hints, &ai) != 0 || !ai) die("worker: cannot resolve " + host);
    int fd = -1;
    for (addrinfo* p = ai; p; p = p->ai_next) {
        fd = ::socket(p->ai_family, p->ai_socktype, p->ai_protocol);
        if (fd < 0) continue;
        if (::connect(fd, p->ai_addr, p->ai_addrlen) == 0) break;
        ::close(fd); fd = -1;
    }
    freeaddrinfo(ai);
    if (fd < 0) die("worker: cannot connect to " + host + ":" + port);
    mapad_index_t* idx = nullptr;
    mapad_ctx_t* ctx = nullptr;
    std::vector<uint8_t> msg;
    uint64_t n_tasks = 0, n_reads_total = 0;
    for (;;) {
        msg.resize(8);
        if (!read_exact(fd, msg.data(), 8)) break;  // the dispatcher has dropped the connection: done (worker.rs:203-206)
        uint64_t size; std::memcpy(&size, msg.data(), 8);
        // a task is at most one chunk of records (sequence + qualities + tags) plus the parameters: anything beyond a few GiB is a corrupt header,
        // not a task (the dispatcher sends chunk_size records, distributed/dispatcher.rs:223-247)
        if (size < 8 + 8 + 8 + 2 || size > (4ull << 30)) die("worker: implausible task size");
        try { msg.resize(size); } catch (const std::bad_alloc&) { die("worker: implausible task size (allocation failed)"); }
        if (!read_exact(fd, msg.data() + 8, size - 8)) die("worker: connection closed inside a message");
        const wire::Task t = wire::decode_task(msg.data(), msg.size());
        if (!dry && !idx) {  // worker.rs:57-65
            if (!t.has_reference) die("worker: the first task does not name the reference");
            check(mapad_index_open(t.reference_path.c_str(), &idx), "mapad_index_open");
        }
        if (!dry && !ctx) {  // worker.rs:67-75
            if (!t.has_params) die("worker: the first task carries no alignment parameters");
            check(mapad_ctx_create(idx, &t.params, devices[0], &ctx), "mapad_ctx_create");
            check(mapad_ctx_set_fetch_d_arrays(ctx, 0), "mapad_ctx_set_fetch_d_arrays");
        }
        std::vector<int64_t> read_of(t.records.size(), -1);
        std::vector<uint8_t> seqs, quals;
        std::vector<uint64_t> offsets{0};
        for (size_t i = 0; i < t.records.size(); ++i) {
            const wire::RecordView& r = t.records[i];
            if (r.len == 0 || r.len > MAPAD_MAX_READ_LEN || r.qual_len != r.len) continue;
            read_of[i] = (int64_t)offsets.size() - 1;
            seqs.insert(seqs.end(), r.seq, r.seq + r.len); quals.insert(quals.end(), r.qual, r.qual + r.len);
            offsets.push_back(seqs.size());
        }
        const auto t0 = std::chrono::steady_clock::now();
        mapad_batch_result_t* res = nullptr;
        if (!dry) check(mapad_map_batch(ctx, seqs.data(), quals.data(), offsets.data(), offsets.size() - 1, &res), "mapad_map_batch");
        const double per_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)std::max<size_t>(t.records.size(), 1);
        const std::vector<uint8_t> out = wire::encode_result(t, res, read_of, per_read);
        if (res) mapad_batch_result_free(res);
        for (size_t sent = 0; sent < out.size();) {
            const ssize_t k = ::write(fd, out.data() + sent, out.size() - sent);
            if (k < 0) { if (errno == EINTR) continue; die(std::string("worker: write: ") + std::strerror(errno)); }
            sent += (size_t)k;
        }
        n_tasks += 1; n_reads_total += t.records.size();
    }
    ::close(fd);
    std::fprintf(stderr, "mapad-amd worker: %llu task(s), %llu reads; the dispatcher closed the connection\n", (unsigned long long)n_tasks, (unsigned long long)n_reads_total);
    if (ctx) mapad_ctx_destroy(ctx);
    if (idx) mapad_index_free(idx);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    static const std::map<std::string, std::string> shorts = {
        {"-g", "reference"}, {"-r", "reads"}, {"-o", "output"}, {"-p", "poisson_prob"}, {"-c", "as_cutoff"}, {"-e", "as_cutoff_exponent"}, {"-l", "library"},
        {"-f", "five_prime_overhang"}, {"-t", "three_prime_overhang"}, {"-d", "ds_deamination_rate"}, {"-s", "ss_deamination_rate"}, {"-D", "divergence"},
        {"-i", "indel_rate"}, {"-x", "gap_extension_penalty"}, {"-R", "read_group"}};
    static const std::vector<std::string> bool_flags = {"ignore_base_quality", "no_search_limit_recovery", "force_overwrite", "host_index", "dry_run"};
    std::string cmdline, sub;
    for (int i = 0; i < argc; ++i) cmdline += std::string(i ? " " : "") + argv[i];
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k == "index" || k == "map" || k == "recode" || k == "worker") { sub = k; continue; }
        if (k == "-v" || k == "-vv" || k == "-vvv") continue;
        if (k == "--batch_size") k = "--chunk_size";
        std::string key;
        if (shorts.count(k)) key = shorts.at(k);
        else if (k.rfind("--", 0) == 0) key = k.substr(2);
        else die("unexpected argument " + k);
        bool is_flag = false;
        for (auto& b : bool_flags) is_flag |= b == key;
        if (is_flag) a.flags.push_back(key);
        else { if (i + 1 >= argc) die("missing value for " + k); a.kv[key] = argv[++i]; }
    }
    // every chunk in flight has its own HIP stream; the runtime multiplexes streams onto 4 hardware queues unless told otherwise, and a
    // chunk's long tail would then hold back the launches queued behind it (must be set before the first HIP call)
    setenv("GPU_MAX_HW_QUEUES", "20", 0);
    const uint64_t seed = std::strtoull(a.get("seed", "1234").c_str(), nullptr, 10);
    const std::vector<int> devices = parse_devices(a.get("devices", a.get("device", "0")));
    try {
        if (sub == "index") return cmd_index(a, seed, devices[0]);
        if (sub == "map") return cmd_map(a, seed, devices, cmdline);
        if (sub == "recode") return cmd_recode(a);
        if (sub == "worker") return cmd_worker(a, devices);
        die("usage: mapad-amd [--seed N] [--devices 0[,1,...|-7]] index|map|worker ...");
    } catch (const std::exception& e) { die(e.what()); }
}
