// mapad-amd — command line with the `mapad index` / `mapad map` flag surface (src/main.rs:57-302) on top of the C ABI
// (include/mapad_amd.h).  Everything a Rust `mapad` would do around the hot path is done here through the same extern "C"
// calls a Rust caller would bind: index open/build, parameters, mapad_map_batch (GPU), mapad_hits_to_records_gpu (SA locate on the GPU), BAM output.
//
//   mapad-amd [--seed N] [--device K] index -g ref.fa
//   mapad-amd [--seed N] [--device K] map -r reads.{bam,fastq,fastq.gz} -g ref.fa -o out.bam -l single_stranded|double_stranded
//             -p 0.03 | (-c CUTOFF [-e EXP]) -f F -t T -d D -s S [-D 0.02] -i I [-x 1.0] [--batch_size 250000] [--ignore_base_quality]
//             [--gap_dist_ends 5] [--max_num_gaps_open 2] [--no_search_limit_recovery] [--force_overwrite] [-R ID]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../../include/mapad_amd.h"
#include "bam_io.hpp"

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

int cmd_index(const Args& a, uint64_t seed) {
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
    check(mapad_index_build(np.data(), sp.data(), lens.data(), (uint32_t)names.size(), seed, &idx), "mapad_index_build");
    check(mapad_index_save(idx, ref.c_str()), "mapad_index_save");  // files are named <reference>.{tbw,...} (indexing.rs:110-208)
    mapad_index_free(idx);
    return 0;
}

int cmd_map(const Args& a, uint64_t seed, int device, const std::string& cmdline) {
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
    mapad_index_t* idx = nullptr;
    check(mapad_index_open(a.get("reference").c_str(), &idx), "mapad_index_open");
    mapad_ctx_t* ctx = nullptr;
    check(mapad_ctx_create(idx, &prm, device, &ctx), "mapad_ctx_create");
    check(mapad_ctx_set_fetch_d_arrays(ctx, 0), "mapad_ctx_set_fetch_d_arrays");
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
    uint64_t n_total = 0, n_mapped = 0, chunk_no = 0;
    std::vector<InRecord> chunk;
    std::vector<uint8_t> seqs, quals, enc;
    std::vector<uint64_t> offsets;
    std::vector<uint16_t> flags;
    bool more = true;
    while (more) {  // run_inner's chunk loop (mapping.rs:151-294): map a chunk, write it in input order
        chunk.clear(); seqs.clear(); quals.clear(); offsets.assign(1, 0); flags.clear();
        InRecord r;
        while (chunk.size() < prm.chunk_size && (more = src.next(r))) {
            if (r.seq.size() > MAPAD_MAX_READ_LEN) { std::fprintf(stderr, "Skip record due to an error: read \"%s\" is longer than %d bp\n", r.name.c_str(), MAPAD_MAX_READ_LEN); continue; }
            seqs.insert(seqs.end(), r.seq.begin(), r.seq.end());
            quals.insert(quals.end(), r.qual.begin(), r.qual.end());
            offsets.push_back(seqs.size());
            flags.push_back(r.flags);
            chunk.push_back(std::move(r));
        }
        if (chunk.empty()) break;
        const auto t0 = std::chrono::steady_clock::now();
        mapad_batch_result_t* res = nullptr;
        check(mapad_map_batch(ctx, seqs.data(), quals.data(), offsets.data(), chunk.size(), &res), "mapad_map_batch");
        const float per_read_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count() / (float)chunk.size();
        mapad_records_t* recs = nullptr;
        check(mapad_hits_to_records_gpu(ctx, res, seqs.data(), quals.data(), offsets.data(), flags.data(), seed + chunk_no, &recs), "mapad_hits_to_records_gpu");
        enc.clear();
        for (size_t i = 0; i < chunk.size(); ++i) {
            const mapad_record_t& c = recs->recs[i];
            OutFields f;
            f.mapped = c.mapped; f.reverse = c.reverse; f.flags = c.flags; f.tid = c.tid; f.pos = c.pos; f.mapq = c.mapq;
            f.cigar.assign(recs->text + c.cigar_off, c.cigar_len); f.md.assign(recs->text + c.md_off, c.md_len); f.xa.assign(recs->text + c.xa_off, c.xa_len);
            f.as = c.as_score; f.xs = c.xs_score; f.nm = c.nm; f.x0 = c.x0; f.x1 = c.x1; f.has_xs = c.has_xs; f.has_alt = c.mapped; f.xt = c.xt;
            f.xd = per_read_s;  // the reference stores the wall time of each read's search (mapping.rs:912-918); here: chunk time / reads
            encode_bam_record(chunk[i], f, rg, enc);
            n_mapped += c.mapped;
        }
        out.write(enc.data(), enc.size());
        n_total += chunk.size();
        chunk_no += 1;
        mapad_records_free(recs);
        mapad_batch_result_free(res);
    }
    out.close();
    std::fprintf(stderr, "mapad-amd: %llu reads, %llu mapped\n", (unsigned long long)n_total, (unsigned long long)n_mapped);
    mapad_ctx_destroy(ctx);
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

}  // namespace

int main(int argc, char** argv) {
    static const std::map<std::string, std::string> shorts = {
        {"-g", "reference"}, {"-r", "reads"}, {"-o", "output"}, {"-p", "poisson_prob"}, {"-c", "as_cutoff"}, {"-e", "as_cutoff_exponent"}, {"-l", "library"},
        {"-f", "five_prime_overhang"}, {"-t", "three_prime_overhang"}, {"-d", "ds_deamination_rate"}, {"-s", "ss_deamination_rate"}, {"-D", "divergence"},
        {"-i", "indel_rate"}, {"-x", "gap_extension_penalty"}, {"-R", "read_group"}};
    static const std::vector<std::string> bool_flags = {"ignore_base_quality", "no_search_limit_recovery", "force_overwrite"};
    std::string cmdline, sub;
    for (int i = 0; i < argc; ++i) cmdline += std::string(i ? " " : "") + argv[i];
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k == "index" || k == "map" || k == "recode") { sub = k; continue; }
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
    const uint64_t seed = std::strtoull(a.get("seed", "1234").c_str(), nullptr, 10);
    const int device = std::atoi(a.get("device", "0").c_str());
    try {
        if (sub == "index") return cmd_index(a, seed);
        if (sub == "map") return cmd_map(a, seed, device, cmdline);
        if (sub == "recode") return cmd_recode(a);
        die("usage: mapad-amd [--seed N] [--device K] index|map ...");
    } catch (const std::exception& e) { die(e.what()); }
}
