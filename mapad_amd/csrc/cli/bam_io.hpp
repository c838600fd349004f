// bam_io.hpp — minimal BGZF / BAM / FASTQ plumbing for the `mapad-amd` command line (host I/O, not on the accelerated path).
//
// Stands in for the parts of noodles that `mapad map` uses (src/map/input_chunk_reader.rs:42-172, src/map/record.rs:138-215,
// src/map/mapping.rs:92-110,292): BAM / CRAM 3.0 (cram_io.hpp) / FASTQ(.gz) records in, BAM records out.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cctype>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cram_io.hpp"

namespace mapad {
namespace cli {

// ---- BGZF -------------------------------------------------------------------------------------------------------------
class BgzfWriter {
public:
    explicit BgzfWriter(const std::string& path, bool force) {
        f_ = std::fopen(path.c_str(), force ? "wb" : "wbx");  // create_new(!force_overwrite) (mapping.rs:93-100)
        if (!f_) throw std::runtime_error("cannot create output file " + path + (force ? "" : " (exists? use --force_overwrite)"));
    }
    ~BgzfWriter() { try { close(); } catch (...) {} }
    void write(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        while (n) {
            const size_t k = std::min(n, kBlock - buf_.size());
            buf_.insert(buf_.end(), b, b + k);
            b += k; n -= k;
            if (buf_.size() == kBlock) flush_block();
        }
    }
    // the same stream as write(), with the full blocks of a large buffer deflated by `threads` threads (BGZF blocks are independent)
    void write_parallel(const void* p, size_t n, unsigned threads) {
        const uint8_t* b = (const uint8_t*)p;
        if (!buf_.empty()) {  // complete the block that is open
            const size_t k = std::min(n, kBlock - buf_.size());
            write(b, k);
            b += k; n -= k;
        }
        const size_t n_blocks = n / kBlock;
        if (n_blocks >= 2 && threads > 1) {
            std::vector<std::vector<uint8_t>> outs(n_blocks);
            threads = (unsigned)std::min<size_t>(threads, n_blocks);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < threads; ++t)
                pool.emplace_back([&, t] { for (size_t i = t; i < n_blocks; i += threads) outs[i] = deflate_block(b + i * kBlock, kBlock); });
            for (auto& th : pool) th.join();
            for (auto& o : outs) if (std::fwrite(o.data(), 1, o.size(), f_) != o.size()) throw std::runtime_error("write failed");
            b += n_blocks * kBlock; n -= n_blocks * kBlock;
        }
        write(b, n);
    }
    // BGZF blocks of a whole buffer (full blocks and a last shorter one), appended to `out`: for callers that deflate on their own threads
    static void compress_all(const uint8_t* p, size_t n, std::vector<uint8_t>& out) {
        Deflater d;
        for (size_t off = 0; off < n; off += kBlock) d.block(p + off, std::min(kBlock, n - off), out);
    }
    // blocks made by compress_all, in stream order (closes the block that write() may have left open first)
    void write_compressed(const std::vector<uint8_t>& blocks) {
        if (!buf_.empty()) flush_block();
        if (std::fwrite(blocks.data(), 1, blocks.size(), f_) != blocks.size()) throw std::runtime_error("write failed");
    }
    void close() {
        if (!f_) return;
        if (!buf_.empty()) flush_block();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(eof, 1, 28, f_);
        std::fclose(f_);
        f_ = nullptr;
    }

private:
    static constexpr size_t kBlock = 0xff00;
    FILE* f_ = nullptr;
    std::vector<uint8_t> buf_;
    struct Deflater {  // one zlib state for many blocks (deflateInit2 allocates a quarter of a megabyte)
        z_stream zs{};
        Deflater() { if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2"); }
        ~Deflater() { deflateEnd(&zs); }
        Deflater(const Deflater&) = delete;
        void block(const uint8_t* data, size_t len, std::vector<uint8_t>& out) {
            const size_t base = out.size();
            out.resize(base + kBlock + 1024);
            deflateReset(&zs);
            zs.next_in = const_cast<uint8_t*>(data); zs.avail_in = (uInt)len;
            zs.next_out = out.data() + base + 18; zs.avail_out = (uInt)(kBlock + 1024 - 18 - 8);
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate");
            const size_t clen = zs.total_out, bsize = clen + 18 + 8;
            const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
            std::memcpy(out.data() + base, hdr, 18);
            const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), data, (uInt)len), isize = (uint32_t)len;
            std::memcpy(out.data() + base + 18 + clen, &crc, 4);
            std::memcpy(out.data() + base + 18 + clen + 4, &isize, 4);
            out.resize(base + bsize);
        }
    };
    static std::vector<uint8_t> deflate_block(const uint8_t* data, size_t len) {
        std::vector<uint8_t> out(kBlock + 1024);
        z_stream zs{};
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
        zs.next_in = const_cast<uint8_t*>(data); zs.avail_in = (uInt)len;
        zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 18 - 8);
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate");
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        const size_t bsize = clen + 18 + 8;
        const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
        std::memcpy(out.data(), hdr, 18);
        const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), data, (uInt)len), isize = (uint32_t)len;
        std::memcpy(out.data() + 18 + clen, &crc, 4);
        std::memcpy(out.data() + 18 + clen + 4, &isize, 4);
        out.resize(bsize);
        return out;
    }
    void flush_block() {
        const std::vector<uint8_t> out = deflate_block(buf_.data(), buf_.size());
        if (std::fwrite(out.data(), 1, out.size(), f_) != out.size()) throw std::runtime_error("write failed");
        buf_.clear();
    }
};

// gz / BGZF / plain reader through zlib's gz layer (BGZF is a valid multi-member gzip stream)
class GzReader {
public:
    explicit GzReader(const std::string& path) {
        g_ = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
        if (!g_) throw std::runtime_error("The given input file could not be found: " + path);
        gzbuffer(g_, 1 << 20);
        buf_.resize(4u << 20);
    }
    ~GzReader() { if (g_) gzclose(g_); }
    bool read_exact(void* p, size_t n) {
        uint8_t* b = (uint8_t*)p;
        while (n) {
            if (pos_ == end_ && !fill()) return false;
            const size_t k = std::min(n, end_ - pos_);
            std::memcpy(b, buf_.data() + pos_, k);
            pos_ += k; b += k; n -= k;
        }
        return true;
    }
    // One line without its terminator as a view into the read buffer (valid until the next call); false at end of input.
    // The FASTQ parser works on these views: no per-line string, one memchr per line.
    bool line_view(const char*& p, size_t& n) {
        for (;;) {
            const void* nl = pos_ < end_ ? std::memchr(buf_.data() + pos_, '\n', end_ - pos_) : nullptr;
            if (nl) {
                const size_t e = (const uint8_t*)nl - buf_.data();
                p = (const char*)buf_.data() + pos_; n = e - pos_;
                pos_ = e + 1;
                if (n && p[n - 1] == '\r') n -= 1;
                return true;
            }
            // no terminator in the buffer: move the partial line to the front and read on
            if (pos_ > 0) { std::memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
            if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
            const int k = eof_ ? 0 : gzread(g_, buf_.data() + end_, (unsigned)std::min<size_t>(buf_.size() - end_, 1u << 30));
            if (k <= 0) {
                eof_ = true;
                if (end_ == 0) return false;
                p = (const char*)buf_.data(); n = end_; pos_ = end_ = 0;  // last line without a newline
                if (n && p[n - 1] == '\r') n -= 1;
                return true;
            }
            end_ += (size_t)k;
        }
    }
    // Up to `max_lines` whole lines as (offset, length) pairs into one contiguous block (valid until the next call): the text of a chunk of
    // FASTQ records, to be parsed by several threads.  Fewer lines only at the end of the input.
    const char* lines_block(size_t max_lines, std::vector<std::pair<uint32_t, uint32_t>>& lines) {
        lines.clear();
        if (pos_ > 0) { std::memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
        size_t scan = 0;
        while (lines.size() < max_lines) {
            const void* nl = scan < end_ ? std::memchr(buf_.data() + scan, '\n', end_ - scan) : nullptr;
            if (nl) {
                const size_t e = (const uint8_t*)nl - buf_.data();
                size_t n = e - scan;
                if (n && buf_[e - 1] == '\r') n -= 1;
                lines.emplace_back((uint32_t)scan, (uint32_t)n);
                scan = e + 1;
                continue;
            }
            if (end_ == buf_.size()) { if (buf_.size() >= (1ull << 31)) throw std::runtime_error("input chunk larger than 2 GiB of text"); buf_.resize(buf_.size() * 2); }
            const int k = eof_ ? 0 : gzread(g_, buf_.data() + end_, (unsigned)std::min<size_t>(buf_.size() - end_, 1u << 30));
            if (k <= 0) {
                eof_ = true;
                if (scan < end_) { size_t n = end_ - scan; if (buf_[end_ - 1] == '\r') n -= 1; lines.emplace_back((uint32_t)scan, (uint32_t)n); scan = end_; }
                break;
            }
            end_ += (size_t)k;
        }
        pos_ = scan;
        return (const char*)buf_.data();
    }
    void unread_from(uint32_t offset) { pos_ = offset; }  // hand the tail of the last lines_block() back (offset = start of the first unused line)
    bool getline(std::string& s) {
        const char* p; size_t n;
        if (!line_view(p, n)) { s.clear(); return false; }
        s.assign(p, n);
        return true;
    }
    int peek() { if (pos_ == end_ && !fill()) return -1; return buf_[pos_]; }

private:
    gzFile g_ = nullptr;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    bool fill() {
        if (eof_) return false;
        pos_ = end_ = 0;
        const int k = gzread(g_, buf_.data(), (unsigned)std::min<size_t>(buf_.size(), 1u << 30));
        if (k <= 0) { eof_ = true; return false; }
        end_ = (size_t)k;
        return true;
    }
};

// ---- records ------------------------------------------------------------------------------------------------------------
struct InRecord {  // Record (src/map/record.rs:129-136)
    std::string name;
    bool has_name = false;
    uint16_t flags = 0;
    std::string seq;            // mapping orientation (un-reversed if the input had 0x10, record.rs:157-160)
    std::vector<uint8_t> qual;  // raw Phred
    std::vector<uint8_t> aux;   // raw BAM aux bytes of the input record
};

inline char comp(char c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        default: return c;
    }
}
inline std::string revcomp(const std::string& s) { std::string r(s.rbegin(), s.rend()); for (auto& c : r) c = comp(c); return r; }

class ReadSource {
public:
    explicit ReadSource(const std::string& path) : in_(path) {
        // format sniffing (input_chunk_reader.rs:42-135): CRAM magic, BAM magic after gunzip, else FASTQ
        const int c = in_.peek();
        if (c == 'C') {
            char magic[4];
            if (!in_.read_exact(magic, 4) || std::memcmp(magic, "CRAM", 4) != 0) throw std::runtime_error("unrecognised input format");
            cram_.reset(new cram::Reader([this](uint8_t* p, size_t n) { return in_.read_exact(p, n); }, true));
            header_text_ = cram_->header_text();
            is_bam_ = true;  // records with flags and tags, not FASTQ text
        } else if (c == 'B') {
            char magic[4];
            if (!in_.read_exact(magic, 4) || std::memcmp(magic, "BAM\1", 4) != 0) throw std::runtime_error("unrecognised input format");
            is_bam_ = true;
            uint32_t l_text;
            if (!in_.read_exact(&l_text, 4)) throw std::runtime_error("truncated BAM header");
            header_text_.resize(l_text);
            if (l_text && !in_.read_exact(&header_text_[0], l_text)) throw std::runtime_error("truncated BAM header");
            while (!header_text_.empty() && header_text_.back() == '\0') header_text_.pop_back();
            uint32_t n_ref;
            if (!in_.read_exact(&n_ref, 4)) throw std::runtime_error("truncated BAM header");
            for (uint32_t i = 0; i < n_ref; ++i) {
                uint32_t l_name, l_ref;
                if (!in_.read_exact(&l_name, 4)) throw std::runtime_error("truncated BAM header");
                std::string nm(l_name, '\0');
                if (!in_.read_exact(&nm[0], l_name) || !in_.read_exact(&l_ref, 4)) throw std::runtime_error("truncated BAM header");
            }
        } else if (c == '@' || c < 0) {
            is_bam_ = false;
        } else throw std::runtime_error("unrecognised input format (expected BAM or FASTQ, optionally gzip-compressed)");
    }
    bool is_bam() const { return is_bam_; }
    const std::string& header_text() const { return header_text_; }

    // next record; false at end of input.  Malformed records are reported and skipped (input_chunk_reader.rs:200-214).
    bool next(InRecord& r) {
        return cram_ ? next_cram(r) : is_bam_ ? next_bam(r) : next_fastq(r);
    }
    // FASTQ only: the lines of up to `max_records` records (4 lines each) for parse_fastq_record(); nullptr for BAM input
    const char* fastq_block(size_t max_records, std::vector<std::pair<uint32_t, uint32_t>>& lines) {
        if (is_bam_) return nullptr;
        return in_.lines_block(4 * max_records, lines);
    }
    void fastq_unread_from(uint32_t offset) { in_.unread_from(offset); }
    // one record from its four lines; false if they do not form a FASTQ record
    static bool parse_fastq_record(const char* base, const std::pair<uint32_t, uint32_t>* ln, InRecord& r) {
        const char* h = base + ln[0].first; const size_t hn = ln[0].second;
        const char* sq = base + ln[1].first; const size_t sn = ln[1].second;
        const char* pl = base + ln[2].first; const size_t pn = ln[2].second;
        const char* q = base + ln[3].first; const size_t qn = ln[3].second;
        if (hn == 0 || h[0] != '@' || pn == 0 || pl[0] != '+' || sn != qn) return false;
        r = InRecord();
        size_t e = 1;
        while (e < hn && h[e] != ' ' && h[e] != '\t') ++e;
        r.name.assign(h + 1, e - 1);
        r.has_name = true; r.flags = 0;
        r.seq.resize(sn);
        for (size_t i = 0; i < sn; ++i) { const char ch = sq[i]; r.seq[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 'a' + 'A') : ch; }
        r.qual.resize(qn);
        for (size_t i = 0; i < qn; ++i) r.qual[i] = (uint8_t)(q[i] - 33);
        return true;
    }

private:
    GzReader in_;
    bool is_bam_ = false;
    std::unique_ptr<cram::Reader> cram_;
    std::string header_text_;

    bool next_cram(InRecord& r) {  // the same conversion as for a BAM record (record.rs:138-183)
        cram::Rec c;
        for (;;) {
            if (!cram_->next(c)) return false;
            if (!c.error.empty()) { std::fprintf(stderr, "Skip record due to an error: CRAM record \"%s\": %s\n", c.name.c_str(), c.error.c_str()); continue; }
            r = InRecord();
            r.name = std::move(c.name); r.has_name = c.has_name; r.flags = c.flags;
            r.seq = std::move(c.seq); r.qual = std::move(c.qual); r.aux = std::move(c.aux);
            for (auto& ch : r.seq) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
            if (r.flags & 0x10) { r.seq = revcomp(r.seq); std::reverse(r.qual.begin(), r.qual.end()); }
            return true;
        }
    }

    bool next_fastq(InRecord& r) {  // TryFrom<fastq::Record> (record.rs:185-215)
        const char* p; size_t n;
        for (;;) {
            if (!in_.line_view(p, n)) return false;
            if (n == 0) continue;
            const bool head_ok = p[0] == '@';
            r = InRecord();
            size_t e = 1;
            while (e < n && p[e] != ' ' && p[e] != '\t') ++e;
            if (head_ok) r.name.assign(p + 1, e - 1);
            r.has_name = true; r.flags = 0;
            if (!in_.line_view(p, n)) return false;
            r.seq.resize(n);
            for (size_t i = 0; i < n; ++i) { const char ch = p[i]; r.seq[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 'a' + 'A') : ch; }
            if (!in_.line_view(p, n)) return false;
            const bool plus_ok = n > 0 && p[0] == '+';
            if (!in_.line_view(p, n)) return false;
            if (!head_ok || !plus_ok || n != r.seq.size()) { std::fprintf(stderr, "Skip record due to an error: malformed FASTQ record\n"); continue; }
            r.qual.resize(n);
            for (size_t i = 0; i < n; ++i) r.qual[i] = (uint8_t)(p[i] - 33);
            return true;
        }
    }
    bool next_bam(InRecord& r) {  // TryFrom<&dyn sam::alignment::Record> (record.rs:138-183)
        uint32_t block_size;
        if (!in_.read_exact(&block_size, 4)) return false;
        std::vector<uint8_t> b(block_size);
        if (!in_.read_exact(b.data(), block_size) || block_size < 32) throw std::runtime_error("truncated BAM record");
        const uint8_t l_read_name = b[8];
        uint16_t n_cigar, flag;
        uint32_t l_seq;
        std::memcpy(&n_cigar, &b[12], 2); std::memcpy(&flag, &b[14], 2); std::memcpy(&l_seq, &b[16], 4);
        size_t o = 32;
        r = InRecord();
        r.name.assign((const char*)&b[o], l_read_name ? l_read_name - 1 : 0);
        r.has_name = !(r.name.empty() || r.name == "*");
        o += l_read_name + 4ull * n_cigar;
        static const char code[] = "=ACMGRSVTWYHKDBN";
        r.seq.resize(l_seq);
        for (uint32_t i = 0; i < l_seq; ++i) r.seq[i] = code[(b[o + i / 2] >> (i % 2 ? 0 : 4)) & 0xF];
        o += (l_seq + 1) / 2;
        r.qual.assign(b.begin() + o, b.begin() + o + l_seq);
        o += l_seq;
        r.aux.assign(b.begin() + o, b.end());
        r.flags = flag;
        if (flag & 0x10) { r.seq = revcomp(r.seq); std::reverse(r.qual.begin(), r.qual.end()); }
        return true;
    }
};

// size in bytes of one BAM aux field starting at p (tag[2] type value...), 0 if malformed
inline size_t aux_field_size(const uint8_t* p, size_t left) {
    if (left < 3) return 0;
    auto fixed = [](char t) -> size_t { switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; } };
    const char t = (char)p[2];
    if (size_t f = fixed(t)) return 3 + f <= left ? 3 + f : 0;
    if (t == 'Z' || t == 'H') { for (size_t i = 3; i < left; ++i) if (p[i] == 0) return i + 1; return 0; }
    if (t == 'B') {
        if (left < 8) return 0;
        const size_t f = fixed((char)p[3]);
        uint32_t n; std::memcpy(&n, p + 4, 4);
        return f && 8 + f * n <= left ? 8 + f * n : 0;
    }
    return 0;
}

inline uint16_t reg2bin(int64_t beg, int64_t end) {  // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct OutFields {  // what create_bam_record (mapping.rs:722-927) puts into one record
    bool mapped = false, reverse = false;
    uint16_t flags = 0;
    int32_t tid = -1;
    int64_t pos = -1;
    uint8_t mapq = 0;
    std::string cigar, md, xa;
    float as = 0, xs = 0, xd = 0;
    int32_t nm = 0, x0 = 0, x1 = 0;
    bool has_xs = false, has_alt = false;
    char xt = 'N';
};

inline void encode_bam_record(const InRecord& in, const OutFields& f, const std::string& read_group, std::vector<uint8_t>& out) {
    std::vector<uint8_t> rec(32, 0);
    const std::string name = in.has_name ? in.name : "*";
    // CIGAR string -> ops
    std::vector<uint32_t> ops;
    int64_t ref_len = 0;
    for (size_t i = 0; i < f.cigar.size();) {
        uint32_t n = 0;
        while (i < f.cigar.size() && std::isdigit((unsigned char)f.cigar[i])) n = n * 10 + (uint32_t)(f.cigar[i++] - '0');
        const char k = f.cigar[i++];
        const uint32_t code = k == 'M' ? 0 : k == 'I' ? 1 : 2;
        ops.push_back(n << 4 | code);
        if (k != 'I') ref_len += n;
    }
    const std::string seq = f.reverse ? revcomp(in.seq) : in.seq;  // mapping.rs:795-819
    std::vector<uint8_t> qual = in.qual;
    if (f.reverse) std::reverse(qual.begin(), qual.end());
    const int32_t pos32 = (int32_t)f.pos, tid = f.tid, next = -1, tlen = 0;
    const uint32_t l_seq = (uint32_t)seq.size();
    const uint16_t bin = reg2bin(f.pos < 0 ? -1 : f.pos, f.pos < 0 ? 0 : f.pos + (ref_len ? ref_len : 1)), n_cig = (uint16_t)ops.size();
    std::memcpy(&rec[0], &tid, 4); std::memcpy(&rec[4], &pos32, 4);
    rec[8] = (uint8_t)(name.size() + 1); rec[9] = f.mapq;
    std::memcpy(&rec[10], &bin, 2); std::memcpy(&rec[12], &n_cig, 2); std::memcpy(&rec[14], &f.flags, 2); std::memcpy(&rec[16], &l_seq, 4);
    std::memcpy(&rec[20], &next, 4); std::memcpy(&rec[24], &next, 4); std::memcpy(&rec[28], &tlen, 4);
    rec.insert(rec.end(), name.begin(), name.end()); rec.push_back(0);
    for (uint32_t op : ops) { const uint8_t* p = (const uint8_t*)&op; rec.insert(rec.end(), p, p + 4); }
    static const auto nib = [](char c) -> uint8_t { const char* t = "=ACMGRSVTWYHKDBN"; const char* q = std::strchr(t, c); return q ? (uint8_t)(q - t) : 15; };
    for (uint32_t i = 0; i < l_seq; i += 2) rec.push_back((uint8_t)(nib(seq[i]) << 4 | (i + 1 < l_seq ? nib(seq[i + 1]) : 0)));
    rec.insert(rec.end(), qual.begin(), qual.end());
    // aux: input tags minus the BWA/mapAD specific ones (mapping.rs:834-848), then the new ones in the reference's order (:850-918)
    static const char* drop[] = {"AS", "MD", "NM", "X0", "X1", "XA", "XD", "XE", "XF", "XG", "XM", "XN", "XO", "XS", "XT"};
    for (size_t o = 0; o < in.aux.size();) {
        const size_t n = aux_field_size(in.aux.data() + o, in.aux.size() - o);
        if (!n) break;
        bool skip = !read_group.empty() && in.aux[o] == 'R' && in.aux[o + 1] == 'G';
        for (const char* d : drop) skip |= in.aux[o] == (uint8_t)d[0] && in.aux[o + 1] == (uint8_t)d[1];
        if (!skip) rec.insert(rec.end(), in.aux.begin() + o, in.aux.begin() + o + n);
        o += n;
    }
    auto tag_z = [&](const char* t, const std::string& v) { rec.push_back(t[0]); rec.push_back(t[1]); rec.push_back('Z'); rec.insert(rec.end(), v.begin(), v.end()); rec.push_back(0); };
    auto tag_f = [&](const char* t, float v) { rec.push_back(t[0]); rec.push_back(t[1]); rec.push_back('f'); const uint8_t* p = (const uint8_t*)&v; rec.insert(rec.end(), p, p + 4); };
    auto tag_i = [&](const char* t, int32_t v) { rec.push_back(t[0]); rec.push_back(t[1]); rec.push_back('i'); const uint8_t* p = (const uint8_t*)&v; rec.insert(rec.end(), p, p + 4); };
    if (!read_group.empty()) tag_z("RG", read_group);
    if (f.mapped) { tag_f("AS", f.as); tag_i("NM", f.nm); tag_z("MD", f.md); }
    if (f.has_alt) {
        if (!f.xa.empty()) tag_z("XA", f.xa);
        tag_i("X0", f.x0); tag_i("X1", f.x1);
        if (f.x1 > 0) tag_f("XS", f.xs);
        rec.push_back('X'); rec.push_back('T'); rec.push_back('A'); rec.push_back((uint8_t)f.xt);
    }
    tag_f("XD", f.xd);
    const uint32_t bs = (uint32_t)rec.size();
    const uint8_t* p = (const uint8_t*)&bs;
    out.insert(out.end(), p, p + 4);
    out.insert(out.end(), rec.begin(), rec.end());
}

}  // namespace cli
}  // namespace mapad
