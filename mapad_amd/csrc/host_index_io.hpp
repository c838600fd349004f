// host_index_io.hpp — the reference's on-disk index: 7 files, each a snappy frame stream of bincode 1.3
// (little-endian, fixed-width ints, u64 lengths) of Item{version: u8 = 5, data: T}
// (src/index/versioned_index.rs:11-55; writers src/index/indexing.rs:110-208; loaders src/index/mod.rs:212-239).
//
//   .tbw Vec<u8> rank BWT        .tle Vec<u64> Less            .toc Occ{occ: Vec<Vec<u64>>, k: u32} (rust-bio layout)
//   .trt RankTransform{ranks: VecMap<u8>} (map u64 key -> u8)   .tsa {sample: Vec<u64>, sampling_rate: u64, extra_rows: map<u64,u64>, sentinel: u8}
//   .tpi {Vec<{start: u64, end: u64, identifier: String}>}      .tos map<u64, u8>
//
// Reading accepts compressed and uncompressed snappy chunks; writing emits uncompressed chunks (valid frame format,
// readable by snap::read::FrameDecoder).  The fork's Occ byte layout is unpinned: .toc is only validated
// (outer length, inner lengths, k) and its counts are skipped — ranks are answered from the block layout rebuilt from .tbw.
// The large files (.tbw, .toc, .tsa) are streamed chunk by chunk in both directions; nothing is held twice.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/mapad_amd.h"
#include "host_index.hpp"

namespace mapad {
namespace host {
namespace io {

inline uint32_t crc32c_table(uint32_t c, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t v = i; for (int k = 0; k < 8; ++k) v = (v & 1) ? (v >> 1) ^ 0x82F63B78u : v >> 1; table[i] = v; }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) inline uint32_t crc32c_hw(uint32_t c, const uint8_t* p, size_t n) {
    uint64_t c64 = c;
    while (n >= 8) { uint64_t v; std::memcpy(&v, p, 8); c64 = __builtin_ia32_crc32di(c64, v); p += 8; n -= 8; }
    c = (uint32_t)c64;
    while (n--) c = __builtin_ia32_crc32qi(c, *p++);
    return c;
}
#endif
inline uint32_t crc32c(const uint8_t* p, size_t n) {  // a 6 GB .tbw is checked at memory speed with the CPU's CRC32C instruction
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
#endif
    return crc32c_table(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
}
inline uint32_t masked_crc(const uint8_t* p, size_t n) { const uint32_t c = crc32c(p, n); return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// raw snappy block -> bytes
inline bool snappy_uncompress(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
    size_t i = 0; uint64_t ulen = 0; int shift = 0;
    for (;;) { if (i >= n) return false; const uint8_t b = in[i++]; ulen |= (uint64_t)(b & 0x7F) << shift; if (!(b & 0x80)) break; shift += 7; if (shift > 35) return false; }
    const size_t base = out.size();
    out.reserve(base + ulen);
    while (i < n) {
        const uint8_t tag = in[i++];
        size_t len, off;
        switch (tag & 3) {
            case 0: {
                len = (tag >> 2) + 1;
                if (len > 60) { const size_t nb = len - 60; if (i + nb > n) return false; len = 0; for (size_t k = 0; k < nb; ++k) len |= (size_t)in[i + k] << (8 * k); len += 1; i += nb; }
                if (i + len > n) return false;
                out.insert(out.end(), in + i, in + i + len); i += len;
                continue;
            }
            case 1: if (i >= n) return false; len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i++]; break;
            case 2: if (i + 2 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8); i += 2; break;
            default: if (i + 4 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4;
        }
        if (off == 0 || off > out.size() - base) return false;
        for (size_t k = 0; k < len; ++k) out.push_back(out[out.size() - off]);
    }
    return out.size() - base == ulen;
}

// Decoded byte stream of a snappy frame file, one chunk (<= 64 KiB of payload) in memory at a time: a 6 GB .tbw is copied straight into its
// destination and the 2.25 GB .toc is skipped without being materialised.
class FrameStream {
public:
    explicit FrameStream(const std::string& path) : f_(std::fopen(path.c_str(), "rb")) { if (f_) std::setvbuf(f_, nullptr, _IOFBF, 1 << 22); }
    ~FrameStream() { if (f_) std::fclose(f_); }
    bool is_open() const { return f_ != nullptr; }
    int error() const { return err_; }
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos_ == chunk_.size() && !next_chunk()) { if (!err_) err_ = MAPAD_ERR_PARSE; return false; }
            const size_t k = std::min(n, chunk_.size() - pos_);
            if (d) { std::memcpy(d, chunk_.data() + pos_, k); d += k; }
            pos_ += k; n -= k;
        }
        return true;
    }
    bool skip(size_t n) { return read(nullptr, n); }
    uint8_t u8() { uint8_t v = 0; read(&v, 1); return v; }
    uint32_t u32() { uint32_t v = 0; read(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; read(&v, 8); return v; }
    bool at_end() { return pos_ == chunk_.size() && !next_chunk() && err_ == MAPAD_OK; }

private:
    FILE* f_;
    std::vector<uint8_t> raw_, chunk_;
    size_t pos_ = 0;
    int err_ = MAPAD_OK;
    bool next_chunk() {  // false at end of file or on error (err_)
        for (;;) {
            uint8_t h[4];
            const size_t got = std::fread(h, 1, 4, f_);
            if (got == 0) return false;
            if (got != 4) { err_ = MAPAD_ERR_PARSE; return false; }
            const uint8_t type = h[0];
            const size_t len = h[1] | ((size_t)h[2] << 8) | ((size_t)h[3] << 16);
            raw_.resize(len);
            if (len && std::fread(raw_.data(), 1, len, f_) != len) { err_ = MAPAD_ERR_PARSE; return false; }
            if (type == 0xFF) { if (len != 6 || std::memcmp(raw_.data(), "sNaPpY", 6) != 0) { err_ = MAPAD_ERR_PARSE; return false; } continue; }
            if (type == 0x00 || type == 0x01) {
                if (len < 4) { err_ = MAPAD_ERR_PARSE; return false; }
                const uint32_t want = raw_[0] | (raw_[1] << 8) | (raw_[2] << 16) | ((uint32_t)raw_[3] << 24);
                chunk_.clear(); pos_ = 0;
                if (type == 0x01) chunk_.assign(raw_.begin() + 4, raw_.end());
                else if (!snappy_uncompress(raw_.data() + 4, len - 4, chunk_)) { err_ = MAPAD_ERR_PARSE; return false; }
                if (masked_crc(chunk_.data(), chunk_.size()) != want) { err_ = MAPAD_ERR_PARSE; return false; }
                if (chunk_.empty()) continue;
                return true;
            }
            if (type >= 0x02 && type <= 0x7F) { err_ = MAPAD_ERR_PARSE; return false; }  // reserved unskippable
            // 0x80..0xFE: skippable
        }
    }
};
inline int read_frames(const std::string& path, std::vector<uint8_t>& out) {  // small files
    FrameStream fs(path);
    if (!fs.is_open()) return MAPAD_ERR_IO;
    uint8_t buf[4096];
    for (;;) {
        if (fs.at_end()) break;
        size_t k = 0;
        while (k < sizeof buf && !fs.at_end()) { if (!fs.read(buf + k, 1)) return fs.error() ? fs.error() : MAPAD_ERR_PARSE; ++k; }
        out.insert(out.end(), buf, buf + k);
    }
    return fs.error();
}
// frame writer: uncompressed chunks, fed incrementally
class FrameWriter {
public:
    explicit FrameWriter(const std::string& path) : f_(std::fopen(path.c_str(), "wb")) {
        if (f_) { std::setvbuf(f_, nullptr, _IOFBF, 1 << 22); static const uint8_t ident[10] = {0xFF, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y'}; ok_ = std::fwrite(ident, 1, 10, f_) == 10; }
        buf_.reserve(65536);
    }
    ~FrameWriter() { if (f_) std::fclose(f_); }
    void bytes(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        while (n) {
            const size_t k = std::min(n, (size_t)65536 - buf_.size());
            buf_.insert(buf_.end(), b, b + k);
            b += k; n -= k;
            if (buf_.size() == 65536) flush();
        }
    }
    void u8(uint8_t v) { bytes(&v, 1); }
    void u32(uint32_t v) { bytes(&v, 4); }
    void u64(uint64_t v) { bytes(&v, 8); }
    int close() {
        if (!f_) return MAPAD_ERR_IO;
        if (!buf_.empty()) flush();
        const bool good = ok_ && std::fclose(f_) == 0;
        f_ = nullptr;
        return good ? MAPAD_OK : MAPAD_ERR_IO;
    }
private:
    FILE* f_;
    bool ok_ = false;
    std::vector<uint8_t> buf_;
    void flush() {
        const size_t n = buf_.size();
        const uint32_t crc = masked_crc(buf_.data(), n);
        const size_t len = n + 4;
        const uint8_t hdr[8] = {0x01, (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24)};
        ok_ = ok_ && std::fwrite(hdr, 1, 8, f_) == 8 && std::fwrite(buf_.data(), 1, n, f_) == n;
        buf_.clear();
    }
};
inline int write_frames(const std::string& path, const std::vector<uint8_t>& data) {
    FrameWriter w(path);
    w.bytes(data.data(), data.size());
    return w.close();
}

struct Reader {
    const std::vector<uint8_t>& b; size_t i = 0; bool ok = true;
    explicit Reader(const std::vector<uint8_t>& v) : b(v) {}
    uint8_t u8() { if (i + 1 > b.size()) { ok = false; return 0; } return b[i++]; }
    uint32_t u32() { if (i + 4 > b.size()) { ok = false; return 0; } uint32_t v; std::memcpy(&v, &b[i], 4); i += 4; return v; }
    uint64_t u64() { if (i + 8 > b.size()) { ok = false; return 0; } uint64_t v; std::memcpy(&v, &b[i], 8); i += 8; return v; }
    bool bytes(size_t n, const uint8_t*& p) { if (i + n > b.size()) { ok = false; return false; } p = &b[i]; i += n; return true; }
};
struct Writer {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u32(uint32_t v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + 4); }
    void u64(uint64_t v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + 8); }
    void bytes(const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};
constexpr uint8_t kIndexVersion = 5;  // versioned_index.rs:19

}  // namespace io

inline int load_index(const std::string& prefix, Index& ix) {
    using namespace io;
    std::vector<uint8_t> buf;
    int rc;
    auto open = [&](const char* ext) -> int {
        buf.clear();
        if ((rc = read_frames(prefix + ext, buf))) return rc;
        if (buf.empty()) return MAPAD_ERR_PARSE;
        if (buf[0] != kIndexVersion) { std::fprintf(stderr, "mapad_amd: index version mismatch in %s%s: on disk %u, expected %u\n", prefix.c_str(), ext, buf[0], kIndexVersion); return MAPAD_ERR_INDEX_VERSION; }
        return MAPAD_OK;
    };
    ix = Index();
    auto open_stream = [&](FrameStream& fs, const char* ext) -> int {
        if (!fs.is_open()) return MAPAD_ERR_IO;
        const uint8_t version = fs.u8();
        if (fs.error()) return fs.error();
        if (version != kIndexVersion) { std::fprintf(stderr, "mapad_amd: index version mismatch in %s%s: on disk %u, expected %u\n", prefix.c_str(), ext, version, kIndexVersion); return MAPAD_ERR_INDEX_VERSION; }
        return MAPAD_OK;
    };
    {   // .tbw: streamed straight into the BWT vector
        FrameStream fs(prefix + ".tbw");
        if ((rc = open_stream(fs, ".tbw"))) return rc;
        const uint64_t n = fs.u64();
        if (fs.error() || n >= (1ull << 40)) return MAPAD_ERR_PARSE;
        ix.bwt.resize(n); ix.n = n;
        if (!fs.read(ix.bwt.data(), n) || !fs.at_end()) return MAPAD_ERR_PARSE;
    }
    std::vector<uint64_t> less_disk;
    {   // .tle
        if ((rc = open(".tle"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64();
        if (n > 64) return MAPAD_ERR_PARSE;
        for (uint64_t i = 0; i < n; ++i) less_disk.push_back(r.u64());
        if (!r.ok) return MAPAD_ERR_PARSE;
    }
    {   // .toc — shape validated (outer length, inner lengths against n / k, k), counts skipped: ranks are answered from the blocks rebuilt from .tbw
        FrameStream fs(prefix + ".toc");
        if ((rc = open_stream(fs, ".toc"))) return rc;
        const uint64_t outer = fs.u64();
        if (fs.error() || outer == 0 || outer > 256) return MAPAD_ERR_PARSE;
        uint64_t inner0 = 0;
        for (uint64_t sidx = 0; sidx < outer; ++sidx) {
            const uint64_t inner = fs.u64();
            if (fs.error() || inner > ix.n + 1 || (sidx && inner != inner0)) return MAPAD_ERR_PARSE;
            inner0 = inner;
            if (!fs.skip(inner * 8)) return MAPAD_ERR_PARSE;
        }
        const uint32_t k = fs.u32();
        if (fs.error() || k == 0 || !fs.at_end()) return MAPAD_ERR_PARSE;
        if (inner0 != (ix.n + k - 1) / k) return MAPAD_ERR_PARSE;  // one checkpoint per k rows
    }
    {   // .trt — must be the $ACGTX (or $ACGT) rank transform
        if ((rc = open(".trt"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64();
        static const char expect[] = "$ACGTX";
        if (n < 5 || n > 6) return MAPAD_ERR_PARSE;
        for (uint64_t i = 0; i < n; ++i) { const uint64_t key = r.u64(); const uint8_t val = r.u8(); if (!r.ok || key != (uint64_t)expect[i] || val != i) return MAPAD_ERR_PARSE; }
    }
    {   // .tsa: the sample array is streamed into place
        FrameStream fs(prefix + ".tsa");
        if ((rc = open_stream(fs, ".tsa"))) return rc;
        const uint64_t n = fs.u64();
        if (fs.error() || n > ix.n + 1) return MAPAD_ERR_PARSE;
        ix.sa_sample.resize(n);
        if (!fs.read(ix.sa_sample.data(), n * 8)) return MAPAD_ERR_PARSE;
        ix.sa_rate = fs.u64();
        const uint64_t m = fs.u64();
        if (fs.error() || m > 1024) return MAPAD_ERR_PARSE;
        for (uint64_t i = 0; i < m; ++i) { const uint64_t k = fs.u64(), v = fs.u64(); ix.extra_rows[k] = v; }
        const uint8_t sentinel = fs.u8();
        if (fs.error() || ix.sa_rate == 0 || sentinel != 0 || !fs.at_end()) return MAPAD_ERR_PARSE;
    }
    {   // .tpi
        if ((rc = open(".tpi"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64();
        for (uint64_t i = 0; i < n && r.ok; ++i) {
            Contig c; c.start = r.u64(); c.end = r.u64();
            const uint64_t l = r.u64(); const uint8_t* p;
            if (!r.bytes(l, p)) return MAPAD_ERR_PARSE;
            c.name.assign((const char*)p, l);
            ix.contigs.push_back(c);
        }
        if (!r.ok) return MAPAD_ERR_PARSE;
    }
    {   // .tos
        if ((rc = open(".tos"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64();
        for (uint64_t i = 0; i < n && r.ok; ++i) { const uint64_t k = r.u64(); const uint8_t v = r.u8(); ix.original_symbols[k] = v; }
        if (!r.ok) return MAPAD_ERR_PARSE;
    }
    try { build_blocks(ix); } catch (const std::exception& e) { std::fprintf(stderr, "mapad_amd: %s\n", e.what()); return MAPAD_ERR_PARSE; }
    for (size_t c = 0; c < less_disk.size() && c < 7; ++c) if (less_disk[c] != ix.less[c]) return MAPAD_ERR_PARSE;  // .tle must agree with .tbw
    if (ix.sa_sample.size() != (ix.n + ix.sa_rate - 1) / ix.sa_rate) return MAPAD_ERR_PARSE;
    return MAPAD_OK;
}

inline int save_index(const std::string& prefix, const Index& ix) {
    using namespace io;
    int rc;
    {   FrameWriter w(prefix + ".tbw"); w.u8(kIndexVersion); w.u64(ix.bwt.size()); w.bytes(ix.bwt.data(), ix.bwt.size());
        if ((rc = w.close())) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(7); for (int c = 0; c < 7; ++c) w.u64(ix.less[c]);  // less: max_symbol + 2 entries
        if ((rc = write_frames(prefix + ".tle", w.b))) return rc; }
    {   // Occ::new(bwt, 128, alphabet 0..6) (indexing.rs:188): per symbol, the inclusive count at every row i with i % k == 0
        const uint32_t k = 128;
        const uint64_t n_cp = (ix.n + k - 1) / k;
        FrameWriter w(prefix + ".toc"); w.u8(kIndexVersion); w.u64(6);
        std::vector<uint64_t> col(n_cp);
        for (int sym = 0; sym < 6; ++sym) {  // one pass per symbol keeps the memory at one column (n / 128 x 8 B)
            uint64_t cur = 0;
            for (uint64_t i = 0; i < ix.n; ++i) { cur += ix.bwt[i] == sym; if (i % k == 0) col[i / k] = cur; }
            w.u64(n_cp); w.bytes(col.data(), n_cp * 8);
        }
        w.u32(k);
        if ((rc = w.close())) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(6); static const char sym[] = "$ACGTX"; for (int i = 0; i < 6; ++i) { w.u64((uint64_t)sym[i]); w.u8((uint8_t)i); }
        if ((rc = write_frames(prefix + ".trt", w.b))) return rc; }
    {   FrameWriter w(prefix + ".tsa"); w.u8(kIndexVersion); w.u64(ix.sa_sample.size()); w.bytes(ix.sa_sample.data(), ix.sa_sample.size() * 8); w.u64(ix.sa_rate);
        w.u64(ix.extra_rows.size()); for (auto& kv : ix.extra_rows) { w.u64(kv.first); w.u64(kv.second); } w.u8(0);
        if ((rc = w.close())) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.contigs.size());
        for (auto& c : ix.contigs) { w.u64(c.start); w.u64(c.end); w.u64(c.name.size()); w.bytes(c.name.data(), c.name.size()); }
        if ((rc = write_frames(prefix + ".tpi", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.original_symbols.size()); for (auto& kv : ix.original_symbols) { w.u64(kv.first); w.u8(kv.second); }
        if ((rc = write_frames(prefix + ".tos", w.b))) return rc; }
    return MAPAD_OK;
}

}  // namespace host
}  // namespace mapad
