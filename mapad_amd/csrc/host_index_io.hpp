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
// (outer length, k) and otherwise ignored — ranks are answered from the block layout rebuilt from .tbw.
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

inline uint32_t crc32c(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
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

inline int read_frames(const std::string& path, std::vector<uint8_t>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return MAPAD_ERR_IO;
    std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t i = 0;
    while (i + 4 <= raw.size()) {
        const uint8_t type = raw[i];
        const size_t len = raw[i + 1] | ((size_t)raw[i + 2] << 8) | ((size_t)raw[i + 3] << 16);
        i += 4;
        if (i + len > raw.size()) return MAPAD_ERR_PARSE;
        if (type == 0xFF) { if (len != 6 || std::memcmp(&raw[i], "sNaPpY", 6) != 0) return MAPAD_ERR_PARSE; }
        else if (type == 0x00 || type == 0x01) {
            if (len < 4) return MAPAD_ERR_PARSE;
            const uint32_t want = raw[i] | (raw[i + 1] << 8) | (raw[i + 2] << 16) | ((uint32_t)raw[i + 3] << 24);
            const size_t before = out.size();
            if (type == 0x01) out.insert(out.end(), raw.begin() + i + 4, raw.begin() + i + len);
            else if (!snappy_uncompress(&raw[i + 4], len - 4, out)) return MAPAD_ERR_PARSE;
            if (masked_crc(out.data() + before, out.size() - before) != want) return MAPAD_ERR_PARSE;
        } else if (type >= 0x02 && type <= 0x7F) return MAPAD_ERR_PARSE;  // reserved unskippable
        i += len;
    }
    return i == raw.size() ? MAPAD_OK : MAPAD_ERR_PARSE;
}
inline int write_frames(const std::string& path, const std::vector<uint8_t>& data) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) return MAPAD_ERR_IO;
    static const uint8_t ident[10] = {0xFF, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y'};
    f.write((const char*)ident, 10);
    for (size_t i = 0; i < data.size() || (i == 0 && data.empty()); i += 65536) {
        const size_t n = std::min<size_t>(65536, data.size() - i);
        if (n == 0) break;
        const uint32_t crc = masked_crc(data.data() + i, n);
        const size_t len = n + 4;
        const uint8_t hdr[8] = {0x01, (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24)};
        f.write((const char*)hdr, 8);
        f.write((const char*)data.data() + i, (std::streamsize)n);
    }
    return f ? MAPAD_OK : MAPAD_ERR_IO;
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
    {   // .tbw
        if ((rc = open(".tbw"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64(); const uint8_t* p;
        if (!r.bytes(n, p) || r.i != buf.size()) return MAPAD_ERR_PARSE;
        ix.bwt.assign(p, p + n); ix.n = n;
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
    {   // .toc — validated, not used
        if ((rc = open(".toc"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t outer = r.u64();
        if (!r.ok || outer == 0 || outer > 256) return MAPAD_ERR_PARSE;
        for (uint64_t s = 0; s < outer; ++s) { const uint64_t inner = r.u64(); const uint8_t* p; if (!r.ok || !r.bytes(inner * 8, p)) return MAPAD_ERR_PARSE; }
        const uint32_t k = r.u32();
        if (!r.ok || k == 0) return MAPAD_ERR_PARSE;
    }
    {   // .trt — must be the $ACGTX (or $ACGT) rank transform
        if ((rc = open(".trt"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64();
        static const char expect[] = "$ACGTX";
        if (n < 5 || n > 6) return MAPAD_ERR_PARSE;
        for (uint64_t i = 0; i < n; ++i) { const uint64_t key = r.u64(); const uint8_t val = r.u8(); if (!r.ok || key != (uint64_t)expect[i] || val != i) return MAPAD_ERR_PARSE; }
    }
    {   // .tsa
        if ((rc = open(".tsa"))) return rc;
        Reader r(buf); r.u8();
        const uint64_t n = r.u64(); const uint8_t* p;
        if (!r.bytes(n * 8, p)) return MAPAD_ERR_PARSE;
        ix.sa_sample.resize(n); std::memcpy(ix.sa_sample.data(), p, n * 8);
        ix.sa_rate = r.u64();
        const uint64_t m = r.u64();
        for (uint64_t i = 0; i < m && r.ok; ++i) { const uint64_t k = r.u64(), v = r.u64(); ix.extra_rows[k] = v; }
        const uint8_t sentinel = r.u8();
        if (!r.ok || ix.sa_rate == 0 || sentinel != 0) return MAPAD_ERR_PARSE;
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
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.bwt.size()); w.bytes(ix.bwt.data(), ix.bwt.size());
        if ((rc = write_frames(prefix + ".tbw", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(7); for (int c = 0; c < 7; ++c) w.u64(ix.less[c]);  // less: max_symbol + 2 entries
        if ((rc = write_frames(prefix + ".tle", w.b))) return rc; }
    {   // Occ::new(bwt, 128, alphabet 0..6) (indexing.rs:188)
        const uint32_t k = 128;
        std::vector<std::vector<uint64_t>> occ(6);
        uint64_t cur[6] = {0, 0, 0, 0, 0, 0};
        for (uint64_t i = 0; i < ix.n; ++i) { cur[ix.bwt[i]] += 1; if (i % k == 0) for (int s = 0; s < 6; ++s) occ[s].push_back(cur[s]); }
        Writer w; w.u8(kIndexVersion); w.u64(6);
        for (int s = 0; s < 6; ++s) { w.u64(occ[s].size()); w.bytes(occ[s].data(), occ[s].size() * 8); }
        w.u32(k);
        if ((rc = write_frames(prefix + ".toc", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(6); static const char sym[] = "$ACGTX"; for (int i = 0; i < 6; ++i) { w.u64((uint64_t)sym[i]); w.u8((uint8_t)i); }
        if ((rc = write_frames(prefix + ".trt", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.sa_sample.size()); w.bytes(ix.sa_sample.data(), ix.sa_sample.size() * 8); w.u64(ix.sa_rate);
        w.u64(ix.extra_rows.size()); for (auto& kv : ix.extra_rows) { w.u64(kv.first); w.u64(kv.second); } w.u8(0);
        if ((rc = write_frames(prefix + ".tsa", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.contigs.size());
        for (auto& c : ix.contigs) { w.u64(c.start); w.u64(c.end); w.u64(c.name.size()); w.bytes(c.name.data(), c.name.size()); }
        if ((rc = write_frames(prefix + ".tpi", w.b))) return rc; }
    {   Writer w; w.u8(kIndexVersion); w.u64(ix.original_symbols.size()); for (auto& kv : ix.original_symbols) { w.u64(kv.first); w.u8(kv.second); }
        if ((rc = write_frames(prefix + ".tos", w.b))) return rc; }
    return MAPAD_OK;
}

}  // namespace host
}  // namespace mapad
