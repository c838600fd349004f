// cram_io.hpp — CRAM 3.0 records in, for the `mapad-amd` command line (host I/O, not on the accelerated path).
//
// The reference reads CRAM through noodles 0.98 (Cargo.toml:28; its noodles-cram component; `cram::io::Reader::new(file)` WITHOUT a reference-sequence repository,
// src/map/input_chunk_reader.rs:27,86-95,160-168) and turns every alignment record into its `Record` (name, flags, bases, qualities, tags:
// src/map/record.rs:138-183).  noodles is a dependency that is not part of /root/reference; this file restates the published format (CRAM format
// specification 3.0, samtools/hts-specs) for what that call can deliver: the file definition, containers, blocks (raw, gzip, rANS 4x8 orders 0 and 1),
// the compression header (preservation map, data-series and tag encodings: EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA, GAMMA, SUBEXP),
// slices and records.  Bases come from the record's own data (unmapped reads, the normal input of a mapper; mapped reads whose features spell out
// every base) or from a reference embedded in the slice; a record that needs an external reference is reported and skipped, as the reference's
// reader without a repository cannot decode it either.  bzip2 / lzma blocks (no such library in this image) and the CRAM 3.1 codecs are refused by name.
// PARITY UNPINNED: there is no CRAM file and no CRAM writer in /root/reference or in this image; tests/cram_util.py writes files from the same reading
// of the specification.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <array>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mapad {
namespace cli {
namespace cram {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
// sizes a file may state before anything has been checked against them: beyond these it is taken for corrupt rather than allocated
constexpr int64_t kMaxBlockBytes = 1ll << 30, kMaxSliceRecords = 1 << 24, kMaxReadLen = 1 << 26;

// ---- byte cursor with the format's integer codes -------------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p = nullptr;
    size_t n = 0, at = 0;
    Cursor() = default;
    Cursor(const uint8_t* d, size_t len) : p(d), n(len) {}
    uint8_t u8() { if (at >= n) throw Error("CRAM: read past the end of a block"); return p[at++]; }
    const uint8_t* bytes(size_t k) { if (k > n - at) throw Error("CRAM: read past the end of a block"); const uint8_t* r = p + at; at += k; return r; }
    int32_t i32le() { const uint8_t* b = bytes(4); return (int32_t)((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24); }
    int32_t itf8() {  // specification 2.3: the number of leading one bits of the first byte says how many bytes follow
        const uint32_t b0 = u8();
        if (b0 < 0x80) return (int32_t)b0;
        if (b0 < 0xC0) return (int32_t)(((b0 & 0x3F) << 8) | u8());
        if (b0 < 0xE0) { const uint32_t b1 = u8(), b2 = u8(); return (int32_t)(((b0 & 0x1F) << 16) | (b1 << 8) | b2); }
        if (b0 < 0xF0) { const uint32_t b1 = u8(), b2 = u8(), b3 = u8(); return (int32_t)(((b0 & 0x0F) << 24) | (b1 << 16) | (b2 << 8) | b3); }
        const uint32_t b1 = u8(), b2 = u8(), b3 = u8(), b4 = u8();
        return (int32_t)(((b0 & 0x0F) << 28) | (b1 << 20) | (b2 << 12) | (b3 << 4) | (b4 & 0x0F));
    }
    int64_t ltf8() {
        const uint32_t b0 = u8();
        int extra = 0;
        while (extra < 8 && (b0 & (0x80u >> extra))) ++extra;
        uint64_t v = extra >= 7 ? 0 : (b0 & (0xFFu >> (extra + 1)));
        for (int k = 0; k < extra; ++k) v = (v << 8) | u8();
        return (int64_t)v;
    }
    bool done() const { return at >= n; }
};

// ---- block payloads ----------------------------------------------------------------------------------------------------------------------------
inline std::vector<uint8_t> gunzip(const uint8_t* src, size_t n, size_t raw) {
    std::vector<uint8_t> out(raw);
    z_stream zs{};
    if (inflateInit2(&zs, 15 + 32) != Z_OK) throw Error("CRAM: inflateInit2");
    zs.next_in = const_cast<uint8_t*>(src); zs.avail_in = (uInt)n;
    zs.next_out = out.data(); zs.avail_out = (uInt)raw;
    const int rc = raw ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
    const size_t got = zs.total_out;
    inflateEnd(&zs);
    if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw) throw Error("CRAM: a gzip block does not inflate to its stated size");
    return out;
}

// rANS 4x8 (specification 3.0, section 13; the static-frequency codec of CRAM 3.0): a 9-byte prefix {order, compressed size, raw size}, the
// frequency table(s) with run-length coded symbol lists, four interleaved 32-bit states, 12-bit frequencies, renormalisation byte-wise below 2^23.
struct RansTable {
    uint16_t freq[256], start[256];
    uint8_t sym_of[4096];
};
inline void rans_read_table(Cursor& c, RansTable& t) {
    std::memset(t.freq, 0, sizeof t.freq); std::memset(t.start, 0, sizeof t.start); std::memset(t.sym_of, 0, sizeof t.sym_of);
    uint32_t x = 0, rle = 0;
    uint32_t j = c.u8();
    do {
        uint32_t f = c.u8();
        if (f >= 128) f = ((f & 127) << 8) | c.u8();
        if (x + f > 4096) throw Error("CRAM: rANS frequencies exceed 4096");
        t.freq[j] = (uint16_t)(f & 0xFFFF); t.start[j] = (uint16_t)x;
        std::memset(t.sym_of + x, (int)j, f);
        x += f;
        if (!rle && c.at < c.n && j + 1 == c.p[c.at]) { j = c.u8(); rle = c.u8(); }
        else if (rle) { rle -= 1; j += 1; if (j > 255) throw Error("CRAM: rANS symbol run beyond 255"); }
        else j = c.u8();
    } while (j != 0);
}
inline std::vector<uint8_t> rans4x8_decode(const uint8_t* src, size_t n, size_t expect_raw) {
    Cursor c(src, n);
    const uint8_t order = c.u8();
    const uint32_t comp = (uint32_t)c.i32le(), raw = (uint32_t)c.i32le();
    if ((size_t)comp + 9 > n) throw Error("CRAM: truncated rANS block");
    if (raw != expect_raw) throw Error("CRAM: rANS stream and block header disagree about the uncompressed size");
    std::vector<uint8_t> out(raw);
    if (raw == 0) return out;
    auto renorm = [&](uint32_t& r) { while (r < (1u << 23)) r = (r << 8) | c.u8(); };
    if (order == 0) {
        std::unique_ptr<RansTable> t(new RansTable);
        rans_read_table(c, *t);
        uint32_t R[4];
        for (auto& r : R) r = (uint32_t)c.i32le();
        auto step = [&](uint32_t& r) -> uint8_t {
            const uint32_t m = r & 0xFFF;
            const uint8_t s = t->sym_of[m];
            r = (uint32_t)t->freq[s] * (r >> 12) + m - t->start[s];
            return s;
        };
        const size_t full = raw & ~(size_t)3;
        for (size_t i = 0; i < full; i += 4) {
            for (int k = 0; k < 4; ++k) out[i + k] = step(R[k]);
            for (int k = 0; k < 4; ++k) renorm(R[k]);
        }
        for (size_t i = full, k = 0; i < raw; ++i, ++k) out[i] = t->sym_of[R[k] & 0xFFF];
        return out;
    }
    if (order != 1) throw Error("CRAM: unknown rANS order");
    std::vector<std::unique_ptr<RansTable>> T(256);
    {
        uint32_t rle = 0;
        uint32_t i = c.u8();
        do {
            T[i].reset(new RansTable);
            rans_read_table(c, *T[i]);
            if (!rle && c.at < c.n && i + 1 == c.p[c.at]) { i = c.u8(); rle = c.u8(); }
            else if (rle) { rle -= 1; i += 1; if (i > 255) throw Error("CRAM: rANS context run beyond 255"); }
            else i = c.u8();
        } while (i != 0);
    }
    uint32_t R[4];
    for (auto& r : R) r = (uint32_t)c.i32le();
    const size_t q = raw >> 2;
    size_t idx[4] = {0, q, 2 * q, 3 * q};
    uint8_t last[4] = {0, 0, 0, 0};
    auto step = [&](int k) {
        const RansTable* t = T[last[k]].get();
        if (!t) throw Error("CRAM: rANS order-1 context without a table");
        const uint32_t m = R[k] & 0xFFF;
        const uint8_t s = t->sym_of[m];
        R[k] = (uint32_t)t->freq[s] * (R[k] >> 12) + m - t->start[s];
        out[idx[k]++] = s;
        last[k] = s;
    };
    for (size_t i = 0; i < q; ++i) {
        for (int k = 0; k < 4; ++k) step(k);
        for (int k = 0; k < 4; ++k) renorm(R[k]);
    }
    while (idx[3] < raw) { step(3); renorm(R[3]); }  // the last quarter takes the remainder
    return out;
}

struct Block {
    int method = 0, content_type = 0, content_id = 0;
    std::vector<uint8_t> data;
};
inline Block read_block(Cursor& c) {
    Block b;
    const size_t begin = c.at;
    b.method = c.u8(); b.content_type = c.u8(); b.content_id = c.itf8();
    const int32_t comp = c.itf8(), raw = c.itf8();
    if (comp < 0 || raw < 0) throw Error("CRAM: negative block size");
    if (raw > kMaxBlockBytes) throw Error("CRAM: block larger than 1 GiB");
    const uint8_t* d = c.bytes((size_t)comp);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), c.p + begin, (uInt)(c.at - begin));
    if ((uint32_t)c.i32le() != crc) throw Error("CRAM: block checksum mismatch");  // CRC32 over the block, header included (version 3)
    switch (b.method) {
        case 0: b.data.assign(d, d + comp); break;
        case 1: b.data = gunzip(d, (size_t)comp, (size_t)raw); break;
        case 4: b.data = rans4x8_decode(d, (size_t)comp, (size_t)raw); break;
        case 2: throw Error("CRAM: bzip2-compressed block (not supported by this build)");
        case 3: throw Error("CRAM: lzma-compressed block (not supported by this build)");
        case 5: case 6: case 7: case 8: throw Error("CRAM: block uses a CRAM 3.1 codec (rANS Nx16 / arithmetic / fqzcomp / name tokeniser): write the file as CRAM 3.0");
        default: throw Error("CRAM: unknown block compression method");
    }
    if (b.data.size() != (size_t)raw) throw Error("CRAM: block does not expand to its stated size");
    return b;
}

// ---- encodings ------------------------------------------------------------------------------------------------------------------------------------
struct BitReader {  // the core data block: most significant bit first
    const uint8_t* p = nullptr;
    size_t n = 0, bit = 0;
    uint32_t get(int k) {
        uint32_t v = 0;
        for (int i = 0; i < k; ++i) {
            if ((bit >> 3) >= n) throw Error("CRAM: read past the end of the core block");
            v = (v << 1) | ((p[bit >> 3] >> (7 - (bit & 7))) & 1u);
            bit += 1;
        }
        return v;
    }
};
struct Streams {
    BitReader core;
    std::map<int, Cursor> ext;
    Cursor& external(int id) { auto it = ext.find(id); if (it == ext.end()) throw Error("CRAM: external block " + std::to_string(id) + " is missing from the slice"); return it->second; }
};
struct Encoding {
    int kind = 0;  // 0 NULL, 1 EXTERNAL, 3 HUFFMAN, 4 BYTE_ARRAY_LEN, 5 BYTE_ARRAY_STOP, 6 BETA, 7 SUBEXP, 9 GAMMA
    int id = 0, offset = 0, k = 0;
    uint8_t stop = 0;
    std::vector<int32_t> sym, len;       // HUFFMAN: sorted by (length, symbol) = canonical order
    std::vector<uint32_t> code;
    std::unique_ptr<Encoding> a, b;      // BYTE_ARRAY_LEN: lengths, values

    static Encoding parse(Cursor& c) {
        Encoding e;
        e.kind = c.itf8();
        const int32_t plen = c.itf8();
        if (plen < 0) throw Error("CRAM: negative encoding parameter size");
        Cursor p(c.bytes((size_t)plen), (size_t)plen);
        switch (e.kind) {
            case 0: break;
            case 1: e.id = p.itf8(); break;
            case 3: {
                const int32_t na = p.itf8();
                std::vector<int32_t> al((size_t)std::max(na, 0));
                for (auto& v : al) v = p.itf8();
                const int32_t nl = p.itf8();
                if (nl != na) throw Error("CRAM: Huffman alphabet and code lengths differ in number");
                std::vector<int32_t> ln((size_t)nl);
                for (auto& v : ln) { v = p.itf8(); if (v < 0 || v > 31) throw Error("CRAM: Huffman code length out of range"); }
                std::vector<size_t> o(al.size());
                for (size_t i = 0; i < o.size(); ++i) o[i] = i;
                std::sort(o.begin(), o.end(), [&](size_t x, size_t y) { return ln[x] != ln[y] ? ln[x] < ln[y] : al[x] < al[y]; });
                uint32_t code = 0;
                int prev = o.empty() ? 0 : ln[o[0]];
                for (size_t i : o) {
                    code <<= (ln[i] - prev); prev = ln[i];
                    e.sym.push_back(al[i]); e.len.push_back(ln[i]); e.code.push_back(code);
                    code += 1;
                }
                break;
            }
            case 4: e.a.reset(new Encoding(parse(p))); e.b.reset(new Encoding(parse(p))); break;
            case 5: e.stop = p.u8(); e.id = p.itf8(); break;
            case 6: e.offset = p.itf8(); e.k = p.itf8(); if (e.k < 0 || e.k > 32) throw Error("CRAM: BETA width out of range"); break;
            case 7: e.offset = p.itf8(); e.k = p.itf8(); if (e.k < 0 || e.k > 31) throw Error("CRAM: SUBEXP k out of range"); break;
            case 9: e.offset = p.itf8(); break;
            case 2: case 8: throw Error("CRAM: Golomb / Golomb-Rice encodings are not supported");
            default: throw Error("CRAM: unknown encoding " + std::to_string(e.kind));
        }
        return e;
    }
    int32_t huffman(Streams& s) const {
        if (sym.empty()) throw Error("CRAM: empty Huffman alphabet");
        if (sym.size() == 1 && len[0] == 0) return sym[0];
        uint32_t v = 0;
        int have = 0;
        for (size_t i = 0; i < sym.size(); ++i) {
            if (len[i] > have) { v = (v << (len[i] - have)) | s.core.get(len[i] - have); have = len[i]; }
            if (code[i] == v) return sym[i];
        }
        throw Error("CRAM: bits in the core block match no Huffman code");
    }
    int32_t get_int(Streams& s) const {
        switch (kind) {
            case 1: return s.external(id).itf8();
            case 3: return huffman(s);
            case 6: return (int32_t)s.core.get(k) - offset;
            case 9: { int z = 0; while (s.core.get(1) == 0) { if (++z > 31) throw Error("CRAM: GAMMA run too long"); } return (int32_t)((1u << z) | s.core.get(z)) - offset; }
            case 7: {
                int u = 0;
                while (s.core.get(1) == 1) { if (++u > 31) throw Error("CRAM: SUBEXP run too long"); }
                const int nb = u == 0 ? k : u + k - 1;
                if (nb > 31) throw Error("CRAM: SUBEXP value too wide");
                const uint32_t v = s.core.get(nb);
                return (int32_t)(u == 0 ? v : ((1u << nb) | v)) - offset;
            }
            default: throw Error("CRAM: an integer data series uses encoding " + std::to_string(kind));
        }
    }
    uint8_t get_byte(Streams& s) const {
        switch (kind) {
            case 1: return s.external(id).u8();
            case 3: return (uint8_t)huffman(s);
            case 6: return (uint8_t)((int32_t)s.core.get(k) - offset);
            default: throw Error("CRAM: a byte data series uses encoding " + std::to_string(kind));
        }
    }
    void get_bytes(Streams& s, std::vector<uint8_t>& out) const {  // a whole byte array
        out.clear();
        if (kind == 4) {
            const int32_t n = a->get_int(s);
            if (n < 0 || n > kMaxReadLen) throw Error("CRAM: byte-array length out of range");
            if (b->kind == 1) { const uint8_t* d = s.external(b->id).bytes((size_t)n); out.assign(d, d + n); }
            else for (int32_t i = 0; i < n; ++i) out.push_back(b->get_byte(s));
        } else if (kind == 5) {
            Cursor& c = s.external(id);
            for (;;) { const uint8_t v = c.u8(); if (v == stop) break; out.push_back(v); }
        } else throw Error("CRAM: a byte-array data series uses encoding " + std::to_string(kind));
    }
};

struct CompressionHeader {
    bool names_preserved = true, ap_delta = true, ref_required = true;
    uint8_t sub[5][4];                                      // substitution matrix: [reference base ACGTN][code] -> read base
    std::vector<std::vector<std::array<uint8_t, 3>>> tag_lines;  // TD
    std::map<uint16_t, Encoding> series;                    // key = two letters
    std::map<int32_t, Encoding> tags;                       // key = tag << 8 | type
    const Encoding* get(const char* k) const { auto it = series.find((uint16_t)(k[0] << 8 | k[1])); return it == series.end() ? nullptr : &it->second; }
    const Encoding& need(const char* k) const { const Encoding* e = get(k); if (!e || e->kind == 0) throw Error(std::string("CRAM: data series ") + k + " is used but has no encoding"); return *e; }
};
inline CompressionHeader parse_compression_header(const std::vector<uint8_t>& d) {
    CompressionHeader h;
    static const char bases[] = "ACGTN";
    for (int r = 0; r < 5; ++r) { int k = 0; for (int b = 0; b < 5; ++b) if (b != r) h.sub[r][k++] = (uint8_t)bases[b]; }
    Cursor c(d.data(), d.size());
    {   // preservation map
        const int32_t size = c.itf8();
        Cursor m(c.bytes((size_t)std::max(size, 0)), (size_t)std::max(size, 0));
        const int32_t n = m.itf8();
        for (int32_t i = 0; i < n; ++i) {
            const uint8_t k0 = m.u8(), k1 = m.u8();
            if (k0 == 'R' && k1 == 'N') h.names_preserved = m.u8() != 0;
            else if (k0 == 'A' && k1 == 'P') h.ap_delta = m.u8() != 0;
            else if (k0 == 'R' && k1 == 'R') h.ref_required = m.u8() != 0;
            else if (k0 == 'S' && k1 == 'M') {
                for (int r = 0; r < 5; ++r) {  // the byte of reference base r: four 2-bit codes, one for each other base in ACGTN order
                    const uint8_t v = m.u8();
                    int k = 0;
                    for (int b = 0; b < 5; ++b) if (b != r) { h.sub[r][(v >> (6 - 2 * k)) & 3] = (uint8_t)bases[b]; ++k; }
                }
            } else if (k0 == 'T' && k1 == 'D') {
                const int32_t len = m.itf8();
                const uint8_t* t = m.bytes((size_t)std::max(len, 0));
                std::vector<std::array<uint8_t, 3>> line;
                for (int32_t o = 0; o < len;) {
                    if (t[o] == 0) { h.tag_lines.push_back(line); line.clear(); o += 1; continue; }
                    if (o + 3 > len) throw Error("CRAM: truncated tag dictionary");
                    line.push_back({t[o], t[o + 1], t[o + 2]});
                    o += 3;
                }
                if (!line.empty()) h.tag_lines.push_back(line);
            } else throw Error("CRAM: unknown preservation-map key");
        }
    }
    {   // data series encodings
        const int32_t size = c.itf8();
        Cursor m(c.bytes((size_t)std::max(size, 0)), (size_t)std::max(size, 0));
        const int32_t n = m.itf8();
        for (int32_t i = 0; i < n; ++i) { const uint8_t k0 = m.u8(), k1 = m.u8(); h.series[(uint16_t)(k0 << 8 | k1)] = Encoding::parse(m); }
    }
    {   // tag encodings
        const int32_t size = c.itf8();
        Cursor m(c.bytes((size_t)std::max(size, 0)), (size_t)std::max(size, 0));
        const int32_t n = m.itf8();
        for (int32_t i = 0; i < n; ++i) { const int32_t key = m.itf8(); h.tags[key] = Encoding::parse(m); }
    }
    return h;
}

// one decoded record, in the terms of InRecord (bam_io.hpp)
struct Rec {
    std::string name;
    bool has_name = false;
    uint16_t flags = 0;
    std::string seq;
    std::vector<uint8_t> qual, aux;
    std::string error;  // non-empty: the record cannot be delivered (reported and skipped by the caller)
};

class Reader {
public:
    // `read` fills a buffer with exactly n bytes and returns false at a clean end of input; the four magic bytes have been consumed by the caller
    template <class ReadExact>
    Reader(ReadExact read, bool magic_consumed) : read_(read) {
        uint8_t def[26];
        const size_t skip = magic_consumed ? 4 : 0;
        if (!read_(def + skip, 26 - skip)) throw Error("CRAM: truncated file definition");
        major_ = def[4];
        if (major_ != 3) throw Error("CRAM: version " + std::to_string((int)def[4]) + "." + std::to_string((int)def[5]) + " files are not supported (3.0 is)");
        // the first container holds the SAM header
        std::vector<uint8_t> body;
        ContainerHeader ch;
        if (!next_container(ch, body)) throw Error("CRAM: no header container");
        Cursor c(body.data(), body.size());
        const Block b = read_block(c);
        if (b.content_type != 0) throw Error("CRAM: the first container does not hold the file header");
        Cursor t(b.data.data(), b.data.size());
        const int32_t len = t.i32le();
        if (len < 0 || (size_t)len > b.data.size() - 4) throw Error("CRAM: header text longer than its block");
        header_.assign((const char*)t.bytes((size_t)len), (size_t)len);
        while (!header_.empty() && header_.back() == '\0') header_.pop_back();
        for (size_t o = 0; o < header_.size();) {  // read-group ids in header order: the RG data series is an index into them
            size_t e = header_.find('\n', o);
            if (e == std::string::npos) e = header_.size();
            if (header_.compare(o, 4, "@RG\t") == 0) {
                std::string id;
                for (size_t f = o + 4; f < e;) {
                    size_t g = header_.find('\t', f);
                    if (g == std::string::npos || g > e) g = e;
                    if (header_.compare(f, 3, "ID:") == 0) id = header_.substr(f + 3, g - f - 3);
                    f = g + 1;
                }
                read_groups_.push_back(id);
            }
            o = e + 1;
        }
    }
    const std::string& header_text() const { return header_; }

    // next record; false at the end of the file
    bool next(Rec& r) {
        while (at_ >= recs_.size()) {
            if (!load_container()) return false;
        }
        r = std::move(recs_[at_++]);
        return true;
    }

private:
    struct ContainerHeader { int32_t length = 0, ref_id = 0, start = 0, span = 0, n_records = 0, n_blocks = 0; };
    std::function<bool(uint8_t*, size_t)> read_;
    int major_ = 3;
    std::string header_;
    std::vector<std::string> read_groups_;
    std::vector<Rec> recs_;
    size_t at_ = 0;

    bool next_container(ContainerHeader& h, std::vector<uint8_t>& body) {
        uint8_t l4[4];
        if (!read_(l4, 4)) return false;
        h.length = (int32_t)((uint32_t)l4[0] | (uint32_t)l4[1] << 8 | (uint32_t)l4[2] << 16 | (uint32_t)l4[3] << 24);
        if (h.length < 0 || h.length > kMaxBlockBytes) throw Error("CRAM: container length out of range");
        // the rest of the header is a sequence of variable-length integers: read them byte by byte
        std::vector<uint8_t> hdr(l4, l4 + 4);  // the header's bytes, for its checksum
        auto byte = [&]() -> uint8_t { uint8_t b; if (!read_(&b, 1)) throw Error("CRAM: truncated container header"); hdr.push_back(b); return b; };
        auto itf8 = [&]() -> int32_t {
            uint8_t buf[5]; buf[0] = byte();
            const int extra = buf[0] < 0x80 ? 0 : buf[0] < 0xC0 ? 1 : buf[0] < 0xE0 ? 2 : buf[0] < 0xF0 ? 3 : 4;
            for (int k = 0; k < extra; ++k) buf[1 + k] = byte();
            Cursor c(buf, 5); return c.itf8();
        };
        auto ltf8 = [&]() -> int64_t {
            uint8_t buf[9]; buf[0] = byte();
            int extra = 0; while (extra < 8 && (buf[0] & (0x80u >> extra))) ++extra;
            for (int k = 0; k < extra; ++k) buf[1 + k] = byte();
            Cursor c(buf, 9); return c.ltf8();
        };
        h.ref_id = itf8(); h.start = itf8(); h.span = itf8(); h.n_records = itf8();
        (void)ltf8(); (void)ltf8();  // record counter, bases
        h.n_blocks = itf8();
        const int32_t n_land = itf8();
        if (n_land < 0) throw Error("CRAM: negative landmark count");
        for (int32_t i = 0; i < n_land; ++i) (void)itf8();
        uint8_t crc[4];
        if (!read_(crc, 4)) throw Error("CRAM: truncated container header");
        const uint32_t want = (uint32_t)crc[0] | (uint32_t)crc[1] << 8 | (uint32_t)crc[2] << 16 | (uint32_t)crc[3] << 24;
        if (want != (uint32_t)crc32(crc32(0, nullptr, 0), hdr.data(), (uInt)hdr.size())) throw Error("CRAM: container header checksum mismatch");
        body.resize((size_t)h.length);
        if (h.length && !read_(body.data(), body.size())) throw Error("CRAM: truncated container");
        return true;
    }

    bool load_container() {
        recs_.clear(); at_ = 0;
        ContainerHeader ch;
        std::vector<uint8_t> body;
        for (;;) {
            if (!next_container(ch, body)) return false;
            if (ch.n_records > 0) break;  // the end-of-file container and empty containers hold no records
        }
        Cursor c(body.data(), body.size());
        const Block hb = read_block(c);
        if (hb.content_type != 1) throw Error("CRAM: a data container does not start with a compression header");
        const CompressionHeader H = parse_compression_header(hb.data);
        while (!c.done()) {
            const Block sh = read_block(c);
            if (sh.content_type != 2) throw Error("CRAM: expected a slice header");
            decode_slice(H, sh, c);
        }
        return true;
    }

    void decode_slice(const CompressionHeader& H, const Block& sh, Cursor& c) {
        Cursor s(sh.data.data(), sh.data.size());
        const int32_t ref_id = s.itf8(), start = s.itf8();
        (void)s.itf8();  // alignment span
        const int32_t n_records = s.itf8();
        (void)s.ltf8();
        const int32_t n_blocks = s.itf8(), n_ids = s.itf8();
        for (int32_t i = 0; i < n_ids; ++i) (void)s.itf8();
        const int32_t embedded = s.itf8();
        // (reference MD5 and optional tags follow)
        if (n_records < 0 || n_records > kMaxSliceRecords || n_blocks < 0) throw Error("CRAM: slice header out of range");
        std::vector<Block> blocks;
        for (int32_t i = 0; i < n_blocks; ++i) blocks.push_back(read_block(c));
        Streams st;
        const std::vector<uint8_t>* emb = nullptr;
        bool have_core = false;
        for (const Block& b : blocks) {
            if (b.content_type == 5) { st.core.p = b.data.data(); st.core.n = b.data.size(); have_core = true; }
            else if (b.content_type == 4) {
                st.ext[b.content_id] = Cursor(b.data.data(), b.data.size());
                if (embedded >= 0 && b.content_id == embedded) emb = &b.data;
            }
        }
        if (!have_core) throw Error("CRAM: slice without a core data block");
        int32_t prev_pos = start;
        auto ref_base = [&](int64_t pos1) -> int {  // 1-based reference position -> base, -1 if no reference is at hand
            if (!emb || ref_id < 0) return -1;
            const int64_t o = pos1 - start;
            if (o < 0 || o >= (int64_t)emb->size()) return -1;
            return (*emb)[(size_t)o];
        };
        std::vector<uint8_t> tmp;
        // A data series with a zero-bit HUFFMAN encoding consumes no input, so the sizes a slice states do not bound what its records decode to: the
        // total of decoded bases per slice is capped (a real slice holds ~10 K records; 2^28 bases is three orders of magnitude above that).
        int64_t slice_bases = 0;
        for (int32_t i = 0; i < n_records; ++i) {
            Rec r;
            const int32_t bf = H.need("BF").get_int(st), cf = H.need("CF").get_int(st);
            r.flags = (uint16_t)bf;
            if (ref_id == -2) (void)H.need("RI").get_int(st);
            const int32_t rl = H.need("RL").get_int(st);
            if (rl < 0 || rl > kMaxReadLen) throw Error("CRAM: read length out of range");
            if ((slice_bases += rl) > (int64_t)1 << 28) throw Error("CRAM: a slice decodes to more than 2^28 bases");
            int32_t ap = H.need("AP").get_int(st);
            if (H.ap_delta) { ap += prev_pos; prev_pos = ap; }
            const int32_t rg = H.need("RG").get_int(st);
            auto take_name = [&]() { H.need("RN").get_bytes(st, tmp); r.name.assign(tmp.begin(), tmp.end()); r.has_name = !(r.name.empty() || r.name == "*"); };
            if (H.names_preserved) take_name();
            if (cf & 0x2) {  // detached: the mate's fields are in this record
                (void)H.need("MF").get_int(st);
                if (!H.names_preserved) take_name();
                (void)H.need("NS").get_int(st); (void)H.need("NP").get_int(st); (void)H.need("TS").get_int(st);
            } else if (cf & 0x4) (void)H.need("NF").get_int(st);
            const int32_t tl = H.need("TL").get_int(st);
            if (tl < 0 || (size_t)tl >= H.tag_lines.size()) throw Error("CRAM: tag line index outside the dictionary");
            for (const auto& t : H.tag_lines[(size_t)tl]) {
                const int32_t key = (int32_t)t[0] << 16 | (int32_t)t[1] << 8 | t[2];
                auto it = H.tags.find(key);
                if (it == H.tags.end()) throw Error("CRAM: tag without an encoding");
                it->second.get_bytes(st, tmp);
                r.aux.push_back(t[0]); r.aux.push_back(t[1]); r.aux.push_back(t[2]);
                r.aux.insert(r.aux.end(), tmp.begin(), tmp.end());
            }
            if (rg >= 0 && (size_t)rg < read_groups_.size()) {  // the read group travels as an index; records carry it as RG:Z
                const std::string& id = read_groups_[(size_t)rg];
                r.aux.push_back('R'); r.aux.push_back('G'); r.aux.push_back('Z');
                r.aux.insert(r.aux.end(), id.begin(), id.end()); r.aux.push_back(0);
            }
            const bool unknown_seq = (cf & 0x8) != 0;
            r.seq.assign(unknown_seq ? 0 : (size_t)rl, 'N');
            r.qual.assign((size_t)rl, 0xFF);
            if (!(bf & 0x4)) {  // mapped: the bases are the reference plus the read's features
                const int32_t fn = H.need("FN").get_int(st);
                if (fn < 0 || fn > 4 * rl + 16) throw Error("CRAM: more read features than the read can hold");
                std::vector<uint8_t> known((size_t)rl, 0);
                int64_t ref_pos = ap;
                int32_t read_pos = 1, prev = 0;
                bool need_ref = false;
                auto put = [&](int32_t p1, uint8_t base) { if (p1 >= 1 && p1 <= rl) { if (!unknown_seq) r.seq[(size_t)p1 - 1] = (char)base; known[(size_t)p1 - 1] = 1; } };
                auto fill_to = [&](int32_t upto) {  // read positions [read_pos, upto) match the reference
                    for (; read_pos < upto && read_pos <= rl; ++read_pos, ++ref_pos) {
                        const int b = ref_base(ref_pos);
                        if (b < 0) need_ref = true; else put(read_pos, (uint8_t)b);
                    }
                };
                for (int32_t k = 0; k < fn; ++k) {
                    const uint8_t code = H.need("FC").get_byte(st);
                    const int32_t fp = H.need("FP").get_int(st);
                    if (fp < 0 || fp > rl + 1 || prev > rl + 1) throw Error("CRAM: read feature position outside the read");
                    const int32_t pos = prev + fp;
                    if (pos < 1 || pos > rl + 1) throw Error("CRAM: read feature position outside the read");
                    prev = pos;
                    fill_to(pos);
                    switch (code) {
                        case 'b': { H.need("BB").get_bytes(st, tmp); for (size_t x = 0; x < tmp.size(); ++x) put(pos + (int32_t)x, tmp[x]); read_pos = pos + (int32_t)tmp.size(); ref_pos += (int64_t)tmp.size(); break; }
                        case 'q': { H.need("QQ").get_bytes(st, tmp); for (size_t x = 0; x < tmp.size(); ++x) if (pos + (int32_t)x >= 1 && pos + (int32_t)x <= rl) r.qual[(size_t)pos - 1 + x] = tmp[x]; break; }
                        case 'B': { const uint8_t b = H.need("BA").get_byte(st), q = H.need("QS").get_byte(st); put(pos, b); if (pos >= 1 && pos <= rl) r.qual[(size_t)pos - 1] = q; read_pos = pos + 1; ref_pos += 1; break; }
                        case 'X': {
                            const uint8_t sc = H.need("BS").get_byte(st);
                            const int b = ref_base(ref_pos);
                            if (b < 0) need_ref = true;
                            else { const char* q = std::strchr("ACGTN", std::toupper(b)); put(pos, H.sub[q ? (int)(q - "ACGTN") : 4][sc & 3]); }
                            read_pos = pos + 1; ref_pos += 1; break;
                        }
                        case 'I': { H.need("IN").get_bytes(st, tmp); for (size_t x = 0; x < tmp.size(); ++x) put(pos + (int32_t)x, tmp[x]); read_pos = pos + (int32_t)tmp.size(); break; }
                        case 'S': { H.need("SC").get_bytes(st, tmp); for (size_t x = 0; x < tmp.size(); ++x) put(pos + (int32_t)x, tmp[x]); read_pos = pos + (int32_t)tmp.size(); break; }
                        case 'i': { put(pos, H.need("BA").get_byte(st)); read_pos = pos + 1; break; }
                        case 'D': ref_pos += H.need("DL").get_int(st); break;
                        case 'N': ref_pos += H.need("RS").get_int(st); break;
                        case 'H': (void)H.need("HC").get_int(st); break;
                        case 'P': (void)H.need("PD").get_int(st); break;
                        case 'Q': { const uint8_t q = H.need("QS").get_byte(st); if (pos >= 1 && pos <= rl) r.qual[(size_t)pos - 1] = q; break; }
                        default: throw Error("CRAM: unknown read feature");
                    }
                }
                fill_to(rl + 1);
                (void)H.need("MQ").get_int(st);
                if (cf & 0x1) for (int32_t x = 0; x < rl; ++x) r.qual[(size_t)x] = H.need("QS").get_byte(st);
                if (need_ref && !unknown_seq) r.error = "its bases are stored as differences to a reference sequence that is not embedded in the file";
            } else {
                if (!unknown_seq) for (int32_t x = 0; x < rl; ++x) r.seq[(size_t)x] = (char)H.need("BA").get_byte(st);
                if (cf & 0x1) for (int32_t x = 0; x < rl; ++x) r.qual[(size_t)x] = H.need("QS").get_byte(st);
            }
            recs_.push_back(std::move(r));
        }
    }
};

}  // namespace cram
}  // namespace cli
}  // namespace mapad
