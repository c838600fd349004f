// host_index.hpp — host representation of a mapAD index and its construction (`mapad index` equivalent).
//
// Follows src/index/indexing.rs:43-212 (FASTA -> uppercase -> IUPAC replacement -> text$revcomp$ -> rank transform ->
// suffix array -> BWT -> SA sample 1/32 + extra rows -> Less) and src/index/mod.rs (SampledSuffixArray, FastaIdPositions,
// OriginalSymbols).  Instead of rust-bio's Occ (u64 checkpoints every 128 rows, byte BWT) the rank structure is the
// 64-byte-block layout of fmd_device.hpp, shared by host (SA walks) and device.
#pragma once
#include "host_cpus.hpp"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fmd_device.hpp"

namespace mapad {
namespace host {

struct Contig { uint64_t start, end; std::string name; };  // FastaIdPosition (src/index/mod.rs:30-35), end inclusive

struct Index {
    uint64_t n = 0;
    std::vector<uint8_t> bwt;        // reference ranks $=0 A=1 C=2 G=3 T=4 X=5
    uint64_t less[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t sentinel[2] = {0, 0};
    std::vector<uint64_t> blocks;    // device layout (fmd_device.hpp): kBlockWords u64 per kBlockRows rows
    std::vector<uint64_t> x_counts;  // per block b: X symbols in rows [0, kBlockRows b) (only needed for LF steps through 'X')
    std::vector<uint64_t> sa_sample;
    uint64_t sa_rate = 32;
    std::map<uint64_t, uint64_t> extra_rows;
    std::vector<Contig> contigs;
    std::map<uint64_t, uint8_t> original_symbols;

    DevIndex view() const {
        DevIndex v;
        v.blocks = blocks.data(); v.n = n; v.n_blocks = blocks.size() / kBlockWords;
        for (int i = 0; i < 8; ++i) v.less[i] = less[i];
        v.sentinel[0] = sentinel[0]; v.sentinel[1] = sentinel[1];
        return v;
    }
    // occurrences of reference rank `a` in bwt[0..=r]
    uint64_t occ_rank(uint64_t r, int a) const {
        if (a >= 1 && a <= 4) return occ_scalar(view(), r, a - 1);
        if (a == 0) return (uint64_t)(r >= sentinel[0]) + (uint64_t)(r >= sentinel[1]);
        return occ_x_scalar(view(), x_counts.empty() ? nullptr : x_counts.data(), r);  // 'X'
    }
    // SampledSuffixArray::get (src/index/mod.rs:160-187)
    bool sa_get(uint64_t index, uint64_t& out) const {
        if (index >= n) return false;
        uint64_t pos = index, offset = 0;
        for (;;) {
            if (pos % sa_rate == 0) { out = sa_sample[pos / sa_rate] + offset; return true; }
            const uint8_t c = bwt[pos];
            if (c == 0) { out = extra_rows.at(pos) + offset; return true; }
            pos = less[c] + occ_rank(pos - 1, c);
            offset += 1;
        }
    }
    // FastaIdPositions::get_reference_identifier (src/index/mod.rs:55-75)
    bool contig_of(uint64_t position, uint64_t pattern_length, uint32_t& tid, uint64_t& rel) const {
        for (size_t i = 0; i < contigs.size(); ++i)
            if (contigs[i].start <= position && position + pattern_length - 1 <= contigs[i].end) { tid = (uint32_t)i; rel = position - contigs[i].start; return true; }
        return false;
    }
    bool original_symbol(uint64_t pos, uint8_t& out) const {
        auto it = original_symbols.find(pos);
        if (it == original_symbols.end()) return false;
        out = it->second;
        return true;
    }
};

inline uint8_t complement(uint8_t a) {  // bio::alphabets::dna::complement
    switch (a) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k';
        case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
        default: return a;
    }
}

// ---- suffix array by induced sorting (SA-IS, Nong/Zhang/Chan 2009), implicit sentinel smaller than every symbol ----------
// Any correct construction reproduces rust-bio's suffix_array(): the order of the suffixes of text$ is unique.
template <class S, class I>
void sais(const S* s, I* SA, I n, I K);

namespace detail {
struct TypeBits {
    std::vector<uint64_t> b;
    explicit TypeBits(size_t n) : b((n + 64) / 64, 0) {}
    bool get(size_t i) const { return (b[i >> 6] >> (i & 63)) & 1; }
    void set(size_t i) { b[i >> 6] |= 1ull << (i & 63); }
};
template <class S, class I>
void bucket_bounds(const S* s, I n, I K, std::vector<I>& B, bool ends) {
    std::fill(B.begin(), B.end(), (I)0);
    for (I i = 0; i < n; ++i) B[s[i]] += 1;
    I sum = 0;
    for (I c = 0; c < K; ++c) { const I cnt = B[c]; if (ends) { sum += cnt; B[c] = sum; } else { B[c] = sum; sum += cnt; } }
}
template <class S, class I>
void induce(const S* s, I* SA, I n, I K, const TypeBits& t, std::vector<I>& B) {
    bucket_bounds(s, n, K, B, false);
    SA[B[s[n - 1]]++] = n - 1;  // predecessor of the implicit sentinel is L-type
    for (I i = 0; i < n; ++i) {
        const I j = SA[i];
        if (j > 0 && !t.get((size_t)j - 1)) SA[B[s[j - 1]]++] = j - 1;
    }
    bucket_bounds(s, n, K, B, true);
    for (I i = n; i-- > 0;) {
        const I j = SA[i];
        if (j > 0 && t.get((size_t)j - 1)) SA[--B[s[j - 1]]] = j - 1;
    }
}
}  // namespace detail

template <class S, class I>
void sais(const S* s, I* SA, I n, I K) {
    using namespace detail;
    if (n == 0) return;
    if (n == 1) { SA[0] = 0; return; }
    TypeBits t((size_t)n);  // bit = S-type
    for (I i = n - 1; i-- > 0;) if (s[i] < s[i + 1] || (s[i] == s[i + 1] && t.get((size_t)i + 1))) t.set((size_t)i);
    auto is_lms = [&](I i) { return i > 0 && t.get((size_t)i) && !t.get((size_t)i - 1); };
    std::vector<I> B((size_t)K);
    const I EMPTY = (I)-1;
    // stage 1: sort the LMS substrings
    std::fill(SA, SA + n, EMPTY);
    bucket_bounds(s, n, K, B, true);
    for (I i = n - 1; i > 0; --i) if (is_lms(i)) SA[--B[s[i]]] = i;
    {   // induce with EMPTY-aware scans
        bucket_bounds(s, n, K, B, false);
        SA[B[s[n - 1]]++] = n - 1;
        for (I i = 0; i < n; ++i) { const I j = SA[i]; if (j != EMPTY && j > 0 && !t.get((size_t)j - 1)) SA[B[s[j - 1]]++] = j - 1; }
        bucket_bounds(s, n, K, B, true);
        for (I i = n; i-- > 0;) { const I j = SA[i]; if (j != EMPTY && j > 0 && t.get((size_t)j - 1)) SA[--B[s[j - 1]]] = j - 1; }
    }
    I n1 = 0;
    for (I i = 0; i < n; ++i) if (SA[i] != EMPTY && is_lms(SA[i])) SA[n1++] = SA[i];
    std::fill(SA + n1, SA + n, EMPTY);
    I name = 0, prev = EMPTY;
    for (I i = 0; i < n1; ++i) {
        const I pos = SA[i];
        bool diff = prev == EMPTY;
        if (!diff) {
            for (I d = 0;; ++d) {
                if (pos + d >= n || prev + d >= n) { diff = true; break; }
                if (s[pos + d] != s[prev + d] || t.get((size_t)(pos + d)) != t.get((size_t)(prev + d))) { diff = true; break; }
                if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
            }
        }
        if (diff) { name += 1; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    {
        I j = n - 1;
        for (I i = n - 1; i >= n1; --i) { if (SA[i] != EMPTY) SA[j--] = SA[i]; if (i == n1) break; }
    }
    I* s1 = SA + (n - n1);
    I* SA1 = SA;
    if (name < n1) sais<I, I>(s1, SA1, n1, name);
    else for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
    // stage 2: sorted LMS suffixes -> full SA
    {
        I j = 0;
        for (I i = 1; i < n; ++i) if (is_lms(i)) s1[j++] = i;
    }
    for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
    std::fill(SA + n1, SA + n, EMPTY);
    bucket_bounds(s, n, K, B, true);
    for (I i = n1; i-- > 0;) {
        const I j = SA[i];
        SA[i] = EMPTY;
        SA[--B[s[j]]] = j;
    }
    {
        bucket_bounds(s, n, K, B, false);
        SA[B[s[n - 1]]++] = n - 1;
        for (I i = 0; i < n; ++i) { const I j = SA[i]; if (j != EMPTY && j > 0 && !t.get((size_t)j - 1)) SA[B[s[j - 1]]++] = j - 1; }
        bucket_bounds(s, n, K, B, true);
        for (I i = n; i-- > 0;) { const I j = SA[i]; if (j != EMPTY && j > 0 && t.get((size_t)j - 1)) SA[--B[s[j - 1]]] = j - 1; }
    }
}

// ---- device block layout from a rank BWT -----------------------------------------------------------------------------------
inline void build_blocks(Index& ix) {
    const uint64_t n = ix.n;
    const uint64_t n_blocks = (n + kBlockRows - 1) / kBlockRows + 1;  // one spare block keeps hi-row prefetches in bounds
    ix.blocks.assign(n_blocks * kBlockWords, 0);
    ix.x_counts.clear();
    static const int CODE[6] = {0, 4, 5, 6, 7, 1};  // rank -> device symbol code
    // pass 1 (threads over block ranges): bit planes + the block's own symbol counts, parked in the count words
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(cpu_share(), 32u), n_blocks / 4096 + 1));
    std::vector<uint64_t> xs(n_blocks, 0);
    std::vector<std::vector<uint64_t>> sent(T);
    std::vector<int> bad(T, 0);
    auto work = [&](unsigned t) {
        const uint64_t b0 = n_blocks * t / T, b1 = n_blocks * (t + 1) / T;
        for (uint64_t b = b0; b < b1; ++b) {
            uint64_t* blk = ix.blocks.data() + b * kBlockWords;
            uint64_t cnt[4] = {0, 0, 0, 0}, xcnt = 0;
            uint32_t pl[4][3] = {};
            const uint64_t r0 = b * kBlockRows;
            for (uint64_t r = r0; r < std::min(n, r0 + kBlockRows); ++r) {
                const uint8_t a = ix.bwt[r];
                if (a > 5) { bad[t] = 1; continue; }
                const int code = CODE[a];
                const int w = (int)(r - r0) / kSubRows, bit = (int)(r - r0) % kSubRows;
                for (int k = 0; k < 3; ++k) if (code & (1 << k)) pl[w][k] |= 1u << bit;
                if (a >= 1 && a <= 4) cnt[a - 1] += 1;
                else if (a == 5) xcnt += 1;
                else if (sent[t].size() < 4) sent[t].push_back(r);
            }
            for (int w = 0; w < 4; ++w) { blk[2 * w] = pack_sub0(cnt[w], pl[w][0]); blk[2 * w + 1] = pack_sub1(pl[w][1], pl[w][2]); }
            xs[b] = xcnt;
        }
    };
    if (T == 1) work(0);
    else { std::vector<std::thread> pool; for (unsigned t = 0; t < T; ++t) pool.emplace_back(work, t); for (auto& th : pool) th.join(); }
    for (unsigned t = 0; t < T; ++t) if (bad[t]) throw std::runtime_error("BWT symbol out of range");
    // pass 2: exclusive prefix sums over the blocks
    uint64_t cnt[4] = {0, 0, 0, 0}, xcnt = 0;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        uint64_t* blk = ix.blocks.data() + b * kBlockWords;
        for (int w = 0; w < 4; ++w) { const uint64_t c = blk[2 * w] & kCountMask; blk[2 * w] = pack_sub0(cnt[w], sub_p0(blk[2 * w])); cnt[w] += c; }
        const uint64_t x = xs[b]; xs[b] = xcnt; xcnt += x;
    }
    std::vector<uint64_t> all_sent;
    for (auto& v : sent) all_sent.insert(all_sent.end(), v.begin(), v.end());
    if (all_sent.size() != 2) throw std::runtime_error("BWT must contain exactly two sentinels");
    ix.sentinel[0] = std::min(all_sent[0], all_sent[1]); ix.sentinel[1] = std::max(all_sent[0], all_sent[1]);
    if (xcnt) ix.x_counts = std::move(xs);
    // Less: less[c] = number of symbols < c
    uint64_t per[6] = {2, cnt[0], cnt[1], cnt[2], cnt[3], xcnt};
    uint64_t acc = 0;
    for (int c = 0; c < 6; ++c) { ix.less[c] = acc; acc += per[c]; }
    ix.less[6] = acc; ix.less[7] = acc;
}

// run_apply (src/index/indexing.rs:215-256): ambiguous runs shorter than min_run_len -> non_run_fun (original kept),
// longer runs -> run_fun.
template <class F, class G>
void run_apply(std::vector<uint8_t>& seq, size_t min_run_len, F non_run_fun, G run_fun, std::map<uint64_t, uint8_t>& original) {
    size_t i = 0;
    while (i < seq.size()) {
        const uint8_t sym = seq[i];
        size_t run = 1;
        while (i + run < seq.size() && seq[i + run] == sym) ++run;
        if (base_index(sym) > 3) {
            if (run < min_run_len) for (size_t j = 0; j < run; ++j) { original[i + j] = seq[i + j]; seq[i + j] = non_run_fun(seq[i + j]); }
            else for (size_t j = 0; j < run; ++j) seq[i + j] = run_fun(seq[i + j]);
        }
        i += run;
    }
}

// The random source of `mapad index`: rand 0.9 `StdRng::seed_from_u64(seed)` + `slice.choose(rng)` (src/index/indexing.rs:30,78-92), restated from
// the published algorithms (the crates are not vendored): rand_core 0.9 seed_from_u64 expands the u64 into the 32-byte key with PCG32 (XSH-RR output of an
// LCG advanced first); StdRng is ChaCha12 with a 64-bit block counter from 0 and stream 0, whose output words are consumed in order; choose() draws
// an index by `random_range(0..len as u32)`, which in rand 0.9 is Canon's method: the high half of (next_u32 x len), corrected by one more draw when the
// low half falls within `len` of 2^32.  Checked against the reference where it can be: the ChaCha core against the RFC 7539 key stream, and the one draw the
// reference's integration test pins — StdRng(1234) must turn the N of Chromosome_02 into 'A' (tests/integration_tests.rs, MD 4C5N11 with MAPQ 37).
struct StdRngCompat {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    explicit StdRngCompat(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 8; ++i) {  // rand_core::SeedableRng::seed_from_u64
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    void refill() {  // one ChaCha12 block
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                          (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = s[i];
        auto qr = [&](int a, int b, int c, int d) {
            w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
            w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
        };
        for (int r = 0; r < 6; ++r) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) buf[i] = w[i] + s[i];
        counter += 1;
        idx = 0;
    }
    uint32_t next_u32() { if (idx == 16) refill(); return buf[idx++]; }
    uint32_t index_below(uint32_t n) {  // rand 0.9 UniformInt<u32>::sample_single_inclusive(0, n - 1)
        const uint64_t m = (uint64_t)next_u32() * n;
        uint32_t result = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - n)) {
            const uint32_t new_hi = (uint32_t)(((uint64_t)next_u32() * n) >> 32);
            if ((uint64_t)lo + new_hi > 0xFFFFFFFFull) result += 1;
        }
        return result;
    }
};

inline bool is_iupac(uint8_t c) {
    switch (c) { case 'A': case 'C': case 'G': case 'T': case 'U': case 'R': case 'Y': case 'K': case 'M': case 'S': case 'W': case 'B': case 'D': case 'H': case 'V': case 'N': return true; default: return false; }
}

// indexing.rs:43-160: contigs -> uppercase -> IUPAC check -> replacement of ambiguity codes -> text $ revcomp $ as ranks ($=0 A=1 C=2 G=3 T=4 X=5).
// Fills ix.contigs, ix.original_symbols and ix.n.  `fixed_replacement` != 0 forces every short-run replacement to that base (a test hook for texts
// whose tests want a known base; the reference's own stream is reproduced by StdRngCompat).
inline std::vector<uint8_t> prepare_text(const std::vector<std::string>& names, const uint8_t* const* seqs, const uint64_t* lens, uint64_t seed, uint8_t fixed_replacement, Index& ix) {
    std::vector<uint8_t> text;
    uint64_t end = 0;
    for (size_t c = 0; c < names.size(); ++c) end += lens[c];
    text.resize(end);
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(cpu_share(), 32u), text.size() / (1u << 22) + 1));
    auto parallel = [&](const std::function<void(size_t, size_t, unsigned)>& fn, size_t total) {
        if (T == 1) { fn(0, total, 0); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t) pool.emplace_back([&, t] { fn(total * t / T, total * (t + 1) / T, t); });
        for (auto& th : pool) th.join();
    };
    end = 0;
    for (size_t c = 0; c < names.size(); ++c) {  // :115-137
        uint8_t* dst = text.data() + end;
        const uint8_t* src = seqs[c];
        auto upper = [&](size_t lo, size_t hi, unsigned) {
            for (size_t i = lo; i < hi; ++i) { uint8_t ch = src[i]; if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 'a' + 'A'); dst[i] = ch; }
        };
        if (lens[c] >= (1u << 22)) parallel(upper, lens[c]); else upper(0, lens[c], 0);  // no thread launches for scaffold-sized contigs
        end += lens[c];
        ix.contigs.push_back({end - lens[c], end - 1, names[c]});
    }
    std::vector<int> flag(T, 0);  // 1 ambiguous, 2 non-IUPAC
    parallel([&](size_t lo, size_t hi, unsigned t) {
        int fl = 0;
        for (size_t i = lo; i < hi; ++i) { const uint8_t ch = text[i]; if (base_index(ch) <= 3) continue; fl |= is_iupac(ch) ? 1 : 2; }
        flag[t] = fl;
    }, text.size());
    bool ambiguous = false;
    for (int fl : flag) { if (fl & 2) throw std::runtime_error("Found non-IUPAC symbol in reference sequence"); ambiguous |= (fl & 1) != 0; }  // :71
    if (ambiguous) {
        StdRngCompat rng(seed);
        auto pick = [&](const char* set) -> uint8_t { const size_t k = std::char_traits<char>::length(set); return (uint8_t)set[rng.index_below((uint32_t)k)]; };
        auto replace = [&](uint8_t b) -> uint8_t {  // :78-92
            if (b == 'U') return 'T';
            if (fixed_replacement) return fixed_replacement;
            switch (b) {
                case 'R': return pick("AG"); case 'Y': return pick("CT"); case 'K': return pick("GT"); case 'M': return pick("AC");
                case 'S': return pick("CG"); case 'W': return pick("AT"); case 'B': return pick("CGT"); case 'D': return pick("AGT");
                case 'H': return pick("ACT"); case 'V': return pick("ACG"); default: return pick("ACGT");
            }
        };
        run_apply(text, 20, replace, [](uint8_t) -> uint8_t { return 'X'; }, ix.original_symbols);  // :98-107
    }
    // :139-160 text $ revcomp $ -> ranks
    const size_t g = text.size();
    std::vector<uint8_t> t(2 * g + 2);
    auto rank = [](uint8_t c) -> uint8_t { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 5; };
    parallel([&](size_t lo, size_t hi, unsigned) {
        for (size_t i = lo; i < hi; ++i) { const uint8_t r = rank(text[i]); t[i] = r; t[2 * g - i] = (uint8_t)(r == 5 ? 5 : 5 - r); }  // complement of a rank: A<->T, C<->G, X stays
    }, g);
    t[g] = 0; t[2 * g + 1] = 0;
    ix.n = t.size();
    return t;
}

// indexing.rs:163-195 on the host: suffix array (SA-IS) -> BWT, SA sample 1/32 + extra rows, Less, rank blocks.
inline void finish_on_host(const std::vector<uint8_t>& t, Index& ix) {
    ix.bwt.resize(ix.n);
    auto finish = [&](auto* sa) {
        for (uint64_t i = 0; i < ix.n; ++i) {  // :166, :168-182
            const uint64_t p = (uint64_t)sa[i];
            ix.bwt[i] = p > 0 ? t[p - 1] : t[ix.n - 1];
            if (i % ix.sa_rate == 0) ix.sa_sample.push_back(p);
            else if (ix.bwt[i] == 0) ix.extra_rows[i] = p;
        }
    };
    const char* force64 = std::getenv("MAPAD_INDEX_FORCE_64");  // test hook: exercise the 64-bit suffix sorter (texts >= 2^31 rows) on small inputs
    if (ix.n < (1ull << 31) && !(force64 && force64[0] == '1')) { std::vector<int32_t> sa(ix.n); sais<uint8_t, int32_t>(t.data(), sa.data(), (int32_t)ix.n, 6); finish(sa.data()); }
    else { std::vector<int64_t> sa(ix.n); sais<uint8_t, int64_t>(t.data(), sa.data(), (int64_t)ix.n, 6); finish(sa.data()); }
    build_blocks(ix);
}

inline Index build_index(const std::vector<std::string>& names, const uint8_t* const* seqs, const uint64_t* lens, uint64_t seed, uint8_t fixed_replacement = 0) {
    Index ix;
    const std::vector<uint8_t> t = prepare_text(names, seqs, lens, seed, fixed_replacement, ix);
    finish_on_host(t, ix);
    return ix;
}

}  // namespace host
}  // namespace mapad
