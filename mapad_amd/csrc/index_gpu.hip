// index_gpu.hip — suffix sorting of the index text on the MI355X (`mapad index` at hg19 scale in seconds instead of a quarter of an hour).
//
// Replaces, for texts of any length < 2^40: `suffix_array(&ref_seq)`, `bwt(&ref_seq, &suffix_array)`, the 1/32 SA sample with its
// extra rows, `less` and the rank structure (src/index/indexing.rs:163-195).  The suffix order of text$revcomp$ is unique, so any
// correct construction reproduces rust-bio's; tests compare every product of this path byte for byte with the host SA-IS path.
//
// Method (prefix doubling in the manner of Larsson & Sadakane, sized for 288 GB of HBM — the whole SA and its inverse stay resident):
//   1. one counting pass + one scatter pass bucket all n suffixes by their first kPrefix symbols (6^4 = 1296 buckets);
//   2. every bucket is radix-sorted (rocPRIM) by the next 21 symbols packed 3 bits each into a 64-bit key; equal keys form a
//      group whose members share >= 25 symbols, rank(suffix) = first row of its group (written to the inverse array);
//   3. while groups of more than one suffix remain: sort each group by rank(suffix + h) (radix sort by that rank, then a stable radix
//      sort by group: no per-group launches, so one huge group — a run of N's — costs the same per element as millions of pairs), split it
//      where the keys differ, double h.  Ranks are always monotone in the true suffix order, so refining some groups deeper than others is
//      harmless.  i.i.d. genomes leave ~n^2 / 4^25 suffixes for step 3 (a few thousand at n = 6e9); repeats cost log(repeat length) rounds.
//   4. BWT, SA sample and the 64-byte rank blocks (fmd_device.hpp layout) are produced on the device from the finished SA.
// HBM traffic is dominated by the radix passes of step 2 (8 passes x 32 B per suffix) and three random 8-byte accesses per suffix.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/mapad_amd.h"
#include "host_index.hpp"

namespace mapad {
namespace gpuidx {

namespace {

constexpr int kPrefix = 4;                 // symbols that select the bucket
constexpr int kBuckets = 6 * 6 * 6 * 6;    // 6^kPrefix
constexpr int kKeySyms = 21;               // symbols per 64-bit sort key (3 bits each)
constexpr int kTextPad = 64;               // zero bytes behind the text so that keys near the end read '$'-like padding
constexpr uint64_t kMaxSortChunk = 1ull << 30;  // elements per rocPRIM call

#define GI_TRY(expr)                                                                                               \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

template <class T>
struct Buf {
    T* p = nullptr;
    size_t n = 0;
    Buf() = default;
    explicit Buf(size_t count) { alloc(count); }
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        if (hipMalloc((void**)&p, count * sizeof(T)) != hipSuccess) { p = nullptr; throw std::bad_alloc(); }
        n = count;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~Buf() { release(); }
};

__device__ __forceinline__ uint32_t prefix_code(const uint8_t* t, uint64_t i) {  // text is padded with zeros: positions >= n read 0
    return ((uint32_t)t[i] * 6u + t[i + 1]) * 36u + (uint32_t)t[i + 2] * 6u + t[i + 3];
}

// ---- step 1: bucket histogram and scatter ----------------------------------------------------------------------------------------
constexpr int kTile = 8192;  // positions per workgroup
__global__ void __launch_bounds__(1024) bucket_count_kernel(const uint8_t* __restrict__ t, uint64_t n, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[kBuckets];
    for (int k = threadIdx.x; k < kBuckets; k += 1024) h[k] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
    for (int k = threadIdx.x; k < kTile; k += 1024) {
        const uint64_t i = base + k;
        if (i < n) atomicAdd(&h[prefix_code(t, i)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kBuckets; k += 1024) if (h[k]) atomicAdd(&hist[k], (unsigned long long)h[k]);
}
__global__ void __launch_bounds__(1024) bucket_scatter_kernel(const uint8_t* __restrict__ t, uint64_t n, unsigned long long* __restrict__ cursor, uint64_t* __restrict__ sa) {
    __shared__ uint32_t h[kBuckets];
    __shared__ unsigned long long start[kBuckets];
    for (int k = threadIdx.x; k < kBuckets; k += 1024) h[k] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
    uint32_t code[kTile / 1024], off[kTile / 1024];
#pragma unroll
    for (int q = 0; q < kTile / 1024; ++q) {
        const uint64_t i = base + q * 1024 + threadIdx.x;
        code[q] = 0; off[q] = 0;
        if (i < n) { code[q] = prefix_code(t, i); off[q] = atomicAdd(&h[code[q]], 1u); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kBuckets; k += 1024) if (h[k]) start[k] = atomicAdd(&cursor[k], (unsigned long long)h[k]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kTile / 1024; ++q) {
        const uint64_t i = base + q * 1024 + threadIdx.x;
        if (i < n) sa[start[code[q]] + off[q]] = i;
    }
}

// ---- step 2: per-bucket keys ------------------------------------------------------------------------------------------------------
// key = symbols [p + kPrefix, p + kPrefix + 21) packed 3 bits each, most significant first (four aligned 8-byte loads cover the 21 bytes)
__global__ void __launch_bounds__(256) bucket_keys_kernel(const uint8_t* __restrict__ t, const uint64_t* __restrict__ sa, uint64_t m, uint64_t* __restrict__ key, uint64_t* __restrict__ val) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t p = sa[j];
    const uint64_t a = p + kPrefix;
    const uint64_t* w = reinterpret_cast<const uint64_t*>(t + (a & ~7ull));
    const uint64_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    const int sh = (int)(a & 7) * 8;
    // bytes a .. a+23 as three little-endian words
    const uint64_t b0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    const uint64_t b1 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
    const uint64_t b2 = sh ? (w2 >> sh) | (w3 << (64 - sh)) : w2;
    uint64_t k = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) k = (k << 3) | ((b0 >> (8 * s)) & 7);
#pragma unroll
    for (int s = 0; s < 8; ++s) k = (k << 3) | ((b1 >> (8 * s)) & 7);
#pragma unroll
    for (int s = 0; s < kKeySyms - 16; ++s) k = (k << 3) | ((b2 >> (8 * s)) & 7);
    key[j] = k;
    val[j] = p;
}

// head marker of a sorted run: position j if its key differs from its predecessor's (or j is a segment start), else 0;
// an inclusive max-scan of these markers gives every position the first position of its group
struct HeadMark {
    const uint64_t* key;
    const uint8_t* seg_start;  // may be nullptr: one segment starting at 0
    __device__ uint32_t operator()(uint32_t j) const {
        if (j == 0) return 0;
        if (seg_start && seg_start[j]) return j;
        return key[j] != key[j - 1] ? j : 0;
    }
};
// rows [row0, row0 + m): SA and inverse SA from a sorted chunk; `is_head[row]` = the row starts a group
__global__ void __launch_bounds__(256) write_chunk_kernel(const uint64_t* __restrict__ val, const uint32_t* __restrict__ head, uint64_t m, uint64_t row0, const uint64_t* __restrict__ rows,
                                                          uint64_t* __restrict__ sa, uint64_t* __restrict__ isa, uint8_t* __restrict__ is_head) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint32_t hd = head[j];
    const uint64_t row = rows ? rows[j] : row0 + j, head_row = rows ? rows[hd] : row0 + hd;
    const uint64_t p = val[j];
    sa[row] = p;
    isa[p] = head_row;
    is_head[row] = hd == (uint32_t)j;
}

// rows whose group has more than one member: !(head(row) && head(row + 1))
struct UnresolvedAbs {
    const uint8_t* is_head;
    uint64_t n;
    __device__ bool operator()(uint64_t row) const {
        const bool next_head = row + 1 >= n || is_head[row + 1];
        return !(is_head[row] && next_head);
    }
};
struct UnresolvedCount {
    UnresolvedAbs u;
    uint64_t base;
    __device__ unsigned long long operator()(uint64_t r) const { return u(base + r) ? 1ull : 0ull; }
};
struct RowAt {
    uint64_t base;
    __device__ uint64_t operator()(uint64_t r) const { return base + r; }
};

// ---- step 3: doubling round over the unresolved rows U (ascending; groups are runs of consecutive rows) -----------------------------
__global__ void __launch_bounds__(256) dbl_keys_kernel(const uint64_t* __restrict__ rows, uint64_t m, const uint64_t* __restrict__ sa, const uint64_t* __restrict__ isa, const uint8_t* __restrict__ is_head,
                                                       uint64_t n, uint64_t h, uint64_t* __restrict__ key, uint64_t* __restrict__ val, uint8_t* __restrict__ seg_start) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t row = rows[j], p = sa[row];
    key[j] = p + h < n ? isa[p + h] + 1 : 0;  // a suffix that ends within h symbols sorts first
    val[j] = p;
    seg_start[j] = is_head[row];
}
struct SegMark {  // position j if a group starts there, else 0: max-scan -> group id (= first position of the group) of every position
    const uint8_t* seg_start;
    __device__ uint32_t operator()(uint32_t j) const { return seg_start[j] ? j : 0u; }
};
__global__ void __launch_bounds__(256) gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t m, uint32_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < m) out[j] = src[idx[j]];
}
__global__ void __launch_bounds__(256) gather_pairs_kernel(const uint64_t* __restrict__ key, const uint64_t* __restrict__ val, const uint32_t* __restrict__ idx, uint32_t m, uint64_t* __restrict__ key_out,
                                                           uint64_t* __restrict__ val_out) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < m) { const uint32_t q = idx[j]; key_out[j] = key[q]; val_out[j] = val[q]; }
}
// chunk cuts: last group head in (lo, hi] / first group head in (lo, hi)
__global__ void __launch_bounds__(256) find_last_head_kernel(const uint64_t* __restrict__ rows, const uint8_t* __restrict__ is_head, uint64_t lo, uint64_t hi, unsigned long long* out) {
    const uint64_t j = lo + 1 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j <= hi && is_head[rows[j]]) atomicMax(out, (unsigned long long)j);
}
__global__ void __launch_bounds__(256) find_first_head_kernel(const uint64_t* __restrict__ rows, const uint8_t* __restrict__ is_head, uint64_t lo, uint64_t hi, unsigned long long* out) {
    const uint64_t j = lo + 1 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < hi && is_head[rows[j]]) atomicMin(out, (unsigned long long)j);
}
struct StillTied {  // after a round: positions of the chunk whose (new) group still has more than one member
    const uint32_t* head;
    uint32_t m;
    __device__ bool operator()(uint32_t j) const {
        const bool is_h = head[j] == j;
        const bool next_h = j + 1 >= m || head[j + 1] == j + 1;
        return !(is_h && next_h);
    }
};
__global__ void __launch_bounds__(256) gather_rows_kernel(const uint64_t* __restrict__ rows, const uint32_t* __restrict__ idx, uint64_t m, uint64_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) out[j] = rows[idx[j]];
}

// ---- step 4: products ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bwt_kernel(const uint8_t* __restrict__ t, const uint64_t* __restrict__ sa, uint64_t n, uint32_t rate_shift, uint8_t* __restrict__ bwt, uint64_t* __restrict__ sample,
                                                  unsigned long long* __restrict__ extra /* [0] = count, then (row, value) pairs */) {
    // grid-stride: a launch may not have 2^32 or more threads
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
        const uint64_t p = sa[r];
        const uint8_t c = p ? t[p - 1] : t[n - 1];  // indexing.rs:166
        bwt[r] = c;
        if ((r & ((1ull << rate_shift) - 1)) == 0) sample[r >> rate_shift] = p;  // :168-182
        else if (c == 0) { const unsigned long long k = atomicAdd(&extra[0], 1ull); if (k < 4) { extra[1 + 2 * k] = r; extra[2 + 2 * k] = p; } }
        if (c == 0) { const unsigned long long k = atomicAdd(&extra[16], 1ull); if (k < 4) extra[17 + k] = r; }  // '$' rows of the BWT
    }
}
// one thread per 96-row block: bit planes + symbol counts of the block
__global__ void __launch_bounds__(256) block_planes_kernel(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t n_blocks, uint64_t* __restrict__ blocks, uint32_t* __restrict__ counts /* [5][n_blocks] */) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t c[6] = {0, 0, 0, 0, 0, 0};
    uint64_t* blk = blocks + b * kBlockWords;
    for (int w = 0; w < 4; ++w) {
        uint32_t p0 = 0, p1 = 0, p2 = 0;
        const uint64_t r0 = b * kBlockRows + (uint64_t)kSubRows * w;
        if (r0 < n) {
            const uint2* src = reinterpret_cast<const uint2*>(bwt + r0);  // the BWT buffer is padded to a multiple of 96 rows; 24 rows = 3 x 8 bytes
            for (int q = 0; q < 3; ++q) {
                const uint2 v = src[q];
                const uint32_t ws[2] = {v.x, v.y};
                for (int d = 0; d < 2; ++d)
                    for (int e = 0; e < 4; ++e) {
                        const int bit = q * 8 + d * 4 + e;
                        if (r0 + bit >= n) continue;
                        const uint32_t a = (ws[d] >> (8 * e)) & 0xFF;  // rank: $=0 A=1 C=2 G=3 T=4 X=5 -> code 0,4,5,6,7,1
                        const uint32_t code = a == 0 ? 0u : a == 5 ? 1u : a + 3u;
                        p0 |= (code & 1) << bit; p1 |= ((code >> 1) & 1) << bit; p2 |= (code >> 2) << bit;
                        c[a] += 1;
                    }
            }
        }
        blk[2 * w] = pack_sub0(0, p0); blk[2 * w + 1] = pack_sub1(p1, p2);
    }
    for (int k = 0; k < 5; ++k) counts[(uint64_t)k * n_blocks + b] = c[k + 1];
}
__global__ void __launch_bounds__(256) block_counts_kernel(const uint64_t* __restrict__ prefix /* [5][n_blocks] */, uint64_t n_blocks, uint64_t* __restrict__ blocks, uint64_t* __restrict__ x_counts) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    for (int w = 0; w < 4; ++w) blocks[b * kBlockWords + 2 * w] |= prefix[(uint64_t)w * n_blocks + b] & kCountMask;
    x_counts[b] = prefix[4 * n_blocks + b];
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Scratch {  // temporary storage for rocPRIM calls, grown on demand
    Buf<uint8_t> b;
    void* get(size_t bytes) { if (bytes > b.n) b.alloc(bytes + bytes / 4 + 256); return b.p; }
};

inline uint32_t grid_for(uint64_t m, uint32_t block) {  // one thread per element; callers keep m * 1 thread below 2^32 threads per launch
    if (m >= (1ull << 32)) throw std::runtime_error("launch of 2^32 or more threads");
    return (uint32_t)((m + block - 1) / block);
}

// Sorts one chunk [0, m) of (key, val) pairs — as a whole, or inside the groups marked by `seg_start` — and computes, per position,
// the first position of its (new) group in `head`.  Input in key[0] / val[0]; returns the index of the buffers that hold the result.
struct ChunkSorter {
    Buf<uint64_t> key[2], val[2];
    Buf<uint32_t> head, u32[6], n_sel;
    Buf<uint8_t> seg_start;
    Scratch tmp;
    size_t cap = 0, cap_grouped = 0;
    void reserve(size_t m, bool grouped) {
        if (m > cap) {
            const size_t c = m + m / 8 + 1024;
            for (int i = 0; i < 2; ++i) { key[i].alloc(c + 8); val[i].alloc(c + 8); }
            head.alloc(c); seg_start.alloc(c + 8);
            if (!n_sel.p) n_sel.alloc(2);
            cap = c;
            cap_grouped = 0;
        }
        if (grouped && cap > cap_grouped) { for (auto& b : u32) b.alloc(cap); cap_grouped = cap; }
    }
    void heads(int cur, uint32_t m, bool grouped, hipStream_t s) {  // inclusive max-scan of the head markers
        size_t bytes = 0;
        auto marks = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), HeadMark{key[cur].p, grouped ? seg_start.p : nullptr});
        GI_TRY(rocprim::inclusive_scan(nullptr, bytes, marks, head.p, m, rocprim::maximum<uint32_t>(), s));
        GI_TRY(rocprim::inclusive_scan(tmp.get(bytes), bytes, marks, head.p, m, rocprim::maximum<uint32_t>(), s));
    }
    int sort_plain(uint32_t m, unsigned bits, hipStream_t s) {
        rocprim::double_buffer<uint64_t> k(key[0].p, key[1].p), v(val[0].p, val[1].p);
        size_t bytes = 0;
        GI_TRY(rocprim::radix_sort_pairs(nullptr, bytes, k, v, m, 0, bits, s));
        GI_TRY(rocprim::radix_sort_pairs(tmp.get(bytes), bytes, k, v, m, 0, bits, s));
        const int cur = k.current() == key[0].p ? 0 : 1;
        if ((v.current() == val[0].p ? 0 : 1) != cur) throw std::runtime_error("rocPRIM double buffers out of step");
        heads(cur, m, false, s);
        return cur;
    }
    int sort_grouped(uint32_t m, unsigned key_bits, hipStream_t s) {
        uint32_t *grp = u32[0].p, *perm1 = u32[1].p, *gkey = u32[2].p, *gkey2 = u32[3].p, *perm2 = u32[4].p, *iota = u32[5].p;
        size_t bytes = 0;
        // group id of every position
        auto marks = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), SegMark{seg_start.p});
        GI_TRY(rocprim::inclusive_scan(nullptr, bytes, marks, grp, m, rocprim::maximum<uint32_t>(), s));
        GI_TRY(rocprim::inclusive_scan(tmp.get(bytes), bytes, marks, grp, m, rocprim::maximum<uint32_t>(), s));
        // order by key (input buffers are left intact), then stably by group
        auto ident = rocprim::counting_iterator<uint32_t>(0);
        GI_TRY(rocprim::transform(ident, iota, m, rocprim::identity<uint32_t>(), s));
        GI_TRY(rocprim::radix_sort_pairs(nullptr, bytes, key[0].p, key[1].p, iota, perm1, m, 0, key_bits, s));
        GI_TRY(rocprim::radix_sort_pairs(tmp.get(bytes), bytes, key[0].p, key[1].p, iota, perm1, m, 0, key_bits, s));
        hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, grp, perm1, m, gkey);
        unsigned gbits = 1;
        while ((1ull << gbits) < m) ++gbits;
        GI_TRY(rocprim::radix_sort_pairs(nullptr, bytes, gkey, gkey2, perm1, perm2, m, 0, gbits, s));
        GI_TRY(rocprim::radix_sort_pairs(tmp.get(bytes), bytes, gkey, gkey2, perm1, perm2, m, 0, gbits, s));
        hipLaunchKernelGGL(gather_pairs_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, key[0].p, val[0].p, perm2, m, key[1].p, val[1].p);
        GI_TRY(hipGetLastError());
        heads(1, m, true, s);
        return 1;
    }
};

}  // namespace

// Fills bwt, sa_sample, extra_rows, blocks, x_counts, less, sentinel of `ix` (ix.n and the text `t` of n rank bytes are given).
void suffix_products(const uint8_t* t_host, host::Index& ix, int device, bool verbose) {
    const uint64_t n = ix.n;
    if (n < 8 || n >= (1ull << 40)) throw std::length_error("text length out of range for the GPU indexer");
    GI_TRY(hipSetDevice(device));
    hipStream_t s = nullptr;
    const double t_begin = now_s();
    double t_mark = t_begin;
    auto lap = [&](const char* what) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const double t = now_s();
        std::fprintf(stderr, "[index_gpu] %-28s %8.3f s\n", what, t - t_mark);
        t_mark = t;
    };

    Buf<uint8_t> d_t(n + kTextPad);
    GI_TRY(hipMemsetAsync(d_t.p + n, 0, kTextPad, s));
    GI_TRY(hipMemcpyAsync(d_t.p, t_host, n, hipMemcpyHostToDevice, s));
    Buf<uint64_t> d_sa(n), d_isa(n);
    Buf<uint8_t> d_head(n + 1);
    Buf<unsigned long long> d_hist(2 * kBuckets);
    GI_TRY(hipMemsetAsync(d_hist.p, 0, 2 * kBuckets * sizeof(unsigned long long), s));
    lap("upload + alloc");

    // ---- 1. buckets ----
    const uint32_t tiles = (uint32_t)((n + kTile - 1) / kTile);
    hipLaunchKernelGGL(bucket_count_kernel, dim3(tiles), dim3(1024), 0, s, d_t.p, n, d_hist.p);
    std::vector<unsigned long long> hist(kBuckets), start(kBuckets + 1, 0);
    GI_TRY(hipMemcpyAsync(hist.data(), d_hist.p, kBuckets * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    GI_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < kBuckets; ++k) start[k + 1] = start[k] + hist[k];
    if (start[kBuckets] != n) throw std::runtime_error("bucket histogram does not add up");
    GI_TRY(hipMemcpyAsync(d_hist.p + kBuckets, start.data(), kBuckets * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(tiles), dim3(1024), 0, s, d_t.p, n, d_hist.p + kBuckets, d_sa.p);
    GI_TRY(hipGetLastError());
    lap("bucket count + scatter");

    // ---- 2. sort inside buckets ----
    ChunkSorter cs;
    {
        uint64_t biggest = 0;
        for (int k = 0; k < kBuckets; ++k) biggest = std::max<uint64_t>(biggest, hist[k]);
        if (biggest > kMaxSortChunk) throw std::length_error("a prefix bucket holds more than 2^30 suffixes (not supported)");
        cs.reserve((size_t)std::max<uint64_t>(biggest, 1), false);
    }
    for (int k = 0; k < kBuckets; ++k) {
        const uint64_t m = hist[k], row0 = start[k];
        if (m == 0) continue;
        hipLaunchKernelGGL(bucket_keys_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, d_t.p, d_sa.p + row0, m, cs.key[0].p, cs.val[0].p);
        const int cur = cs.sort_plain((uint32_t)m, 3 * kKeySyms, s);
        hipLaunchKernelGGL(write_chunk_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, cs.val[cur].p, cs.head.p, m, row0, (const uint64_t*)nullptr, d_sa.p, d_isa.p, d_head.p);
    }
    GI_TRY(hipGetLastError());
    lap("bucket sorts");

    // ---- 3. doubling over the unresolved rows ----
    Buf<uint64_t> d_rows[2];
    Buf<unsigned long long> d_cnt(2);
    uint64_t m_unres = 0;
    {
        // count (one pass), then collect the rows in ascending order
        std::vector<uint64_t> chunk_cnt;
        Scratch tmp;
        uint64_t total = 0;
        const UnresolvedAbs pred{d_head.p, n};
        for (int pass = 0; pass < 2; ++pass) {
            uint64_t out_off = 0;
            if (pass == 1) { d_rows[0].alloc(std::max<uint64_t>(total, 1)); d_rows[1].alloc(std::max<uint64_t>(total, 1)); }
            for (uint64_t base = 0; base < n; base += kMaxSortChunk) {
                const uint64_t mm = std::min<uint64_t>(kMaxSortChunk, n - base);
                size_t bytes = 0;
                if (pass == 0) {
                    auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), UnresolvedCount{pred, base});
                    GI_TRY(rocprim::reduce(nullptr, bytes, flags, d_cnt.p, 0ull, mm, rocprim::plus<unsigned long long>(), s));
                    GI_TRY(rocprim::reduce(tmp.get(bytes), bytes, flags, d_cnt.p, 0ull, mm, rocprim::plus<unsigned long long>(), s));
                    unsigned long long c = 0;
                    GI_TRY(hipMemcpyAsync(&c, d_cnt.p, 8, hipMemcpyDeviceToHost, s));
                    GI_TRY(hipStreamSynchronize(s));
                    chunk_cnt.push_back(c);
                    total += c;
                } else {
                    const uint64_t c = chunk_cnt[base / kMaxSortChunk];
                    if (c == 0) continue;
                    auto rows_it = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), RowAt{base});
                    GI_TRY(rocprim::select(nullptr, bytes, rows_it, d_rows[0].p + out_off, d_cnt.p, mm, pred, s));
                    GI_TRY(rocprim::select(tmp.get(bytes), bytes, rows_it, d_rows[0].p + out_off, d_cnt.p, mm, pred, s));
                    out_off += c;
                }
            }
        }
        m_unres = total;
    }
    lap("collect unresolved");
    unsigned key_bits = 1;
    while ((1ull << key_bits) <= n + 1) ++key_bits;
    uint64_t h = kPrefix + kKeySyms;
    int cur_rows = 0, rounds = 0;
    Buf<uint32_t> d_keep;
    constexpr uint64_t kChunk = 1ull << 28;
    while (m_unres > 0) {
        if (++rounds > 64) throw std::runtime_error("prefix doubling did not converge");
        if (verbose) std::fprintf(stderr, "[index_gpu] round %d: h = %llu, %llu unresolved suffixes\n", rounds, (unsigned long long)h, (unsigned long long)m_unres);
        const uint64_t* rows = d_rows[cur_rows].p;
        uint64_t* rows_next = d_rows[1 - cur_rows].p;
        uint64_t next_m = 0;
        for (uint64_t lo = 0; lo < m_unres;) {
            uint64_t hi = std::min<uint64_t>(m_unres, lo + kChunk);
            if (hi < m_unres) {  // cut at a group boundary: the last group head in (lo, hi], else the first one behind it
                unsigned long long z = 0;
                GI_TRY(hipMemcpyAsync(d_cnt.p, &z, 8, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(find_last_head_kernel, dim3(grid_for(hi - lo, 256)), dim3(256), 0, s, rows, d_head.p, lo, hi, d_cnt.p);
                GI_TRY(hipMemcpyAsync(&z, d_cnt.p, 8, hipMemcpyDeviceToHost, s));
                GI_TRY(hipStreamSynchronize(s));
                if (z > lo) hi = z;
                else {
                    const uint64_t lim = std::min<uint64_t>(m_unres, lo + kMaxSortChunk);
                    z = ~0ull;
                    GI_TRY(hipMemcpyAsync(d_cnt.p, &z, 8, hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(find_first_head_kernel, dim3(grid_for(lim - hi, 256)), dim3(256), 0, s, rows, d_head.p, hi, lim, d_cnt.p);
                    GI_TRY(hipMemcpyAsync(&z, d_cnt.p, 8, hipMemcpyDeviceToHost, s));
                    GI_TRY(hipStreamSynchronize(s));
                    if (z != ~0ull) hi = z;
                    else if (lim == m_unres) hi = m_unres;
                    else throw std::length_error("a group of more than 2^30 equal suffixes (not supported)");
                }
            }
            const uint32_t m = (uint32_t)(hi - lo);
            cs.reserve(m, true);
            hipLaunchKernelGGL(dbl_keys_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, rows + lo, (uint64_t)m, d_sa.p, d_isa.p, d_head.p, n, h, cs.key[0].p, cs.val[0].p, cs.seg_start.p);
            const int cur = cs.sort_grouped(m, key_bits, s);
            hipLaunchKernelGGL(write_chunk_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, cs.val[cur].p, cs.head.p, (uint64_t)m, (uint64_t)0, rows + lo, d_sa.p, d_isa.p, d_head.p);
            // rows of this chunk that are still tied
            if (d_keep.n < (size_t)m + 8) d_keep.alloc((size_t)m + m / 8 + 1024);
            size_t bytes = 0;
            GI_TRY(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), d_keep.p, cs.n_sel.p, m, StillTied{cs.head.p, m}, s));
            GI_TRY(rocprim::select(cs.tmp.get(bytes), bytes, rocprim::counting_iterator<uint32_t>(0), d_keep.p, cs.n_sel.p, m, StillTied{cs.head.p, m}, s));
            uint32_t kept = 0;
            GI_TRY(hipMemcpyAsync(&kept, cs.n_sel.p, 4, hipMemcpyDeviceToHost, s));
            GI_TRY(hipStreamSynchronize(s));
            if (kept) hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(kept, 256)), dim3(256), 0, s, rows + lo, d_keep.p, (uint64_t)kept, rows_next + next_m);
            next_m += kept;
            lo = hi;
        }
        GI_TRY(hipGetLastError());
        m_unres = next_m;
        cur_rows = 1 - cur_rows;
        h *= 2;
    }
    lap("prefix doubling");
    d_rows[0].release(); d_rows[1].release(); d_isa.release(); d_head.release(); d_keep.release();
    for (int i = 0; i < 2; ++i) { cs.key[i].release(); cs.val[i].release(); }
    cs.head.release(); cs.seg_start.release();
    for (auto& b : cs.u32) b.release();

    // ---- 4. BWT, SA sample, rank blocks ----
    uint32_t rate_shift = 0;
    while ((1ull << rate_shift) < ix.sa_rate) ++rate_shift;
    if ((1ull << rate_shift) != ix.sa_rate) throw std::runtime_error("SA sampling rate must be a power of two");
    const uint64_t n_samples = (n + ix.sa_rate - 1) / ix.sa_rate;
    const uint64_t n_blocks = (n + kBlockRows - 1) / kBlockRows + 1;  // one spare block (host_index.hpp: build_blocks)
    Buf<uint8_t> d_bwt(n_blocks * kBlockRows);
    Buf<uint64_t> d_sample(n_samples);
    Buf<unsigned long long> d_extra(32);
    GI_TRY(hipMemsetAsync(d_extra.p, 0, 32 * 8, s));
    GI_TRY(hipMemsetAsync(d_bwt.p + n, 0, n_blocks * kBlockRows - n, s));
    hipLaunchKernelGGL(bwt_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, s, d_t.p, d_sa.p, n, rate_shift, d_bwt.p, d_sample.p, d_extra.p);
    GI_TRY(hipGetLastError());
    unsigned long long extra[32];
    GI_TRY(hipMemcpyAsync(extra, d_extra.p, sizeof extra, hipMemcpyDeviceToHost, s));
    GI_TRY(hipStreamSynchronize(s));
    d_sa.release(); d_t.release();
    if (extra[16] != 2) throw std::runtime_error("BWT must contain exactly two sentinels");
    ix.sentinel[0] = std::min(extra[17], extra[18]); ix.sentinel[1] = std::max(extra[17], extra[18]);
    ix.extra_rows.clear();
    for (unsigned long long k = 0; k < extra[0] && k < 4; ++k) ix.extra_rows[extra[1 + 2 * k]] = extra[2 + 2 * k];
    lap("BWT + SA sample");

    Buf<uint64_t> d_blocks(n_blocks * kBlockWords), d_prefix(5 * n_blocks), d_xc(n_blocks);
    Buf<uint32_t> d_counts(5 * n_blocks);
    hipLaunchKernelGGL(block_planes_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, d_bwt.p, n, n_blocks, d_blocks.p, d_counts.p);
    GI_TRY(hipGetLastError());
    uint64_t totals[5];
    for (int k = 0; k < 5; ++k) {
        size_t bytes = 0;
        const uint32_t* in = d_counts.p + (uint64_t)k * n_blocks;
        uint64_t* out = d_prefix.p + (uint64_t)k * n_blocks;
        GI_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, (uint64_t)0, n_blocks, rocprim::plus<uint64_t>(), s));
        GI_TRY(rocprim::exclusive_scan(cs.tmp.get(bytes), bytes, in, out, (uint64_t)0, n_blocks, rocprim::plus<uint64_t>(), s));
        GI_TRY(hipMemcpyAsync(&totals[k], out + n_blocks - 1, 8, hipMemcpyDeviceToHost, s));  // the spare block is empty: its prefix is the total
    }
    hipLaunchKernelGGL(block_counts_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, d_prefix.p, n_blocks, d_blocks.p, d_xc.p);
    GI_TRY(hipGetLastError());
    GI_TRY(hipStreamSynchronize(s));
    lap("rank blocks");

    ix.bwt.resize(n);
    GI_TRY(hipMemcpy(ix.bwt.data(), d_bwt.p, n, hipMemcpyDeviceToHost));
    ix.sa_sample.resize(n_samples);
    GI_TRY(hipMemcpy(ix.sa_sample.data(), d_sample.p, n_samples * 8, hipMemcpyDeviceToHost));
    ix.blocks.resize(n_blocks * kBlockWords);
    GI_TRY(hipMemcpy(ix.blocks.data(), d_blocks.p, n_blocks * kBlockBytes, hipMemcpyDeviceToHost));
    ix.x_counts.clear();
    if (totals[4]) { ix.x_counts.resize(n_blocks); GI_TRY(hipMemcpy(ix.x_counts.data(), d_xc.p, n_blocks * 8, hipMemcpyDeviceToHost)); }
    const uint64_t per[6] = {2, totals[0], totals[1], totals[2], totals[3], totals[4]};  // Less
    uint64_t acc = 0;
    for (int c = 0; c < 6; ++c) { ix.less[c] = acc; acc += per[c]; }
    ix.less[6] = acc; ix.less[7] = acc;
    if (acc != n) throw std::runtime_error("symbol counts do not add up to the text length");
    lap("download");
    if (verbose) std::fprintf(stderr, "[index_gpu] n = %llu rows in %.3f s (%d doubling rounds)\n", (unsigned long long)n, now_s() - t_begin, rounds);
}

}  // namespace gpuidx
}  // namespace mapad

