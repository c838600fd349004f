// randline.hip — what the memory system gives the search kernel's access shape: one-wavefront blocks, a quad (4 lanes x 16 B) reads one random
// 64-byte line (or 8 lanes one 128-byte line) of a buffer far larger than every cache, CHAIN dependent round trips with INFLIGHT independent lines each,
// optionally one 32-byte store to another random line per trip.  Prints lines/s and TB/s for several blocks-per-CU.  Diagnostic, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int INFLIGHT, int LANES, bool STORE>
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ buf, uint4* __restrict__ wbuf, uint64_t n_lines, int trips, uint32_t* sink) {
    const uint32_t lane = threadIdx.x, grp = lane / LANES, sub = lane % LANES;
    uint64_t s = (uint64_t)(blockIdx.x * (64 / LANES) + grp) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int t = 0; t < trips; ++t) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = (s >> 20) % n_lines;
            v[k] = buf[line * (LANES) + sub];  // LANES x 16 B = one 64-byte (4) or 128-byte (8) line
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc += v[k].x ^ v[k].w;
        s ^= (uint64_t)__shfl(acc, grp * LANES) & 1u;  // the next addresses depend on what came back
        if (STORE) {
            const uint64_t line = ((s >> 24) * 2654435761ull) % n_lines;
            if (sub < 2) wbuf[line * LANES + sub] = make_uint4(acc, lane, t, 0);  // 32 bytes: one node
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int INFLIGHT, int LANES, bool STORE>
void run(const uint4* buf, uint4* wbuf, uint64_t bytes, uint32_t* sink, int cus) {
    const uint64_t n_lines = bytes / (16 * LANES);
    for (int per_cu : {4, 8, 11, 16, 32}) {
        const int blocks = cus * per_cu, trips = 4000;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        chase<INFLIGHT, LANES, STORE><<<blocks, 64>>>(buf, wbuf, n_lines, 200, sink);
        CK(hipEventRecord(a));
        chase<INFLIGHT, LANES, STORE><<<blocks, 64>>>(buf, wbuf, n_lines, trips, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double lines = (double)blocks * (64 / LANES) * trips * INFLIGHT;
        const double wl = STORE ? (double)blocks * (64 / LANES) * trips : 0;
        std::printf("line %3d B  in flight %d  store %d  blocks/CU %2d: %7.2f ms  %6.1f G lines/s read (%5.2f TB/s)%s  trip %6.0f ns\n", 16 * LANES, INFLIGHT, (int)STORE, per_cu, ms,
                    lines / ms * 1e-6, lines * 16 * LANES / ms * 1e-9, STORE ? "" : "", ms * 1e6 / trips);
        (void)wl;
    }
}

int main(int argc, char** argv) {
    char* suffix = nullptr;
    const uint64_t gb = argc > 1 ? std::strtoull(argv[1], &suffix, 10) : 64;
    const uint64_t bytes = (suffix && *suffix == 'M') ? gb << 20 : gb << 30;  // "64" = 64 GiB (beyond every cache), "128M" = 128 MiB (Infinity-Cache resident)
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    uint4 *buf, *wbuf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&wbuf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(wbuf, 0, bytes));
    std::printf("%s, %d CUs, buffers 2 x %llu MiB\n", p.name, p.multiProcessorCount, (unsigned long long)(bytes >> 20));
    run<1, 4, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<2, 4, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<4, 4, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<8, 4, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<2, 8, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<4, 8, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<4, 4, true>(buf, wbuf, bytes, sink, p.multiProcessorCount);
    run<4, 2, false>(buf, wbuf, bytes, sink, p.multiProcessorCount);  // 32-byte pieces (a node)
    return 0;
}
