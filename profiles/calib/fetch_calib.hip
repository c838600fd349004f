// fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE on gfx950 for THIS path's access pattern (MI355X_MICROARCH.md §HBM: "calibrate on
// a known byte count in your own access pattern before trusting an absolute").
//   gather_quads : every quad reads one random 128-byte block of an 8 GiB table, lane w loading 2 x 16 B at +32w (fmd_device.hpp::quad_occ)
//   gather_lanes : every lane reads one random, 32-byte aligned 32-byte record (search_core.hpp node / PosInfo loads)
//   gather_quads64 : every quad reads one random 64-byte block, lane w loading 16 B at +16w (the 96-row index block of late round 5): does the L2 fetch 64 or
//                    128 bytes for it, and what does FETCH_SIZE tally?
// Known bytes: gather_quads N_q x 128 B, gather_lanes N_l x 32 B (table >> Infinity Cache, so every access misses).
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void gather_quads(const ulonglong2* t, uint64_t n_blocks, uint64_t per_quad, uint64_t* sink) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, quad = tid >> 2, w = tid & 3;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_quad; ++i) {
        const uint64_t b = mix(quad * per_quad + i + 1) % n_blocks;
        const ulonglong2 v0 = t[b * 8 + 2 * w], v1 = t[b * 8 + 2 * w + 1];
        acc += v0.x ^ v0.y ^ v1.x ^ v1.y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void gather_quads64(const ulonglong2* t, uint64_t n_blocks, uint64_t per_quad, uint64_t* sink) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, quad = tid >> 2, w = tid & 3;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_quad; ++i) {
        const uint64_t b = mix(quad * per_quad + i + 3) % n_blocks;
        const ulonglong2 v = t[b * 4 + w];
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// the same with eight independent blocks in flight per quad and iteration: is the rate above the memory system's, or the loop's?
__global__ void gather_quads64_x8(const ulonglong2* t, uint64_t n_blocks, uint64_t per_quad, uint64_t* sink) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, quad = tid >> 2, w = tid & 3;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_quad; i += 8) {
        ulonglong2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[(mix(quad * per_quad + i + k + 5) % n_blocks) * 4 + w];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void gather_lanes(const ulonglong2* t, uint64_t n_recs, uint64_t per_lane, uint64_t* sink) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_lane; ++i) {
        const uint64_t r = mix(tid * per_lane + i + 7) % n_recs;
        const ulonglong2 v0 = t[r * 2], v1 = t[r * 2 + 1];
        acc += v0.x ^ v0.y ^ v1.x ^ v1.y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
int main() {
    const uint64_t bytes = 8ull << 30;
    void* t; uint64_t* sink;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc((void**)&sink, 8) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(t, 1, bytes);
    hipDeviceSynchronize();
    const uint32_t grid = 256 * 32, block = 64;
    const uint64_t per = 512;
    hipLaunchKernelGGL(gather_quads, dim3(grid), dim3(block), 0, 0, (const ulonglong2*)t, bytes / 128, per, sink);
    hipLaunchKernelGGL(gather_lanes, dim3(grid), dim3(block), 0, 0, (const ulonglong2*)t, bytes / 32, per, sink);
    hipLaunchKernelGGL(gather_quads64_x8, dim3(grid), dim3(block), 0, 0, (const ulonglong2*)t, bytes / 64, per, sink);
    hipLaunchKernelGGL(gather_quads64, dim3(grid), dim3(block), 0, 0, (const ulonglong2*)t, bytes / 64, per, sink);
    hipDeviceSynchronize();
    std::printf("gather_quads64 known_bytes %llu (x2 if the L2 fills whole 128-byte lines)\n", (unsigned long long)((uint64_t)grid * block / 4 * per * 64));
    std::printf("gather_quads known_bytes %llu\n", (unsigned long long)((uint64_t)grid * block / 4 * per * 128));
    std::printf("gather_lanes known_bytes %llu\n", (unsigned long long)((uint64_t)grid * block * per * 32));
    return 0;
}
