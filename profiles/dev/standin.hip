// standin.hip — a kernel with the resource shape of RCCL's transfer kernel, for the 1-GPU rehearsal of the multi-GPU gather (profiles/dev/rccl_standin.py).
// ncclDevKernel_Generic_{1,2,4} in this image's librccl.so.1.0.70200 (gfx950 code object, read with clang-offload-bundler + llvm-readelf --notes):
// 248-256 VGPRs, 37 664 bytes of LDS, 696-712 bytes of scratch per lane, up to 512 threads per block.  This kernel asks for the same: 256 threads, 248 VGPRs,
// 37 664 bytes of LDS, a scratch array; `blocks` blocks (RCCL: one per channel) copy `words` 16-byte words from src to dst.  No GPU-to-GPU link is involved: what
// is measured is how long such a kernel waits for a CU beside the persistent search wavefronts, and what its presence costs the search.
#include <hip/hip_runtime.h>
#include <cstdint>

extern "C" __global__ void __launch_bounds__(256) standin_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t words, unsigned long long* stamp) {
    extern __shared__ uint32_t lds[];  // 37 664 bytes, asked for at the launch
    volatile uint32_t spill[174];  // 696 bytes of scratch per lane
    for (int i = 0; i < 174; i += 29) spill[i] = threadIdx.x + i;
    asm volatile("v_mov_b32 v247, 0" ::: "v247");  // the register file a RCCL wave takes: 248 VGPRs -> two waves per SIMD
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = wall_clock64();  // when the first block became resident (100 MHz constant clock)
    for (int i = threadIdx.x; i < 37664 / 4; i += 256) lds[i] = i;
    __syncthreads();
    uint32_t acc = lds[threadIdx.x] + spill[29];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256) {
        uint4 v = src[i];
        v.x ^= acc & 0u;  // (keeps lds / scratch alive without changing the data)
        dst[i] = v;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}

extern "C" int standin_launch(void* stream, const void* src, void* dst, uint64_t bytes, unsigned blocks, void* stamp) {
    hipLaunchKernelGGL(standin_copy, dim3(blocks), dim3(256), 37664, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes / 16, (unsigned long long*)stamp);
    return (int)hipGetLastError();
}
