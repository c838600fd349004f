// Checks the assumption behind the XCD-partitioned arena pools: HW_REG_XCC_ID[3:0] identifies the XCD a wavefront runs on
// (8 distinct values; blocks b and b + 8 of a launch share one).  hipcc --offload-arch=gfx950 xcc_check.hip -o xcc_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }
int main() {
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> ids(h.begin(), h.end());
    int same = 0; for (int b = 0; b + 8 < n; ++b) same += h[b] == h[b + 8];
    std::printf("distinct xcc ids: %zu (", ids.size()); for (unsigned v : ids) std::printf("%u ", v);
    std::printf("), blocks b and b+8 on the same XCD: %d of %d\n", same, n - 8);
    return ids.size() == 8 ? 0 : 1;
}
