// simd_map.hip -- which SIMD does wave w of a workgroup land on?  (lab; decides whether per-wave roles of unequal
// cost should rotate with the block index).  Records HW_ID (gfx9 layout: wave [3:0], simd [5:4], cu [11:8], sh [12],
// se [15:13]) for every wave of blocks of 256 threads that stay resident long enough to overlap (4 blocks per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256) probe(unsigned* out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID, 32 bits
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | (x == 7.f ? 1u << 31 : 0);
}
int main()
{
    const int nb = 256 * 4 * 3;
    unsigned* d; hipMalloc(&d, nb * 4 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 30000, 0, d, 20000);       // 30 KB LDS: 4-5 blocks per CU
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    std::map<int, int> pat; int same_cu = 0;
    for (int b = 0; b < nb; ++b) {
        int key = 0; bool one = true;
        for (int w = 0; w < 4; ++w) {
            key = key * 4 + ((h[b * 4 + w] >> 4) & 3);
            one = one && ((h[b * 4 + w] >> 8) & 0xff) == ((h[b * 4] >> 8) & 0xff);
        }
        pat[key]++; same_cu += one;
    }
    printf("blocks %d, all waves on one CU: %d\nSIMD of waves (w0 w1 w2 w3) : count\n", nb, same_cu);
    for (auto& kv : pat) printf("  %d %d %d %d : %d\n", (kv.first >> 6) & 3, (kv.first >> 4) & 3, (kv.first >> 2) & 3, kv.first & 3, kv.second);
    for (int b = 0; b < 12; ++b) printf("block %d: %08x %08x %08x %08x\n", b, h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    return 0;
}
