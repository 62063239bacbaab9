// prints how workgroups of a (512 x 256 threads, ~200 VGPR) launch land on CUs / wave slots
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/hwid.hip -o tools/exp/hwid   (the binary is not kept in git)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ void __launch_bounds__(256, 2) k(unsigned* out, float* sink) {
    extern __shared__ float lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
    float acc = 0.f;
    for (int i = 0; i < 20000; ++i) acc = fmaf(acc, 1.0001f, 0.5f);   // stay resident a while
    lds[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (acc == 12345.f) sink[0] = lds[(threadIdx.x + 1) & 255];
}
int main() {
    const int nb = 512;
    unsigned* d; float* s;
    hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned)); hipMalloc(&s, 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 30 * 1024, 0, d, s);
    hipDeviceSynchronize();
    static unsigned h[nb * 4 * 2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::map<unsigned, int> wid;
    for (int b = 0; b < 8; ++b) {
        printf("block %3d:", b);
        for (int w = 0; w < 4; ++w) { unsigned v = h[(b * 4 + w) * 2]; printf("  [xcc %u se %u sh %u cu %2u simd %u wave %2u]", h[(b*4+w)*2+1] & 15, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15, (v >> 4) & 3, v & 15); }
        printf("\n");
    }
    std::map<unsigned long long, int> percu;
    for (int b = 0; b < nb; ++b) { unsigned v = h[(b * 4) * 2]; unsigned long long key = ((unsigned long long)(h[(b*4)*2+1] & 15) << 32) | (v & 0xff00); percu[key]++; wid[v & 15]++; }
    std::map<int,int> hist; for (auto& kv : percu) hist[kv.second]++;
    printf("distinct CUs used: %zu; workgroups-per-CU histogram:", percu.size()); for (auto& kv : hist) printf(" %d WG x %d CUs;", kv.first, kv.second); printf("\n");
    printf("wave-slot ids of wave 0:"); for (auto& kv : wid) printf(" slot %u x %d;", kv.first, kv.second); printf("\n");
    // co-resident pairs: which block ids share a CU
    std::map<unsigned long long, int> first; int shown = 0;
    for (int b = 0; b < nb && shown < 6; ++b) { unsigned v = h[(b*4)*2]; unsigned long long key = ((unsigned long long)(h[(b*4)*2+1] & 15) << 32) | (v & 0xff00); if (first.count(key)) { printf("blocks %d and %d share a CU (wave slots %u, %u)\n", first[key], b, h[(first[key]*4)*2] & 15, v & 15); ++shown; } else first[key] = b; }
    return 0;
}
