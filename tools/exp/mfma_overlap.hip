// Does a wave that issues f32 MFMAs back to back leave issue slots for another wave's VALU / LDS work on the
// same SIMD?  One workgroup of 8 waves (2 per SIMD): waves 0-3 run an MFMA loop with CH independent accumulator
// chains, waves 4-7 run a VALU (or LDS) loop; clock64() around each.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_overlap.hip -o tools/exp/mfma_overlap   (the binary is not kept in git)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void __launch_bounds__(512, 1) k(int mfma_on, int valu_kind, long long* out, float* sink) {
    __shared__ float lds[4096];
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
    __syncthreads();
    long long t0 = clock64();
    float res = 0.f;
    if (w < 4) {
        if (mfma_on) {
            f32x16 acc[CH];
            for (int c = 0; c < CH; ++c) acc[c] = (f32x16){0};
            float a = threadIdx.x * 0.001f, b = 1.0f;
            for (int it = 0; it < 2048 / CH; ++it) {
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            }
            for (int c = 0; c < CH; ++c) res += acc[c][0] + acc[c][15];
        }
    } else {
        if (valu_kind == 1) {           // dependent-free VALU stream
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
            for (int it = 0; it < 4096; ++it) {
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
            }
            res = x0 + x1 + x2 + x3;
        } else if (valu_kind == 2) {    // LDS reads + adds
            float s = 0.f; int idx = threadIdx.x & 1023;
            for (int it = 0; it < 2048; ++it) { s += lds[idx]; idx = (idx + 65) & 4095; }
            res = s;
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { out[w * 2] = t1 - t0; out[w * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
    if (res == 12345.678f) sink[0] = res;
}
template <int CH> void run(long long* d, float* s) {
    long long h[16];
    for (int vk = 1; vk <= 2; ++vk) for (int on = 0; on <= 1; ++on) {
        hipLaunchKernelGGL(k<CH>, dim3(1), dim3(512), 0, 0, on, vk, d, s);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("chains=%d %s mfma=%d : mfma waves", CH, vk == 1 ? "VALU" : "LDS ", on);
        for (int w = 0; w < 4; ++w) printf(" %lld(simd%lld)", h[w * 2], (h[w * 2 + 1] >> 4) & 3);
        printf(" | other waves");
        for (int w = 4; w < 8; ++w) printf(" %lld(simd%lld)", h[w * 2], (h[w * 2 + 1] >> 4) & 3);
        printf("\n");
    }
}
int main() {
    long long* d; float* s;
    (void)hipMalloc(&d, 16 * sizeof(long long)); (void)hipMalloc(&s, 4);
    run<1>(d, s); run<2>(d, s); run<4>(d, s);
    return 0;
}
