// Where does a conv3x3_wgrad_x3 workgroup spend its time?  Builds the kernel with its phase clock and prints, for wave 0
// of one workgroup of a 128 -> 128, 64 x 64 x 64 weight gradient: ticks per phase (summed over its 32 rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX3W_TIMING=40 -Wno-unused-result -I scm_gan_amd/csrc -I include \
//         tools/exp/x3w_phases.hip -o tools/exp/bin/x3w_phases
#include "../../scm_gan_amd/csrc/conv3x3_wgrad_x3.hip"
#include <cstdio>
#include <vector>

// the two helpers of api.cpp the kernel file needs (the harness does not link the library)
void scm_set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int scm_check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return e != hipSuccess;
}

int main() {
    const int B = 64, C = 128, H = 64, W = 64;
    const size_t n = (size_t)B * C * H * W;
    float *x, *dy, *y, *part;
    unsigned* mask;
    hipMalloc(&x, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&y, n * 4); hipMalloc(&mask, n / 32 * 4);
    std::vector<float> hx(n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(y, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(mask, 0x5a, n / 32 * 4);
    WgradPlan p = {};
    p.mt = 4; p.nt = 2; p.spi = 2; p.rows_per_split = 32; p.nseg = 1; p.PS = 64; p.co_groups = 1; p.ci_groups = 2;
    p.nsplit = B * p.spi;
    hipMalloc(&part, (size_t)p.nsplit * 9 * C * C * 4);
    WgradKArgs a = {};
    a.x1 = x; a.c1 = C; a.dy = dy; a.gate = y; a.gate_act = SCM_ACT_LRELU; a.cout = C; a.part = part;
    a.B = B; a.H = H; a.W = W; a.pad_mode = SCM_PAD_CIRCULAR;
    a.spi = p.spi; a.rows_per_split = p.rows_per_split; a.nseg = p.nseg; a.PS = p.PS; a.gate_mask = mask;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(x3w_ticks), z, sizeof z);
        hipEventRecord(e0, nullptr);
        if (scm_launch_wgrad_x3(a, p, nullptr)) { printf("launch failed\n"); return 1; }
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[8];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(x3w_ticks), sizeof t);
        unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += t[i];
        printf("kernel %.1f us | ticks: prologue %llu  split+store x %llu  split+store dy %llu  barrier1 %llu  (unused) %llu  mfma+fetch %llu  barrier2 %llu  write-out %llu | total %llu\n",
               ms * 1e3, t[0], t[7], t[1], t[2], t[3], t[4], t[5], t[6], tot);
    }
    return 0;
}
