// Where does a conv3x3_wgrad_wino workgroup spend its time?  Phase clock of wave 0 of one workgroup of the 128 -> 128,
// 64 x 64 x 64 weight gradient (1-bit gate): ticks of prologue / K loop / epilogue.  -DWW_ABL=n removes work from the
// loop (1: global loads, 2: + transforms, 3: + LDS writes, 4: + LDS operand reads); -DWW_UNGATED: no gate (the product's form).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWW_TIMING=20 -I scm_gan_amd/csrc -I include tools/exp/ww_phases.hip \
//         -o tools/exp/bin/ww_phases
#include "../../scm_gan_amd/csrc/conv3x3_wgrad_wino.hip"
#include <cstdio>
#include <vector>

void scm_set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int scm_check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return e != hipSuccess;
}

int main() {
    const int B = 64, H = 64, W = 64, C = 128;
    const size_t n = (size_t)B * C * H * W;
    float *x, *dy, *part, *cs;
    unsigned* mask;
    hipMalloc(&x, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&mask, n / 32 * 4 * 1); hipMalloc(&part, (size_t)128 * 9 * C * C * 4);
    hipMalloc(&cs, 128 * C * 4);
    std::vector<float> hx(n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    for (auto& v : hx) v = rnd() * 1e-3f;
    hipMemcpy(dy, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(mask, 0x5a, n / 32 * 4);
    WgradKArgs a = {};
    a.x1 = x; a.c1 = C; a.dy = dy; a.gate = dy; a.gate_act = SCM_ACT_LRELU; a.cout = C; a.part = part; a.csum_part = cs;
    a.B = B; a.H = H; a.W = W; a.pad_mode = SCM_PAD_CIRCULAR; a.gate_mask = mask;
#ifdef WW_UNGATED        // the form the product runs since the producers gate
    a.gate = nullptr; a.gate_act = 0; a.gate_mask = nullptr;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        unsigned long long z[4] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(ww_ticks), z, sizeof z);
        int ns = 0, spi = 0;
        hipEventRecord(e0, nullptr);
        if (scm_launch_wgrad_wino(a, 128, &ns, &spi, nullptr)) { printf("launch failed\n"); return 1; }
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[4];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(ww_ticks), sizeof t);
        printf("kernel %.1f us (%d splits) | ticks: prologue %llu  loop %llu (%llu per chunk)  epilogue %llu | total %llu\n", ms * 1e3, ns,
               t[0], t[1], t[1] / 128, t[2], t[0] + t[1] + t[2]);
    }
    return 0;
}
