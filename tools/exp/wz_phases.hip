// Where does a conv3x3_wgrad_winoz workgroup spend its time?  Phase clock of wave 0 of one workgroup of the ungated
// 128 -> 128, 64 x 64 x 64 weight gradient: ticks of prologue / K loop / epilogue, circular and zero padding, next to the
// launch's wall time (-> the clock the chip holds).  -DWZ_ABL=n removes work from the loop (1: global loads,
// 2: + transforms, 3: + LDS stores, 4: + LDS operand reads).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWZ_TIMING=20 -I scm_gan_amd/csrc -I include tools/exp/wz_phases.hip \
//         -o tools/exp/bin/wz_phases
#include "../../scm_gan_amd/csrc/conv3x3_wgrad_winoz.hip"
#include <cstdio>
#include <vector>

void scm_set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int scm_check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return e != hipSuccess;
}

int main(int argc, char** argv) {
    const int B = 64, H = 64, W = 64, C = 128;
    const size_t n = (size_t)B * C * H * W;
    float *x, *dy, *part, *cs;
    hipMalloc(&x, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&part, (size_t)128 * 9 * C * C * 4);
    hipMalloc(&cs, 128 * C * 4);
    std::vector<float> hx(n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    for (auto& v : hx) v = rnd() * 1e-3f;
    hipMemcpy(dy, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pad = 1; pad >= 0; --pad) {
        WgradKArgs a = {};
        a.x1 = x; a.c1 = C; a.dy = dy; a.gate = nullptr; a.gate_act = 0; a.cout = C; a.part = part; a.csum_part = cs;
        a.B = B; a.H = H; a.W = W; a.pad_mode = pad ? SCM_PAD_CIRCULAR : SCM_PAD_ZERO; a.gate_mask = nullptr;
        a.spi = 1; a.rows_per_split = H / 2; a.ci_groups = 2;
        for (int it = 0; it < 4; ++it) {
            unsigned long long z[4] = {0};
            hipMemcpyToSymbol(HIP_SYMBOL(wz_ticks), z, sizeof z);
            hipEventRecord(e0, nullptr);
            if (scm_launch_wgrad_winoz(a, dim3(B, 4), nullptr)) { printf("launch failed\n"); return 1; }
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t[4];
            hipMemcpyFromSymbol(t, HIP_SYMBOL(wz_ticks), sizeof t);
            const unsigned long long tot = t[0] + t[1] + t[2];
            printf("pad %d: kernel %.1f us | ticks: prologue %llu  loop %llu (%llu per chunk)  epilogue %llu | total %llu -> %.2f GHz if one round\n",
                   pad, ms * 1e3, t[0], t[1], t[1] / 128, t[2], tot, tot / (ms * 1e3) / 1e3);
        }
    }
    return 0;
}
