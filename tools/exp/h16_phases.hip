// Where does a conv3x3_h16 workgroup spend its time?  Builds the kernel with its phase clock and prints, for wave 0 of one
// mid-grid workgroup of a 128 -> 128, 64 x 64 x 64 fp16-operand convolution: ticks per phase (summed over the 8 chunks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DH16_TIMING=300 -Wno-unused-result -I scm_gan_amd/csrc -I include \
//         tools/exp/h16_phases.hip -o tools/exp/bin/h16_phases
#include "../../scm_gan_amd/csrc/conv3x3_h16.hip"
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
    float *x, *y, *w, *bias;
    void* pack;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, (size_t)C * C * 9 * 4); hipMalloc(&bias, C * 4);
    std::vector<float> hx(n), hw((size_t)C * C * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, C * 4);
    hipMalloc(&pack, scm_conv3x3_pack_h16_bytes(C, C));
    if (scm_pack_conv3x3_weights_h16(w, C, C, 0, 0, 0, pack, SCM_PREC_F16, nullptr, SCM_PREC_BF16, nullptr)) { printf("pack failed\n"); return 1; }
    scm_conv3x3_desc d = {};
    d.x1 = x; d.c1 = C; d.wpack = (const float*)pack; d.cout = C; d.bias = bias; d.y1 = y; d.co1 = C;
    d.B = B; d.H = H; d.W = W; d.pad_mode = SCM_PAD_CIRCULAR; d.act = SCM_ACT_LRELU;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(h16_ticks), z, sizeof z);
        hipEventRecord(e0, nullptr);
        if (scm_conv3x3_h16(&d, SCM_PREC_F16, nullptr)) { printf("launch failed\n"); return 1; }
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[8];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(h16_ticks), sizeof t);
        unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += t[i];
        printf("kernel %.1f us | ticks: prologue %llu  mfma taps 0-3 %llu  store next %llu  mfma taps 4-8 + loads %llu  barrier %llu  epilogue %llu | total %llu\n",
               ms * 1e3, t[0], t[1], t[2], t[3], t[4], t[5], tot);
    }
    return 0;
}
