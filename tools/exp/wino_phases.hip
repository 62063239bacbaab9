// Where does a conv3x3_wino workgroup spend its time?  Builds the kernel with its phase clock and prints, for wave 0
// of one mid-grid workgroup of a cin -> 128, 64 x 64 x 64 convolution: ticks of prologue / K loop / epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWINO_TIMING=300 -I scm_gan_amd/csrc -I include tools/exp/wino_phases.hip \
//         -o tools/exp/bin/wino_phases
#ifdef WINO_SRC            // (A/B against another revision of the kernel file: -DWINO_SRC='"bin/wino_old.hip"')
#include WINO_SRC
#else
#include "../../scm_gan_amd/csrc/conv3x3_wino.hip"
#endif
#include "../../scm_gan_amd/csrc/conv3x3_wino16.hip"      // (the entry point dispatches the 16-channel forms)
#include "../../scm_gan_amd/csrc/conv3x3_wino_k16.hip"
#include <cstdio>
#include <algorithm>
#include <map>
#include <vector>

// the two helpers of api.cpp the kernel file needs (the harness does not link the library)
void scm_set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int scm_check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return e != hipSuccess;
}

int main() {
    const int B = 64, H = 64, W = 64, CO = 128;
    for (int C : {128, 256}) {
        const size_t n = (size_t)B * C * H * W, no = (size_t)B * CO * H * W;
        float *x, *y, *w, *bias, *pack;
        hipMalloc(&x, n * 4); hipMalloc(&y, no * 4); hipMalloc(&w, (size_t)C * CO * 9 * 4); hipMalloc(&bias, CO * 4);
        std::vector<float> hx(n), hw((size_t)C * CO * 9);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.1f;
        hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemset(bias, 0, CO * 4);
        hipMalloc(&pack, scm_conv3x3_pack_wino_floats(C, CO) * 4);
        if (scm_pack_conv3x3_weights_wino(w, C, CO, 0, 0, 0, pack, nullptr, nullptr)) { printf("pack failed\n"); return 1; }
        scm_conv3x3_desc d = {};
        d.x1 = x; d.c1 = C; d.wpack = pack; d.cout = CO; d.bias = bias; d.y1 = y; d.co1 = CO;
        d.B = B; d.H = H; d.W = W; d.pad_mode = SCM_PAD_CIRCULAR; d.act = SCM_ACT_LRELU;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it) {
            unsigned long long z[4] = {0};
            hipMemcpyToSymbol(HIP_SYMBOL(wino_ticks), z, sizeof z);
            hipEventRecord(e0, nullptr);
            if (scm_conv3x3_wino(&d, nullptr)) { printf("launch failed\n"); return 1; }
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t[4];
            hipMemcpyFromSymbol(t, HIP_SYMBOL(wino_ticks), sizeof t);
            printf("cin %d: kernel %.1f us | ticks: prologue %llu  loop %llu (%llu per chunk)  epilogue %llu | total %llu\n", C,
                   ms * 1e3, t[0], t[1], t[1] / (C / 8), t[2], t[0] + t[1] + t[2]);
            if (it == 2) {
                // per-CU timeline: lifetime of a workgroup and the idle time between consecutive workgroups of one CU
                static unsigned long long tr[3 * 8192];
                hipMemcpyFromSymbol(tr, HIP_SYMBOL(wino_trace), sizeof tr);
                const int nwg = CO / 128 * (B * (H / 2) * (W / 2) / 32) < 8192 ? CO / 128 * (B * (H / 2) * (W / 2) / 32) : 8192;
                std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
                for (int i = 0; i < nwg; ++i)
                    cu[((tr[3 * i] >> 32) & 15) << 16 | (tr[3 * i] & 0xff00)].push_back({tr[3 * i + 1], tr[3 * i + 2]});
                double life = 0, gap = 0; size_t nl = 0, ng = 0; unsigned long long span = 0;
                for (auto& kv : cu) {
                    auto& v = kv.second;
                    std::sort(v.begin(), v.end());
                    for (size_t i = 0; i < v.size(); ++i) {
                        life += v[i].second - v[i].first; ++nl;
                        if (i) { gap += (double)v[i].first - (double)v[i - 1].second; ++ng; }
                    }
                    span = std::max(span, v.back().second - v.front().first);
                }
                printf("  %zu CUs, %zu workgroups: mean lifetime %.0f ticks, mean gap to the next workgroup of the CU %.0f ticks, "
                       "longest CU span %llu ticks\n", cu.size(), nl, life / nl, ng ? gap / ng : 0.0, span);
            }
        }
        hipFree(x); hipFree(y); hipFree(w); hipFree(bias); hipFree(pack);
    }
    return 0;
}
