// 3x3 same-size convolution for cout <= 16 on v_mfma_f32_16x16x4_f32.
//
// The 16-channel latent z makes several layers of the path 16 outputs wide:
// Transition.conv6 (256->16) and Encoder.conv4 (128->16) forward, the input
// gradients of Transition.conv1 (128->16), Decoder.conv1 (64->16) and
// RewardPredictor.conv1 (32->16), and the Decoder's factor-summed conv2 (64->C).
// On the 32x32x2 tile of conv3x3_mfma.hip half (or 29/32) of every MFMA is padding;
// the 16x16x4 instruction has the same MAC rate and a 16-row tile.
//
//   MFMA A = weights  A[i = co (lane&15)][k = ci (lane>>4)]      4 input channels / instruction
//   MFMA B = pixels   B[k = ci (lane>>4)][j = pixel (lane&15)]
//   D[i][j]: lane&15 -> pixel, row = 4*(lane>>4) + reg -> co: 64-byte row segments of NCHW.
//
// Workgroup = 4 waves x 8 tiles x 16 pixels = 512 consecutive pixels of one image,
// <= 128 VGPRs so several workgroups share a CU and hide each other's staging.
// Same staging as the 32-wide kernel: K chunks of 8 channels, halo tile in LDS with
// the im2col as an address offset, register prefetch of the next chunk, unconditional
// clamped loads with select masks, act' gate applied on load for the backward pass.
#include "conv_args.h"
#include <type_traits>

#define M16_PT 8     // 16-pixel tiles per wave

// GATED: 0 plain input, 1 input * act'(gate tensor), 2 input * leaky-relu slope from the 1-bit gate mask
template <int NJ, int GATED>
__global__ void __launch_bounds__(256, 2) conv3x3_m16_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = NJ * 256 + 16;       // +16: the four ci groups of a wave land on different banks
    constexpr int WCH = KC * 9 * 16;            // weights of a chunk: [ci][tap][16 co]
    constexpr int WR = (WCH + 255) / 256;
    constexpr int NPW = M16_PT * 16, NP = 4 * NPW;
    float* wl = smem;                           // [KC][9][16]  (padded to WR*256)
    float* xl = smem + WR * 256;                // [KC][PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int W = a.W, H = a.H, HW = H * W;
    const int ROWPL = W + 2;

    const int b = blockIdx.x / a.tiles_per_img;
    const int tile = blockIdx.x - b * a.tiles_per_img;
    const int p0 = tile * NP;
    const int r0 = p0 / W;
    int plast = p0 + NP;
    if (plast > HW) plast = HW;
    const int nrows = (plast - 1) / W - r0 + 3;
    const int nelem = nrows * ROWPL;

    int goff[NJ];
    unsigned okmask = 0u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = tid + 256 * j;
        int g = 0;
        if (e < nelem) {
            const int r = e / ROWPL;
            const int c = e - r * ROWPL;
            int iy = r0 - 1 + r, ix = c - 1;
            bool ok = true;
            if (a.pad_mode == SCM_PAD_CIRCULAR) {
                iy = iy < 0 ? iy + H : (iy >= H ? iy - H : iy);
                ix = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
            } else {
                ok = (iy >= 0 && iy < H && ix >= 0 && ix < W);
            }
            iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            g = iy * W + ix;
            if (ok) okmask |= 1u << j;
        }
        goff[j] = g;
    }

    int baddr[M16_PT];
#pragma unroll
    for (int t = 0; t < M16_PT; ++t) {
        int p = p0 + wave * NPW + t * 16 + l15;
        if (p > HW - 1) p = HW - 1;
        const int y = p / W;
        const int x = p - y * W;
        baddr[t] = lq * PLANE + (y - r0) * ROWPL + x;
    }

    f32x4 acc[M16_PT];
#pragma unroll
    for (int t = 0; t < M16_PT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cin = a.c1 + a.c2;
    const int gate_act = GATED ? a.gate_act : 0;
    // the pack of a cout <= 32 layer is [ci][tap][32 lanes]: this kernel uses lanes 0..15
    const float* wsrc = a.wpack;
    const int wmax = cin * 9 * 32 - 1;

    float pw[WR], px[KC][NJ], pg[GATED == 1 ? KC : 1][GATED == 1 ? NJ : 1];
    unsigned pm[GATED == 2 ? NJ : 1];
    const int mask_nw = (a.c1 + 31) >> 5;
    // per-image bases of the two concatenated inputs, indexed by the CONCATENATED channel; pinned in SGPRs (left to
    // itself the compiler selects between the kernarg slots and re-loads the pointer, with a wait, per channel) and
    // spelled out as global address space (behind the asm it would otherwise fall back to flat loads)
    typedef const __attribute__((address_space(1))) float* gfp;
    gfp xb1 = (gfp)(a.x1 + (size_t)b * a.c1 * HW);
    gfp xb2 = a.c2 ? (gfp)(a.x2 + ((size_t)b * a.c2 - a.c1) * HW) : xb1;
    gfp gb1 = GATED == 1 ? (gfp)(a.gate1 + (size_t)b * a.c1 * HW) : nullptr;
    gfp gb2 = (GATED == 1 && a.c2) ? (gfp)(a.gate2 + ((size_t)b * a.c2 - a.c1) * HW) : gb1;
    asm volatile("" : "+s"(xb1), "+s"(xb2));
    if constexpr (GATED == 1) asm volatile("" : "+s"(gb1), "+s"(gb2));
    auto fetch = [&](int ci0) {
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            const int i = tid + 256 * k;                 // (cil, tap, co) flattened, co fastest
            const int co = i & 15, ct = i >> 4;          // ct = cil * 9 + tap
            int gi = (ci0 * 9 + ct) * 32 + co;
            gi = gi < wmax ? gi : wmax;
            pw[k] = wsrc[gi];
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int ci = ci0 + c;
            const int cc = ci < cin ? ci : 0;
            const bool first = cc < a.c1;
            const gfp src = (first ? xb1 : xb2) + (size_t)cc * HW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) px[c][j] = src[goff[j]];
            if constexpr (GATED == 1) {
                const gfp gsrc = (first ? gb1 : gb2) + (size_t)cc * HW;
#pragma unroll
                for (int j = 0; j < NJ; ++j) pg[c][j] = gsrc[goff[j]];
            }
        }
        if constexpr (GATED == 2) {
            const int w = (ci0 < cin ? ci0 : 0) >> 5;
            const unsigned* mrow = a.mask_in + ((size_t)b * mask_nw + w) * HW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) pm[j] = mrow[goff[j]];
        }
    };
    auto store = [&](int ci0) {
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            const int i = tid + 256 * k;
            const int cil = (i >> 4) / 9;
            wl[i] = (ci0 + cil < cin && i < WCH) ? pw[k] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const bool cvalid = ci0 + c < cin;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v = px[c][j];
                if constexpr (GATED == 1) v *= scm_act_grad_from_out(pg[c][j], gate_act);
                if constexpr (GATED == 2) v *= ((pm[j] >> ((ci0 & 31) + c)) & 1u) ? 1.f : 0.01f;
                xl[c * PLANE + tid + 256 * j] = (cvalid && (okmask & (1u << j))) ? v : 0.f;
            }
        }
    };

    fetch(0);
    for (int ci0 = 0; ci0 < cin; ci0 += KC) {
        __syncthreads();
        store(ci0);
        __syncthreads();
        if (ci0 + KC < cin) fetch(ci0 + KC);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * ROWPL + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const float av = wl[((4 * ks + lq) * 9 + tap) * 16 + l15];
                float bv[M16_PT];
#pragma unroll
                for (int t = 0; t < M16_PT; ++t) bv[t] = xl[baddr[t] + 4 * ks * PLANE + toff];
#pragma unroll
                for (int t = 0; t < M16_PT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;
    const float slope = scm_act_slope(a.act);
    float badd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * lq + r;
        const int coc = co < a.cout ? co : a.cout - 1;
        badd[r] = 0.f;
        if (a.bias) badd[r] += a.bias[coc];
        if (a.sbias) badd[r] += a.sbias[(size_t)b * a.cout + coc];
    }
    const int co2n = a.cout - a.co1;
    // accumulate_y1: the values already in y1, all requested before anything waits on one of them
    float old[4][M16_PT];
    if (a.acc_y1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * lq + r;
            const float* src = a.y1 + ((size_t)b * a.co1 + (co < a.co1 ? co : 0)) * HW;
#pragma unroll
            for (int t = 0; t < M16_PT; ++t) {
                const int p = p0 + wave * NPW + t * 16 + l15;
                old[r][t] = (co < a.co1 && co < a.cout && p < HW) ? src[p] : 0.f;
            }
        }
    }
    auto write_out = [&](auto sigc) {
        constexpr bool SIG = decltype(sigc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * lq + r;
            if (co < a.cout) {
                float* dst = co < a.co1 ? a.y1 + ((size_t)b * a.co1 + co) * HW
                                        : a.y2 + ((size_t)b * co2n + (co - a.co1)) * HW;
#pragma unroll
                for (int t = 0; t < M16_PT; ++t) {
                    const int p = p0 + wave * NPW + t * 16 + l15;
                    if (p < HW) {
                        const float u = acc[t][r] * sc + badd[r];
                        const float v = SIG ? scm_sigmoid(u) : scm_act_sel(u, slope);
                        dst[p] = (a.acc_y1 && co < a.co1) ? old[r][t] + v : v;
                    }
                }
            }
        }
    };
    if (sigmoid) write_out(std::true_type{});
    else write_out(std::false_type{});
}

template <int NJ, int GATED>
static int launch_m16(const ConvKArgs& a, hipStream_t s) {
    constexpr int WR = (KC * 9 * 16 + 255) / 256;
    const size_t lds = ((size_t)WR * 256 + (size_t)KC * (NJ * 256 + 16)) * sizeof(float);
    auto kern = conv3x3_m16_kernel<NJ, GATED>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_img), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_m16_kernel");
    return SCM_OK;
}

int scm_launch_conv3x3_m16(const ConvKArgs& k, hipStream_t s) {
    constexpr int NP = 4 * M16_PT * 16;
    ConvKArgs a = k;
    const int HW = a.H * a.W;
    a.tiles_per_img = (HW + NP - 1) / NP;
    int rows = (NP % a.W == 0) ? NP / a.W : (NP + a.W - 2) / a.W + 1;
    if (rows > a.H) rows = a.H;
    a.plane = (rows + 2) * (a.W + 2);
    const int g = a.gate_act == 0 ? 0 : ((a.mask_in && a.gate_act == SCM_ACT_LRELU && a.c2 == 0) ? 2 : 1);
#define SCM_M16_G(NJ_) (g == 2 ? launch_m16<NJ_, 2>(a, s) : (g == 1 ? launch_m16<NJ_, 1>(a, s) : launch_m16<NJ_, 0>(a, s)))
    if (a.plane <= 3 * 256) return SCM_M16_G(3);
    if (a.plane <= 5 * 256) return SCM_M16_G(5);
    if (a.plane <= 8 * 256) return SCM_M16_G(8);
#undef SCM_M16_G
    scm_set_error("scm_conv3x3: image width %d too large for the halo staging map", a.W);
    return SCM_EUNSUPPORTED;
}
