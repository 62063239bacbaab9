// 3x3 stride-1 same-size convolution as an implicit GEMM on the fp32 MFMA pipe
// (v_mfma_f32_32x32x2_f32), NCHW in / NCHW out, zero or circular halo.
//
// Replaces (reference /root/reference): nn.Conv2d / nn.ConvTranspose2d forward
// and input-gradient at models.py:76-102 (Transition), :145-155 (Encoder),
// :274-278 (Decoder).
//
// GEMM view per image b:   D[co][p] = sum_{ci,tap} Wc[co][ci][tap] * Xpad[ci][p + tap]
//   MFMA A operand = weights  (lane -> co, k -> ci)   read from LDS as one vector
//   MFMA B operand = pixels   (lane -> p,  k -> ci)   read from the LDS halo tile
//   D: lane&31 -> pixel, so every accumulator register stores 2 x 128 B rows
//   of the NCHW output.
//
// Workgroup = 4 waves = NP = 128*PT consecutive pixels (flattened y*W+x) of one
// image x one group of CT*32 output channels.  Wave w owns pixels
// [w*32*PT, (w+1)*32*PT) and all CT*32 couts: CT*PT accumulator tiles.
// K loop: chunks of KC=8 input channels; per chunk the halo tile
// [KC][rows+2][W+2] and the weight slab [KC][9][32][CT] are staged in LDS; the
// im2col is only an address offset (ky*(W+2)+kx) into the halo tile.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_args.h"

// One wave per SIMD with the whole 512-register budget (__launch_bounds__(256, 1)):
// CT*PT accumulator tiles (256 VGPRs at 4x4) plus the prefetch registers of the
// next K chunk.  Per chunk:
//   barrier | registers -> LDS (masks, act' gate applied here) | barrier |
//   issue the next chunk's global loads into registers | 36 x CT*PT MFMAs
// Every global load is unconditional on a clamped address (a load under a
// per-lane condition makes hipcc branch around it and serialise the loads).
typedef unsigned u32x4m __attribute__((ext_vector_type(4)));

// GATED: 0 = plain input, 1 = input * act'(gate tensor), 2 = input * leaky-relu slope from the 1-bit
// gate mask (one word per pixel and 32 channels instead of 32 floats).
template <int CT, int PT, int NJ, int GATED, int WPS = 1>
__global__ void __launch_bounds__(256, WPS) conv3x3_mfma_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CW = CT * 32;
    constexpr int WCHUNK = KC * 9 * CW;
    constexpr int W4 = WCHUNK / 4;                 // float4 per weight chunk
    constexpr int WR = (W4 + 255) / 256;           // float4 per thread
    constexpr int PER_CI4 = 9 * CW / 4;
    constexpr int NPW = PT * 32, NP = 4 * NPW;
    constexpr int PLANE = NJ * 256;                // fixed LDS plane pitch: every thread writes NJ slots, no branches
    constexpr int WLDS = WR * 256 * 4;             // weight region padded to whole thread passes
    constexpr int BUF = WLDS + KC * PLANE;         // floats per LDS stage
    constexpr bool DB = (NJ == 4) && WPS == 1;     // two LDS stages when they fit in 160 KB (one WG per CU)
    constexpr int BIAS_F = (DB ? 2 : 1) * BUF;     // bias table (CW floats) behind the stage buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = a.W, H = a.H, HW = H * W;
    const int ROWPL = W + 2;

    const int b = blockIdx.x / a.tiles_per_img;
    const int tile = blockIdx.x - b * a.tiles_per_img;
    const int cg = blockIdx.y;
    const int p0 = tile * NP;
    const int r0 = p0 / W;
    int plast = p0 + NP;
    if (plast > HW) plast = HW;
    const int nrows = (plast - 1) / W - r0 + 3;
    const int nelem = nrows * ROWPL;

    if (tid < CW) {      // bias table for the epilogue (visible after the main loop's barriers)
        const int co = blockIdx.y * CW + tid;
        const int coc = co < a.cout ? co : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        smem[BIAS_F + tid] = v;
    }

    // staging map: LDS element e = tid + 256*j  <-  plane offset goff[j] (clamped);
    // bit j of okmask: a real (or wrapped) pixel; other elements are written as 0
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

    int baddr[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int p = p0 + wave * NPW + pt * 32 + l31;
        if (p > HW - 1) p = HW - 1;
        const int y = p / W;
        const int x = p - y * W;
        baddr[pt] = lh * PLANE + (y - r0) * ROWPL + x;
    }

    f32x16 acc[CT][PT];
#pragma unroll
    for (int m = 0; m < CT; ++m)
#pragma unroll
        for (int n = 0; n < PT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int cin = a.c1 + a.c2;
    const f32x4* wsrc4 = reinterpret_cast<const f32x4*>(a.wpack + (size_t)cg * cin * 9 * CW);
    const int wtotal4 = cin * PER_CI4;
    const int gate_act = GATED ? a.gate_act : 0;

    float pw[WR][4];      // plain floats: an ext_vector array captured by the lambdas lands in scratch
    float px[KC][NJ], pg[GATED == 1 ? KC : 1][GATED == 1 ? NJ : 1];
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
            int gi = ci0 * PER_CI4 + tid + 256 * k;
            gi = gi < wtotal4 ? gi : wtotal4 - 1;
            const f32x4 t = wsrc4[gi];
            pw[k][0] = t[0]; pw[k][1] = t[1]; pw[k][2] = t[2]; pw[k][3] = t[3];
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
    auto store = [&](int ci0, int buf) {
        float* wl = smem + buf * BUF;
        float* xl = wl + WLDS;
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            const int i = tid + 256 * k;
            const int cil = i / PER_CI4;
            const bool wv = ci0 + cil < cin;
            f32x4 v;
            v[0] = wv ? pw[k][0] : 0.f; v[1] = wv ? pw[k][1] : 0.f;
            v[2] = wv ? pw[k][2] : 0.f; v[3] = wv ? pw[k][3] : 0.f;
            reinterpret_cast<f32x4*>(wl)[i] = v;
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

    // MFMA over taps [t0, t1) of the chunk staged in LDS buffer `buf`
    auto mfma_taps = [&](int buf, auto t0c, auto t1c) {
        const float* wl = smem + buf * BUF;
        const float* xl = wl + WLDS;
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
#pragma unroll
        for (int tap = T0; tap < T1; ++tap) {
            const int toff = (tap / 3) * ROWPL + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) {
                const float* wp = wl + ((2 * ks + lh) * 9 + tap) * CW + l31 * CT;
                float av[CT];
                if constexpr (CT == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(wp);
                    av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
                } else if constexpr (CT == 2) {
                    const f32x2 t = *reinterpret_cast<const f32x2*>(wp);
                    av[0] = t[0]; av[1] = t[1];
                } else {
                    av[0] = wp[0];
                }
                float bv[PT];
#pragma unroll
                for (int n = 0; n < PT; ++n) bv[n] = xl[baddr[n] + 2 * ks * PLANE + toff];
#pragma unroll
                for (int m = 0; m < CT; ++m)
#pragma unroll
                    for (int n = 0; n < PT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using I9 = std::integral_constant<int, 9>;

    if constexpr (DB) {
        // Two LDS stages: chunk k+1 is written into the idle stage in the middle of chunk
        // k's MFMAs (its loads were issued a whole chunk earlier), then chunk k+2's loads
        // are issued; one barrier per chunk.
        fetch(0);
        store(0, 0);
        if (KC < cin) fetch(KC);
        __syncthreads();
        int cur = 0;
        for (int ci0 = 0; ci0 < cin; ci0 += KC) {
            mfma_taps(cur, I0{}, I4{});
            if (ci0 + KC < cin) store(ci0 + KC, cur ^ 1);
            if (ci0 + 2 * KC < cin) fetch(ci0 + 2 * KC);
            mfma_taps(cur, I4{}, I9{});
            __syncthreads();
            cur ^= 1;
        }
    } else {
        fetch(0);
        for (int ci0 = 0; ci0 < cin; ci0 += KC) {
            __syncthreads();                 // the previous chunk's MFMAs are done with the tiles
            store(ci0, 0);
            __syncthreads();
            if (ci0 + KC < cin) fetch(ci0 + KC);   // flies under this chunk's MFMAs
            mfma_taps(0, I0{}, I9{});
        }
    }

    // ---- epilogue: scale, bias, per-sample bias, activation, NCHW store.
    // The 16*CT bias values of this lane are loaded first, unconditionally (clamped
    // index) and all in flight together: loading bias[co] inside the guarded store
    // loop cost one serialized L2 round trip per accumulator register (45 us / WG).
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;     // one branch per loop nest, not per value (common.h)
    const float slope = scm_act_slope(a.act);
    const int co2n = a.cout - a.co1;
    // bias + per-sample bias come from the LDS table the prologue filled: 16*CT broadcast LDS
    // reads instead of up to 32*CT global loads per thread
    float badd[CT][16];
    {
        const float* bl = smem + BIAS_F;
#pragma unroll
        for (int m = 0; m < CT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) badd[m][r] = bl[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    if constexpr (PT <= 4 && CT >= 2) {
        // cout > 32 variants (their LDS stage is larger than the four slabs): transpose each wave's 32-channel x NPW-pixel slab through its own
        // LDS region and store float4s (whole row runs, a quarter of the store instructions; the
        // lane -> pixel / register -> channel accumulator layout costs one store instruction per
        // register otherwise).
        constexpr int EP = NPW + 4;
        float* ep = smem + wave * (32 * EP);
        // 16-byte stores need 16-byte aligned rows: H*W % 4 == 0 and aligned base pointers (wave-uniform)
        const bool vec = (HW & 3) == 0 && ((reinterpret_cast<size_t>(a.y1) | reinterpret_cast<size_t>(a.y2) |
                                             reinterpret_cast<size_t>(a.mask_out)) & 15) == 0;
        __syncthreads();                                // every wave is done reading the stage
#pragma unroll
        for (int m = 0; m < CT; ++m) {
            constexpr int Q4 = NPW / 4;                 // float4s per channel row of the slab
            constexpr int RP = 64 / Q4;                 // rows per wave instruction
            // accumulate_y1: the values already in y1, ALL requested before anything waits on one of them (one
            // load-add-store per row in turn costs a memory round trip each)
            f32x4 old[32 / RP];
            if (a.acc_y1 && vec) {
#pragma unroll
                for (int k = 0; k < 32 / RP; ++k) {
                    const int co = cg * CW + m * 32 + RP * k + lane / Q4;
                    const int p = p0 + wave * NPW + 4 * (lane % Q4);
                    const bool take = co < a.co1 && co < a.cout && p + 3 < HW;
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    old[k] = take ? *reinterpret_cast<const f32x4*>(a.y1 + ((size_t)b * a.co1 + co) * HW + p) : zero;
                }
            }
            if (sigmoid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                    for (int n = 0; n < PT; ++n) ep[row * EP + n * 32 + l31] = scm_sigmoid(acc[m][n][r] * sc + badd[m][r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                    for (int n = 0; n < PT; ++n) ep[row * EP + n * 32 + l31] = scm_act_sel(acc[m][n][r] * sc + badd[m][r], slope);
                }
            }
            __syncthreads();
            unsigned bits[4] = {0u, 0u, 0u, 0u};        // sign bits of this lane's rows, per pixel of its float4
#pragma unroll
            for (int k = 0; k < 32 / RP; ++k) {
                const int row = RP * k + lane / Q4;
                const int q = lane % Q4;
                const int co = cg * CW + m * 32 + row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * EP + 4 * q);
                const int p = p0 + wave * NPW + 4 * q;
#pragma unroll
                for (int i = 0; i < 4; ++i) bits[i] |= (v[i] > 0.f ? 1u : 0u) << row;
                if (co < a.cout) {
                    float* dst = (co < a.co1 ? a.y1 + ((size_t)b * a.co1 + co) * HW
                                             : a.y2 + ((size_t)b * co2n + (co - a.co1)) * HW) + p;
                    const bool add = a.acc_y1 && co < a.co1;
                    if (vec && p + 3 < HW) {
                        f32x4 o = v;
                        if (add) o += old[k];
                        *reinterpret_cast<f32x4*>(dst) = o;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (p + i < HW) dst[i] = add ? dst[i] + v[i] : v[i];
                    }
                }
            }
            if (a.mask_out) {                           // wave-uniform: the 1-bit leaky-relu gate of these 32 channels
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int o = Q4; o < 64; o <<= 1) bits[i] |= __shfl_xor(bits[i], o);
                }
                const int q = lane % Q4;
                const int p = p0 + wave * NPW + 4 * q;
                const int nw = (a.cout + 31) >> 5, w = cg * CT + m;
                if (lane < Q4 && w < nw) {
                    unsigned* dst = a.mask_out + ((size_t)b * nw + w) * HW + p;
                    if (vec && p + 3 < HW) {
                        u32x4m t;
                        t[0] = bits[0]; t[1] = bits[1]; t[2] = bits[2]; t[3] = bits[3];
                        *reinterpret_cast<u32x4m*>(dst) = t;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (p + i < HW) dst[i] = bits[i];
                    }
                }
            }
            if (m + 1 < CT) __syncthreads();
        }
        return;
    }
    const int pbase = p0 + wave * NPW + l31;
    auto write_out = [&](auto sigc) {
        constexpr bool SIG = decltype(sigc)::value;
#pragma unroll
        for (int m = 0; m < CT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * CW + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.cout) {
                    float* dst = co < a.co1 ? a.y1 + ((size_t)b * a.co1 + co) * HW
                                            : a.y2 + ((size_t)b * co2n + (co - a.co1)) * HW;
#pragma unroll
                    for (int n = 0; n < PT; ++n) {
                        const int p = pbase + n * 32;
                        if (p < HW) {
                            const float t = acc[m][n][r] * sc + badd[m][r];
                            const float v = SIG ? scm_sigmoid(t) : scm_act_sel(t, slope);
                            dst[p] = (a.acc_y1 && co < a.co1) ? dst[p] + v : v;
                        }
                    }
                }
            }
        }
    };
    if (sigmoid) write_out(std::true_type{});
    else write_out(std::false_type{});
}

// ------------------------------------------------------------------ pack
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src,
                                    int wct, int wc0, float* __restrict__ fwd, float* __restrict__ dgr) {
    // element accessor of the equivalent conv weight Wc[co][ci][t]; the stored
    // tensor has wct input channels, of which [wc0, wc0+cin) are used
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (fwd) {
        const int ct = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
        const int cw = ct * 32;
        const int ng = (cout + cw - 1) / cw;
        const int total = ng * cin * 9 * cw;
        if (idx < total) {
            int e = idx;
            const int t = e % ct; e /= ct;
            const int l = e % 32; e /= 32;
            const int tap = e % 9; e /= 9;
            const int ci = e % cin; e /= cin;
            const int g = e;
            const int co = g * cw + t * 32 + l;
            fwd[idx] = co < cout ? Wc(co, ci, tap) : 0.f;
        }
    }
    if (dgr) {
        // dgrad conv: input channels = cout, output channels = cin, taps flipped
        const int ct = cin <= 32 ? 1 : (cin <= 64 ? 2 : 4);
        const int cw = ct * 32;
        const int ng = (cin + cw - 1) / cw;
        const int total = ng * cout * 9 * cw;
        if (idx < total) {
            int e = idx;
            const int t = e % ct; e /= ct;
            const int l = e % 32; e /= 32;
            const int tap = e % 9; e /= 9;
            const int co = e % cout; e /= cout;
            const int g = e;
            const int ci = g * cw + t * 32 + l;
            dgr[idx] = ci < cin ? Wc(co, ci, 8 - tap) : 0.f;
        }
    }
}

// ------------------------------------------------------------------ host
static inline int conv_ct(int cout) { return cout <= 32 ? 1 : (cout <= 64 ? 2 : 4); }

extern "C" size_t scm_conv3x3_pack_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    const int cw = conv_ct(cout) * 32;
    const int ng = (cout + cw - 1) / cw;
    return (size_t)ng * cin * 9 * cw;
}

extern "C" int scm_pack_conv3x3_weights(const float* w, int cin, int cout, int transposed_src, int w_cin_total,
                                        int w_ci0, float* fwd_pack, float* dgrad_pack, void* stream) {
    SCM_REQUIRE(w && cin > 0 && cout > 0, "scm_pack_conv3x3_weights: bad arguments (cin=%d cout=%d)", cin, cout);
    if (w_cin_total == 0) w_cin_total = cin;
    SCM_REQUIRE(w_ci0 >= 0 && w_ci0 + cin <= w_cin_total, "scm_pack_conv3x3_weights: bad channel slice");
    SCM_REQUIRE(fwd_pack || dgrad_pack, "scm_pack_conv3x3_weights: no output requested");
    size_t n = 0;
    if (fwd_pack) n = scm_conv3x3_pack_floats(cin, cout);
    if (dgrad_pack) {
        const size_t m = scm_conv3x3_pack_floats(cout, cin);
        if (m > n) n = m;
    }
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, fwd_pack, dgrad_pack);
    SCM_LAUNCH_CHECK("pack_conv3x3_kernel");
    return SCM_OK;
}

template <int CT, int PT, int NJ, int GATED, int WPS = 1>
static int launch_conv_nj(const ConvKArgs& a, int groups, hipStream_t s) {
    constexpr int W4 = KC * 9 * CT * 32 / 4, WR = (W4 + 255) / 256;
    const size_t lds = ((size_t)WR * 256 * 4 + (size_t)KC * NJ * 256) * sizeof(float) * ((NJ == 4 && WPS == 1) ? 2 : 1) +
                       CT * 32 * sizeof(float);       // + bias table
    auto kern = conv3x3_mfma_kernel<CT, PT, NJ, GATED, WPS>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    if (lds > 160 * 1024) {
        scm_set_error("scm_conv3x3: halo tile of %zu bytes exceeds the LDS", lds);
        return SCM_EUNSUPPORTED;
    }
    dim3 grid(a.B * a.tiles_per_img, groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_mfma_kernel");
    return SCM_OK;
}

// gate mode of a launch: 0 none, 1 gate tensor, 2 1-bit mask (leaky-relu only)
static inline int gate_mode(const ConvKArgs& a) {
    if (a.gate_act == 0) return 0;
    return (a.mask_in && a.gate_act == SCM_ACT_LRELU && a.c2 == 0) ? 2 : 1;
}
#define SCM_LAUNCH_G(CT_, PT_, NJ_, ...)                                                      \
    (g == 2 ? launch_conv_nj<CT_, PT_, NJ_, 2, ##__VA_ARGS__>(a, groups, s)                   \
            : (g == 1 ? launch_conv_nj<CT_, PT_, NJ_, 1, ##__VA_ARGS__>(a, groups, s)         \
                      : launch_conv_nj<CT_, PT_, NJ_, 0, ##__VA_ARGS__>(a, groups, s)))

template <int CT, int PT>
static int launch_conv(const ConvKArgs& k, int groups, hipStream_t s) {
    constexpr int NP = 128 * PT;
    ConvKArgs a = k;
    const int HW = a.H * a.W;
    a.tiles_per_img = (HW + NP - 1) / NP;
    int rows = (NP % a.W == 0) ? NP / a.W : (NP + a.W - 2) / a.W + 1;
    if (rows > a.H) rows = a.H;
    a.plane = (rows + 2) * (a.W + 2);
    const int g = gate_mode(a);
    if constexpr (CT == 4 && PT == 1) {
        // 128-pixel tiles for grids that would leave CUs idle at 256 pixels (small batches:
        // the planner's 25 beams, BASELINE configs[0] B = 8)
        if (a.plane <= 1 * 256) return SCM_LAUNCH_G(4, 1, 1, 2);
        if (a.plane <= 2 * 256) return SCM_LAUNCH_G(4, 1, 2, 2);
        scm_set_error("scm_conv3x3: 128-pixel variant needs a halo tile <= 512 elements");
        return SCM_EUNSUPPORTED;
    }
    if constexpr (CT == 4 && PT == 2) {
        if (a.plane <= 2 * 256) return SCM_LAUNCH_G(4, 2, 2, 2);
        if (a.plane <= 3 * 256) return SCM_LAUNCH_G(4, 2, 3, 2);
        scm_set_error("scm_conv3x3: 2-WG/CU variant needs a halo tile <= 768 elements");
        return SCM_EUNSUPPORTED;
    }
    if (a.plane <= 4 * 256) return SCM_LAUNCH_G(CT, PT, 4);
    if (a.plane <= MAXJ * 256) return SCM_LAUNCH_G(CT, PT, MAXJ);
    scm_set_error("scm_conv3x3: image width %d too large for the halo staging map", a.W);
    return SCM_EUNSUPPORTED;
}

extern "C" int scm_conv3x3(const scm_conv3x3_desc* d, void* stream) {
    SCM_REQUIRE(d != nullptr, "scm_conv3x3: null descriptor");
    SCM_REQUIRE(!d->x_half && !d->y_half, "scm_conv3x3: IEEE-half tensors (x_half / y_half) are taken by scm_conv3x3_h16 only");
    SCM_REQUIRE(d->x1 && d->c1 > 0, "scm_conv3x3: x1/c1 required");
    SCM_REQUIRE((d->x2 != nullptr) == (d->c2 > 0) && d->c2 >= 0, "scm_conv3x3: x2/c2 mismatch");
    SCM_REQUIRE(d->wpack && d->cout > 0, "scm_conv3x3: wpack/cout required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "scm_conv3x3: bad shape B=%d H=%d W=%d", d->B, d->H, d->W);
    SCM_REQUIRE(d->y1 && d->co1 > 0 && d->co1 <= d->cout, "scm_conv3x3: y1/co1 invalid");
    SCM_REQUIRE((d->co1 == d->cout) || d->y2, "scm_conv3x3: y2 required when co1 < cout");
    SCM_REQUIRE(d->pad_mode == SCM_PAD_ZERO || d->pad_mode == SCM_PAD_CIRCULAR, "scm_conv3x3: bad pad_mode %d",
                d->pad_mode);
    SCM_REQUIRE(d->act >= 0 && d->act <= 2 && d->gate_act >= 0 && d->gate_act <= 2, "scm_conv3x3: bad act");
    if (d->gate_act) {
        SCM_REQUIRE(d->gate1 && (d->c2 == 0 || d->gate2), "scm_conv3x3: gate tensors required with gate_act");
    }
    ConvKArgs k;
    k.x1 = d->x1; k.x2 = d->x2; k.c1 = d->c1; k.c2 = d->c2;
    k.gate1 = d->gate1; k.gate2 = d->gate2; k.gate_act = d->gate_act;
    k.wpack = d->wpack; k.cout = d->cout;
    k.scale = d->scale; k.bias = d->bias; k.sbias = d->sbias;
    k.y1 = d->y1; k.y2 = d->y2; k.co1 = d->co1;
    k.B = d->B; k.H = d->H; k.W = d->W; k.pad_mode = d->pad_mode; k.act = d->act;
    k.tiles_per_img = 0; k.plane = 0;
    k.mask_in = d->gate_mask_in;
    k.acc_y1 = d->accumulate_y1;
    k.ymask1 = k.ymask2 = nullptr;
    SCM_REQUIRE(!d->gate_mask_y1 && !d->gate_mask_y2, "scm_conv3x3: output gates (gate_mask_y1 / _y2) are applied by scm_conv3x3_wino only");
    k.mask_out = (d->act == SCM_ACT_LRELU && d->cout > 32) ? d->gate_mask_out : nullptr;
    SCM_REQUIRE(!d->gate_mask_out || k.mask_out, "scm_conv3x3: gate_mask_out needs act == LRELU and cout > 32");
    if (d->cout <= 16) return scm_launch_conv3x3_m16(k, (hipStream_t)stream);
    const int ct = conv_ct(d->cout);
    const int groups = (d->cout + ct * 32 - 1) / (ct * 32);
    hipStream_t s = (hipStream_t)stream;
    if (ct == 4) {
        // Two workgroups per CU (256-pixel tiles, <= 256 VGPRs) let one workgroup's epilogue
        // stores and staging bubbles hide under the other's MFMAs: +4..6 % over the 512-pixel
        // one-wave-per-SIMD tile when the halo tile fits 512 LDS slots (W = 64 images).
        static const int one_wg = getenv("SCM_CONV_1WG") ? atoi(getenv("SCM_CONV_1WG")) : 0;
        int rows = (256 % d->W == 0) ? 256 / d->W : (256 + d->W - 2) / d->W + 1;
        if (rows > d->H) rows = d->H;
        if (!one_wg && (rows + 2) * (d->W + 2) <= 768) {
            // fewer 256-pixel workgroups than the chip holds at two per CU: halve the tile
            static const int no_small = getenv("SCM_CONV_NO128") ? atoi(getenv("SCM_CONV_NO128")) : 0;
            const long wgs = (long)d->B * ((d->H * d->W + 255) / 256) * groups;
            int rows1 = (128 % d->W == 0) ? 128 / d->W : (128 + d->W - 2) / d->W + 1;
            if (rows1 > d->H) rows1 = d->H;
            if (!no_small && wgs < 512 && (rows1 + 2) * (d->W + 2) <= 512) return launch_conv<4, 1>(k, groups, s);
            return launch_conv<4, 2>(k, groups, s);
        }
        return launch_conv<4, 4>(k, groups, s);
    }
    if (ct == 2) return launch_conv<2, 4>(k, groups, s);
    return launch_conv<1, 8>(k, groups, s);
}
