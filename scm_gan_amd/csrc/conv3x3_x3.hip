// fp32 convolution on the bf16 MFMA pipe: every fp32 operand is split EXACTLY into three bf16 values
// (x = hi + mid + lo, 8 + 8 + 8 significand bits), and a product a*b is evaluated as the six partial
// products of total order <= 2,
//     a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi,
// each exact in the fp32 accumulator's input (8 x 8 bits), summed in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped
// terms (a_mid b_lo, a_lo b_mid, a_lo b_lo) are <= 2^-23 of the product: one fp32 rounding.  Against a float64
// reference the result is as close as the native v_mfma_f32_32x32x2_f32 chain (tests/test_gpu_x3_mode.py measures
// both); it is NOT a reduced-precision mode.  Six bf16 MFMAs cost 6/16 of one f32 MFMA step, so the MFMA-bound
// 128 -> 128 convolutions of the transition run 1.6-1.7x faster (348 / 381 us against 595 / 608: the split, the LDS
// stores, the barriers and the epilogue do not shrink with the MFMAs).  Opt-in (ops.set_compute_precision('fp32x3')).
//
// Same implicit GEMM, tensors, epilogue and gate handling as conv3x3_h16.hip (which see); differences:
//   * three LDS images of the halo tile and of the weight chunk (hi / mid / lo), split while staging;
//   * 512 pixels x up to 128 output channels per workgroup, one workgroup per CU.  The three images of a 16-channel
//     halo tile (74 KB) stay in LDS while the weights pass through in two 64-channel halves (55 KB each): the tile is
//     fetched and split once per 128 output channels, 135 KB of LDS in all.  Launches that would leave CUs idle use
//     128-pixel workgroups instead (PT = 1);
//   * the next stage's global loads are dealt out over the taps of the running one, LDS operands are read one step
//     ahead of the MFMAs that use them (what the phase clock below showed).
#include "conv_args.h"
#include <type_traits>
#include <stdlib.h>

#define XKC 16
#define X3_PT 4
#define X3_NJ 3

typedef unsigned u32x4x __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8x __attribute__((ext_vector_type(8)));
typedef __bf16 b16x2x __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4x a, u32x4x b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8x, a), __builtin_bit_cast(b16x8x, b), c, 0, 0, 0);
}
// (hi, mid, lo) of two floats, each packed as two bf16 in one dword
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    b16x2x h = {(__bf16)x0, (__bf16)x1};
    const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
    b16x2x m = {(__bf16)r0, (__bf16)r1};
    const float s0 = r0 - (float)m[0], s1 = r1 - (float)m[1];
    b16x2x l = {(__bf16)s0, (__bf16)s1};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

// Phase clock for tools/exp/x3_phases.hip (compiled in only with -DX3_TIMING): shader-clock ticks per phase,
// summed over the chunks of one wave.
#ifdef X3_TIMING
__device__ unsigned long long x3_ticks[8];
#define X3_TICK(i)                                      \
    do {                                                \
        const unsigned long long t_ = clock64();        \
        t_acc[i] += t_ - t_last;                        \
        t_last = t_;                                    \
    } while (0)
#else
#define X3_TICK(i)
#endif

// GATED: 0 plain input, 1 input * act'(gate tensor), 2 input * leaky-relu slope from the 1-bit gate mask
// CT: 32-channel row tiles per weight half (1 or 2); NH: halves per workgroup (1 or 2)
// PT: 32-pixel tiles per wave (4: 512-pixel workgroups; 1: 128-pixel workgroups for launches that would leave CUs idle)
// NJ: halo-tile slots per thread and channel (256 * NJ staged elements)
template <int CT, int NH, int GATED, int PT = X3_PT, int NJ = X3_NJ>
__global__ void __launch_bounds__(256, 1) conv3x3_x3_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4x smx[];
    constexpr int CW = NH * CT * 32;               // output channels of a workgroup
    constexpr int WU = 9 * 2 * CT * 32;            // 16-byte units of one component of a weight chunk half: [tap][h][m][lane]
    constexpr int WR = (WU + 255) / 256;
    constexpr int NPW = PT * 32, NP = 4 * NPW;
    constexpr int PLANE = NJ * 256;                // units per channel half
    constexpr int WCOMP = WR * 256, XCOMP = 2 * PLANE;
    constexpr int STAGE = 3 * (WCOMP + XCOMP);
    constexpr int EP_U = 4 * 32 * (NPW + 4) / 4;
    constexpr int BIAS_U = STAGE > EP_U ? STAGE : EP_U;

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

    if (tid < CW) {
        const int co = cg * CW + tid;
        const int coc = co < a.cout ? co : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        reinterpret_cast<float*>(smx + BIAS_U)[tid] = v;
    }

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

    f32x16 acc[NH * CT][PT];
#pragma unroll
    for (int m = 0; m < NH * CT; ++m)
#pragma unroll
        for (int n = 0; n < PT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int cin = a.c1 + a.c2;
    const int nchunks = (cin + XKC - 1) / XKC;
    // pack: [64- (or 32-) channel group][chunk][component][WU units]; this workgroup's halves are groups cg*NH + h
    const u32x4x* wsrc = reinterpret_cast<const u32x4x*>(a.wpack) + (size_t)cg * NH * nchunks * 3 * WU;
    const int gate_act = GATED ? a.gate_act : 0;

    u32x4x pw[3][WR];
    float px[XKC][NJ], pg[GATED == 1 ? XKC : 1][GATED == 1 ? NJ : 1];
    unsigned pm[GATED == 2 ? NJ : 1];
    const int mask_nw = (a.c1 + 31) >> 5;

    // weight load i of the 3 * WR that make a chunk half; fetch_w issues all of them
    auto fetch_w_part = [&](int ci0, int half, int i) {
        const u32x4x* wc = wsrc + ((size_t)half * nchunks + ci0 / XKC) * 3 * WU;
        const int comp = i / WR, k = i % WR;
        int gi = tid + 256 * k;
        gi = gi < WU ? gi : WU - 1;
        pw[comp][k] = wc[comp * WU + gi];
    };
    auto fetch_w = [&](int ci0, int half) {
#pragma unroll
        for (int i = 0; i < 3 * WR; ++i) fetch_w_part(ci0, half, i);
    };
    // per-image bases of the two concatenated inputs, indexed by the CONCATENATED channel; pinned in SGPRs (left to
    // itself the compiler selects between the kernarg slots and re-loads the pointer, with a wait, per channel)
    // (global address space spelled out: behind the asm the compiler would otherwise fall back to flat loads, which
    // also count on lgkmcnt and so serialise with the LDS reads of the MFMA loop)
    typedef const __attribute__((address_space(1))) float* gfp;
    gfp xb1 = (gfp)(a.x1 + (size_t)b * a.c1 * HW);
    gfp xb2 = a.c2 ? (gfp)(a.x2 + ((size_t)b * a.c2 - a.c1) * HW) : xb1;
    gfp gb1 = GATED == 1 ? (gfp)(a.gate1 + (size_t)b * a.c1 * HW) : nullptr;
    gfp gb2 = (GATED == 1 && a.c2) ? (gfp)(a.gate2 + ((size_t)b * a.c2 - a.c1) * HW) : gb1;
    asm volatile("" : "+s"(xb1), "+s"(xb2));
    if constexpr (GATED == 1) asm volatile("" : "+s"(gb1), "+s"(gb2));
    auto fetch_x_ch = [&](int ci0, int c) {
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
    };
    auto fetch_mask = [&](int ci0) {
        if constexpr (GATED == 2) {
            const int w = (ci0 < cin ? ci0 : 0) >> 5;
            const unsigned* mrow = a.mask_in + ((size_t)b * mask_nw + w) * HW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) pm[j] = mrow[goff[j]];
        }
    };
    auto fetch_x = [&](int ci0) {
#pragma unroll
        for (int c = 0; c < XKC; ++c) fetch_x_ch(ci0, c);
        fetch_mask(ci0);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int comp = 0; comp < 3; ++comp)
#pragma unroll
            for (int k = 0; k < WR; ++k) smx[comp * WCOMP + tid + 256 * k] = pw[comp][k];
    };
    auto store_x = [&](int ci0) {
        u32x4x* xl = smx + 3 * WCOMP;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool ok = (okmask >> j) & 1u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = 8 * h + i;
                    float t = px[c][j];
                    if constexpr (GATED == 1) t *= scm_act_grad_from_out(pg[c][j], gate_act);
                    if constexpr (GATED == 2) t *= ((pm[j] >> ((ci0 & 31) + c)) & 1u) ? 1.f : 0.01f;
                    v[i] = (ok && ci0 + c < cin) ? t : 0.f;
                }
                u32x4x uh, um, ul;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned hi, mid, lo;
                    split3_pair(v[2 * q], v[2 * q + 1], hi, mid, lo);
                    uh[q] = hi; um[q] = mid; ul[q] = lo;
                }
                const int o = h * PLANE + tid + 256 * j;
                xl[o] = uh;
                xl[XCOMP + o] = um;
                xl[2 * XCOMP + o] = ul;
            }
        }
    };

    auto mfma_half = [&](int half, auto&& after_tap) {
        const u32x4x* wl = smx;
        const u32x4x* xl = smx + 3 * WCOMP;
        // Operands are read ahead of the products that use them (one wave per SIMD: nothing else would cover the LDS
        // latency): the pixels of the next 32-pixel tile are in flight under the current tile's 6*CT MFMAs, and each
        // weight component of tap t+1 replaces tap t's as soon as that one's last product has been issued.  Products
        // per accumulator stay in the fixed order lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; consecutive MFMAs go
        // to different accumulators.
        u32x4x av[3][CT], bv[2][3];
        auto load_a = [&](int tap, int comp) {
#pragma unroll
            for (int m = 0; m < CT; ++m) av[comp][m] = wl[comp * WCOMP + ((tap * 2 + lh) * CT + m) * 32 + l31];
        };
        auto load_b = [&](int step) {
            const int tap = step / PT, n = step % PT;
            const int toff = (tap / 3) * ROWPL + (tap % 3);
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) bv[step & 1][comp] = xl[comp * XCOMP + baddr[n] + toff];
        };
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        constexpr int LAST_USE[3] = {5, 3, 0};          // product after which a weight component is dead
        load_a(0, 0); load_a(0, 1); load_a(0, 2);
        load_b(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                const int step = tap * PT + n;
                if (step + 1 < 9 * PT) load_b(step + 1);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
#pragma unroll
                    for (int m = 0; m < CT; ++m)
                        acc[half * CT + m][n] = mfma_bf16(av[PA[q]][m], bv[step & 1][PB[q]], acc[half * CT + m][n]);
                    if (n == PT - 1 && tap + 1 < 9 && q == LAST_USE[PA[q]]) load_a(tap + 1, PA[q]);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the read-ahead at one step (registers)
            }
            after_tap(tap);
        }
    };

#ifdef X3_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    fetch_x(0);
    fetch_w(0, 0);
    X3_TICK(0);
    for (int ci0 = 0; ci0 < cin; ci0 += XKC) {
        __syncthreads();                    // the previous chunk's products are done with both images
        X3_TICK(1);
        store_x(ci0);
        store_w();
        X3_TICK(2);
        __syncthreads();
        X3_TICK(1);
        // The global loads of the next stage are spread over the taps of the running one: issued in one burst they
        // fill the memory pipeline's queues and the wave waits ~50 cycles per load with its MFMAs stopped
        // (tools/exp/x3_phases.hip); a few per tap go out for free.
        if constexpr (NH == 2) {
            mfma_half(0, [&](int tap) {     // second half of this chunk's weights, needed when this half is done
                constexpr int PER = (3 * WR + 4) / 5;
                if (tap < 5) {
#pragma unroll
                    for (int i = tap * PER; i < (tap + 1) * PER && i < 3 * WR; ++i) fetch_w_part(ci0, 1, i);
                }
            });
            X3_TICK(3);
            __syncthreads();                // every wave is done with the first half's weights
            X3_TICK(1);
            store_w();
            X3_TICK(4);
            __syncthreads();
            X3_TICK(1);
        }
        // Past the last chunk the loads re-read it: unconditional, because under a branch the compiler waits for the
        // loads on the spot to merge the registers.
        const int cnext = ci0 + XKC < cin ? ci0 + XKC : ci0;
        X3_TICK(5);
        mfma_half(NH - 1, [&](int tap) {    // the next chunk's pixels and first weight half
            constexpr int CPT = XKC / 8, WPT = (3 * WR + 7) / 8;
            if (tap < 8) {
#pragma unroll
                for (int c = tap * CPT; c < (tap + 1) * CPT; ++c) fetch_x_ch(cnext, c);
#pragma unroll
                for (int i = tap * WPT; i < (tap + 1) * WPT && i < 3 * WR; ++i) fetch_w_part(cnext, 0, i);
            } else {
                fetch_mask(cnext);
            }
        });
        X3_TICK(6);
    }

    // ---- epilogue (as conv3x3_h16.hip)
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;
    const float slope = scm_act_slope(a.act);
    const int co2n = a.cout - a.co1;
    float badd[NH * CT][16];
    {
        const float* bl = reinterpret_cast<const float*>(smx + BIAS_U);
#pragma unroll
        for (int m = 0; m < NH * CT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) badd[m][r] = bl[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    constexpr int EP = NPW + 4;
    float* ep = reinterpret_cast<float*>(smx) + wave * (32 * EP);
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<size_t>(a.y1) | reinterpret_cast<size_t>(a.y2) |
                                         reinterpret_cast<size_t>(a.mask_out)) & 15) == 0;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NH * CT; ++m) {
        // accumulate_y1: the values already in y1, ALL requested before anything waits on one of them (one
        // load-add-store per row in turn costs a memory round trip each: +200 us on a 128 -> 128 input gradient)
        constexpr int Q4 = NPW / 4;                 // float4s per channel row of the slab
        constexpr int RP = 64 / Q4;                 // rows per wave instruction
        const int q4 = lane % Q4, rsub = lane / Q4;
        f32x4 old[32 / RP];
        if (a.acc_y1 && vec) {
#pragma unroll
            for (int k = 0; k < 32 / RP; ++k) {
                const int co = cg * CW + m * 32 + RP * k + rsub;
                const int p = p0 + wave * NPW + 4 * q4;
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
        unsigned bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 32 / RP; ++k) {
            const int row = RP * k + rsub;
            const int co = cg * CW + m * 32 + row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * EP + 4 * q4);
            const int p = p0 + wave * NPW + 4 * q4;
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
        if (a.mask_out) {
            const int p = p0 + wave * NPW + 4 * q4;
            const int nw = (a.cout + 31) >> 5, w = cg * NH * CT + m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int o = Q4; o < 64; o <<= 1) bits[i] |= __shfl_xor(bits[i], o);
            }
            if (lane < Q4 && w < nw) {
                unsigned* dst = a.mask_out + ((size_t)b * nw + w) * HW + p;
                if (vec && p + 3 < HW) {
                    u32x4x t;
                    t[0] = bits[0]; t[1] = bits[1]; t[2] = bits[2]; t[3] = bits[3];
                    *reinterpret_cast<u32x4x*>(dst) = t;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p + i < HW) dst[i] = bits[i];
                }
            }
        }
        if (m + 1 < NH * CT) __syncthreads();
    }
    X3_TICK(7);
#ifdef X3_TIMING
    if (blockIdx.x == X3_TIMING && blockIdx.y == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) x3_ticks[i] = t_acc[i];
#endif
}

// ------------------------------------------------------------------ pack
// [group of CT*32 couts][chunk of 16 ci][component hi/mid/lo][tap][h][m][32 lanes][8]
static inline int x3_ct(int cout) { return cout <= 32 ? 1 : 2; }
static inline int x3_groups(int cout) {           // 64-channel groups, an even number beyond 64 channels (two per workgroup)
    const int ct = x3_ct(cout), cw = ct * 32;
    const int ng = (cout + cw - 1) / cw;
    return cout > 64 ? (ng + 1) / 2 * 2 : ng;
}
static size_t x3_pack_halves(int cin, int cout) {
    const int ct = x3_ct(cout), nch = (cin + XKC - 1) / XKC;
    return (size_t)x3_groups(cout) * nch * 3 * 9 * 2 * ct * 32 * 8;
}
extern "C" size_t scm_conv3x3_pack_x3_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return x3_pack_halves(cin, cout) * 2;
}

__global__ void pack_conv3x3_x3_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src, int wct,
                                       int wc0, unsigned short* __restrict__ fwd, unsigned short* __restrict__ dgr, int nf,
                                       int nd) {
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    auto comp_of = [](float v, int comp) -> unsigned short {
        const __bf16 h = (__bf16)v;
        if (comp == 0) return __builtin_bit_cast(unsigned short, h);
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        if (comp == 1) return __builtin_bit_cast(unsigned short, m);
        return __builtin_bit_cast(unsigned short, (__bf16)(r1 - (float)m));
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    auto decode = [&](int e, int ko, int ki, int& o, int& i, int& tap, int& comp) {
        const int ct = ko <= 32 ? 1 : 2;
        const int nch = (ki + XKC - 1) / XKC;
        const int j = e % 8; e /= 8;
        const int l = e % 32; e /= 32;
        const int m = e % ct; e /= ct;
        const int h = e % 2; e /= 2;
        tap = e % 9; e /= 9;
        comp = e % 3; e /= 3;
        const int ch = e % nch; e /= nch;
        o = e * ct * 32 + m * 32 + l;
        i = ch * XKC + 8 * h + j;
    };
    if (fwd && idx < nf) {
        int co, ci, tap, comp;
        decode(idx, cout, cin, co, ci, tap, comp);
        fwd[idx] = comp_of((co < cout && ci < cin) ? Wc(co, ci, tap) : 0.f, comp);
    }
    if (dgr && idx < nd) {
        int o, i, tap, comp;
        decode(idx, cin, cout, o, i, tap, comp);
        dgr[idx] = comp_of((o < cin && i < cout) ? Wc(i, o, 8 - tap) : 0.f, comp);
    }
}

extern "C" int scm_pack_conv3x3_weights_x3(const float* w, int cin, int cout, int transposed_src, int w_cin_total,
                                           int w_ci0, void* fwd_pack, void* dgrad_pack, void* stream) {
    SCM_REQUIRE(w && cin > 0 && cout > 0, "scm_pack_conv3x3_weights_x3: bad arguments (cin=%d cout=%d)", cin, cout);
    if (w_cin_total == 0) w_cin_total = cin;
    SCM_REQUIRE(w_ci0 >= 0 && w_ci0 + cin <= w_cin_total, "scm_pack_conv3x3_weights_x3: bad channel slice");
    SCM_REQUIRE(fwd_pack || dgrad_pack, "scm_pack_conv3x3_weights_x3: no output requested");
    const int nf = fwd_pack ? (int)x3_pack_halves(cin, cout) : 0;
    const int nd = dgrad_pack ? (int)x3_pack_halves(cout, cin) : 0;
    const int n = nf > nd ? nf : nd;
    hipLaunchKernelGGL(pack_conv3x3_x3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, (unsigned short*)fwd_pack, (unsigned short*)dgrad_pack, nf, nd);
    SCM_LAUNCH_CHECK("pack_conv3x3_x3_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------------ launch
template <int CT, int NH, int GATED, int PT, int NJ>
static int launch_x3(const ConvKArgs& a, int groups, hipStream_t s) {
    constexpr int WR = (9 * 2 * CT * 32 + 255) / 256;
    constexpr size_t stage = (size_t)3 * (WR * 256 + 2 * NJ * 256) * 16;
    constexpr size_t ep_bytes = 4 * 32 * (PT * 32 + 4) * sizeof(float);
    constexpr size_t lds = (stage > ep_bytes ? stage : ep_bytes) + NH * CT * 32 * sizeof(float);
    static_assert(lds <= 160 * 1024, "stage exceeds the LDS");
    auto kern = conv3x3_x3_kernel<CT, NH, GATED, PT, NJ>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_img, groups), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_x3_kernel");
    return SCM_OK;
}
template <int CT, int NH, int PT, int NJ>
static int launch_x3_g(const ConvKArgs& a, int groups, hipStream_t s) {
    if (a.gate_act == 0) return launch_x3<CT, NH, 0, PT, NJ>(a, groups, s);
    if (a.mask_in && a.gate_act == SCM_ACT_LRELU && a.c2 == 0) return launch_x3<CT, NH, 2, PT, NJ>(a, groups, s);
    return launch_x3<CT, NH, 1, PT, NJ>(a, groups, s);
}
// halo-tile geometry of NP-pixel workgroups; returns the elements per channel to stage
static int x3_tile_geometry(ConvKArgs& k, int NP) {
    const int HW = k.H * k.W;
    k.tiles_per_img = (HW + NP - 1) / NP;
    int rows = (NP % k.W == 0) ? NP / k.W : (NP + k.W - 2) / k.W + 1;
    if (rows > k.H) rows = k.H;
    k.plane = (rows + 2) * (k.W + 2);
    return k.plane;
}

extern "C" int scm_conv3x3_x3(const scm_conv3x3_desc* d, void* stream) {
    SCM_REQUIRE(d != nullptr, "scm_conv3x3_x3: null descriptor");
    SCM_REQUIRE(!d->x_half && !d->y_half, "scm_conv3x3_x3: IEEE-half tensors (x_half / y_half) are taken by scm_conv3x3_h16 only");
    SCM_REQUIRE(d->x1 && d->c1 > 0, "scm_conv3x3_x3: x1/c1 required");
    SCM_REQUIRE((d->x2 != nullptr) == (d->c2 > 0) && d->c2 >= 0, "scm_conv3x3_x3: x2/c2 mismatch");
    SCM_REQUIRE(d->wpack && d->cout > 0, "scm_conv3x3_x3: wpack/cout required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "scm_conv3x3_x3: bad shape B=%d H=%d W=%d", d->B, d->H, d->W);
    SCM_REQUIRE(d->y1 && d->co1 > 0 && d->co1 <= d->cout, "scm_conv3x3_x3: y1/co1 invalid");
    SCM_REQUIRE((d->co1 == d->cout) || d->y2, "scm_conv3x3_x3: y2 required when co1 < cout");
    SCM_REQUIRE(d->pad_mode == SCM_PAD_ZERO || d->pad_mode == SCM_PAD_CIRCULAR, "scm_conv3x3_x3: bad pad_mode %d", d->pad_mode);
    SCM_REQUIRE(d->act >= 0 && d->act <= 2 && d->gate_act >= 0 && d->gate_act <= 2, "scm_conv3x3_x3: bad act");
    if (d->gate_act) SCM_REQUIRE(d->gate1 && (d->c2 == 0 || d->gate2), "scm_conv3x3_x3: gate tensors required with gate_act");
    ConvKArgs k;
    k.x1 = d->x1; k.x2 = d->x2; k.c1 = d->c1; k.c2 = d->c2;
    k.gate1 = d->gate1; k.gate2 = d->gate2; k.gate_act = d->gate_act;
    k.wpack = d->wpack; k.cout = d->cout;
    k.scale = d->scale; k.bias = d->bias; k.sbias = d->sbias;
    k.y1 = d->y1; k.y2 = d->y2; k.co1 = d->co1;
    k.acc_y1 = d->accumulate_y1;
    k.ymask1 = k.ymask2 = nullptr;
    SCM_REQUIRE(!d->gate_mask_y1 && !d->gate_mask_y2, "scm_conv3x3_x3: output gates (gate_mask_y1 / _y2) are applied by scm_conv3x3_wino only");
    k.B = d->B; k.H = d->H; k.W = d->W; k.pad_mode = d->pad_mode; k.act = d->act;
    k.mask_in = d->gate_mask_in;
    k.mask_out = (d->act == SCM_ACT_LRELU && d->cout > 32) ? d->gate_mask_out : nullptr;
    SCM_REQUIRE(!d->gate_mask_out || k.mask_out, "scm_conv3x3_x3: gate_mask_out needs act == LRELU and cout > 32");
    hipStream_t s = (hipStream_t)stream;
    // 128 output channels in a launch of fewer than 160 512-pixel workgroups on the 256 CUs (small batches: BASELINE
    // configs[0], the planner's beams, the eval rollout): 128-pixel workgroups, four times as many.  Measured on the
    // B x 64 x 64, horizon-5 iteration: B = 16 (128 workgroups) 5 993 vs 5 672 frames/s in favour of the small form,
    // B = 24 (192) 6 236 vs 6 586 and B = 32 (256) 7 388 vs 8 257 against it.
    if (d->cout > 64) {
        x3_tile_geometry(k, 512);
        const int groups = (d->cout + 127) / 128;
        static const int small_below = getenv("SCM_X3_SMALL_BELOW") ? atoi(getenv("SCM_X3_SMALL_BELOW")) : 160;
        if (d->B * k.tiles_per_img * groups < small_below && x3_tile_geometry(k, 128) <= 2 * 256)
            return launch_x3_g<2, 2, 1, 2>(k, groups, s);
    }
    if (x3_tile_geometry(k, 512) > X3_NJ * 256) {
        scm_set_error("scm_conv3x3_x3: a 512-pixel halo tile of a %d-wide image (%d elements) exceeds the %d staged per channel",
                      d->W, k.plane, X3_NJ * 256);
        return SCM_EUNSUPPORTED;
    }
    if (d->cout <= 32) return launch_x3_g<1, 1, X3_PT, X3_NJ>(k, 1, s);
    if (d->cout <= 64) return launch_x3_g<2, 1, X3_PT, X3_NJ>(k, 1, s);
    return launch_x3_g<2, 2, X3_PT, X3_NJ>(k, (d->cout + 127) / 128, s);   // the pack is padded to whole 128-channel workgroups
}
