// 16-bit-input / fp32-accumulate variant of the 3x3 same-size convolution
// (SURVEY.md section 7 step 10, section 8(d) "cfg5 (fp16 MFMA)"), opt-in.
//
// Same implicit GEMM, same NCHW fp32 tensors in HBM and the same epilogue as
// conv3x3_mfma.hip; only the MFMA operands change: activations are rounded to
// fp16 (forward) or bf16 (gradients: fp32's exponent range, no loss scaling)
// while they are staged into LDS, weights are packed in that type once per
// optimizer step, products accumulate in fp32 on v_mfma_f32_32x32x16_{f16,bf16}
// (16x the f32 MFMA rate, so the kernel is HBM/L2-bound, not MFMA-bound).
//
//   MFMA A = weights  A[i = co (lane&31)][k = ci 8*(lane>>5)+j]   8 halves (16 B) per lane
//   MFMA B = pixels   B[k = ci 8*(lane>>5)+j][j = pixel (lane&31)]
//   D as in the f32 kernel: col = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = co.
//
// K chunks of 16 input channels.  A thread stages halo element e for all 16
// channels of the chunk, so the channel-innermost LDS image the B operand needs
// ([ci half h][element e][8 channels] = one 16-byte unit per element) is a plain
// ds_write_b128 of the thread's own registers: no transpose pass.  Consecutive
// lanes read/write consecutive 16-byte units: conflict-free.
//
// Workgroup = 4 waves x PT pixel tiles x CT*32 output channels.  128 output channels on frames up to 64 wide:
// PT = 2 (256 pixels), one LDS stage, TWO workgroups per CU -- they drift apart, so one's staging, barriers and
// epilogue (a third of a workgroup's time: the whole chip writing at once) run under the other's MFMAs.
// Otherwise PT = 4 (512 pixels), one workgroup per CU (256 accumulator registers + the prefetch registers of a
// 16-channel chunk), two LDS stages when they fit.  The next chunk's global loads are dealt out over the taps, the
// LDS operands of tap t+1 are read under tap t's MFMAs.
#include <type_traits>
#include <stdlib.h>
#include "conv_args.h"

#define HKC 16

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));

template <int PREC>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    if constexpr (PREC == SCM_PREC_F16) {
        h16x2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(unsigned, v);
    } else {
        b16x2 v = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
}

template <int PREC>
__device__ __forceinline__ f32x16 mfma_h16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (PREC == SCM_PREC_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
}

// Phase clock for tools/exp/h16_phases.hip (compiled in only with -DH16_TIMING): shader-clock ticks per phase of wave 0
#ifdef H16_TIMING
__device__ unsigned long long h16_ticks[8];
#define H16_TICK(i)                                     \
    do {                                                \
        const unsigned long long t_ = clock64();        \
        t_acc[i] += t_ - t_last;                        \
        t_last = t_;                                    \
    } while (0)
#else
#define H16_TICK(i)
#endif

// GATED: 0 plain input, 1 input * act'(gate tensor), 2 input * leaky-relu slope from the 1-bit gate mask
// WPS: workgroups per CU (= waves per SIMD).  1: 512-pixel tiles, 256 accumulator registers, two LDS stages when they fit.
// 2: 256-pixel tiles (PT = 2), one LDS stage -- two workgroups share a CU and drift apart, so one's staging, barriers and
// epilogue (a third of a workgroup's time at 128 -> 128: the whole chip writing at once) run under the other's MFMAs.
// XH: the inputs are IEEE-half tensors (scm_conv3x3_desc.x_half; forward launches of the fp16 precision only)
template <int PREC, int CT, int PT, int NJ, int GATED, int WPS = 1, int XH = 0>
__global__ void __launch_bounds__(256, WPS) conv3x3_h16_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    constexpr int CW = CT * 32;
    constexpr int WU = 9 * 2 * CT * 32;            // 16-byte units of a weight chunk: [tap][h][m][lane]
    constexpr int WR = (WU + 255) / 256;
    constexpr int NPW = PT * 32, NP = 4 * NPW;
    constexpr int PLANE = NJ * 256;                // units per channel half
    constexpr int STAGE = WR * 256 + 2 * PLANE;    // units per LDS stage
    constexpr bool DB = WPS == 1 && 2 * STAGE * 16 <= 160 * 1024;
    constexpr int MAIN_U = (DB ? 2 : 1) * STAGE, EP_U = 4 * 32 * (NPW + 4) / 4;
    constexpr int BIAS_U = MAIN_U > EP_U ? MAIN_U : EP_U;     // bias table behind the stages and the epilogue slabs

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
        const int co = cg * CW + tid;
        const int coc = co < a.cout ? co : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        reinterpret_cast<float*>(smem16 + BIAS_U)[tid] = v;
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

    f32x16 acc[CT][PT];
#pragma unroll
    for (int m = 0; m < CT; ++m)
#pragma unroll
        for (int n = 0; n < PT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int cin = a.c1 + a.c2;
    const int nchunks = (cin + HKC - 1) / HKC;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)cg * nchunks * WU;
    const int wtotal = nchunks * WU;
    const int gate_act = GATED ? a.gate_act : 0;

    unsigned pw[WR][4];
    float px[HKC][NJ], pg[GATED == 1 ? HKC : 1][GATED == 1 ? NJ : 1];
    unsigned pm[GATED == 2 ? NJ : 1];
    const int mask_nw = (a.c1 + 31) >> 5;

    // per-image bases of the two concatenated inputs, indexed by the CONCATENATED channel; pinned in SGPRs (left to
    // itself the compiler selects between the kernarg slots and re-loads the pointer, with a wait, per channel) and
    // spelled out as global address space (behind the asm it would otherwise fall back to flat loads, which also count
    // on lgkmcnt and so serialise with the LDS reads of the MFMA loop)
    typedef const __attribute__((address_space(1))) float* gfp;
    gfp xb1 = (gfp)(a.x1 + (size_t)b * a.c1 * HW);
    gfp xb2 = a.c2 ? (gfp)(a.x2 + ((size_t)b * a.c2 - a.c1) * HW) : xb1;
    gfp gb1 = GATED == 1 ? (gfp)(a.gate1 + (size_t)b * a.c1 * HW) : nullptr;
    gfp gb2 = (GATED == 1 && a.c2) ? (gfp)(a.gate2 + ((size_t)b * a.c2 - a.c1) * HW) : gb1;
    asm volatile("" : "+s"(xb1), "+s"(xb2));
    if constexpr (GATED == 1) asm volatile("" : "+s"(gb1), "+s"(gb2));
    auto fetch_w_part = [&](int ci0, int k) {
        int gi = (ci0 / HKC) * WU + tid + 256 * k;
        gi = gi < wtotal ? gi : wtotal - 1;
        const u32x4 t = wsrc[gi];
        pw[k][0] = t[0]; pw[k][1] = t[1]; pw[k][2] = t[2]; pw[k][3] = t[3];
    };
    // x_half: the inputs are IEEE-half tensors (written by a call with y_half): two bytes per element cross HBM; the
    // value staged is the same fp16 number either way
    typedef const __attribute__((address_space(1))) _Float16* ghp;
    constexpr bool xh = XH != 0;
    ghp hb1 = (ghp)(gfp)(a.x1) + (size_t)b * a.c1 * HW;
    ghp hb2 = a.c2 ? (ghp)(gfp)(a.x2) + ((size_t)b * a.c2 - a.c1) * HW : hb1;
    auto fetch_x_ch = [&](int ci0, int c) {
        const int ci = ci0 + c;
        const int cc = ci < cin ? ci : 0;
        const bool first = cc < a.c1;
        if constexpr (xh) {
            // two channels share a register (px of the even one): the 2-byte loads land in its low / high half
            // (global_load_short_d16 / _d16_hi) and the LDS image takes it as it is -- no conversion, no packing
            const ghp hsrc = (first ? hb1 : hb2) + (size_t)cc * HW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                h16x2 t = __builtin_bit_cast(h16x2, px[c & ~1][j]);
                t[c & 1] = hsrc[goff[j]];
                px[c & ~1][j] = __builtin_bit_cast(float, t);
            }
        } else {
            const gfp src = (first ? xb1 : xb2) + (size_t)cc * HW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) px[c][j] = src[goff[j]];
        }
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
    auto fetch = [&](int ci0) {
#pragma unroll
        for (int k = 0; k < WR; ++k) fetch_w_part(ci0, k);
#pragma unroll
        for (int c = 0; c < HKC; ++c) fetch_x_ch(ci0, c);
        fetch_mask(ci0);
    };
    // part `i` of `n` equal shares of a chunk's global loads (dealt out over MFMA taps: issued in one burst they fill
    // the memory pipeline's queues and the wave waits ~50 cycles per load with its MFMAs stopped)
    auto fetch_share = [&](int ci0, int i, int n) {
#pragma unroll
        for (int c = 0; c < HKC; ++c)
            if (c * n / HKC == i) fetch_x_ch(ci0, c);
#pragma unroll
        for (int k = 0; k < WR; ++k)
            if (k * n / WR == i) fetch_w_part(ci0, k);
        if (i == n - 1) fetch_mask(ci0);
    };
    auto store = [&](int ci0, int buf) {
        u32x4* wl = smem16 + buf * STAGE;
        u32x4* xl = wl + WR * 256;
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            u32x4 v;
            v[0] = pw[k][0]; v[1] = pw[k][1]; v[2] = pw[k][2]; v[3] = pw[k][3];
            wl[tid + 256 * k] = v;                 // channels >= cin are zero in the pack
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool ok = (okmask >> j) & 1u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (xh) {
                    // (channels beyond cin are zero in the weight pack; what is staged for them only has to be finite:
                    // the loads re-read channel 0)
                    u32x4 u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i] = ok ? __float_as_uint(px[8 * h + 2 * i][j]) : 0u;
                    xl[h * PLANE + tid + 256 * j] = u;
                    continue;
                }
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = 8 * h + i;
                    float t = px[c][j];
                    if constexpr (GATED == 1) t *= scm_act_grad_from_out(pg[c][j], gate_act);
                    if constexpr (GATED == 2) t *= ((pm[j] >> ((ci0 & 31) + c)) & 1u) ? 1.f : 0.01f;
                    v[i] = (ok && ci0 + c < cin) ? t : 0.f;
                }
                u32x4 u;
                u[0] = pack2<PREC>(v[0], v[1]); u[1] = pack2<PREC>(v[2], v[3]);
                u[2] = pack2<PREC>(v[4], v[5]); u[3] = pack2<PREC>(v[6], v[7]);
                xl[h * PLANE + tid + 256 * j] = u;
            }
        }
    };
    // MFMA over taps [T0, T1) of the chunk staged in LDS buffer `buf`.  The operands of tap t+1 are read while tap t's
    // CT*PT MFMAs run (one wave per SIMD: nothing else covers the LDS latency); after_tap(t) is where the caller deals
    // out its global loads.
    auto mfma_taps = [&](int buf, auto t0c, auto t1c, auto&& after_tap) {
        const u32x4* wl = smem16 + buf * STAGE;
        const u32x4* xl = wl + WR * 256;
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
        constexpr bool AHEAD = NJ <= 4 && WPS == 1;    // (no registers for it with the 6-slot halo map or two waves per SIMD)
        u32x4 av[AHEAD ? 2 : 1][CT], bv[AHEAD ? 2 : 1][PT];
        auto load_ops = [&](int tap) {
            const int toff = (tap / 3) * ROWPL + (tap % 3);
            const int s = AHEAD ? tap & 1 : 0;
#pragma unroll
            for (int m = 0; m < CT; ++m) av[s][m] = wl[((tap * 2 + lh) * CT + m) * 32 + l31];
#pragma unroll
            for (int n = 0; n < PT; ++n) bv[s][n] = xl[baddr[n] + toff];
        };
        if constexpr (AHEAD) load_ops(T0);
#pragma unroll
        for (int tap = T0; tap < T1; ++tap) {
            if constexpr (AHEAD) {
                if (tap + 1 < T1) load_ops(tap + 1);
            } else {
                load_ops(tap);
            }
            const int s = AHEAD ? tap & 1 : 0;
#pragma unroll
            for (int m = 0; m < CT; ++m)
#pragma unroll
                for (int n = 0; n < PT; ++n) acc[m][n] = mfma_h16<PREC>(av[s][m], bv[s][n], acc[m][n]);
            after_tap(tap);
        }
    };
    auto no_hook = [](int) {};
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using I9 = std::integral_constant<int, 9>;

#ifdef H16_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    if constexpr (DB) {
        fetch(0);
        store(0, 0);
        if (HKC < cin) fetch(HKC);
        __syncthreads();
        H16_TICK(0);
        int cur = 0;
        for (int ci0 = 0; ci0 < cin; ci0 += HKC) {
            mfma_taps(cur, I0{}, I4{}, no_hook);
            H16_TICK(1);
            if (ci0 + HKC < cin) store(ci0 + HKC, cur ^ 1);
            H16_TICK(2);
            // the staging registers are free again: chunk + 2, a share per tap.  Past the last chunk the loads re-read
            // it (unconditional: under a branch the compiler waits for them on the spot to merge the registers)
            if constexpr (NJ <= 4) {
                const int cnext = ci0 + 2 * HKC < cin ? ci0 + 2 * HKC : ci0;
                mfma_taps(cur, I4{}, I9{}, [&](int tap) { fetch_share(cnext, tap - 4, 5); });
            } else {                                        // 128-wide frames: no registers to spare, one burst
                if (ci0 + 2 * HKC < cin) fetch(ci0 + 2 * HKC);
                mfma_taps(cur, I4{}, I9{}, no_hook);
            }
            H16_TICK(3);
            __syncthreads();
            H16_TICK(4);
            cur ^= 1;
        }
    } else {
        fetch(0);
        for (int ci0 = 0; ci0 < cin; ci0 += HKC) {
            __syncthreads();
            store(ci0, 0);
            __syncthreads();
            if constexpr (NJ <= 4) {
                const int cnext = ci0 + HKC < cin ? ci0 + HKC : ci0;   // (unconditional loads, as above)
                mfma_taps(0, I0{}, I9{}, [&](int tap) { fetch_share(cnext, tap, 9); });
            } else {
                if (ci0 + HKC < cin) fetch(ci0 + HKC);
                mfma_taps(0, I0{}, I9{}, no_hook);
            }
        }
    }

    // ---- epilogue (as conv3x3_mfma.hip): scale, bias, per-sample bias, activation, NCHW store
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;     // one branch per loop nest, not per value (common.h)
    const float slope = scm_act_slope(a.act);
    const int co2n = a.cout - a.co1;
    // bias + per-sample bias of this workgroup's channels were put into an LDS table by the
    // prologue: 16*CT broadcast LDS reads here instead of 32*CT global loads per thread (~22
    // cycles of the CU's memory pipe each, with nothing to hide them at one wave per SIMD)
    float badd[CT][16];
    {
        const float* bl = reinterpret_cast<const float*>(smem16 + BIAS_U);
#pragma unroll
        for (int m = 0; m < CT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) badd[m][r] = bl[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    // The accumulator layout (lane -> pixel, register -> channel) gives 128-byte store segments
    // and one store instruction per register: ~85 cycles of the CU's store path each, 36 us per
    // workgroup at CT = 4.  Each wave transposes its 32-channel x 128-pixel slab through its own
    // LDS region instead and stores float4s: whole 512-byte row runs, a quarter of the instructions.
    constexpr int EP = NPW + 4;                     // floats per channel row (+16 B: rows on different banks)
    float* ep = reinterpret_cast<float*>(smem16) + wave * (32 * EP);
    // 16-byte stores need 16-byte aligned rows: H*W % 4 == 0 and aligned base pointers (wave-uniform)
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<size_t>(a.y1) | reinterpret_cast<size_t>(a.y2) |
                                         reinterpret_cast<size_t>(a.mask_out)) & 15) == 0;
    __syncthreads();                                // every wave is done reading the stage buffers
#pragma unroll
    for (int m = 0; m < CT; ++m) {
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
        unsigned bits[4] = {0u, 0u, 0u, 0u};        // sign bits of this lane's rows, per pixel of its float4
#pragma unroll
        for (int k = 0; k < 32 / RP; ++k) {
            const int row = RP * k + rsub;
            const int co = cg * CW + m * 32 + row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * EP + 4 * q4);
            const int p = p0 + wave * NPW + 4 * q4;
#pragma unroll
            for (int i = 0; i < 4; ++i) bits[i] |= (v[i] > 0.f ? 1u : 0u) << row;
            if (co < a.cout && a.y_half) {          // (wave-uniform) the activation leaves as IEEE half: 8-byte lanes
                _Float16* dst = (co < a.co1 ? reinterpret_cast<_Float16*>(a.y1) + ((size_t)b * a.co1 + co) * HW
                                            : reinterpret_cast<_Float16*>(a.y2) + ((size_t)b * co2n + (co - a.co1)) * HW) + p;
                if (vec && p + 3 < HW) {
                    u32x2h t;
                    t[0] = pack2<SCM_PREC_F16>(v[0], v[1]); t[1] = pack2<SCM_PREC_F16>(v[2], v[3]);
                    *reinterpret_cast<u32x2h*>(dst) = t;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p + i < HW) dst[i] = (_Float16)v[i];
                }
            } else if (co < a.cout) {
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
            const int p = p0 + wave * NPW + 4 * q4;
            const int nw = (a.cout + 31) >> 5, w = cg * CT + m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int o = Q4; o < 64; o <<= 1) bits[i] |= __shfl_xor(bits[i], o);
            }
            if (lane < Q4 && w < nw) {
                unsigned* dst = a.mask_out + ((size_t)b * nw + w) * HW + p;
                if (vec && p + 3 < HW) {
                    u32x4 t;
                    t[0] = bits[0]; t[1] = bits[1]; t[2] = bits[2]; t[3] = bits[3];
                    *reinterpret_cast<u32x4*>(dst) = t;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p + i < HW) dst[i] = bits[i];
                }
            }
        }
        if (m + 1 < CT) __syncthreads();
    }
    H16_TICK(5);
#ifdef H16_TIMING
    if (blockIdx.x == H16_TIMING && blockIdx.y == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) h16_ticks[i] = t_acc[i];
#endif
}

// ------------------------------------------------------------------ pack
// [group][chunk of 16 ci][tap][h][m][32 lanes][8]: element j of the unit of lane l holds
// Wc[co = g*CW + m*32 + l][ci = 16*chunk + 8*h + j][tap]; zero beyond cout / cin.
static inline int h16_ct(int cout) { return cout <= 32 ? 1 : (cout <= 64 ? 2 : 4); }

static size_t h16_pack_halves(int cin, int cout) {
    const int ct = h16_ct(cout), cw = ct * 32;
    const int ng = (cout + cw - 1) / cw, nch = (cin + HKC - 1) / HKC;
    return (size_t)ng * nch * 9 * 2 * ct * 32 * 8;
}

__global__ void pack_conv3x3_h16_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src, int wct,
                                        int wc0, unsigned short* __restrict__ fwd, int fprec,
                                        unsigned short* __restrict__ dgr, int dprec, int nf, int nd) {
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    auto cvt = [](float v, int prec) -> unsigned short {
        if (prec == SCM_PREC_F16) return __builtin_bit_cast(unsigned short, (_Float16)v);
        return __builtin_bit_cast(unsigned short, (__bf16)v);
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // decode idx for a conv with `ko` output and `ki` input channels
    auto decode = [&](int e, int ko, int ki, int& o, int& i, int& tap) {
        const int ct = ko <= 32 ? 1 : (ko <= 64 ? 2 : 4);
        const int nch = (ki + HKC - 1) / HKC;
        const int j = e % 8; e /= 8;
        const int l = e % 32; e /= 32;
        const int m = e % ct; e /= ct;
        const int h = e % 2; e /= 2;
        tap = e % 9; e /= 9;
        const int ch = e % nch; e /= nch;
        o = e * ct * 32 + m * 32 + l;
        i = ch * HKC + 8 * h + j;
    };
    if (fwd && idx < nf) {
        int co, ci, tap;
        decode(idx, cout, cin, co, ci, tap);
        fwd[idx] = cvt((co < cout && ci < cin) ? Wc(co, ci, tap) : 0.f, fprec);
    }
    if (dgr && idx < nd) {
        // dgrad conv: input channels = cout, output channels = cin, taps flipped
        int o, i, tap;
        decode(idx, cin, cout, o, i, tap);
        dgr[idx] = cvt((o < cin && i < cout) ? Wc(i, o, 8 - tap) : 0.f, dprec);
    }
}

extern "C" size_t scm_conv3x3_pack_h16_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return h16_pack_halves(cin, cout) * 2;
}

extern "C" int scm_pack_conv3x3_weights_h16(const float* w, int cin, int cout, int transposed_src, int w_cin_total,
                                            int w_ci0, void* fwd_pack, int fwd_precision, void* dgrad_pack,
                                            int dgrad_precision, void* stream) {
    SCM_REQUIRE(w && cin > 0 && cout > 0, "scm_pack_conv3x3_weights_h16: bad arguments (cin=%d cout=%d)", cin, cout);
    if (w_cin_total == 0) w_cin_total = cin;
    SCM_REQUIRE(w_ci0 >= 0 && w_ci0 + cin <= w_cin_total, "scm_pack_conv3x3_weights_h16: bad channel slice");
    SCM_REQUIRE(fwd_pack || dgrad_pack, "scm_pack_conv3x3_weights_h16: no output requested");
    SCM_REQUIRE(!fwd_pack || fwd_precision == SCM_PREC_F16 || fwd_precision == SCM_PREC_BF16,
                "scm_pack_conv3x3_weights_h16: bad forward precision %d", fwd_precision);
    SCM_REQUIRE(!dgrad_pack || dgrad_precision == SCM_PREC_F16 || dgrad_precision == SCM_PREC_BF16,
                "scm_pack_conv3x3_weights_h16: bad dgrad precision %d", dgrad_precision);
    const int nf = fwd_pack ? (int)h16_pack_halves(cin, cout) : 0;
    const int nd = dgrad_pack ? (int)h16_pack_halves(cout, cin) : 0;
    const int n = nf > nd ? nf : nd;
    hipLaunchKernelGGL(pack_conv3x3_h16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, (unsigned short*)fwd_pack, fwd_precision,
                       (unsigned short*)dgrad_pack, dgrad_precision, nf, nd);
    SCM_LAUNCH_CHECK("pack_conv3x3_h16_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------------ launch
template <int PREC, int CT, int PT, int NJ, int GATED, int WPS>
static int launch_h16_nj(const ConvKArgs& a, int groups, hipStream_t s) {
    constexpr int WR = (9 * 2 * CT * 32 + 255) / 256;
    constexpr size_t stage = ((size_t)WR * 256 + 2 * NJ * 256) * 16;
    constexpr size_t ep_bytes = 4 * 32 * (PT * 32 + 4) * sizeof(float);   // epilogue transpose slabs
    constexpr size_t lds_main = (WPS == 1 && 2 * stage <= 160 * 1024) ? 2 * stage : stage;
    constexpr size_t lds = (lds_main > ep_bytes ? lds_main : ep_bytes) + CT * 32 * sizeof(float);   // + bias table
    static_assert(lds * WPS <= 160 * 1024, "halo tile exceeds the LDS");
    if constexpr (PREC == SCM_PREC_F16 && GATED == 0) {
        if (a.x_half) {
            auto kernh = conv3x3_h16_kernel<PREC, CT, PT, NJ, GATED, WPS, 1>;
            static std::atomic<uint64_t> attr_done_h{0};
            if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kernh), attr_done_h, 160 * 1024)) return rc;
            hipLaunchKernelGGL(kernh, dim3(a.B * a.tiles_per_img, groups), dim3(256), lds, s, a);
            SCM_LAUNCH_CHECK("conv3x3_h16_kernel");
            return SCM_OK;
        }
    }
    auto kern = conv3x3_h16_kernel<PREC, CT, PT, NJ, GATED, WPS>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_img, groups), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_h16_kernel");
    return SCM_OK;
}

template <int PREC, int CT, int PT, int NJ, int WPS>
static int launch_h16_g(const ConvKArgs& a, int groups, hipStream_t s) {
    if (a.gate_act == 0) return launch_h16_nj<PREC, CT, PT, NJ, 0, WPS>(a, groups, s);
    if (a.mask_in && a.gate_act == SCM_ACT_LRELU && a.c2 == 0) return launch_h16_nj<PREC, CT, PT, NJ, 2, WPS>(a, groups, s);
    return launch_h16_nj<PREC, CT, PT, NJ, 1, WPS>(a, groups, s);
}

static void h16_tile_geometry(ConvKArgs& a, int NP) {
    const int HW = a.H * a.W;
    a.tiles_per_img = (HW + NP - 1) / NP;
    int rows = (NP % a.W == 0) ? NP / a.W : (NP + a.W - 2) / a.W + 1;
    if (rows > a.H) rows = a.H;
    a.plane = (rows + 2) * (a.W + 2);
}

template <int PREC, int CT>
static int launch_h16(const ConvKArgs& k, int groups, hipStream_t s) {
    ConvKArgs a = k;
    if constexpr (CT == 4) {
        // 128 output channels: two 256-pixel workgroups per CU (see WPS) when the halo tile fits two slots per thread
        h16_tile_geometry(a, 256);
        if (a.plane <= 2 * 256 && !getenv("SCM_H16_ONE_WG")) return launch_h16_g<PREC, CT, 2, 2, 2>(a, groups, s);
        // (128-wide frames -- BASELINE configs[4]: two rows of 128 pixels + halo = 520 elements: three slots per thread)
        // -- ungated and mask-gated launches only: the float-gated form spills 41 registers at two workgroups per CU
        if (a.plane <= 3 * 256 && !getenv("SCM_H16_ONE_WG")) {
            if (a.gate_act == 0) return launch_h16_nj<PREC, CT, 2, 3, 0, 2>(a, groups, s);
            if (a.mask_in && a.gate_act == SCM_ACT_LRELU && a.c2 == 0) return launch_h16_nj<PREC, CT, 2, 3, 2, 2>(a, groups, s);
        }
    }
    h16_tile_geometry(a, 512);
    if (a.plane <= 3 * 256) return launch_h16_g<PREC, CT, 4, 3, 1>(a, groups, s);
    if (a.plane <= 4 * 256) return launch_h16_g<PREC, CT, 4, 4, 1>(a, groups, s);
    if (a.plane <= 6 * 256) return launch_h16_g<PREC, CT, 4, 6, 1>(a, groups, s);
    scm_set_error("scm_conv3x3_h16: image width %d too large for the halo staging map", a.W);
    return SCM_EUNSUPPORTED;
}

template <int PREC>
static int launch_h16_ct(const ConvKArgs& k, hipStream_t s) {
    const int ct = h16_ct(k.cout);
    const int groups = (k.cout + ct * 32 - 1) / (ct * 32);
    if (ct == 4) return launch_h16<PREC, 4>(k, groups, s);
    if (ct == 2) return launch_h16<PREC, 2>(k, groups, s);
    return launch_h16<PREC, 1>(k, groups, s);
}

extern "C" int scm_conv3x3_h16(const scm_conv3x3_desc* d, int precision, void* stream) {
    SCM_REQUIRE(d != nullptr, "scm_conv3x3_h16: null descriptor");
    SCM_REQUIRE(precision == SCM_PREC_F16 || precision == SCM_PREC_BF16, "scm_conv3x3_h16: bad precision %d", precision);
    SCM_REQUIRE(d->x1 && d->c1 > 0, "scm_conv3x3_h16: x1/c1 required");
    SCM_REQUIRE((d->x2 != nullptr) == (d->c2 > 0) && d->c2 >= 0, "scm_conv3x3_h16: x2/c2 mismatch");
    SCM_REQUIRE(d->wpack && d->cout > 0, "scm_conv3x3_h16: wpack/cout required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "scm_conv3x3_h16: bad shape B=%d H=%d W=%d", d->B, d->H, d->W);
    SCM_REQUIRE(d->y1 && d->co1 > 0 && d->co1 <= d->cout, "scm_conv3x3_h16: y1/co1 invalid");
    SCM_REQUIRE((d->co1 == d->cout) || d->y2, "scm_conv3x3_h16: y2 required when co1 < cout");
    SCM_REQUIRE(d->pad_mode == SCM_PAD_ZERO || d->pad_mode == SCM_PAD_CIRCULAR, "scm_conv3x3_h16: bad pad_mode %d",
                d->pad_mode);
    SCM_REQUIRE(d->act >= 0 && d->act <= 2 && d->gate_act >= 0 && d->gate_act <= 2, "scm_conv3x3_h16: bad act");
    if (d->gate_act) {
        SCM_REQUIRE(d->gate1 && (d->c2 == 0 || d->gate2), "scm_conv3x3_h16: gate tensors required with gate_act");
    }
    ConvKArgs k;
    k.x1 = d->x1; k.x2 = d->x2; k.c1 = d->c1; k.c2 = d->c2;
    k.gate1 = d->gate1; k.gate2 = d->gate2; k.gate_act = d->gate_act;
    k.wpack = d->wpack; k.cout = d->cout;
    k.scale = d->scale; k.bias = d->bias; k.sbias = d->sbias;
    k.y1 = d->y1; k.y2 = d->y2; k.co1 = d->co1;
    k.acc_y1 = d->accumulate_y1;
    k.ymask1 = k.ymask2 = nullptr;
    SCM_REQUIRE(!d->gate_mask_y1 && !d->gate_mask_y2, "scm_conv3x3_h16: output gates (gate_mask_y1 / _y2) are applied by scm_conv3x3_wino only");
    k.B = d->B; k.H = d->H; k.W = d->W; k.pad_mode = d->pad_mode; k.act = d->act;
    k.tiles_per_img = 0; k.plane = 0;
    k.mask_in = d->gate_mask_in;
    k.mask_out = (d->act == SCM_ACT_LRELU && d->cout > 32) ? d->gate_mask_out : nullptr;
    SCM_REQUIRE(!d->gate_mask_out || k.mask_out, "scm_conv3x3_h16: gate_mask_out needs act == LRELU and cout > 32");
    k.x_half = d->x_half != 0; k.y_half = d->y_half != 0;
    SCM_REQUIRE(!k.y_half || !d->accumulate_y1, "scm_conv3x3_h16: y_half and accumulate_y1 exclude each other");
    SCM_REQUIRE(!k.x_half || (d->gate_act == 0 && precision == SCM_PREC_F16),
                "scm_conv3x3_h16: x_half takes ungated inputs in the fp16 precision (activations, not gradients)");
    SCM_REQUIRE(!k.y_half || ((reinterpret_cast<size_t>(d->y1) | reinterpret_cast<size_t>(d->y2)) & 7) == 0,
                "scm_conv3x3_h16: y_half needs 8-byte aligned outputs");
    hipStream_t s = (hipStream_t)stream;
    return precision == SCM_PREC_F16 ? launch_h16_ct<SCM_PREC_F16>(k, s) : launch_h16_ct<SCM_PREC_BF16>(k, s);
}
