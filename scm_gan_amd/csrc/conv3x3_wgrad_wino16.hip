// Weight gradient of the 3x3 stride-1 same-size convolution by Winograd F(2x2, 3x3) for the layers with at most 16
// channels on ONE side (Transition.conv6: 256 -> 16, Transition.conv1: 16 -> 128, Encoder.conv4: 128 -> 16; reference:
// the autograd backward of nn.Conv2d w.r.t. weight and bias triggered by loss.backward() at /root/reference/main.py:285
// for models.py:51,56,134).  Same operator, same partial-slab / reduce contract as conv3x3_wgrad.hip, the algebra of
// conv3x3_wgrad_wino.hip:
//     dU_f[co][ci] = sum_{b, patch} DY_f[co][patch] * V_f[ci][patch],   DY = A dY A^T,  V = B^T d B,   dg = G^T dU G
// on v_mfma_f32_16x16x4_f32 (a 32-row tile would be half padding): A operand = DY_f (lane & 15 -> co, lane >> 4 -> one of
// 4 patches), B operand = V_f (lane & 15 -> ci, lane >> 4 -> the same patch).  The lane that feeds a patch to the MFMA
// is the lane that transforms it: no TRANSFORMED operand goes through LDS.  A wave owns two 16 x 16 tiles for all 16
// positions (128 accumulator registers): 16 co x 32 ci (XT = 2: thin output side) or 32 co x 16 ci (DT = 2: thin input
// side), for four patches that are neighbours in a patch row (8 output columns), and walks DOWN the patch rows of its
// band: a step needs two new input rows per tile and two output-gradient rows; the other two input rows -- already
// transformed along x -- are the previous step's.  The four waves of a workgroup take four neighbouring 8-column
// ranges of the same rows.  The raw rows reach the lanes through LDS: the MFMA layout puts neighbouring CHANNELS on
// neighbouring lanes, and a load whose lanes each fetch 16 bytes from 64 different channel planes is served one lane
// at a time by the texture path (a first version loaded straight into the feeding lanes: 217 us for 256 -> 16 at
// B = 64, two thirds of it waiting for those loads).  So: thread (channel, 16-byte chunk) requests whole 128-byte row
// segments two steps ahead, writes them (output gradient already gated) into one of two LDS slots a step before their
// use, one barrier per step.  At the end the waves add their accumulators through LDS, apply the signs of A's fourth
// row / column and G^T . G and write the 3x3 partials part[split][tap][co][ci] for the fixed-order reduce of
// conv3x3_wgrad.hip.  Frames: W % 32 == 0, H even.
#include "wgrad_args.h"
#include <type_traits>

#ifndef W16_ABL
#define W16_ABL 0      // ablation builds (tools): 1 no loads in the step loop, 2 no MFMAs, 3 neither
#endif
#define W16_RED_FLOATS (4 * 16 * 4 * 64)     // the cross-wave sum of one tile: [wave][f][reg][lane]

// acc += A x B on v_mfma_f32_16x16x4_f32 with the accumulator TIED to its AGPRs.  hipcc leaves the destination of this
// 4-register MFMA untied (unlike the 16-register ones) and its register allocator then permutes the 32 accumulator
// tiles of a wave through v_accvgpr_read / _mov / _write at every loop iteration (~130 copies per 128 MFMAs).
__device__ __forceinline__ void w16_mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// Packed fp32 adds with their operand selects spelled out (the compiler scalarises a vector subtraction and builds the
// mixed forms from moves): a - b;  (lo.y + hi.x, hi.x - lo.y);  (p.x + p.y, p.x - p.y).
__device__ __forceinline__ f32x2 w16_pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 w16_pk_mid(f32x2 lo, f32x2 hi) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}
__device__ __forceinline__ f32x2 w16_pk_sumdiff(f32x2 p) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}

// XT / DT: 16-channel tiles of the input / output-gradient side a wave owns (2,1 or 1,2).  GATED: dy is multiplied by
// act'(gate), gate = the layer's saved output: 1 leaky_relu, 2 sigmoid.  PADC: circular padding.
template <int XT, int DT, int GATED, int PADC>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_wino16_kernel(const WgradKArgs a) {
    static_assert(XT * DT == 2, "a wave owns two 16 x 16 tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const CSL = smem + W16_RED_FLOATS;        // [2][32] channel sums

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, p = lane >> 4;
    const int W = a.W, H = a.H, HW = H * W;
    const int TRH = H >> 1, TCW = W >> 1;
    const int xparts = (W + 31) >> 5;                // (the last 32-column part of a row may be ragged: W % 4 == 0)

    int split, wg;                                   // wg: 32-channel group of the wide side
    scm_xcd_split_group(a.spi, split, wg);
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int band = si / xparts, xpart = si - band * xparts;
    const int tr0 = band * a.rows_per_split;        // (patch rows)
    int tr1 = tr0 + a.rows_per_split;
    if (tr1 > TRH) tr1 = TRH;
    const int nst = tr1 > tr0 ? tr1 - tr0 : 0;
    const int cin = a.c1 + a.c2;

    const int pl = wave * 4 + p;                     // this lane's patch column within the workgroup's 32 columns
    const int x0 = xpart * 32;

    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gf4;
    typedef const __attribute__((address_space(1))) float* gf1;
    const size_t plane_b = (size_t)HW * 4;
    const unsigned row_b = (unsigned)W * 4u;
    constexpr int XC = 16 * XT, DC = 16 * DT;        // channels of each side in this workgroup
    constexpr int XP = 36;                            // floats of a staged input row: column x0 - 1 + i at index i
    // (a staged output-gradient row holds 32 columns in DP = 36 floats: with a pitch of 32 the 16 channel lanes of an
    // operand read sat on two banks -- PMC: LDS bank-conflict cycles 71-75 % of the LDS-active cycles of this kernel)
    constexpr int DP = 36;
    constexpr int XSL = 2 * XC * XP, DSL = 2 * DC * DP;        // floats of one slot: [row][channel][column]
    float* const XL = smem;                           // two slots of input rows
    float* const DL = smem + 2 * XSL;                 // two slots of (gated) output-gradient rows
    // scalar bases of this image's channel groups + 32-bit lane offsets (channel plane + column)
    const int g0x = XT == 2 ? wg * 32 : 0;
    gcp xb = g0x < a.c1 ? (gcp)(a.x1) + ((size_t)b * a.c1 + g0x) * plane_b
                        : (gcp)(a.x2) + ((size_t)b * a.c2 + (g0x - a.c1)) * plane_b;
    const int g0d = DT == 2 ? wg * 32 : 0;
    gcp db = (gcp)(a.dy) + ((size_t)b * a.cout + g0d) * plane_b;
    gcp gb = GATED ? (gcp)(a.gate) + ((size_t)b * a.cout + g0d) * plane_b : nullptr;

    // ---- staging roles.  A side with 32 channels: thread (channel = tid >> 3, chunk = tid & 7) takes both rows of a
    // pair; with 16 channels: (row = tid >> 7, channel = (tid >> 3) & 15, chunk).  The two border columns of an input
    // row segment (x0 - 1, x0 + 32; wrapped, or zero with zero padding) are two 4-byte loads (one per row) of threads
    // 0 .. 2 * XC - 1: (channel = tid >> 1, side = tid & 1).
    const int sk = tid & 7;
    const int sxc = XT == 2 ? tid >> 3 : (tid >> 3) & 15, sxr = XT == 2 ? 0 : wave >> 1;     // (rows: wave-uniform)
    const int sdc = DT == 2 ? tid >> 3 : (tid >> 3) & 15, sdr = DT == 2 ? 0 : wave >> 1;
    int cxi = g0x + sxc; if (cxi > cin - 1) cxi = cin - 1;      // (a channel past the end: computed, never stored)
    int cdi = g0d + sdc; if (cdi > a.cout - 1) cdi = a.cout - 1;
    // A 4-column chunk past the end of a ragged last part: its output gradient is staged as zeros (the patches do not
    // exist), its input columns are read from columns 0 .. 3 of the row -- the chunk that starts AT column W thereby
    // delivers the wrapped border column (index W - x0 + 1 of the staged row) of the last real patch; with zero padding
    // it is staged as zeros.
    const bool svalid = x0 + 4 * sk < W;
    const int scol = svalid ? x0 + 4 * sk : 0;
    const float sxz = (svalid || PADC) ? 1.f : 0.f, sdz = svalid ? 1.f : 0.f;
    unsigned sxoff = (unsigned)(cxi - g0x) * (unsigned)plane_b + (unsigned)scol * 4u;
    unsigned sdoff = (unsigned)(cdi - g0d) * (unsigned)plane_b + (unsigned)scol * 4u;
    float* const sxl = XL + (sxr * XC + sxc) * XP + 1 + 4 * sk;     // (+ XC * XP for the second row when XT == 2)
    float* const sdl = DL + (sdr * DC + sdc) * DP + 4 * sk;
    const bool hrole = tid < 2 * XC;
    const int hside = tid & 1, hc = (tid >> 1) & (XC - 1);
    int hci = g0x + hc; if (hci > cin - 1) hci = cin - 1;
    int hcol = hside ? x0 + 32 : x0 - 1;
    const float hzero = (PADC == 0 && (hcol < 0 || hcol >= W)) ? 0.f : 1.f;
    hcol = hcol < 0 ? W - 1 : (hcol >= W ? 0 : hcol);
    unsigned hoff = (unsigned)(hci - g0x) * (unsigned)plane_b + (unsigned)hcol * 4u;
    float* const hl = XL + hc * XP + (hside ? 33 : 0);          // (+ XC * XP for the second row)
    // reading lanes: tile channel c, patch pl
    const float* const rxl = XL + c * XP + 2 * pl;
    const float* const rdl = DL + c * DP + 2 * pl;

    f32x4 acc[16][2];
#pragma unroll
    for (int f = 0; f < 16; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float csum = 0.f;

    // Stage s = input rows (2j-1, 2j), j = tr0 + s, together with the output-gradient rows of patch row j - 1: what
    // step s - 1 reads.  Requested at step s - 4 into register set s & 1, written at step s - 2 into LDS slot s & 1.
    f32x4 gx[2][XT == 2 ? 2 : 1], gd[2][DT == 2 ? 2 : 1], gg[2][GATED ? (DT == 2 ? 2 : 1) : 1];
    float gh[2][2];
    f32x2 hx[2][XT][2][2];        // pairs t, t + 1 transformed along x (slot = pair & 1): columns (0, 3) and (1, 2)
#define IC(n) std::integral_constant<int, (n)>{}
#define W16_SB() __builtin_amdgcn_sched_barrier(0)

    auto request = [&](int s, auto q_) {
        constexpr int Q = decltype(q_)::value;
        int j = tr0 + s;
        if (j > TRH) j = TRH;
        int ra = 2 * j - 1, rb = 2 * j;
        if (PADC) { if (ra < 0) ra += H; if (rb >= H) rb -= H; }
        else { if (ra < 0) ra = 0; if (rb >= H) rb = H - 1; }           // (written as zeros)
        int jd = j - 1;
        if (jd < 0) jd = 0;
        if (jd > TRH - 1) jd = TRH - 1;
        gcp pa = xb + (unsigned)ra * row_b, pb = xb + (unsigned)rb * row_b;      // (32-bit products, zero-extended)
        if constexpr (XT == 2) {
            gx[Q][0] = *(gf4)(pa + sxoff);
            gx[Q][1] = *(gf4)(pb + sxoff);
        } else {
            gx[Q][0] = *(gf4)((sxr ? pb : pa) + sxoff);
        }
        gh[Q][0] = *(gf1)(pa + hoff);
        gh[Q][1] = *(gf1)(pb + hoff);
        gcp pd = db + (unsigned)(2 * jd) * row_b;
        if constexpr (DT == 2) {
            gd[Q][0] = *(gf4)(pd + sdoff);
            gd[Q][1] = *(gf4)(pd + row_b + sdoff);
        } else {
            gd[Q][0] = *(gf4)(pd + (sdr ? row_b : 0u) + sdoff);
        }
        if constexpr (GATED != 0) {
            gcp pg = gb + (unsigned)(2 * jd) * row_b;
            if constexpr (DT == 2) {
                gg[Q][0] = *(gf4)(pg + sdoff);
                gg[Q][1] = *(gf4)(pg + row_b + sdoff);
            } else {
                gg[Q][0] = *(gf4)(pg + (sdr ? row_b : 0u) + sdoff);
            }
        }
    };
    auto write = [&](int s, auto q_) {
        constexpr int Q = decltype(q_)::value;
        float* const xs = sxl + Q * XSL;
        float za = 1.f, zb = 1.f;
        if constexpr (PADC == 0) {
            const int j = tr0 + s;
            za = j <= 0 ? 0.f : 1.f;                 // rows -1 and H are zeros
            zb = j >= TRH ? 0.f : 1.f;
        }
#pragma unroll
        for (int i = 0; i < (XT == 2 ? 2 : 1); ++i) {
            f32x4 v = gx[Q][i];
            if constexpr (PADC == 0) v *= (XT == 2 ? (i ? zb : za) : (sxr ? zb : za)) * sxz;
            float* d = xs + i * XC * XP;             // (index 1 + 4k is odd: two ds_write2_b32)
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        if (hrole) {
            float va = gh[Q][0], vb = gh[Q][1];
            if constexpr (PADC == 0) { va *= hzero * za; vb *= hzero * zb; }
            hl[Q * XSL] = va;
            hl[Q * XSL + XC * XP] = vb;
        }
        // the output gradient of patch row j - 1 (step s - 1), gated, zero past the end of the band
        const float lv = (s >= 1 && s - 1 < nst) ? sdz : 0.f;
#pragma unroll
        for (int i = 0; i < (DT == 2 ? 2 : 1); ++i) {
            f32x4 v = gd[Q][i] * lv;
            if constexpr (GATED == 2) { const f32x4 g = gg[Q][i]; v *= g * (1.f - g); }
            if constexpr (GATED == 1) {
                const f32x4 g = gg[Q][i];
                v *= f32x4{g[0] > 0.f ? 1.f : 0.01f, g[1] > 0.f ? 1.f : 0.01f, g[2] > 0.f ? 1.f : 0.01f, g[3] > 0.f ? 1.f : 0.01f};
            }
            csum += (v[0] + v[1]) + (v[2] + v[3]);
            *reinterpret_cast<f32x4*>(sdl + Q * DSL + i * DC * DP) = v;
        }
    };
    // pair s from LDS slot s & 1 -> the patch's four columns, d B (the transform along x): both steps that use the pair
    // share it
    auto conv_x = [&](auto q_) {
        constexpr int Q = decltype(q_)::value;
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = rxl + Q * XSL + (i * XC + xt * 16) * XP;
                const f32x2 lo = *reinterpret_cast<const f32x2*>(src), hi = *reinterpret_cast<const f32x2*>(src + 2);
                hx[Q][xt][i][0] = w16_pk_sub(lo, hi);        // (c0 - c2, c1 - c3)
                hx[Q][xt][i][1] = w16_pk_mid(lo, hi);        // (c1 + c2, c2 - c1)
            }
    };

    // One patch row (step t, Q = t & 1): stage t + 2 written, stage t + 4 requested, pair t + 1 converted, A dY A^T of
    // patch row t (without the signs of its fourth row / column: put back in the epilogue), B^T (d B), 32 MFMAs, barrier.
    auto step = [&](int t, auto q_) {
        constexpr int Q = decltype(q_)::value, Q1 = Q ^ 1;
        // (lane offsets laundered in place: the loads stay `global_load vdst, voff32, s[row base]`)
        asm volatile("" : "+v"(sxoff), "+v"(sdoff), "+v"(hoff));
        write(t + 2, q_);
        if (!(W16_ABL & 1)) request(t + 4, q_);
        conv_x(IC(Q1));
        float vd[DT][16];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float* src = rdl + Q1 * DSL + dt * 16 * DP;
            const f32x2 d0 = *reinterpret_cast<const f32x2*>(src), d1 = *reinterpret_cast<const f32x2*>(src + DC * DP);
            const f32x2 rw[4] = {d0, d0 + d1, w16_pk_sub(d0, d1), d1};      // A dY: rows p, p + r, p - r, r
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 sd = w16_pk_sumdiff(rw[i]);
                vd[dt][4 * i] = rw[i][0]; vd[dt][4 * i + 1] = sd[0]; vd[dt][4 * i + 2] = sd[1]; vd[dt][4 * i + 3] = rw[i][1];
            }
        }
        // B^T (d B) along y: rows 0, 1 from pair t, rows 2, 3 from pair t + 1
        f32x2 tv[XT][4][2];
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 r0 = hx[Q][xt][0][h], r1 = hx[Q][xt][1][h], r2 = hx[Q1][xt][0][h], r3 = hx[Q1][xt][1][h];
                tv[xt][0][h] = w16_pk_sub(r0, r2); tv[xt][1][h] = r1 + r2;
                tv[xt][2][h] = w16_pk_sub(r2, r1); tv[xt][3][h] = w16_pk_sub(r1, r3);
            }
        // The MFMAs are inline asm: the compiler does not see that they read vd / tv and inserts none of the wait states
        // gfx950 wants between a VALU write and an MFMA read of the same register (2).  So: every VALU result of the
        // step exists before the fence, two wait states, then nothing but the 32 MFMAs up to the next fence.
        W16_SB();
        asm volatile("s_nop 1");
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    const int jj = f & 3;       // column jj of the transform: pair (0, 3) or (1, 2)
                    const float bv = tv[xt][f >> 2][(jj == 1 || jj == 2) ? 1 : 0][jj >> 1];
                    if (!(W16_ABL & 2)) w16_mfma(acc[f][xt + dt], vd[dt][f], bv);
                    else acc[f][xt + dt][0] += vd[dt][f] * bv;
                }
        W16_SB();
        __syncthreads();          // stage t + 2 is in LDS; every wave is done with stage t + 1
        W16_SB();
    };

    // ---- prologue (fenced: the compiler's s_waitcnt bookkeeping merges the pending loads of both ways into the loop
    // header, and with another request order here it waits for vmcnt(0) once per trip)
    request(0, IC(0)); W16_SB();
    request(1, IC(1)); W16_SB();
    write(0, IC(0)); W16_SB();
    request(2, IC(0)); W16_SB();
    write(1, IC(1)); W16_SB();
    request(3, IC(1)); W16_SB();
    __syncthreads();
    conv_x(IC(0));
    __syncthreads();
    W16_SB();
    for (int t = 0; t < nst; t += 2) {
        step(t, IC(0));
        step(t + 1, IC(1));
    }
#undef IC
#undef W16_SB

    // (the compiler does not know that the accumulators come from MFMAs: the wait states between the last of them and
    // the first read of a result -- already covered by the order of the reads -- made explicit)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: per tile, the four waves' accumulators meet in LDS; wave w finishes accumulator register w:
    // signs of A's fourth row / column, dg = G^T dU G (G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]), 3x3 partials
    float* pbase = a.part + (size_t)split * 9 * a.cout * cin;
    const size_t ts = (size_t)a.cout * cin;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        __syncthreads();          // (the staging slots / the previous tile's sums are done with)
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) smem[((wave * 16 + f) * 4 + r) * 64 + lane] = acc[f][tile][r];
        __syncthreads();
        float u[4][4];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += smem[((w * 16 + f) * 4 + wave) * 64 + lane];
            const int i = f >> 2, j = f & 3;
            u[i][j] = ((i == 3) != (j == 3)) ? -v : v;
        }
        float tt[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h1 = 0.5f * (u[1][j] + u[2][j]), h2 = 0.5f * (u[1][j] - u[2][j]);
            tt[0][j] = u[0][j] + h1;
            tt[1][j] = h2;
            tt[2][j] = h1 + u[3][j];
        }
        // D[i][j] of the MFMA: i = 4 * (lane >> 4) + register (the DY side), j = lane & 15 (the V side)
        const int co = (DT == 2 ? wg * 32 + tile * 16 : 0) + 4 * p + wave;
        const int ci = (XT == 2 ? wg * 32 + tile * 16 : 0) + c;
        if (co < a.cout && ci < cin) {
            float* dst = pbase + (size_t)co * cin + ci;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float h1 = 0.5f * (tt[q][1] + tt[q][2]), h2 = 0.5f * (tt[q][1] - tt[q][2]);
                dst[(3 * q + 0) * ts] = tt[q][0] + h1;
                dst[(3 * q + 1) * ts] = h2;
                dst[(3 * q + 2) * ts] = h1 + tt[q][3];
            }
        }
    }
    if (a.csum_part && (DT == 2 || wg == 0)) {      // (block-uniform)
        // per-split channel sums of the gated output gradient: the 8 chunk lanes of a staging thread's channel, then
        // (16-channel side) its two rows
        float s = csum;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        __syncthreads();
        if (sk == 0) CSL[sdr * 32 + sdc] = s;
        __syncthreads();
        if (tid < DC) {
            const int co = g0d + tid;
            if (co < a.cout) a.csum_part[(size_t)split * a.cout + co] = DT == 2 ? CSL[tid] : CSL[tid] + CSL[32 + tid];
        }
    }
}

// ------------------------------------------------------------------ host
template <int XT, int DT, int GATED, int PADC>
static int launch_w16(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)(W16_RED_FLOATS + 128) * sizeof(float);
    auto kern = conv3x3_wgrad_wino16_kernel<XT, DT, GATED, PADC>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, (int)lds)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_wino16_kernel");
    return SCM_OK;
}

// Shapes the kernel takes: at most 16 channels on one side, rows of whole 4-pixel groups (at least 32 pixels: the last
// 32-column part may be ragged -- 84 x 84, BASELINE configs[3]), even height, ungated or
// float-gated output gradient (a 1-bit gate goes to the direct kernel), 8-byte aligned output gradient; a wide input
// side split over two tensors in whole 32-channel groups; one split per (image, 32-column part) at least.
bool scm_wgrad_wino16_serves(const WgradKArgs& a, int max_splits) {
    const int cin = a.c1 + a.c2;
    const bool thin_co = a.cout <= 16, thin_ci = cin <= 16;
    if (!thin_co && !thin_ci) return false;
    if (a.W < 32 || (a.W & 3) || a.H < 2 || (a.H & 1)) return false;
    if (a.gate_act != 0 && (a.gate_mask != nullptr || a.gate == nullptr)) return false;
    if ((reinterpret_cast<size_t>(a.dy) | reinterpret_cast<size_t>(a.gate)) & 7) return false;
    if ((reinterpret_cast<size_t>(a.x1) | reinterpret_cast<size_t>(a.x2)) & 3) return false;
    if (thin_co ? (a.c2 != 0 && (a.c1 & 31) != 0) : a.c2 != 0) return false;
    // 32-bit lane offsets: 32 channel planes
    if ((size_t)a.H * a.W * 4 * 32 >= (1ull << 31)) return false;
    return a.B * ((a.W + 31) >> 5) <= max_splits;
}

// One workgroup per (split = image x band of patch rows x 32-column part, 32-channel group of the wide side);
// `max_splits`: what the caller's workspaces were sized for.
int scm_launch_wgrad_wino16(const WgradKArgs& k, int max_splits, int* nsplit_out, int* spi_out, hipStream_t s) {
    WgradKArgs a = k;
    const int cin = a.c1 + a.c2, trh = a.H / 2, xparts = (a.W + 31) / 32;
    const bool thin_co = a.cout <= 16;
    const int groups = thin_co ? (cin + 31) / 32 : (a.cout + 31) / 32;
    const int per_band = a.B * xparts * groups;
    int bands = max_splits / (a.B * xparts);
    const int want = (1024 + per_band - 1) / per_band;      // ~4 workgroups per CU
    if (bands > want) bands = want;
    if (bands > trh / 8) bands = trh / 8;                   // bands of at least 8 patch rows
    if (bands < 1) bands = 1;
    a.rows_per_split = (trh + bands - 1) / bands;
    bands = (trh + a.rows_per_split - 1) / a.rows_per_split;
    a.spi = bands * xparts;
    *nsplit_out = a.B * a.spi;
    *spi_out = a.spi;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int g = a.gate_act == SCM_ACT_SIGMOID ? 2 : (a.gate_act == SCM_ACT_LRELU ? 1 : 0);
    dim3 grid(a.B * a.spi, groups);
#define W16_P(XT_, DT_, G_) (circ ? launch_w16<XT_, DT_, G_, 1>(a, grid, s) : launch_w16<XT_, DT_, G_, 0>(a, grid, s))
#define W16_G(XT_, DT_) (g == 2 ? W16_P(XT_, DT_, 2) : (g == 1 ? W16_P(XT_, DT_, 1) : W16_P(XT_, DT_, 0)))
    return thin_co ? W16_G(2, 1) : W16_G(1, 2);
#undef W16_G
#undef W16_P
}
