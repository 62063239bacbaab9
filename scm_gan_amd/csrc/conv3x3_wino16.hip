// Winograd F(2x2, 3x3) for the 16-output-channel layers, on v_mfma_f32_16x16x4_f32: Transition.conv6 (256 -> 16) and
// Encoder.conv4 (128 -> 16) forward, the input gradient of Transition.conv1 (128 -> 16)
// (reference /root/reference/models.py:76,102,155 and their autograd backward).  Same operator, descriptor and
// epilogue contract as scm_conv3x3 for ungated inputs and a single output tensor; reached through scm_conv3x3_wino.
//
//   M_f[co][patch] = sum_ci U_f[co][ci] V_f[ci][patch]   for the 16 transform positions f  (conv3x3_wino.hip)
//   MFMA A = U  A[i = co (lane & 15)][k = ci (lane >> 4)]        4 input channels per instruction
//   MFMA B = V  B[k = ci (lane >> 4)][j = patch (lane & 15)]
//   D[i][j]: lane & 15 -> patch, row = 4 * (lane >> 4) + reg -> co
// Workgroup = 4 waves = 16 co x 128 patches (512 output pixels); wave w owns patches [32w, 32w+32) as two 16-patch
// blocks, all 16 positions: 128 accumulator registers.  With only 16 output channels a transformed patch is used by
// ONE MFMA column, so V never goes through LDS: the thread that transforms a patch IS the lane that feeds it to the
// MFMA.  Thread (k = lane >> 4, p = lane & 15) of wave w transforms, per K chunk of 8 channels, the four patches
// {32w + p, 32w + 16 + p} x {channel k, channel 4 + k} -- exactly the B operands of its lane for both k-steps and both
// blocks -- and keeps them in registers (two sets: chunk c+1 is transformed while chunk c multiplies).  Only the raw
// halo rows pass through LDS (two stages, aligned pixel pairs as in conv3x3_wino.hip; one barrier per chunk); the
// transformed filters come straight from L2 (pack [chunk][position pair][lane][2 positions x 2 k-steps], shared by the
// four waves through L1).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_args.h"

#define W16_TILES 128
#define W16_USLAB 2048      // floats of one K chunk of the filter pack: [8 position pairs][64 lanes][4]

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wino16_mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// NB: 16-patch blocks per wave.  2: 16 co x 128 patches per workgroup, 128 accumulator registers + two sets of four
// transformed patches (128 registers): one wave per SIMD.  1: 16 co x 64 patches, 64 + 64 registers: the kernel fits a
// CU two or three times, and a workgroup that waits for its rows leaves the SIMD to another (these layers are bound by
// memory latency and instruction issue, not by the MFMA pipe: PMC MFMA-busy 0.40 at one wave per SIMD).
template <int NJ, int PADC, int NB>
__global__ void __launch_bounds__(256, NB == 1 ? 2 : 1) conv3x3_wino16_kernel(const ConvKArgs a) {
    constexpr int TILES = 64 * NB;                    // patches per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = NJ * 512;                   // floats per channel of the raw halo rows
    constexpr int RST = KC * PLANE;                   // one raw stage
    float* const RAW = smem;                          // [2][KC][PLANE]
    float* const BIASL = smem + 2 * RST;              // [16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int W = a.W, H = a.H, HW = H * W;           // W is even
    const int TCW = W >> 1, TRH = (H + 1) >> 1;
    const int ntiles = TCW * TRH;
    const int ROWPL = W + 6;                          // raw row: image column ix at ix + 3 (conv3x3_wino.hip)
    const int HPR = (W + 4) >> 1;

    int b, wt, cg_;
    scm_xcd_tile_map(a.B, a.tiles_per_img, b, wt, cg_);
    const int t0 = wt * TILES;
    const int ty0 = t0 / TCW;
    int tlast = t0 + TILES - 1;
    if (tlast > ntiles - 1) tlast = ntiles - 1;
    const int nrows = 2 * (tlast / TCW - ty0) + 4;
    const int npair = nrows * HPR;

    if (tid < 16) {
        const int coc = tid < a.cout ? tid : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        BIASL[tid] = v;
    }

    // staging map of the raw rows (aligned pixel pairs), as in conv3x3_wino.hip
    unsigned poff[NJ];
    int rawi[NJ];
    unsigned okmask = 0u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = tid + 256 * j;
        const int r = q / HPR;
        int g = 0;
        if (q < npair) {
            int iy = 2 * ty0 - 1 + r, ix = 2 * (q - r * HPR) - 2;
            bool ok = true;
            if (PADC) {
                iy = iy < 0 ? iy + H : (iy >= H ? iy - H : iy);
                ix = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
            } else {
                ok = (iy >= 0 && iy < H && ix >= 0 && ix < W);
            }
            iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= W ? W - 2 : ix);
            g = iy * W + ix;
            if (ok) okmask |= 1u << j;
        }
        poff[j] = (unsigned)g * 4u;
        rawi[j] = q < npair ? r * ROWPL + 2 * (q - r * HPR) + 1 : 0;
    }
    // this lane's two patches (blocks 0 / 1 of the wave): offsets of their 4x4 input patches in a raw plane, and
    // their output positions
    int traw[NB], p00[NB];
    bool tval[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int t = t0 + wave * (16 * NB) + 16 * k + l15;
        tval[k] = t < ntiles;
        const int tc = tval[k] ? t : ntiles - 1;
        const int ty = tc / TCW, tx = tc - ty * TCW;
        traw[k] = 2 * (ty - ty0) * ROWPL + 2 * tx + 2;
        p00[k] = 2 * ty * W + 2 * tx;
    }

    f32x4 acc[16][NB];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[f][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cin = a.c1 + a.c2;
    const int nchunk = (cin + KC - 1) / KC, nchunk_p = (nchunk + 1) & ~1;
    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gfp4;
    typedef const __attribute__((address_space(1))) f32x2* gfp2;
    const size_t plane_b = (size_t)HW * 4;
    gcp xb1 = (gcp)(a.x1) + (size_t)b * a.c1 * plane_b;
    gcp xb2 = a.c2 ? (gcp)(a.x2) + ((size_t)b * a.c2 - a.c1) * plane_b : xb1;      // indexed by the concatenated channel
    gcp ub = (gcp)(a.wpack);
    unsigned uoff = lane * 16;

    float ur[4][4];         // ring of A operands: position pair P of a chunk in ur[P & 3], requested 3 pairs ahead
    float bv[2][2 * NB][16];    // B operands of two chunks: [set][NB * k-step + block][position] (a transform writes 16 in a row)
    f32x2 dlo[4], dhi[4];   // the patch being transformed (read from LDS one MFMA group before it is used)
    float px[KC][NJ][2];    // raw halo pairs of the NEXT chunk to stage

    auto clampc = [&](int chunk) { return chunk < nchunk ? chunk : nchunk - 1; };
    auto vo = [](unsigned o) { asm volatile("" : "+v"(o)); return o; };
    auto u_load = [&](int chunk, auto p_) {
        constexpr int P = decltype(p_)::value;
        const int cu = chunk < nchunk_p ? chunk : nchunk_p - 1;      // (the pack has an even number of slabs, zero past cin)
        gcp base = ub + (unsigned)cu * (unsigned)(W16_USLAB * 4);
        const f32x4 t = *(gfp4)(base + uoff + P * 1024);
        ur[P & 3][0] = t[0]; ur[P & 3][1] = t[1]; ur[P & 3][2] = t[2]; ur[P & 3][3] = t[3];
    };
    gcp xrun = xb1;
    auto x_load = [&](int chunk, auto cc_) {
        constexpr int C = decltype(cc_)::value;
        const int ci0 = clampc(chunk) * KC;
        if constexpr (C == 0) xrun = (ci0 < a.c1 ? xb1 : xb2) + (size_t)ci0 * plane_b;
        else xrun += (ci0 + C < cin ? plane_b : 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x2 t = *(gfp2)(xrun + poff[j]);
            px[C][j][0] = t[0]; px[C][j][1] = t[1];
        }
    };
    auto x_store = [&](int st, auto cc_) {
        constexpr int C = decltype(cc_)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v0 = px[C][j][0], v1 = px[C][j][1];
            if constexpr (!PADC) {
                const bool ok = okmask & (1u << j);
                v0 = ok ? v0 : 0.f; v1 = ok ? v1 : 0.f;
            }
            float* rp = RAW + st * RST + C * PLANE + rawi[j];
            rp[0] = v0; rp[1] = v1;
        }
    };
    // B^T d B of the patch of block BK, local channel 4 * KS + lq, out of raw stage st -> bv[SET][NB * KS + BK][f]:
    // xf_rd requests the patch, xf_math (a couple of MFMAs later) transforms it
    auto xf_rd = [&](int st, auto ks_, auto bk_) {
        constexpr int KS = decltype(ks_)::value, BK = decltype(bk_)::value;
        const float* rp = RAW + st * RST + (4 * KS + lq) * PLANE + traw[BK];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dlo[i] = *reinterpret_cast<const f32x2*>(rp + i * ROWPL);
            dhi[i] = *reinterpret_cast<const f32x2*>(rp + i * ROWPL + 2);
        }
    };
    // (written on the register pairs the 8-byte LDS reads deliver: 8 packed operations along y, then per row two packed
    // operations along x whose operands are one pair and a broadcast half of the other)
    auto xf_math = [&](auto set_, auto ks_, auto bk_) {
        constexpr int SET = decltype(set_)::value, KS = decltype(ks_)::value, BK = decltype(bk_)::value;
        f32x2 tl[4], th[4];
        tl[0] = dlo[0] - dlo[2]; th[0] = dhi[0] - dhi[2];
        tl[1] = dlo[1] + dlo[2]; th[1] = dhi[1] + dhi[2];
        tl[2] = dlo[2] - dlo[1]; th[2] = dhi[2] - dhi[1];
        tl[3] = dlo[1] - dlo[3]; th[3] = dhi[1] - dhi[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 h00 = __builtin_shufflevector(th[i], th[i], 0, 0), l11 = __builtin_shufflevector(tl[i], tl[i], 1, 1);
            const f32x2 sg = {-1.f, 1.f};
            const f32x2 v01 = tl[i] + h00 * sg;                 // (t0 - t2, t1 + t2)
            const f32x2 v23 = th[i] - l11;                      // (t2 - t1, -(t1 - t3)): the pack carries column 3's sign
            bv[SET][NB * KS + BK][4 * i + 0] = v01[0]; bv[SET][NB * KS + BK][4 * i + 1] = v01[1];
            bv[SET][NB * KS + BK][4 * i + 2] = v23[0]; bv[SET][NB * KS + BK][4 * i + 3] = v23[1];
        }
    };
    auto xform = [&](int st, auto set_, auto ks_, auto bk_) { xf_rd(st, ks_, bk_); xf_math(set_, ks_, bk_); };
#define IC(n) std::integral_constant<int, (n)>{}
#define SB() __builtin_amdgcn_sched_barrier(0)
    // Position pair P (positions 2P, 2P+1) of chunk `chunk`, B set SET: 8 MFMAs; behind them this pair's share of the
    // staging work: P = 0..3 raw channels 2P, 2P+1 of chunk+1 into stage (chunk+1) & 1 and chunk+2's requested;
    // P = 4..7 (after the barrier) one of the four patch transforms of chunk+1 each.
    auto pair = [&](int chunk, auto set_, auto p_) {
        constexpr int SET = decltype(set_)::value, P = decltype(p_)::value;
        const int stn = (chunk + 1) & 1;
        // acc += A x B with the accumulator TIED to its AGPRs: hipcc leaves the destination of this 4-register MFMA untied
        // (unlike the 16-register ones) and its register allocator then permutes the accumulator tiles through
        // v_accvgpr_read / _mov / _write every chunk (40-64 of each per 128 MFMAs).  As inline asm the compiler no longer
        // sees an MFMA: the two hazards it would have padded are kept apart by construction -- a B operand is written
        // (xf_math, set SET ^ 1) a whole chunk before it is read, and the two products into one accumulator are three
        // MFMAs apart.
        auto mf = [&](auto f_, auto ks_, auto bk_) {
            constexpr int F = decltype(f_)::value, KS = decltype(ks_)::value, BK = decltype(bk_)::value;
            if constexpr (BK < NB) wino16_mfma(acc[F][BK < NB ? BK : 0], ur[P & 3][2 * (F & 1) + KS], bv[SET][NB * KS + (BK < NB ? BK : 0)][F]);
        };
        // the patch transform (k-step, block) this position pair carries behind its MFMAs: NB = 2: pairs 4..7 one each;
        // NB = 1: pairs 4 and 5
        constexpr int XI = P - 4;
        constexpr bool XF = P >= 4 && XI < 2 * NB;
        constexpr int XKS = XF ? (NB == 2 ? XI >> 1 : XI) : 0, XBK = XF ? (NB == 2 ? XI & 1 : 0) : 0;
        mf(IC(2 * P), IC(0), IC(0)); mf(IC(2 * P), IC(0), IC(1));
        if constexpr (P < 4) { x_store(stn, IC(P < 4 ? 2 * P : 0)); x_store(stn, IC(P < 4 ? 2 * P + 1 : 0)); }
        if constexpr (XF) xf_rd(stn, IC(XKS), IC(XBK));
        SB();
        mf(IC(2 * P + 1), IC(0), IC(0)); mf(IC(2 * P + 1), IC(0), IC(1));
        if constexpr (P < 4) { x_load(chunk + 2, IC(P < 4 ? 2 * P : 0)); x_load(chunk + 2, IC(P < 4 ? 2 * P + 1 : 0)); }
        SB();
        mf(IC(2 * P), IC(1), IC(0)); mf(IC(2 * P), IC(1), IC(1));
        if constexpr (XF) xf_math(IC(SET ^ 1), IC(XKS), IC(XBK));
        SB();
        mf(IC(2 * P + 1), IC(1), IC(0)); mf(IC(2 * P + 1), IC(1), IC(1));
        // ring slot of pair P + 3 (of this chunk or the next) was last read by pair P - 1
        if constexpr (P + 3 < 8) u_load(chunk, IC(P + 3 < 8 ? P + 3 : 0));
        else u_load(chunk + 1, IC(P + 3 >= 8 ? P + 3 - 8 : 0));
        SB();
    };
    auto chunk_body = [&](int chunk, auto set_) {
        // (the lane offsets are laundered in place once per chunk: their zero-extension then stays inside the loop, next to
        // the loads, and folds into the scalar-base addressing mode -- without a register copy per load)
        asm volatile("" : "+v"(uoff));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(poff[j]));
        pair(chunk, set_, IC(0)); pair(chunk, set_, IC(1)); pair(chunk, set_, IC(2)); pair(chunk, set_, IC(3));
        __syncthreads();            // raw stage (chunk+1) & 1 is complete; nobody reads the other one any more
        SB();
        pair(chunk, set_, IC(4)); pair(chunk, set_, IC(5)); pair(chunk, set_, IC(6)); pair(chunk, set_, IC(7));
    };

    // ---- prologue: chunk 0 staged and transformed (set 0), chunk 1 requested, the A ring primed
    x_load(0, IC(0)); x_load(0, IC(1)); x_load(0, IC(2)); x_load(0, IC(3));
    x_load(0, IC(4)); x_load(0, IC(5)); x_load(0, IC(6)); x_load(0, IC(7));
    u_load(0, IC(0)); u_load(0, IC(1)); u_load(0, IC(2));
    x_store(0, IC(0)); x_store(0, IC(1)); x_store(0, IC(2)); x_store(0, IC(3));
    x_store(0, IC(4)); x_store(0, IC(5)); x_store(0, IC(6)); x_store(0, IC(7));
    __syncthreads();
    x_load(1, IC(0)); x_load(1, IC(1)); x_load(1, IC(2)); x_load(1, IC(3));
    x_load(1, IC(4)); x_load(1, IC(5)); x_load(1, IC(6)); x_load(1, IC(7));
    xform(0, IC(0), IC(0), IC(0)); xform(0, IC(0), IC(1), IC(0));
    if constexpr (NB == 2) { xform(0, IC(0), IC(0), IC(NB - 1)); xform(0, IC(0), IC(1), IC(NB - 1)); }
    // (whole pairs of chunks: the accumulators must not cross a conditional.  A chunk past the end multiplies the
    // pack's zero slab by patches of the clamped last raw chunk.)
    SB();
    asm volatile("s_nop 1");          // (set 0 was written by the VALU just above: two wait states before an MFMA reads it)
    SB();
    for (int chunk = 0; chunk < nchunk; chunk += 2) {
        chunk_body(chunk, IC(0));
        chunk_body(chunk + 1, IC(1));
    }
    // (the wait states between the last MFMAs and the first read of a result, which the compiler cannot count any more)
    SB();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    SB();
#undef IC
#undef SB

    // ---- epilogue: output transform, scale, bias, activation, NCHW store (8-byte pairs)
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;
    const float slope = scm_act_slope(a.act);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int oy = p00[k] / W;
        const bool row1 = oy + 1 < H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * lq + r;
            float m[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) m[f] = acc[f][k][r];
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = m[j] + m[4 + j] + m[8 + j];
                s1[j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
            float o[4];
            o[0] = s0[0] + s0[1] + s0[2];
            o[1] = s0[1] - s0[2] - s0[3];
            o[2] = s1[0] + s1[1] + s1[2];
            o[3] = s1[1] - s1[2] - s1[3];
            const float badd = BIASL[co];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float u = fmaf(o[q], sc, badd);
                o[q] = sigmoid ? scm_sigmoid(u) : fmaxf(u, slope * u);
            }
            if (co < a.cout && tval[k]) {
                float* dst = a.y1 + ((size_t)b * a.cout + co) * HW + p00[k];
                f32x2 v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
                *reinterpret_cast<f32x2*>(dst) = v0;
                if (row1) *reinterpret_cast<f32x2*>(dst + W) = v1;
            }
        }
    }
}

// ------------------------------------------------------------------ pack
// [chunk c][position pair P][lane = 16 * kq + co][4 = (pos 2P, k-step 0), (2P, 1), (2P+1, 0), (2P+1, 1)],
// ci = 8c + 4 * k-step + kq; U = G g G^T; missing channels are zero.
__global__ void pack_conv3x3_wino16_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src, int wct,
                                           int wc0, float* __restrict__ fwd, float* __restrict__ dgr) {
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    auto emit = [&](float* out, int n_in, int n_out, bool dgrad) {
        const int nchunk = 2 * ((n_in + 2 * KC - 1) / (2 * KC));     // an even number of slabs: the kernel runs chunk pairs
        if (idx >= nchunk * 128) return;            // one thread per (chunk, local channel, co)
        const int co = idx & 15, cl = (idx >> 4) & 7, c = idx >> 7;
        const int ci = c * KC + cl, ks = cl >> 2, kq = cl & 3;
        float gk[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = 0.f;
            if (ci < n_in && co < n_out) v = dgrad ? Wc(ci, co, 8 - t) : Wc(co, ci, t);
            gk[t / 3][t % 3] = v;
        }
        float gg[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gg[0][k] = gk[0][k];
            gg[1][k] = 0.5f * (gk[0][k] + gk[1][k] + gk[2][k]);
            gg[2][k] = 0.5f * (gk[0][k] - gk[1][k] + gk[2][k]);
            gg[3][k] = gk[2][k];
        }
        float* base = out + (size_t)c * W16_USLAB + (16 * kq + co) * 4 + ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (column 3 negated: the kernel's input transform produces -(B^T d B) there, one packed subtraction cheaper)
            const float u4[4] = {gg[i][0], 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]), 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]), -gg[i][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * i + j;
                base[(f >> 1) * 256 + (f & 1) * 2] = u4[j];
            }
        }
    };
    if (fwd) emit(fwd, cin, cout, false);
    if (dgr) emit(dgr, cout, cin, true);
}

size_t scm_wino16_pack_floats(int cin) { return (size_t)2 * ((cin + 2 * KC - 1) / (2 * KC)) * W16_USLAB; }

int scm_pack_wino16(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0, float* fwd_pack,
                    float* dgrad_pack, hipStream_t s) {
    // fwd_pack: cout <= 16 outputs from cin inputs; dgrad_pack: cin <= 16 outputs from cout inputs
    size_t n = 0;
    if (fwd_pack) n = scm_wino16_pack_floats(cin) / 16;
    if (dgrad_pack) {
        const size_t m = scm_wino16_pack_floats(cout) / 16;
        if (m > n) n = m;
    }
    hipLaunchKernelGGL(pack_conv3x3_wino16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, fwd_pack, dgrad_pack);
    SCM_LAUNCH_CHECK("pack_conv3x3_wino16_kernel");
    return SCM_OK;
}

static inline int w16_plane(int H, int W, int tiles = W16_TILES) {
    const int tcw = W / 2, trh = (H + 1) / 2;
    int trows = (tiles % tcw == 0) ? tiles / tcw : (tiles + tcw - 2) / tcw + 1;
    if (trows > trh) trows = trh;
    return (2 * trows + 2) * (W + 6);
}

// (three pixel pairs per thread and channel at most: 84 x 84 frames -- BASELINE configs[3] -- need 1 080 floats per channel)
bool scm_wino16_serves(int H, int W) { return H >= 2 && W >= 2 && (W & 1) == 0 && w16_plane(H, W) <= 3 * 512; }

template <int NJ, int PADC, int NB = 2>
static int launch_w16(const ConvKArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)2 * KC * NJ * 512 + 16) * sizeof(float);
    auto kern = conv3x3_wino16_kernel<NJ, PADC, NB>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_img), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wino16_kernel");
    return SCM_OK;
}

// cout <= 16, one output tensor, no input gate, no accumulate (checked by scm_conv3x3_wino)
int scm_launch_wino16(const ConvKArgs& k, hipStream_t s) {
    ConvKArgs a = k;
    const int ntiles = ((a.H + 1) / 2) * (a.W / 2);
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    // 64-patch workgroups (one block per wave, two or three workgroups per CU) when their halo rows fit one pixel pair per
    // thread and the grid is large enough to fill the chip that way; SCM_WINO16_NB2=1: the 128-patch form only (A/B)
    static const bool nb2_only = getenv("SCM_WINO16_NB2") && atoi(getenv("SCM_WINO16_NB2")) != 0;
    if (!nb2_only && w16_plane(a.H, a.W, 64) <= 512 && (long)a.B * ((ntiles + 63) / 64) >= 512) {
        a.tiles_per_img = (ntiles + 63) / 64;
        a.plane = w16_plane(a.H, a.W, 64);
        return circ ? launch_w16<1, 1, 1>(a, s) : launch_w16<1, 0, 1>(a, s);
    }
    a.tiles_per_img = (ntiles + W16_TILES - 1) / W16_TILES;
    a.plane = w16_plane(a.H, a.W);
    if (a.plane <= 512) return circ ? launch_w16<1, 1>(a, s) : launch_w16<1, 0>(a, s);
    if (a.plane <= 1024) return circ ? launch_w16<2, 1>(a, s) : launch_w16<2, 0>(a, s);
    return circ ? launch_w16<3, 1>(a, s) : launch_w16<3, 0>(a, s);
}
