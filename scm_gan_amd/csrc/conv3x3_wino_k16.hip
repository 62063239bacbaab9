// Winograd F(2x2, 3x3) for layers with at most 16 INPUT channels and many output channels, on v_mfma_f32_16x16x4_f32:
// Transition.conv1's forward (16 -> 128) and Transition.conv6's input gradient (16 -> 256, sigmoid-gated, two gated
// output tensors) -- reference /root/reference/models.py:76,102 and the autograd backward of :102.  Same operator,
// descriptor and epilogue contract as scm_conv3x3_wino, which dispatches here.
//
// In the wide kernel (conv3x3_wino.hip) such a launch is two K chunks between a prologue and an epilogue made for
// sixteen (36 % of its workgroup's cycles are MFMAs).  Here the roles of the loops are swapped: the whole reduction fits
// one instruction group, so a workgroup transforms its patches ONCE and walks the OUTPUT channels:
//   M_f[co][patch] = sum_{ci < 16} U_f[co][ci] V_f[ci][patch]
//   MFMA A = U   A[i = co (lane & 15)][k = ci (lane >> 4)]      four input channels per instruction, four k-steps
//   MFMA B = V   B[k = ci (lane >> 4)][j = patch (lane & 15)]
//   D[i][j]: lane & 15 -> patch, row = 4 * (lane >> 4) + reg -> co
// Workgroup = 4 waves = 64 patches (256 output pixels); wave w owns the 16 patches [16w, 16w + 16).
// The raw rows of all 16 channels (gated by act'(gate) if the launch is an input gradient) pass through LDS once; the
// thread that transforms a patch is the lane that feeds it to the MFMA (conv3x3_wino16.hip): 4 k-steps x 16
// positions = 64 B-operand registers that stay for the whole workgroup.  Then, per tile of 16 output channels: 16
// 16-byte loads of transformed filters (pack [tile][k-step][position quad][lane][4], L2-resident, requested a k-step ahead), 64 MFMAs (the first of
// each accumulator chain takes the constant 0 as C: nothing to clear), output transform, scale, bias, activation, 1-bit
// gate words (two tiles make a 32-channel word), output gates, store.  The next tile's filters are requested before the
// epilogue of the current one.  MFMAs as inline asm with tied accumulators (conv3x3_wgrad_wino16.hip explains why);
// operands come from loads or from transforms finished long before, the products into one accumulator are 16
// MFMAs apart.
#include "common.h"
#include <type_traits>

#include "conv_args.h"

#define K16_NB 1            // 16-patch blocks per wave (2: 128 B-operand registers -- the epilogue then spills into AGPR copies)
#define K16_TILES (64 * K16_NB)
#define K16_SLAB 4096       // floats of one output-channel tile of the filter pack: [4 k-steps][4 position quads][64 lanes][4]

typedef unsigned u32x2k __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void k16_mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void k16_mfma0(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

// GATED: the input is multiplied by act'(gate1) while it is staged (an input-gradient launch).  PADC: circular padding.
template <int NJ, int GATED, int PADC>
__global__ void __launch_bounds__(256, 1) conv3x3_wino_k16_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = NJ * 512;                   // floats per channel of the raw halo rows
    float* const RAW = smem;                          // [16][PLANE]
    float* const BIASL = smem + 16 * PLANE;           // [cout rounded up to 16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int W = a.W, H = a.H, HW = H * W;           // W is even
    const int TCW = W >> 1, TRH = (H + 1) >> 1;
    const int ntiles = TCW * TRH;
    const int ROWPL = W + 6;                          // raw row: image column ix at ix + 3 (conv3x3_wino.hip)
    const int HPR = (W + 4) >> 1;
    const int ncot = (a.cout + 15) >> 4;

    int b, wt, cg_;
    scm_xcd_tile_map(a.B, a.tiles_per_img, b, wt, cg_);
    b = __builtin_amdgcn_readfirstlane(b);          // (block-uniform by construction; said so for the scalar row bases below)
    wt = __builtin_amdgcn_readfirstlane(wt);
    const int t0 = wt * K16_TILES;
    const int ty0 = t0 / TCW;
    int tlast = t0 + K16_TILES - 1;
    if (tlast > ntiles - 1) tlast = ntiles - 1;
    const int nrows = 2 * (tlast / TCW - ty0) + 4;
    const int npair = nrows * HPR;

    for (int co = tid; co < ncot * 16; co += 256) {
        const int coc = co < a.cout ? co : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        BIASL[co] = v;
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
    constexpr int NB = K16_NB;
    int traw[NB], p00[NB];
    bool tval[NB], row1[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int t = t0 + wave * 16 * NB + 16 * k + l15;
        tval[k] = t < ntiles;
        const int tc = tval[k] ? t : ntiles - 1;
        const int ty = tc / TCW, tx = tc - ty * TCW;
        traw[k] = 2 * (ty - ty0) * ROWPL + 2 * tx + 2;
        p00[k] = 2 * ty * W + 2 * tx;
        row1[k] = 2 * ty + 1 < H;
    }

    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gfp4;
    typedef const __attribute__((address_space(1))) f32x2* gfp2;
    const size_t plane_b = (size_t)HW * 4;
    const int cin = a.c1;
    gcp xb = (gcp)(a.x1) + (size_t)b * cin * plane_b;
    gcp gb = GATED ? (gcp)(a.gate1) + (size_t)b * cin * plane_b : nullptr;
    const int gate_act = a.gate_act;

    // ---- the raw rows of all 16 channels -> LDS (a channel past cin re-reads the last one: its filters are zero)
    {
        f32x2 px[16][NJ], pg[GATED ? 16 : 1][GATED ? NJ : 1];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int cc = c < cin ? c : cin - 1;
            gcp xp = xb + (size_t)cc * plane_b;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                px[c][j] = *(gfp2)(xp + poff[j]);
                if constexpr (GATED != 0) pg[c][j] = *(gfp2)(gb + (size_t)cc * plane_b + poff[j]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x2 v = px[c][j];
                if constexpr (GATED != 0) {
                    const f32x2 g = pg[c][j];
                    if (gate_act == SCM_ACT_SIGMOID) v *= g * (1.f - g);
                    else if (gate_act == SCM_ACT_LRELU) v *= f32x2{g[0] > 0.f ? 1.f : 0.01f, g[1] > 0.f ? 1.f : 0.01f};
                }
                if constexpr (!PADC) {
                    const bool ok = okmask & (1u << j);
                    v = ok ? v : f32x2{0.f, 0.f};
                }
                float* rp = RAW + c * PLANE + rawi[j];
                rp[0] = v[0]; rp[1] = v[1];
            }
    }
    __syncthreads();

    // ---- B^T d B of this lane's patches: bv[k-step][block][position], channel 4 * k-step + lq (the pack carries the sign
    // of transform column 3, conv3x3_wino16.hip)
    float bv[4][NB][16];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int bk = 0; bk < NB; ++bk) {
            const float* rp = RAW + (4 * ks + lq) * PLANE + traw[bk];
            f32x2 dlo[4], dhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dlo[i] = *reinterpret_cast<const f32x2*>(rp + i * ROWPL);
                dhi[i] = *reinterpret_cast<const f32x2*>(rp + i * ROWPL + 2);
            }
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
                const f32x2 v23 = th[i] - l11;                      // (t2 - t1, -(t1 - t3))
                bv[ks][bk][4 * i + 0] = v01[0]; bv[ks][bk][4 * i + 1] = v01[1];
                bv[ks][bk][4 * i + 2] = v23[0]; bv[ks][bk][4 * i + 3] = v23[1];
            }
        }

    // ---- the output-channel tiles
    // (fast epilogue: every patch of the workgroup exists with both of its rows; 32-bit lane offsets of this lane's
    // output patches inside a block of 4 channel planes)
    const bool all_valid = t0 + K16_TILES <= ntiles && (H & 1) == 0 ;
    unsigned yoff[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) yoff[k] = (unsigned)(4 * lq) * (unsigned)plane_b + (unsigned)p00[k] * 4u;
    const float sc = a.scale ? *a.scale : 1.f;
    const float slope = scm_act_slope(a.act);
    const int co2n = a.cout - a.co1;
    const int nwo = (a.cout + 31) >> 5;               // mask_out words per pixel
    gcp ub = (gcp)(a.wpack);
    unsigned uoff = lane * 16;
    f32x4 ua[2][4];           // A operands of one k-step of a tile (16 positions), two buffers: k-step ks in ua[ks & 1]
    unsigned keep[NB][4];     // gate bits of the even tile of a 32-channel word
#pragma unroll
    for (int k = 0; k < NB; ++k) keep[k][0] = keep[k][1] = keep[k][2] = keep[k][3] = 0u;
#define SB() __builtin_amdgcn_sched_barrier(0)
    // the filters of k-step KS of tile ct: four 16-byte loads (positions 4 fq .. 4 fq + 3 each)
    auto u_load = [&](int ct, auto ks_) {
        constexpr int KS = decltype(ks_)::value;
        const int cu = ct < ncot ? ct : ncot - 1;    // (one tile past the end is requested and never used)
        gcp base = ub + (unsigned)cu * (unsigned)(K16_SLAB * 4) + KS * 4096;
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) ua[KS & 1][fq] = *(gfp4)(base + uoff + fq * 1024);
    };
    auto mf_pass = [&](f32x4 (&acc)[16][NB], auto ks_) {
        constexpr int KS = decltype(ks_)::value;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
#pragma unroll
            for (int bk = 0; bk < NB; ++bk) {
                if constexpr (KS == 0) k16_mfma0(acc[f][bk], ua[0][f >> 2][f & 3], bv[0][bk][f]);
                else k16_mfma(acc[f][bk], ua[KS & 1][f >> 2][f & 3], bv[KS][bk][f]);
            }
        }
    };
#define IC(n) std::integral_constant<int, (n)>{}
    u_load(0, IC(0));
    SB();
    asm volatile("s_nop 1");                          // (the last transform above is two wait states away from the first MFMA)
    for (int ct = 0; ct < ncot; ++ct) {
        asm volatile("" : "+v"(uoff));               // (the offset's zero-extension stays next to the loads)
        f32x4 acc[16][NB];
        // a k-step's filters are requested one k-step (32 MFMAs) ahead, the next tile's first ones before the epilogue
        u_load(ct, IC(1)); SB(); mf_pass(acc, IC(0)); SB();
        u_load(ct, IC(2)); SB(); mf_pass(acc, IC(1)); SB();
        u_load(ct, IC(3)); SB(); mf_pass(acc, IC(2)); SB();
        u_load(ct + 1, IC(0)); SB(); mf_pass(acc, IC(3)); SB();
        // (the wait states between the last MFMAs and the first read of a result, which the compiler cannot count)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        SB();

        // ---- epilogue of tile ct: lane -> patch l15 of each block, rows co = 16 * ct + 4 * lq + r.  Two compile-time
        // switches chosen once per tile (block-uniform): MO the launch writes 1-bit gate words, YG it gates what it writes.
        const int cobase = 16 * ct;
        const bool first = cobase < a.co1;             // (a tile lies in one output tensor: host check)
        float* ybase = first ? a.y1 + ((size_t)b * a.co1 + cobase) * HW : a.y2 + ((size_t)b * co2n + (cobase - a.co1)) * HW;
        const unsigned* gm = first ? a.ymask1 : a.ymask2;
        const int cd0 = first ? cobase : cobase - a.co1;          // channel within its tensor
        const bool addp = a.acc_y1 && first;
        auto epi = [&](auto mo_, auto yg_) {
            constexpr bool MO = decltype(mo_)::value, YG = decltype(yg_)::value;
            unsigned bits[NB][4];
#pragma unroll
            for (int k = 0; k < NB; ++k) bits[k][0] = bits[k][1] = bits[k][2] = bits[k][3] = 0u;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                unsigned gw[4] = {~0u, ~0u, ~0u, ~0u};
                if constexpr (YG) {
                    if (tval[k]) {
                        const int gnw = ((first ? a.co1 : co2n) + 31) >> 5;
                        const unsigned* gp = gm + ((size_t)b * gnw + (cd0 >> 5)) * HW + p00[k];
                        const u32x2k g0 = *reinterpret_cast<const u32x2k*>(gp);
                        gw[0] = g0[0]; gw[1] = g0[1];
                        if (row1[k]) { const u32x2k g1 = *reinterpret_cast<const u32x2k*>(gp + W); gw[2] = g1[0]; gw[3] = g1[1]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 4 * lq + r, co = cobase + col;
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
                    const unsigned gbit = 1u << ((cd0 + col) & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float u = fmaf(o[q], sc, badd);
                        o[q] = fmaxf(u, slope * u);          // none (slope 1) / leaky_relu (0.01): bitwise scm_act_sel
                        if constexpr (MO) bits[k][q] |= (o[q] > 0.f ? 1u : 0u) << (co & 31);
                        if constexpr (YG) o[q] = (gw[q] & gbit) ? o[q] : 0.01f * o[q];
                    }
                    if (co < a.cout && tval[k]) {
                        float* dst = ybase + (size_t)col * HW + p00[k];
                        f32x2 v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
                        if (addp) {
                            v0 += *reinterpret_cast<const f32x2*>(dst);
                            if (row1[k]) v1 += *reinterpret_cast<const f32x2*>(dst + W);
                        }
                        *reinterpret_cast<f32x2*>(dst) = v0;
                        if (row1[k]) *reinterpret_cast<f32x2*>(dst + W) = v1;
                    }
                }
            }
            if constexpr (MO) {       // the 1-bit leaky-relu gate: a tile is 16 channels of a 32-channel word
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned v = bits[k][q];
                        v |= __shfl_xor(v, 16);
                        v |= __shfl_xor(v, 32);
                        bits[k][q] = v;
                    }
                const bool odd = ct & 1, last = ct == ncot - 1;
                if (odd || last) {
                    if (lq == 0) {
#pragma unroll
                        for (int k = 0; k < NB; ++k) {
                            if (!tval[k]) continue;
                            unsigned* dst = a.mask_out + ((size_t)b * nwo + (ct >> 1)) * HW + p00[k];
                            const unsigned w0 = bits[k][0] | (odd ? keep[k][0] : 0u), w1 = bits[k][1] | (odd ? keep[k][1] : 0u);
                            const unsigned w2 = bits[k][2] | (odd ? keep[k][2] : 0u), w3 = bits[k][3] | (odd ? keep[k][3] : 0u);
                            *reinterpret_cast<u32x2k*>(dst) = u32x2k{w0, w1};
                            if (row1[k]) *reinterpret_cast<u32x2k*>(dst + W) = u32x2k{w2, w3};
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) keep[k][q] = bits[k][q];
                }
            }
        };
        // Fast path (block-uniform; every launch of the training step takes it): all 64 patches of the workgroup exist
        // with both rows, the tile's 16 channels exist, nothing is accumulated.  As in conv3x3_wino.hip's: scalar row
        // bases + one 32-bit lane offset (no per-store 64-bit address arithmetic), no per-value predicates or divergent
        // branches, the output transform on register pairs -- the general path above held ~1 200 instructions per tile of
        // 64 MFMAs (tools/isa_loop_stats.py), 47 of them divergent branches.
        auto epi_fast = [&](auto mo_, auto yg_) {
            constexpr bool MO = decltype(mo_)::value, YG = decltype(yg_)::value;
            typedef __attribute__((address_space(1))) char* gwp;
            typedef __attribute__((address_space(1))) f32x2* gw2;
            typedef const __attribute__((address_space(1))) u32x2k* ggp;
            // (the tile's base address is block-uniform; the compiler keeps the general path's copy of it in VGPRs, so it
            // is handed to the scalar unit explicitly: two v_readfirstlane per tile)
            const unsigned long long ybv = (unsigned long long)(first ? a.y1 + ((size_t)b * a.co1 + cobase) * HW
                                                                      : a.y2 + ((size_t)b * co2n + (cobase - a.co1)) * HW);
            // (readfirstlane returns a signed int: through `unsigned`, or an address with bit 31 set would sign-extend)
            const unsigned yb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ybv >> 32));
            const unsigned yb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ybv);
            gwp yb = (gwp)(((unsigned long long)yb_hi << 32) | (unsigned long long)yb_lo);
            const unsigned wrow = (unsigned)W * 4u;
            f32x2 sc2, sl2;
            sc2[0] = sc; sc2[1] = sc; sl2[0] = slope; sl2[1] = slope;
            unsigned bits[NB][4];
#pragma unroll
            for (int k = 0; k < NB; ++k) bits[k][0] = bits[k][1] = bits[k][2] = bits[k][3] = 0u;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                unsigned gw[4] = {~0u, ~0u, ~0u, ~0u};
                if constexpr (YG) {
                    const int gnw = ((first ? a.co1 : co2n) + 31) >> 5;
                    ggp gp = (ggp)(gm + ((size_t)b * gnw + (cd0 >> 5)) * HW);
                    const u32x2k g0 = *(ggp)((const __attribute__((address_space(1))) char*)gp + (unsigned)p00[k] * 4u);
                    const u32x2k g1 = *(ggp)((const __attribute__((address_space(1))) char*)gp + (unsigned)p00[k] * 4u + wrow);
                    gw[0] = g0[0]; gw[1] = g0[1]; gw[2] = g1[0]; gw[3] = g1[1];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 4 * lq + r, co = cobase + col;
                    f32x2 m03[4], m12[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        m03[i][0] = acc[4 * i][k][r]; m03[i][1] = acc[4 * i + 3][k][r];
                        m12[i][0] = acc[4 * i + 1][k][r]; m12[i][1] = acc[4 * i + 2][k][r];
                    }
                    const f32x2 s0_03 = scm_pk_add(scm_pk_add(m03[0], m03[1]), m03[2]);
                    const f32x2 s0_12 = scm_pk_add(scm_pk_add(m12[0], m12[1]), m12[2]);
                    const f32x2 s1_03 = scm_pk_sub(scm_pk_sub(m03[1], m03[2]), m03[3]);
                    const f32x2 s1_12 = scm_pk_sub(scm_pk_sub(m12[1], m12[2]), m12[3]);
                    f32x2 oa = scm_pk_addsub(scm_pk_sumdiff(s0_12), s0_03);       // image row oy:     (o0, o1)
                    f32x2 ob = scm_pk_addsub(scm_pk_sumdiff(s1_12), s1_03);       // image row oy + 1: (o2, o3)
                    f32x2 bb;
                    bb[0] = BIASL[co]; bb[1] = bb[0];
                    oa = scm_pk_fma(oa, sc2, bb);
                    ob = scm_pk_fma(ob, sc2, bb);
                    const f32x2 la = scm_pk_mul(oa, sl2), lb = scm_pk_mul(ob, sl2);
                    float o[4];
                    o[0] = fmaxf(oa[0], la[0]); o[1] = fmaxf(oa[1], la[1]);      // none (slope 1) / leaky_relu (0.01)
                    o[2] = fmaxf(ob[0], lb[0]); o[3] = fmaxf(ob[1], lb[1]);
                    const unsigned gbit = 1u << ((cd0 + col) & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (MO) bits[k][q] |= (o[q] > 0.f ? 1u : 0u) << (co & 31);
                        if constexpr (YG) o[q] = (gw[q] & gbit) ? o[q] : 0.01f * o[q];
                    }
                    gwp rb = yb + (size_t)r * plane_b;
                    asm volatile("" : "+s"(rb));
                    const f32x2 v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
                    *(gw2)(rb + yoff[k]) = v0;
                    *(gw2)(rb + wrow + yoff[k]) = v1;
                }
            }
            if constexpr (MO) {       // the 1-bit leaky-relu gate: a tile is 16 channels of a 32-channel word
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned v = bits[k][q];
                        v |= __shfl_xor(v, 16);
                        v |= __shfl_xor(v, 32);
                        bits[k][q] = v;
                    }
                const bool odd = ct & 1, last = ct == ncot - 1;
                if (odd || last) {
                    if (lq == 0) {
#pragma unroll
                        for (int k = 0; k < NB; ++k) {
                            unsigned* dst = a.mask_out + ((size_t)b * nwo + (ct >> 1)) * HW + p00[k];
                            const unsigned w0 = bits[k][0] | (odd ? keep[k][0] : 0u), w1 = bits[k][1] | (odd ? keep[k][1] : 0u);
                            const unsigned w2 = bits[k][2] | (odd ? keep[k][2] : 0u), w3 = bits[k][3] | (odd ? keep[k][3] : 0u);
                            *reinterpret_cast<u32x2k*>(dst) = u32x2k{w0, w1};
                            *reinterpret_cast<u32x2k*>(dst + W) = u32x2k{w2, w3};
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) keep[k][q] = bits[k][q];
                }
            }
        };
        const bool fast = all_valid && !addp && cobase + 16 <= a.cout;
        if (fast) {
            if (a.mask_out) { if (gm) epi_fast(std::true_type{}, std::true_type{}); else epi_fast(std::true_type{}, std::false_type{}); }
            else { if (gm) epi_fast(std::false_type{}, std::true_type{}); else epi_fast(std::false_type{}, std::false_type{}); }
        } else {
            if (a.mask_out) { if (gm) epi(std::true_type{}, std::true_type{}); else epi(std::true_type{}, std::false_type{}); }
            else { if (gm) epi(std::false_type{}, std::true_type{}); else epi(std::false_type{}, std::false_type{}); }
        }
    }
#undef SB
#undef IC
}

// ------------------------------------------------------------------ pack
// [tile ct][k-step ks][position quad f >> 2][lane = 16 * kq + i][f & 3]: U_f[co = 16 * ct + i][ci = 4 * ks + kq], U = G g G^T with transform
// column 3 negated (the kernel's input transform produces -(B^T d B) there); missing channels are zero.
__global__ void pack_conv3x3_wino_k16_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src, int wct,
                                             int wc0, float* __restrict__ fwd, float* __restrict__ dgr) {
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    auto emit = [&](float* out, int n_in, int n_out, bool dgrad) {
        const int nct = (n_out + 15) / 16;
        if (idx >= nct * 256) return;               // one thread per (tile, co in tile, ci)
        const int ci = idx & 15, i = (idx >> 4) & 15, ct = idx >> 8;
        const int co = 16 * ct + i, ks = ci >> 2, kq = ci & 3;
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
        float* base = out + (size_t)ct * K16_SLAB + ks * 1024 + (16 * kq + i) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u4[4] = {gg[r][0], 0.5f * (gg[r][0] + gg[r][1] + gg[r][2]), 0.5f * (gg[r][0] - gg[r][1] + gg[r][2]), -gg[r][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) base[r * 256 + j] = u4[j];
        }
    };
    if (fwd) emit(fwd, cin, cout, false);
    if (dgr) emit(dgr, cout, cin, true);
}

size_t scm_wino_k16_pack_floats(int n_out) { return (size_t)((n_out + 15) / 16) * K16_SLAB; }

// fwd_pack: cin <= 16 inputs, cout outputs; dgrad_pack: cout <= 16 inputs, cin outputs
int scm_pack_wino_k16(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0, float* fwd_pack,
                      float* dgrad_pack, hipStream_t s) {
    size_t n = 0;
    if (fwd_pack) n = (size_t)((cout + 15) / 16) * 256;
    if (dgrad_pack) {
        const size_t m = (size_t)((cin + 15) / 16) * 256;
        if (m > n) n = m;
    }
    hipLaunchKernelGGL(pack_conv3x3_wino_k16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, fwd_pack, dgrad_pack);
    SCM_LAUNCH_CHECK("pack_conv3x3_wino_k16_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------------ host
static inline int k16_plane(int H, int W) {
    const int tcw = W / 2, trh = (H + 1) / 2;
    int trows = (K16_TILES % tcw == 0) ? K16_TILES / tcw : (K16_TILES + tcw - 2) / tcw + 1;
    if (trows > trh) trows = trh;
    return (2 * trows + 2) * (W + 6);
}

// What the kernel takes (anything else stays on the wide kernel, whose pack the builder writes as well): one input tensor
// of at most 16 channels, no 1-bit input gate, no sigmoid epilogue, output tensors split at a multiple of 16, frames whose
// 128-patch row groups fit the raw stage, 8-byte aligned tensors.
bool scm_wino_k16_serves(const ConvKArgs& a) {
    static const bool off = getenv("SCM_NO_WINO_K16") != nullptr && getenv("SCM_NO_WINO_K16")[0] == '1';
    if (off) return false;
    if (a.c2 != 0 || a.c1 > 16 || a.cout <= 16 || a.mask_in != nullptr || a.act == SCM_ACT_SIGMOID) return false;
    if (a.co1 != a.cout && (a.co1 & 15) != 0) return false;
    if (a.H < 2 || a.W < 2 || (a.W & 1) || k16_plane(a.H, a.W) > 2 * 512) return false;
    const size_t al = reinterpret_cast<size_t>(a.x1) | reinterpret_cast<size_t>(a.gate1) | reinterpret_cast<size_t>(a.y1) |
                      reinterpret_cast<size_t>(a.y2) | reinterpret_cast<size_t>(a.mask_out) |
                      reinterpret_cast<size_t>(a.ymask1) | reinterpret_cast<size_t>(a.ymask2);
    return (al & 7) == 0;
}

template <int NJ, int GATED, int PADC>
static int launch_k16(const ConvKArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)16 * NJ * 512 + ((a.cout + 15) & ~15)) * sizeof(float);
    auto kern = conv3x3_wino_k16_kernel<NJ, GATED, PADC>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    if (lds > 160 * 1024) {
        scm_set_error("scm_conv3x3_wino: %d output channels need %zu bytes of LDS in the 16-input-channel form", a.cout, lds);
        return SCM_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_img), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wino_k16_kernel");
    return SCM_OK;
}

int scm_launch_wino_k16(const ConvKArgs& k, hipStream_t s) {
    ConvKArgs a = k;
    const int ntiles = ((a.H + 1) / 2) * (a.W / 2);
    a.tiles_per_img = (ntiles + K16_TILES - 1) / K16_TILES;
    a.plane = k16_plane(a.H, a.W);
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const bool gated = a.gate_act != 0;
#define K16_P(NJ_, G_) (circ ? launch_k16<NJ_, G_, 1>(a, s) : launch_k16<NJ_, G_, 0>(a, s))
    if (a.plane <= 512) return gated ? K16_P(1, 1) : K16_P(1, 0);
    return gated ? K16_P(2, 1) : K16_P(2, 0);
#undef K16_P
}
