// Two tiles per workgroup: the hot instantiation of conv3x3_wino.hip (128 output channels x 32 patches per tile, one raw
// pixel pair per thread, ungated input, the fast epilogue) for frames whose patch rows are 32 patches wide (W = 64), with
// every workgroup taking TWO tiles of an image (half an image apart) one after the other.  What that buys (DESIGN.md 4i:
// setup 1.8 k + prologue 2.5 k + gap 0.5 k ticks per 84 k-tick tile): one kernel-argument wait, one tile map and one set of
// staging maps per pair -- the second tile's maps are the first's moved down half an image -- and the second tile's first
// rows and filters are requested BEFORE the first tile's epilogue, so their latency passes under its 6.5 k ticks.
// Same pack, same arithmetic order per output as conv3x3_wino_kernel<1, 0, PADC, 128>: bitwise the same results.
// Reference: /root/reference/models.py:76-102 (Transition conv2..conv5), :145-155 (Encoder conv2 / conv3) and their
// input-gradient convolutions, as conv3x3_wino.hip.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_args.h"

#define WINO_SLAB 8192    // floats of one (co group, K chunk) weight slab: [16][2][64][4] (conv3x3_wino.hip's pack)
#ifndef WINO_UR
#define WINO_UR 8
#define WINO_UD 6
#endif
#define WINO_ABL 0
#define WINO_TICK(i)

typedef unsigned u32x2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 wino_pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wino_pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wino_pk_sumdiff(f32x2 p) {          // (p.x + p.y, p.x - p.y)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 wino_pk_addsub(f32x2 t, f32x2 p) {   // (t.x + p.x, t.y - p.y)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(t), "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 wino_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 wino_pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int PADC>
__global__ void __launch_bounds__(256, 1) conv3x3_wino_t2_kernel(const ConvKArgs a) {
    constexpr int NJ = 1, GATED = 0, CW = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = NJ * 512;                   // floats per channel of the raw halo rows (NJ pairs per thread)
    float* const RAW = smem;                          // [KC][PLANE]
    constexpr int TW = CW == 64 ? 64 : 32;            // patches per workgroup
    constexpr int VST = 128 * TW;                     // floats of one V stage [16][2][TW][4]
    float* const VB = smem + KC * PLANE;              // two stages of V
    float* const BIASL = VB + 2 * VST;                // [CW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wn = CW == 64 ? (wave & 1) : 0;         // patch half of the wave (CW 64); its 32 channels: co_w below
    const int W = a.W, H = a.H, HW = H * W;           // W is even (scm_conv3x3_wino_serves)
    const int TCW = W >> 1, TRH = (H + 1) >> 1;
    const int ntiles = TCW * TRH;
    const int ROWPL = W + 6;                          // raw row: image column ix sits at ix + 3 (ix = -2 .. W+1; slot 0 unused):
                                                      // global pairs stay 8-byte aligned, patch reads (from ix = 2*tx-1) too

    // (host: TCW == 32 -- a tile is one patch row -- H % 4 == 0, a.tiles_per_img = TILE PAIRS per image)
    int b, wtp, cg;
    scm_xcd_tile_map(a.B, a.tiles_per_img, b, wtp, cg);
    // The pair: tile wtp and the tile HALF AN IMAGE below it.  Workgroups that run at the same time on an XCD then work on
    // vertically adjacent tiles (both their first, later both their second) and meet in its L2 for the two halo rows they
    // share, as in the one-tile kernel; with (2 wtp, 2 wtp + 1) the neighbour's rows were requested 80 k ticks apart: PMC
    // 477 instead of 358 MB of HBM traffic per launch.
    const int k0 = wtp;                               // first tile (= patch row) of this workgroup's pair
    const int kstep = a.tiles_per_img;                // ... and the second one: k0 + (tile pairs per image)
    (void)ntiles;
    const int HPR = (W + 4) >> 1;                     // pairs per raw row
    const int npair = 4 * HPR;                        // a tile's four raw rows

    // this wave's 32 output channels: 64-channel group cgw of the filter pack, half wm of it
    const int cgw = CW == 64 ? cg : 2 * cg + (wave >> 1);
    const int wm = CW == 64 ? (wave >> 1) : (wave & 1);
    const int co_w = cgw * 64 + wm * 32, co_l = co_w - cg * CW;
    if (tid < CW) {           // bias table for the epilogue (visible after the main loop's barriers)
        const int co = cg * CW + tid;
        const int coc = co < a.cout ? co : a.cout - 1;
        float v = 0.f;
        if (a.bias) v += a.bias[coc];
        if (a.sbias) v += a.sbias[(size_t)b * a.cout + coc];
        BIASL[tid] = v;
    }

    // staging map: the raw rows travel as aligned PAIRS of pixels (one 8-byte load per thread, channel and pair):
    // LDS pair q = tid + 256*j holds raw columns (c, c+1), c = 2*(q mod ROWPL/2), of raw row q / (ROWPL/2), i.e.
    // image columns ix0 = c-2, ix0+1.  The pair left of the image (-2,-1) only matters for ix = -1 and the pair right
    // of it (W, W+1) only for ix = W: under circular padding they are the image's last / first pair, under zero
    // padding they are masked.  poff[j]: byte offset of the source pair in a channel plane (always a real pair);
    // bit j of okmask: written as loaded; otherwise written as 0.
    unsigned poff[NJ];
    int rawi[NJ];            // where the pair lands in a raw plane (odd index: two 4-byte stores in one instruction)
    unsigned okmask = 0u;
    const int sq_r = tid / HPR, sq_c = tid - sq_r * HPR;        // this thread's pair: raw row, pair column (one division per pair of tiles)
    const bool sq_live = tid < npair;
    rawi[0] = sq_live ? sq_r * ROWPL + 2 * sq_c + 1 : 0;        // (slots 0, 1 are never read)
    // staging map of tile k: the second tile's is the first's moved down half an image (no division, no tile map)
    auto set_tile = [&](int k) __attribute__((always_inline)) {
        int g = 0;
        okmask = 0u;
        if (sq_live) {
            int iy = 2 * k - 1 + sq_r, ix = 2 * sq_c - 2;
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
            if (ok) okmask = 1u;
        }
        poff[0] = (unsigned)g * 4u;
    };
    set_tile(k0);

    // this thread's transform item: patch tid & 31 of the tile's row x one channel of the chunk
    const int traw = 2 * (tid & 31) + 2;      // offset of its 4x4 patch (raw slot 2*tx + 2 = image column 2*tx - 1, even) in a raw plane

    f32x16 acc[16];          // (first written by chunk 0's MFMAs: slot<FIRST>)

    const int cin = a.c1 + a.c2;
    const int nchunk = (cin + KC - 1) / KC;
    const int gate_act = GATED ? a.gate_act : 0;
    const int mask_nw = (a.c1 + 31) >> 5;

    // Scalar (SGPR) base pointers + 32-bit per-lane byte offsets: every load is `global_load vdst, voff, s[base]` with
    // no vector address arithmetic (VALU instructions do not overlap the f32 MFMAs: every one of them is paid for).
    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gfp4;
    typedef const __attribute__((address_space(1))) f32x2* gfp2;
    typedef const __attribute__((address_space(1))) u32x2w* gup2;
    const size_t plane_b = (size_t)HW * 4;
    gcp xb1 = (gcp)(a.x1) + (size_t)b * a.c1 * plane_b;
    gcp xb2 = a.c2 ? (gcp)(a.x2) + ((size_t)b * a.c2 - a.c1) * plane_b : xb1;      // indexed by the concatenated channel
    gcp gb1 = GATED == 1 ? (gcp)(a.gate1) + (size_t)b * a.c1 * plane_b : nullptr;
    gcp gb2 = (GATED == 1 && a.c2) ? (gcp)(a.gate2) + ((size_t)b * a.c2 - a.c1) * plane_b : gb1;
    gcp mb = GATED == 2 ? (gcp)(a.mask_in) + (size_t)b * mask_nw * plane_b : nullptr;
    // this wave's half (wm) of the co group's transformed filters: per chunk [16 positions][64 lanes] float4
    gcp ub = (gcp)(a.wpack) + ((size_t)cgw * nchunk * 2 + wm) * (WINO_SLAB / 2) * 4;
    unsigned uoff[4];       // lane * 16 + k * 4096: position F is read at uoff[F >> 2] + (F & 3) * 1024 (immediate)
#pragma unroll
    for (int k = 0; k < 4; ++k) uoff[k] = lane * 16 + k * 4096;

    // ---- registers that live across the K loop
    float ur[WINO_UR][4];   // ring of A operands (transformed filters), straight from global / L2: position F of a chunk
                            // sits in ur[F % WINO_UR], requested WINO_UD positions ahead (L2 latency under load: ~1 us)
    float bq[2][4];         // B operands (V) of the current and the next position
    float px[KC][NJ][2];    // raw halo pairs of the NEXT chunk (requested a whole iteration ahead)
    float pg[GATED == 1 ? KC : 1][GATED == 1 ? NJ : 1][2];
    unsigned pm[GATED == 2 ? NJ : 1][2];
    float vv[2][16];

    auto clampc = [&](int chunk) __attribute__((always_inline)) { return chunk < nchunk ? chunk : nchunk - 1; };
    // (32-bit lane offsets are laundered IN PLACE through an empty asm once per chunk -- launder() -- so that their
    // zero-extension stays inside the loop, next to the loads, where instruction selection folds it into the
    // scalar-base addressing mode; a per-load laundering copy costs a v_mov each)
    auto vo = [](unsigned o) { return o; };
    // A operands of position F of `chunk` (clamped: past the end the last slab is re-read and never used)
    auto u_load = [&](int chunk, auto fc) __attribute__((always_inline)) {
        constexpr int F = decltype(fc)::value;
        gcp base = ub + (unsigned)clampc(chunk) * (unsigned)(WINO_SLAB * 4);
        asm volatile("" : "+s"(base));
        const f32x4 t = *(gfp4)(base + vo(uoff[F >> 2]) + (F & 3) * 1024);
        constexpr int R = F % WINO_UR;
        ur[R][0] = t[0]; ur[R][1] = t[1]; ur[R][2] = t[2]; ur[R][3] = t[3];
    };
    // Raw halo pairs of local channel C of `chunk` -> px[C] (+ gate tensor pairs).  The channel pointer runs: set at
    // C = 0 (one 64-bit multiply per chunk), advanced by one plane per channel while the next one exists -- a channel
    // past cin re-reads the last real one: its filters are zero in the pack, what is loaded only has to be finite.
    gcp xrun = xb1, grun = gb1;
    auto x_load = [&](int chunk, auto cc_) __attribute__((always_inline)) {
        constexpr int C = decltype(cc_)::value;
        const int ci0 = clampc(chunk) * KC;
        if constexpr (C == 0) {
            xrun = (ci0 < a.c1 ? xb1 : xb2) + (size_t)ci0 * plane_b;      // (c1 % 8 == 0 when there is a second tensor)
            if constexpr (GATED == 1) grun = (ci0 < a.c1 ? gb1 : gb2) + (size_t)ci0 * plane_b;
        } else {
            const size_t step = ci0 + C < cin ? plane_b : 0;
            xrun += step;
            if constexpr (GATED == 1) grun += step;
        }
        asm volatile("" : "+s"(xrun));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x2 t = *(gfp2)(xrun + vo(poff[j]));
            px[C][j][0] = t[0]; px[C][j][1] = t[1];
        }
        if constexpr (GATED == 1) {
            asm volatile("" : "+s"(grun));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x2 t = *(gfp2)(grun + vo(poff[j]));
                pg[C][j][0] = t[0]; pg[C][j][1] = t[1];
            }
        }
    };
    auto m_load = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (GATED == 2) {
            gcp src = mb + (size_t)((clampc(chunk) * KC) >> 5) * plane_b;
            asm volatile("" : "+s"(src));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const u32x2w t = *(gup2)(src + vo(poff[j]));
                pm[j][0] = t[0]; pm[j][1] = t[1];
            }
        }
    };
    // px[C] of `chunk` -> RAW (activation gate and, under zero padding, the border mask applied here)
    auto x_store = [&](int chunk, auto cc_) __attribute__((always_inline)) {
        constexpr int C = decltype(cc_)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x2 v;
            v[0] = px[C][j][0]; v[1] = px[C][j][1];
            if constexpr (GATED == 1) {
                v[0] *= scm_act_grad_from_out(pg[C][j][0], gate_act);
                v[1] *= scm_act_grad_from_out(pg[C][j][1], gate_act);
            }
            if constexpr (GATED == 2) {
                const unsigned bit = 1u << (((clampc(chunk) * KC) & 31) + C);
                v[0] *= (pm[j][0] & bit) ? 1.f : 0.01f;
                v[1] *= (pm[j][1] & bit) ? 1.f : 0.01f;
            }
            if constexpr (!PADC) {
                const bool ok = okmask & (1u << j);
                v[0] = ok ? v[0] : 0.f;
                v[1] = ok ? v[1] : 0.f;
            }
            float* rp = RAW + C * PLANE + rawi[j];
            rp[0] = v[0]; rp[1] = v[1];
        }
    };
    // Input transform B^T d B of this thread's patches (tile = lane, local channels 2*wave and 2*wave+1), in pieces
    // that are dealt out over the MFMAs:  rd<cc,i>: one patch row from RAW;  col<cc,i>: along x of row i;
    // row<cc,j>: along y of column j -> vv[cc][4*i+j];  vw<f>: the pair of position f -> V[f][half][tile][step]
    // (MFMA k index of local channel c: half = c >> 2, step = c & 3)
    // (CW 128: 32 patches x 8 channels = one (patch, channel) item per thread: only CC = 0 is used)
    const int tch = CW == 64 ? 2 * wave : (tid >> 5);         // first local channel of the item
    // (Packed forms with their operand selects spelled out, as in conv3x3_wgrad_winoz.hip: a patch row arrives as the
    // column pairs lo = (d0, d1), hi = (d2, d3); along x  (d0 - d2, d1 - d3) = lo - hi  and  (d1 + d2, d2 - d1)  are ONE
    // v_pk_add_f32 each, along y every step is elementwise on those pairs: 16 VALU instructions per patch instead of
    // 32 -- each of them is paid for under the f32 MFMAs.)
    f32x2 dlo[2][4], dhi[2][4];          // this thread's patch rows as loaded, then their x-transformed pairs (0, 3) / (1, 2)
    auto t_rd = [&](auto cc_, auto i_) __attribute__((always_inline)) {
        constexpr int CC = decltype(cc_)::value, I = decltype(i_)::value;
        if constexpr (CW == 128 && CC == 1) return;
        const float* rp = RAW + (tch + CC) * PLANE + traw + I * ROWPL;
        dlo[CC][I] = *reinterpret_cast<const f32x2*>(rp);
        dhi[CC][I] = *reinterpret_cast<const f32x2*>(rp + 2);
    };
    auto t_col = [&](auto cc_, auto i_) __attribute__((always_inline)) {
        constexpr int CC = decltype(cc_)::value, I = decltype(i_)::value;
        if constexpr (CW == 128 && CC == 1) return;
        const f32x2 lo = dlo[CC][I], hi = dhi[CC][I];
        f32x2 p03, p12;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p03) : "v"(lo), "v"(hi));                        // (d0 - d2, d1 - d3)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(p12) : "v"(lo), "v"(hi));      // (d1 + d2, d2 - d1)
        dlo[CC][I] = p03; dhi[CC][I] = p12;
    };
    // output row J of the position grid (positions 4J .. 4J + 3) from the x-transformed rows
    auto t_row = [&](auto cc_, auto j_) __attribute__((always_inline)) {
        constexpr int CC = decltype(cc_)::value, J = decltype(j_)::value;
        if constexpr (CW == 128 && CC == 1) return;
        f32x2 o03, o12;
        if constexpr (J == 0) {
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o03) : "v"(dlo[CC][0]), "v"(dlo[CC][2]));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o12) : "v"(dhi[CC][0]), "v"(dhi[CC][2]));
        } else if constexpr (J == 1) {
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(o03) : "v"(dlo[CC][1]), "v"(dlo[CC][2]));
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(o12) : "v"(dhi[CC][1]), "v"(dhi[CC][2]));
        } else if constexpr (J == 2) {
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o03) : "v"(dlo[CC][2]), "v"(dlo[CC][1]));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o12) : "v"(dhi[CC][2]), "v"(dhi[CC][1]));
        } else {
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o03) : "v"(dlo[CC][1]), "v"(dlo[CC][3]));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o12) : "v"(dhi[CC][1]), "v"(dhi[CC][3]));
        }
        vv[CC][4 * J + 0] = o03[0]; vv[CC][4 * J + 1] = o12[0]; vv[CC][4 * J + 2] = o12[1]; vv[CC][4 * J + 3] = o03[1];
    };
    auto t_vw = [&](int buf, auto f_) __attribute__((always_inline)) {
        constexpr int F = decltype(f_)::value;
        if constexpr (CW == 64) {
            float* vp = VB + buf * VST + ((wave >> 1) * 64 + lane) * 4 + (wave & 1) * 2 + F * 512;
            f32x2 o;
            o[0] = vv[0][F]; o[1] = vv[1][F];
            *reinterpret_cast<f32x2*>(vp) = o;
        } else {        // local channel tch: half tch >> 2, step tch & 3; patch tid & 31
            VB[buf * VST + F * 256 + ((tch >> 2) * 32 + (tid & 31)) * 4 + (tch & 3)] = vv[0][F];
        }
    };
    auto b_read = [&](int buf, auto f_) __attribute__((always_inline)) {
        constexpr int F = decltype(f_)::value;
        const f32x4 t = *reinterpret_cast<const f32x4*>(VB + buf * VST + (lh * TW + wn * 32 + l31) * 4 + F * (8 * TW));
        bq[F & 1][0] = t[0]; bq[F & 1][1] = t[1]; bq[F & 1][2] = t[2]; bq[F & 1][3] = t[3];
    };
#define IC(n) std::integral_constant<int, (n)>{}
#define SB() __builtin_amdgcn_sched_barrier(0)
    auto u_prime = [&](auto f_) __attribute__((always_inline)) {           // positions 0 .. WINO_UD-1 of chunk 0
        constexpr int F = decltype(f_)::value;
        u_load(0, IC(F)); if constexpr (F + 1 < WINO_UD) u_load(0, IC(F + 1 < WINO_UD ? F + 1 : 0));
        if constexpr (F + 2 < WINO_UD) u_load(0, IC(F + 2 < WINO_UD ? F + 2 : 0));
        if constexpr (F + 3 < WINO_UD) u_load(0, IC(F + 3 < WINO_UD ? F + 3 : 0));
    };

    // One position F of chunk `chunk` (stage cur): 4 MFMAs, with this position's share of the staging work placed
    // between them.
    //   every position : the A operands of position F+6, the B operands of position F+1
    //   F = 0..7       : raw channel F of chunk+1 from px to RAW, then chunk+2's channel F requested into px
    //   F = 8, 9       : this thread's patch rows of chunk+1 from RAW (after the mid barrier)
    //   F = 10, 11     : the transform arithmetic
    //   F = 12..15     : V of chunk+1 into the idle stage
    auto slot = [&](int chunk, int cur, auto f_, auto first_) __attribute__((always_inline)) {
        constexpr int F = decltype(f_)::value;
        constexpr int R = F % WINO_UR, Q = F & 1;
        constexpr bool FIRST = decltype(first_)::value;
        auto mf = [&](auto s_) __attribute__((always_inline)) {
            constexpr int S = decltype(s_)::value;
            if constexpr (FIRST && S == 0) {
                // chunk 0 (peeled in front of the loop): the accumulator operand of a position's first MFMA is the constant
                // 0 -- 256 v_accvgpr_write of zeros per workgroup (~1 k ticks of an 85 k lifetime) never run
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[R][S], bq[Q][S], zero, 0, 0, 0);
            } else {
                acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[R][S], bq[Q][S], acc[F], 0, 0, 0);
            }
        };
#ifndef WINO_ABL
#define WINO_ABL 0
#endif
        constexpr bool DO_X = WINO_ABL < 1, DO_T = WINO_ABL < 2, DO_U = WINO_ABL < 3, DO_B = WINO_ABL < 4;
        mf(IC(0));
        if constexpr (F < 15 && DO_B) b_read(cur, IC(F < 15 ? F + 1 : 0));
        SB();
        mf(IC(1));
        if constexpr (F < 8 && DO_X) x_store(chunk + 1, IC(F < 8 ? F : 0));
        if constexpr (F == 8 && DO_T) { t_rd(IC(0), IC(0)); t_rd(IC(0), IC(1)); t_rd(IC(0), IC(2)); t_rd(IC(0), IC(3)); }
        if constexpr (F == 9 && DO_T) { t_rd(IC(1), IC(0)); t_rd(IC(1), IC(1)); t_rd(IC(1), IC(2)); t_rd(IC(1), IC(3)); }
        if constexpr (F == 10 && DO_T) { t_col(IC(0), IC(0)); t_col(IC(0), IC(1)); }
        if constexpr (F == 11 && DO_T) { t_col(IC(1), IC(0)); t_col(IC(1), IC(1)); }
        if constexpr (F >= 12 && DO_T) t_vw(cur ^ 1, IC(F >= 12 ? (F - 12) * 4 : 0));
        SB();
        mf(IC(2));
        if constexpr (F < 8 && DO_X) x_load(chunk + 2, IC(F < 8 ? F : 0));
        if constexpr (F == 7 && DO_X) m_load(chunk + 2);
        if constexpr (F == 10 && DO_T) { t_col(IC(0), IC(2)); t_col(IC(0), IC(3)); }
        if constexpr (F == 11 && DO_T) { t_col(IC(1), IC(2)); t_col(IC(1), IC(3)); }
        if constexpr (F >= 12 && DO_T) t_vw(cur ^ 1, IC(F >= 12 ? (F - 12) * 4 + 1 : 0));
        SB();
        mf(IC(3));
        // the ring slot of position F + WINO_UD was last read WINO_UR - WINO_UD positions ago
        if constexpr (F + WINO_UD < 16 && DO_U) u_load(chunk, IC(F + WINO_UD < 16 ? F + WINO_UD : 0));
        if constexpr (F + WINO_UD >= 16 && DO_U) u_load(chunk + 1, IC(F + WINO_UD >= 16 ? F + WINO_UD - 16 : 0));
        if constexpr (F == 10 && DO_T) { t_row(IC(0), IC(0)); t_row(IC(0), IC(1)); t_row(IC(0), IC(2)); t_row(IC(0), IC(3)); }
        if constexpr (F == 11 && DO_T) { t_row(IC(1), IC(0)); t_row(IC(1), IC(1)); t_row(IC(1), IC(2)); t_row(IC(1), IC(3)); }
        if constexpr (F >= 12 && DO_T) { t_vw(cur ^ 1, IC(F >= 12 ? (F - 12) * 4 + 2 : 0)); t_vw(cur ^ 1, IC(F >= 12 ? (F - 12) * 4 + 3 : 0)); }
        SB();
    };

    // this lane's output patch of tile k (its patch row: oy = 2k, ox = 2 (lane & 31)), and -- requested ahead, used in the
    // epilogue -- the gate words of the destination activation for its four pixels (gate_mask_y1 / _y2)
    const int co2n = a.cout - a.co1;
    int p00 = 2 * k0 * W + 2 * l31;
    unsigned gwl[4] = {0u, 0u, 0u, 0u};
    auto gw_load = [&]() __attribute__((always_inline)) {
        const int cw = co_w;
        const bool fst = cw < a.co1;
        const unsigned* gmx = fst ? a.ymask1 : a.ymask2;
        if (gmx) {
            const int gnw = ((fst ? a.co1 : co2n) + 31) >> 5;
            const unsigned* gp = gmx + ((size_t)b * gnw + ((cw - (fst ? 0 : a.co1)) >> 5)) * HW + p00;
            const u32x2w g0 = *reinterpret_cast<const u32x2w*>(gp), g1 = *reinterpret_cast<const u32x2w*>(gp + W);
            gwl[0] = g0[0]; gwl[1] = g0[1]; gwl[2] = g1[0]; gwl[3] = g1[1];
        }
    };
    gw_load();
    // ---- prologue: chunk 0 staged and transformed, chunk 1 requested, the A ring primed -- in two parts: the requests
    // (issued for the second tile BEFORE the first tile's epilogue) and what waits for them
    auto prologue_issue = [&]() __attribute__((always_inline)) {
        m_load(0);
        x_load(0, IC(0)); x_load(0, IC(1)); x_load(0, IC(2)); x_load(0, IC(3));
        x_load(0, IC(4)); x_load(0, IC(5)); x_load(0, IC(6)); x_load(0, IC(7));
        u_prime(IC(0)); if constexpr (WINO_UD > 4) u_prime(IC(4)); if constexpr (WINO_UD > 8) u_prime(IC(WINO_UD > 8 ? 8 : 0));
        if constexpr (WINO_UD > 12) u_prime(IC(WINO_UD > 12 ? 12 : 0));
    };
    auto prologue_finish = [&]() __attribute__((always_inline)) {
        x_store(0, IC(0)); x_store(0, IC(1)); x_store(0, IC(2)); x_store(0, IC(3));
        x_store(0, IC(4)); x_store(0, IC(5)); x_store(0, IC(6)); x_store(0, IC(7));
        __syncthreads();
        m_load(1);
        x_load(1, IC(0)); x_load(1, IC(1)); x_load(1, IC(2)); x_load(1, IC(3));
        x_load(1, IC(4)); x_load(1, IC(5)); x_load(1, IC(6)); x_load(1, IC(7));
        t_rd(IC(0), IC(0)); t_rd(IC(0), IC(1)); t_rd(IC(0), IC(2)); t_rd(IC(0), IC(3));
        t_rd(IC(1), IC(0)); t_rd(IC(1), IC(1)); t_rd(IC(1), IC(2)); t_rd(IC(1), IC(3));
        t_col(IC(0), IC(0)); t_col(IC(0), IC(1)); t_col(IC(0), IC(2)); t_col(IC(0), IC(3));
        t_col(IC(1), IC(0)); t_col(IC(1), IC(1)); t_col(IC(1), IC(2)); t_col(IC(1), IC(3));
        t_row(IC(0), IC(0)); t_row(IC(0), IC(1)); t_row(IC(0), IC(2)); t_row(IC(0), IC(3));
        t_row(IC(1), IC(0)); t_row(IC(1), IC(1)); t_row(IC(1), IC(2)); t_row(IC(1), IC(3));
        t_vw(0, IC(0)); t_vw(0, IC(1)); t_vw(0, IC(2)); t_vw(0, IC(3)); t_vw(0, IC(4)); t_vw(0, IC(5));
        t_vw(0, IC(6)); t_vw(0, IC(7)); t_vw(0, IC(8)); t_vw(0, IC(9)); t_vw(0, IC(10)); t_vw(0, IC(11));
        t_vw(0, IC(12)); t_vw(0, IC(13)); t_vw(0, IC(14)); t_vw(0, IC(15));
        __syncthreads();
    };
    auto chunk_body = [&](int chunk, auto first_) __attribute__((always_inline)) {
        const int cur = chunk & 1;
        asm volatile("" : "+v"(uoff[0]), "+v"(uoff[1]), "+v"(uoff[2]), "+v"(uoff[3]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(poff[j]));
        if (WINO_ABL < 4) b_read(cur, IC(0));
        SB();
        slot(chunk, cur, IC(0), first_); slot(chunk, cur, IC(1), first_); slot(chunk, cur, IC(2), first_); slot(chunk, cur, IC(3), first_);
        slot(chunk, cur, IC(4), first_); slot(chunk, cur, IC(5), first_); slot(chunk, cur, IC(6), first_); slot(chunk, cur, IC(7), first_);
        __syncthreads();            // RAW holds chunk+1
        SB();
        slot(chunk, cur, IC(8), first_); slot(chunk, cur, IC(9), first_); slot(chunk, cur, IC(10), first_); slot(chunk, cur, IC(11), first_);
        slot(chunk, cur, IC(12), first_); slot(chunk, cur, IC(13), first_); slot(chunk, cur, IC(14), first_); slot(chunk, cur, IC(15), first_);
        __syncthreads();            // V[cur ^ 1] holds chunk+1; every wave is done with V[cur] and RAW
        SB();
    };
    auto kloop = [&]() __attribute__((always_inline)) {
        chunk_body(0, std::true_type{});
        for (int chunk = 1; chunk < nchunk; ++chunk) chunk_body(chunk, std::false_type{});
    };

    // ---- epilogue (the fast path of conv3x3_wino_kernel; the host sends only launches that take it): output transform in
    // registers, scale, bias, per-sample bias, activation, NCHW store -- of the tile whose p00 / gate words are passed in
    const float sc = a.scale ? *a.scale : 1.f;
    const float slope = scm_act_slope(a.act);
    auto epilogue = [&](const int p00, const unsigned (&gwl)[4]) {
        typedef __attribute__((address_space(1))) char* gwp;
        typedef __attribute__((address_space(1))) f32x2* gw2;
        typedef __attribute__((address_space(1))) u32x2w* gm2;
        const bool first = co_w < a.co1;
        const size_t plane_b = (size_t)HW * 4;
        gwp ybase = first ? (gwp)(a.y1) + ((size_t)b * a.co1 + co_w) * plane_b
                          : (gwp)(a.y2) + ((size_t)b * co2n + (co_w - a.co1)) * plane_b;
        const unsigned loff = (unsigned)(4 * lh) * (unsigned)plane_b + (unsigned)p00 * 4u;
        const unsigned wrow = (unsigned)W * 4u;
        // leaky-relu gate of the destination activation (gate_mask_y1 / _y2): this wave's 32 channels are one word per
        // pixel; shifted so that bit (r & 3) + 8 * (r >> 2) is this lane's row r
        const unsigned* gm = first ? a.ymask1 : a.ymask2;
        unsigned gw[4] = {gwl[0] >> (4 * lh), gwl[1] >> (4 * lh), gwl[2] >> (4 * lh), gwl[3] >> (4 * lh)};
        auto body = [&](auto add_, auto gate_) __attribute__((always_inline)) {
            constexpr bool ADD = decltype(add_)::value, YG = decltype(gate_)::value;
            float old[ADD ? 16 : 1][4];
            if constexpr (ADD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    gwp rb = ybase + (size_t)((r & 3) + 8 * (r >> 2)) * plane_b;
                    asm volatile("" : "+s"(rb));
                    const f32x2 v0 = *(gw2)(rb + vo(loff)), v1 = *(gw2)(rb + wrow + vo(loff));
                    old[r][0] = v0[0]; old[r][1] = v0[1]; old[r][2] = v1[0]; old[r][3] = v1[1];
                }
            }
            unsigned bits[4] = {0u, 0u, 0u, 0u};
            f32x2 sc2, sl2;
            sc2[0] = sc; sc2[1] = sc; sl2[0] = slope; sl2[1] = slope;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowu = (r & 3) + 8 * (r >> 2);
                // A^T M A in packed form (every VALU instruction of a wave is paid for next to its f32 MFMAs):
                // the 16 positions leave the accumulators as the column pairs (0, 3) and (1, 2) of each grid row -- an
                // accumulator read can land in any register -- so that the sums along y are elementwise on pairs and the
                // two outputs of an image row come out of ONE instruction as the pair the 8-byte store takes:
                //   (o0, o1) = (s0 + (s1 + s2), (s1 - s2) - s3)
                f32x2 m03[4], m12[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m03[i][0] = acc[4 * i][r]; m03[i][1] = acc[4 * i + 3][r];
                    m12[i][0] = acc[4 * i + 1][r]; m12[i][1] = acc[4 * i + 2][r];
                }
                const f32x2 s0_03 = wino_pk_add(wino_pk_add(m03[0], m03[1]), m03[2]);
                const f32x2 s0_12 = wino_pk_add(wino_pk_add(m12[0], m12[1]), m12[2]);
                const f32x2 s1_03 = wino_pk_sub(wino_pk_sub(m03[1], m03[2]), m03[3]);
                const f32x2 s1_12 = wino_pk_sub(wino_pk_sub(m12[1], m12[2]), m12[3]);
                f32x2 oa = wino_pk_addsub(wino_pk_sumdiff(s0_12), s0_03);       // image row oy:     (o0, o1)
                f32x2 ob = wino_pk_addsub(wino_pk_sumdiff(s1_12), s1_03);       // image row oy + 1: (o2, o3)
                f32x2 bb;
                bb[0] = BIASL[co_l + rowu + 4 * lh]; bb[1] = bb[0];
                oa = wino_pk_fma(oa, sc2, bb);
                ob = wino_pk_fma(ob, sc2, bb);
                const f32x2 la = wino_pk_mul(oa, sl2), lb = wino_pk_mul(ob, sl2);
                float o[4];
                o[0] = fmaxf(oa[0], la[0]); o[1] = fmaxf(oa[1], la[1]);      // none (slope 1) / leaky_relu (0.01): bitwise scm_act_sel
                o[2] = fmaxf(ob[0], lb[0]); o[3] = fmaxf(ob[1], lb[1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bits[q] |= (o[q] > 0.f ? 1u : 0u) << (rowu + 4 * lh);
                    if constexpr (YG) o[q] = (gw[q] & (1u << rowu)) ? o[q] : 0.01f * o[q];
                    if constexpr (ADD) o[q] += old[r][q];
                }
                gwp rb = ybase + (size_t)rowu * plane_b;
                asm volatile("" : "+s"(rb));
                f32x2 v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
                *(gw2)(rb + vo(loff)) = v0;
                *(gw2)(rb + wrow + vo(loff)) = v1;
            }
            if (a.mask_out) {         // wave-uniform: the 1-bit leaky-relu gate of this wave's 32 channels
#pragma unroll
                for (int q = 0; q < 4; ++q) bits[q] |= __shfl_xor(bits[q], 32);
                const int nw = (a.cout + 31) >> 5;
                gwp mrow = (gwp)(a.mask_out) + ((size_t)b * nw + (co_w >> 5)) * plane_b;
                if (lh == 0) {
                    u32x2w v0 = {bits[0], bits[1]}, v1 = {bits[2], bits[3]};
                    *(gm2)(mrow + p00 * 4) = v0;
                    *(gm2)(mrow + wrow + p00 * 4) = v1;
                }
            }
        };
        const bool addp = a.acc_y1 && first;
        if (gm) { if (addp) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{}); }
        else { if (addp) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{}); }
    };

    // ---- the pair of tiles
    prologue_issue();
    prologue_finish();
    kloop();
    // second tile: its staging map and its first requests, THEN the first tile's epilogue
    const int p00_first = p00;
    const unsigned gw_first[4] = {gwl[0], gwl[1], gwl[2], gwl[3]};
    set_tile(k0 + kstep);
    prologue_issue();
    epilogue(p00_first, gw_first);
    p00 += 2 * W * kstep;
    gw_load();
    prologue_finish();
    kloop();
    epilogue(p00, gwl);
#undef IC
#undef SB
}

// ------------------------------------------------------------------ host
// Launches scm_conv3x3_wino would send to conv3x3_wino_kernel<1, 0, PADC, 128> with every workgroup on the fast epilogue:
// W == 64 (32-patch rows), H % 4 == 0 (an even number of tiles per image), ungated input, no sigmoid; checked by the caller:
// cout % 128 == 0, co1 % 32 == 0, 8-byte aligned tensors.
bool scm_wino_t2_serves(const ConvKArgs& k) {
    const char* e = getenv("SCM_WINO_T2");          // (read per launch: SCM_WINO_T2=0 -> one tile per workgroup, the A/B; tests switch it)
    const int on = e ? atoi(e) : 1;
    // (... and a grid that still fills the chip twice when every workgroup takes two tiles: at batch 8 the one-tile form's
    // 256 workgroups are one round on 256 CUs, 128 two-tile workgroups would leave half of them idle)
    // (... and whole rounds: one workgroup per CU on MI355X's 256 CUs, 164.4 k ticks per pair against 83.8 k per tile -- at the
    // eval batch of 100, 1 600 pairs are 7 rounds where 3 200 tiles are 12.5: the one-tile form wins by 5 %, measured 1.7 %)
    const long wg2 = (long)k.B * (k.H / 4) * (k.cout / 128), wg1 = 2 * wg2;
    const long t2 = ((wg2 + 255) / 256) * 1644, t1 = ((wg1 + 255) / 256) * 838;
    return on && k.W == 64 && (k.H & 3) == 0 && k.gate_act == 0 && k.act != SCM_ACT_SIGMOID && (k.cout & 127) == 0 &&
           (k.co1 & 31) == 0 && wg2 >= 512 && t2 <= t1;
}

template <int PADC>
static int launch_wino_t2(const ConvKArgs& a, int groups, hipStream_t s) {
    const size_t lds = ((size_t)KC * 512 + 2 * 128 * 32 + 128) * sizeof(float);
    auto kern = conv3x3_wino_t2_kernel<PADC>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    dim3 grid(a.B * a.tiles_per_img, groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wino_t2_kernel");
    return SCM_OK;
}

int scm_launch_wino_t2(const ConvKArgs& k, int groups, hipStream_t s) {
    ConvKArgs a = k;
    a.tiles_per_img = a.H / 4;              // tile PAIRS per image (a tile = one row of 32 patches)
    return a.pad_mode == SCM_PAD_CIRCULAR ? launch_wino_t2<1>(a, groups, s) : launch_wino_t2<0>(a, groups, s);
}
