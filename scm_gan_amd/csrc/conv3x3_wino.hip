// 3x3 stride-1 same-size convolution by Winograd F(2x2, 3x3) on the fp32 MFMA pipe
// (v_mfma_f32_32x32x2_f32), NCHW in / NCHW out, zero or circular halo.
//
// Same operator as conv3x3_mfma.hip (same descriptor, same epilogue contract) for the wide layers of the
// hot path -- reference /root/reference/models.py:76-102 (Transition conv2..conv5), :145-155 (Encoder
// conv2 / conv3) and their input-gradient convolutions -- with 16 multiplications per 2x2 output patch
// and (ci, co) pair instead of 36:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; d = 4x4 input patch, g = 3x3 filter)
//
// fp32 throughout: the f32 MFMA runs at the vector rate on gfx950 (MI355X_MICROARCH.md), so the only way
// to make an fp32 convolution faster than its MFMA roof is to issue fewer products.  The transforms have
// entries 0, +-1, +-1/2 only; the error against float64 is measured next to the direct kernel's in
// tests/test_gpu_wino.py.
//
// GEMM view: 16 independent products, one per transform-domain position f,
//     M_f[co][tile] = sum_ci U_f[co][ci] * V_f[ci][tile]
//   MFMA A operand = U (transformed weights, pre-packed), lane -> co, k -> ci
//   MFMA B operand = V (transformed input patches, built in LDS), lane -> tile, k -> ci
// Workgroup = 4 waves = 64 output channels x 64 tiles (256 output pixels) of one image; wave (wm, wn) owns
// 32 channels x 32 tiles for ALL 16 positions: 16 accumulator tiles = 256 AGPRs, so the output transform
// A^T M A happens in registers (a lane holds the 16 positions of its (tile, channel) pairs).
// K loop: chunks of 8 input channels.  Per chunk: the raw halo rows [8][rows][2*TCW+2] and the weight
// slab U [16][2][64][4] are staged in LDS (same unconditional clamped loads + masks as the direct kernel,
// gates applied here), each thread transforms one (tile, channel pair) patch into V [16][2][64][4], then
// 64 MFMAs per wave with one ds_read_b128 per operand and position.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_args.h"

#define WINO_TILES 64     // tiles per workgroup
#define WINO_CO 64        // output channels per workgroup
#define WINO_SLAB 8192    // floats of one (co group, K chunk) weight slab: [16][2][64][4]
#ifndef WINO_UR
#define WINO_UR 8         // A-operand ring: slots and prefetch distance in positions (measured: 8 / 6, 12 / 10 and
#define WINO_UD 6         // 16 / 14 run the K loop in 5 171 / 5 231 / 5 237 ticks per chunk: latency is covered at 6)
#endif

typedef unsigned u32x2w __attribute__((ext_vector_type(2)));

// Phase clock for tools/exp/wino_phases.hip (compiled in only with -DWINO_TIMING=<workgroup>): shader-clock ticks
// of wave 0 of one workgroup: prologue, K loop, epilogue.
#ifdef WINO_TIMING
__device__ unsigned long long wino_ticks[4];
__device__ unsigned long long wino_trace[3 * 8192];     // per workgroup: (XCC_ID << 32 | HW_ID), entry clock, exit clock
#define WINO_TICK(i)                                    \
    do {                                                \
        const unsigned long long t_ = clock64();        \
        t_acc[i] += t_ - t_last;                        \
        t_last = t_;                                    \
    } while (0)
#else
#define WINO_TICK(i)
#endif

// CW: output channels per workgroup.  64: 2 x 2 waves = 64 co x 64 patches (any frame).  128: 4 x 1 waves = 128 co x 32
// patches -- each transformed patch feeds twice the MFMAs and the raw rows are staged once instead of once per 64-co
// group; for layers with a multiple of 128 output channels on frames whose patch rows divide 32 (W = 16, 32, 64).
// packed fp32 forms with their operand selects spelled out (the compiler scalarises them or builds them from moves)
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

template <int NJ, int GATED, int PADC, int CW>
__global__ void __launch_bounds__(256, 1) conv3x3_wino_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = NJ * 512;                   // floats per channel of the raw halo rows (NJ pairs per thread)
    float* const RAW = smem;                          // [KC][PLANE]
    constexpr int TW = CW == 64 ? 64 : 32;            // patches per workgroup
    constexpr int VST = 128 * TW;                     // floats of one V stage [16][2][TW][4]
    float* const VB = smem + KC * PLANE;              // two stages of V
    float* const BIASL = VB + 2 * VST;                // [CW]
#ifdef WINO_TIMING
    const unsigned long long t_entry = clock64();
#endif

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

    int b, wt, cg;
    scm_xcd_tile_map(a.B, a.tiles_per_img, b, wt, cg);
    const int t0 = wt * TW;
    const int ty0 = t0 / TCW;
    int tlast = t0 + TW - 1;
    if (tlast > ntiles - 1) tlast = ntiles - 1;
    const int nrows = 2 * (tlast / TCW - ty0) + 4;
    const int HPR = (W + 4) >> 1;                     // pairs per raw row
    const int npair = nrows * HPR;

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
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = tid + 256 * j;
        int g = 0;
        if (q < npair) {
            const int r = q / HPR;
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
        { const int r = q / HPR; rawi[j] = q < npair ? r * ROWPL + 2 * (q - r * HPR) + 1 : 0; }   // (slots 0, 1 are never read)
    }

    // this thread's transform item: tile (lane) x channel pair (wave) of the chunk
    int traw;               // offset of its 4x4 patch (raw slot 2*tx + 2 = image column 2*tx - 1, even) in a raw plane
    {
        int t = t0 + (CW == 64 ? lane : (tid & 31));
        if (t > ntiles - 1) t = ntiles - 1;
        const int ty = t / TCW, tx = t - ty * TCW;
        traw = 2 * (ty - ty0) * ROWPL + 2 * tx + 2;
    }

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

    auto clampc = [&](int chunk) { return chunk < nchunk ? chunk : nchunk - 1; };
    // (32-bit lane offsets are laundered IN PLACE through an empty asm once per chunk -- launder() -- so that their
    // zero-extension stays inside the loop, next to the loads, where instruction selection folds it into the
    // scalar-base addressing mode; a per-load laundering copy costs a v_mov each)
    auto vo = [](unsigned o) { return o; };
    // A operands of position F of `chunk` (clamped: past the end the last slab is re-read and never used)
    auto u_load = [&](int chunk, auto fc) {
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
    auto x_load = [&](int chunk, auto cc_) {
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
    auto m_load = [&](int chunk) {
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
    auto x_store = [&](int chunk, auto cc_) {
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
    auto t_rd = [&](auto cc_, auto i_) {
        constexpr int CC = decltype(cc_)::value, I = decltype(i_)::value;
        if constexpr (CW == 128 && CC == 1) return;
        const float* rp = RAW + (tch + CC) * PLANE + traw + I * ROWPL;
        dlo[CC][I] = *reinterpret_cast<const f32x2*>(rp);
        dhi[CC][I] = *reinterpret_cast<const f32x2*>(rp + 2);
    };
    auto t_col = [&](auto cc_, auto i_) {
        constexpr int CC = decltype(cc_)::value, I = decltype(i_)::value;
        if constexpr (CW == 128 && CC == 1) return;
        const f32x2 lo = dlo[CC][I], hi = dhi[CC][I];
        f32x2 p03, p12;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p03) : "v"(lo), "v"(hi));                        // (d0 - d2, d1 - d3)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(p12) : "v"(lo), "v"(hi));      // (d1 + d2, d2 - d1)
        dlo[CC][I] = p03; dhi[CC][I] = p12;
    };
    // output row J of the position grid (positions 4J .. 4J + 3) from the x-transformed rows
    auto t_row = [&](auto cc_, auto j_) {
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
    auto t_vw = [&](int buf, auto f_) {
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
    auto b_read = [&](int buf, auto f_) {
        constexpr int F = decltype(f_)::value;
        const f32x4 t = *reinterpret_cast<const f32x4*>(VB + buf * VST + (lh * TW + wn * 32 + l31) * 4 + F * (8 * TW));
        bq[F & 1][0] = t[0]; bq[F & 1][1] = t[1]; bq[F & 1][2] = t[2]; bq[F & 1][3] = t[3];
    };
#define IC(n) std::integral_constant<int, (n)>{}
#define SB() __builtin_amdgcn_sched_barrier(0)
    auto u_prime = [&](auto f_) {           // positions 0 .. WINO_UD-1 of chunk 0
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
    auto slot = [&](int chunk, int cur, auto f_, auto first_) {
        constexpr int F = decltype(f_)::value;
        constexpr int R = F % WINO_UR, Q = F & 1;
        constexpr bool FIRST = decltype(first_)::value;
        auto mf = [&](auto s_) {
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

    // this lane's output patch (epilogue), and -- requested now, used after the K loop -- the gate words of the
    // destination activation for its four pixels (gate_mask_y1 / _y2; meaningful on the fast epilogue path only)
    const int co2n = a.cout - a.co1;
    const int t = t0 + wn * 32 + l31;
    const bool tvalid = t < ntiles;
    const int tc = tvalid ? t : ntiles - 1;
    const int ty = tc / TCW, tx = tc - ty * TCW;
    const int oy = 2 * ty, ox = 2 * tx;
    const bool row1 = tvalid && oy + 1 < H, col1 = ox + 1 < W;
    const int p00 = oy * W + ox;
    unsigned gwl[4] = {0u, 0u, 0u, 0u};
    {
        const int cw = co_w;
        const bool fst = cw < a.co1;
        const unsigned* gmx = fst ? a.ymask1 : a.ymask2;
        if (gmx && row1 && cw + 32 <= a.cout) {
            const int gnw = ((fst ? a.co1 : co2n) + 31) >> 5;
            const unsigned* gp = gmx + ((size_t)b * gnw + ((cw - (fst ? 0 : a.co1)) >> 5)) * HW + p00;
            const u32x2w g0 = *reinterpret_cast<const u32x2w*>(gp), g1 = *reinterpret_cast<const u32x2w*>(gp + W);
            gwl[0] = g0[0]; gwl[1] = g0[1]; gwl[2] = g1[0]; gwl[3] = g1[1];
        }
    }
#ifdef WINO_TIMING
    unsigned long long t_acc[4] = {0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    // ---- prologue: chunk 0 staged and transformed, chunk 1 requested, the A ring primed
    {
        m_load(0);
        x_load(0, IC(0)); x_load(0, IC(1)); x_load(0, IC(2)); x_load(0, IC(3));
        x_load(0, IC(4)); x_load(0, IC(5)); x_load(0, IC(6)); x_load(0, IC(7));
        u_prime(IC(0)); if constexpr (WINO_UD > 4) u_prime(IC(4)); if constexpr (WINO_UD > 8) u_prime(IC(WINO_UD > 8 ? 8 : 0));
        if constexpr (WINO_UD > 12) u_prime(IC(WINO_UD > 12 ? 12 : 0));
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
    }
    WINO_TICK(0);
    auto chunk_body = [&](int chunk, auto first_) {
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
    chunk_body(0, std::true_type{});
    for (int chunk = 1; chunk < nchunk; ++chunk) chunk_body(chunk, std::false_type{});
#undef IC
#undef SB
    WINO_TICK(1);

    // ---- epilogue: output transform in registers, scale, bias, per-sample bias, activation, NCHW store
    const float sc = a.scale ? *a.scale : 1.f;
    const bool sigmoid = a.act == SCM_ACT_SIGMOID;
    const float slope = scm_act_slope(a.act);
    // 8-byte stores need even row pitch and 8-byte aligned bases (wave-uniform)
    const bool vec = (W & 1) == 0 && ((reinterpret_cast<size_t>(a.y1) | reinterpret_cast<size_t>(a.y2) |
                                        reinterpret_cast<size_t>(a.mask_out)) & 7) == 0;
    // Fast path (wave-uniform; every hot-path launch takes it): all 64 patches of the workgroup exist with both rows,
    // the wave's 32 channels exist and go to one output tensor, no sigmoid.  Scalar row bases + one 32-bit lane offset:
    // no per-store address arithmetic, no per-value predicates; a value already in y1 (accumulate_y1) is requested for
    // all 16 rows before the first one is needed.
    const bool fast = !sigmoid && t0 + TW <= ntiles && (H & 1) == 0 && co_w + 32 <= a.cout &&
                      (co_w + 32 <= a.co1 || co_w >= a.co1) && WINO_ABL != 5;
    if (fast) {
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
        auto body = [&](auto add_, auto gate_) {
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
    } else {
    unsigned bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co_w + row;
        float m[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) m[f] = acc[f][r];
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
        const float badd = BIASL[co_l + row];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float u = o[q] * sc + badd;
            o[q] = sigmoid ? scm_sigmoid(u) : scm_act_sel(u, slope);
            bits[q] |= (o[q] > 0.f ? 1u : 0u) << row;
        }
        if (co < a.cout && tvalid && WINO_ABL != 5) {
            const bool first = co < a.co1;
            if (const unsigned* gmg = first ? a.ymask1 : a.ymask2) {      // output gate of the destination activation
                const int cd = first ? co : co - a.co1;
                const int gnw = ((first ? a.co1 : co2n) + 31) >> 5;
                const unsigned* gp = gmg + ((size_t)b * gnw + (cd >> 5)) * HW + p00;
                const unsigned bit = 1u << (cd & 31);
                o[0] = (gp[0] & bit) ? o[0] : 0.01f * o[0];
                if (col1) o[1] = (gp[1] & bit) ? o[1] : 0.01f * o[1];
                if (row1) {
                    o[2] = (gp[W] & bit) ? o[2] : 0.01f * o[2];
                    if (col1) o[3] = (gp[W + 1] & bit) ? o[3] : 0.01f * o[3];
                }
            }
            float* dst = (first ? a.y1 + ((size_t)b * a.co1 + co) * HW
                                : a.y2 + ((size_t)b * co2n + (co - a.co1)) * HW) + p00;
            const bool add = a.acc_y1 && first;
            if (vec) {
                f32x2 v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
                if (add) {
                    v0 += *reinterpret_cast<const f32x2*>(dst);
                    if (row1) v1 += *reinterpret_cast<const f32x2*>(dst + W);
                }
                *reinterpret_cast<f32x2*>(dst) = v0;
                if (row1) *reinterpret_cast<f32x2*>(dst + W) = v1;
            } else {
                dst[0] = add ? dst[0] + o[0] : o[0];
                if (col1) dst[1] = add ? dst[1] + o[1] : o[1];
                if (row1) {
                    dst[W] = add ? dst[W] + o[2] : o[2];
                    if (col1) dst[W + 1] = add ? dst[W + 1] + o[3] : o[3];
                }
            }
        }
    }
    if (a.mask_out) {         // wave-uniform: the 1-bit leaky-relu gate of this wave's 32 channels
#pragma unroll
        for (int q = 0; q < 4; ++q) bits[q] |= __shfl_xor(bits[q], 32);
        const int nw = (a.cout + 31) >> 5, w = co_w >> 5;
        if (lh == 0 && tvalid && w < nw) {
            unsigned* dst = a.mask_out + ((size_t)b * nw + w) * HW + p00;
            if (vec) {
                u32x2w v0 = {bits[0], bits[1]}, v1 = {bits[2], bits[3]};
                *reinterpret_cast<u32x2w*>(dst) = v0;
                if (row1) *reinterpret_cast<u32x2w*>(dst + W) = v1;
            } else {
                dst[0] = bits[0];
                if (col1) dst[1] = bits[1];
                if (row1) {
                    dst[W] = bits[2];
                    if (col1) dst[W + 1] = bits[3];
                }
            }
        }
    }
    }
#ifdef WINO_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    WINO_TICK(2);
    if (b * a.tiles_per_img + wt == WINO_TIMING && cg == 0 && tid == 0)
        for (int i = 0; i < 4; ++i) wino_ticks[i] = t_acc[i];
    __syncthreads();
    const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0 && wg_ < 8192) {
        wino_trace[3 * wg_] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                              __builtin_amdgcn_s_getreg((31 << 11) | 4);
        wino_trace[3 * wg_ + 1] = t_entry;
        wino_trace[3 * wg_ + 2] = clock64();
    }
#endif
}

// ------------------------------------------------------------------ pack
// U = G g G^T of every (co, ci) filter, laid out as the kernel's LDS slab per (co group, K chunk):
// [group g][chunk c][wm][f = 4i+j][lane = 32*half + co % 32][step 4] with co = 64g + 32wm + lane % 32 and
// ci = 8c + 4*half + step -- per (chunk, wave) 16 contiguous 1 KB rows, each one wave-wide 16-byte load that IS the
// MFMA A operand of four k-steps; missing channels are zero.
__global__ void pack_conv3x3_wino_kernel(const float* __restrict__ w, int cin, int cout, int transposed_src, int wct,
                                         int wc0, float* __restrict__ fwd, float* __restrict__ dgr) {
    // element accessor of the equivalent conv weight Wc[co][ci][t] (as pack_conv3x3_kernel)
    auto Wc = [&](int co, int ci, int t) -> float {
        return transposed_src ? w[((size_t)(wc0 + ci) * cout + co) * 9 + (8 - t)]
                              : w[((size_t)co * wct + wc0 + ci) * 9 + t];
    };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    auto emit = [&](float* out, int n_in, int n_out, bool dgrad) {
        const int nchunk = (n_in + KC - 1) / KC;
        const int ng = (n_out + WINO_CO - 1) / WINO_CO;
        if (idx >= ng * nchunk * 512) return;
        const int s = idx & 3, col = (idx >> 2) & 63, half = (idx >> 8) & 1;
        const int c = (idx >> 9) % nchunk, g = (idx >> 9) / nchunk;
        const int ci = c * KC + half * 4 + s, co = g * WINO_CO + col;
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
        float* dst = out + (size_t)(g * nchunk + c) * WINO_SLAB + (col >> 5) * (WINO_SLAB / 2) +
                     (half * 32 + (col & 31)) * 4 + s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(4 * i + 0) * 256] = gg[i][0];
            dst[(4 * i + 1) * 256] = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
            dst[(4 * i + 2) * 256] = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
            dst[(4 * i + 3) * 256] = gg[i][2];
        }
    };
    if (fwd) emit(fwd, cin, cout, false);
    // dgrad conv: input channels = cout, output channels = cin, taps flipped
    if (dgr) emit(dgr, cout, cin, true);
}

// ------------------------------------------------------------------ host
static inline size_t wino_wide_pack_floats(int cin, int cout) {
    return (size_t)((cout + WINO_CO - 1) / WINO_CO) * ((cin + KC - 1) / KC) * WINO_SLAB;
}

extern "C" size_t scm_conv3x3_pack_wino_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    if (cout <= 16) return scm_wino16_pack_floats(cin);          // the 16-row kernel's layout (conv3x3_wino16.hip)
    // (<= 16 input channels: the pack of conv3x3_wino_k16.hip follows the wide kernel's -- which kernel runs is
    // decided per launch, scm_wino_k16_serves)
    return wino_wide_pack_floats(cin, cout) + (cin <= 16 ? scm_wino_k16_pack_floats(cout) : 0);
}

extern "C" int scm_pack_conv3x3_weights_wino(const float* w, int cin, int cout, int transposed_src, int w_cin_total,
                                             int w_ci0, float* fwd_pack, float* dgrad_pack, void* stream) {
    SCM_REQUIRE(w && cin > 0 && cout > 0, "scm_pack_conv3x3_weights_wino: bad arguments (cin=%d cout=%d)", cin, cout);
    if (w_cin_total == 0) w_cin_total = cin;
    SCM_REQUIRE(w_ci0 >= 0 && w_ci0 + cin <= w_cin_total, "scm_pack_conv3x3_weights_wino: bad channel slice");
    SCM_REQUIRE(fwd_pack || dgrad_pack, "scm_pack_conv3x3_weights_wino: no output requested");
    // a direction with <= 16 output channels gets the 16-row kernel's pack
    const bool f16 = fwd_pack && cout <= 16, d16 = dgrad_pack && cin <= 16;
    if (f16 || d16) {
        if (int rc = scm_pack_wino16(w, cin, cout, transposed_src, w_cin_total, w_ci0, f16 ? fwd_pack : nullptr,
                                     d16 ? dgrad_pack : nullptr, (hipStream_t)stream)) return rc;
        if (f16) fwd_pack = nullptr;
        if (d16) dgrad_pack = nullptr;
        if (!fwd_pack && !dgrad_pack) return SCM_OK;
    }
    // a direction with <= 16 input channels gets the thin-input kernel's pack behind the wide one
    {
        float* fk = (fwd_pack && cin <= 16) ? fwd_pack + wino_wide_pack_floats(cin, cout) : nullptr;
        float* dk = (dgrad_pack && cout <= 16) ? dgrad_pack + wino_wide_pack_floats(cout, cin) : nullptr;
        if (fk || dk)
            if (int rc = scm_pack_wino_k16(w, cin, cout, transposed_src, w_cin_total, w_ci0, fk, dk, (hipStream_t)stream)) return rc;
    }
    size_t n = 0;
    if (fwd_pack) n = wino_wide_pack_floats(cin, cout) / 16;
    if (dgrad_pack) {
        const size_t m = wino_wide_pack_floats(cout, cin) / 16;
        if (m > n) n = m;
    }
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads);
    hipLaunchKernelGGL(pack_conv3x3_wino_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, w, cin, cout,
                       transposed_src, w_cin_total, w_ci0, fwd_pack, dgrad_pack);
    SCM_LAUNCH_CHECK("pack_conv3x3_wino_kernel");
    return SCM_OK;
}

template <int NJ, int GATED, int PADC, int CW>
static int launch_wino(const ConvKArgs& a, int groups, hipStream_t s) {
    const size_t lds = ((size_t)KC * NJ * 512 + 2 * 128 * (CW == 64 ? 64 : 32) + CW) * sizeof(float);
    auto kern = conv3x3_wino_kernel<NJ, GATED, PADC, CW>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    dim3 grid(a.B * a.tiles_per_img, groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wino_kernel");
    return SCM_OK;
}

// halo elements per channel of the worst workgroup (tw consecutive tiles of a TCW-wide tile grid; raw rows of W + 6)
static inline int wino_plane(int H, int W, int tw) {
    const int tcw = W / 2, trh = (H + 1) / 2;
    int trows = (tw % tcw == 0) ? tw / tcw : (tw + tcw - 2) / tcw + 1;
    if (trows > trh) trows = trh;
    return (2 * trows + 2) * (W + 6);
}

extern "C" int scm_conv3x3_wino16_serves(int H, int W) { return scm_wino16_serves(H, W) ? 1 : 0; }

extern "C" int scm_conv3x3_wino_serves(int H, int W) {
    return H >= 2 && W >= 2 && (W & 1) == 0 && wino_plane(H, W, WINO_TILES) <= 2 * 512;
}

extern "C" int scm_conv3x3_wino(const scm_conv3x3_desc* d, void* stream) {
    SCM_REQUIRE(d != nullptr, "scm_conv3x3_wino: null descriptor");
    SCM_REQUIRE(!d->x_half && !d->y_half, "scm_conv3x3_wino: IEEE-half tensors (x_half / y_half) are taken by scm_conv3x3_h16 only");
    SCM_REQUIRE(d->x1 && d->c1 > 0, "scm_conv3x3_wino: x1/c1 required");
    SCM_REQUIRE((d->x2 != nullptr) == (d->c2 > 0) && d->c2 >= 0, "scm_conv3x3_wino: x2/c2 mismatch");
    SCM_REQUIRE(d->wpack && d->cout > 0, "scm_conv3x3_wino: wpack/cout required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "scm_conv3x3_wino: bad shape B=%d H=%d W=%d", d->B, d->H, d->W);
    SCM_REQUIRE(d->y1 && d->co1 > 0 && d->co1 <= d->cout, "scm_conv3x3_wino: y1/co1 invalid");
    SCM_REQUIRE((d->co1 == d->cout) || d->y2, "scm_conv3x3_wino: y2 required when co1 < cout");
    SCM_REQUIRE(d->pad_mode == SCM_PAD_ZERO || d->pad_mode == SCM_PAD_CIRCULAR, "scm_conv3x3_wino: bad pad_mode %d",
                d->pad_mode);
    SCM_REQUIRE(d->act >= 0 && d->act <= 2 && d->gate_act >= 0 && d->gate_act <= 2, "scm_conv3x3_wino: bad act");
    if (d->gate_act) {
        SCM_REQUIRE(d->gate1 && (d->c2 == 0 || d->gate2), "scm_conv3x3_wino: gate tensors required with gate_act");
    }
    if (d->cout <= 16) {
        // the 16-row kernel (conv3x3_wino16.hip): ungated input, one output tensor, plain store
        if (d->gate_act || d->accumulate_y1 || d->gate_mask_y1 || d->gate_mask_y2 || d->gate_mask_out || d->co1 != d->cout ||
            !scm_wino16_serves(d->H, d->W) || (d->c2 != 0 && (d->c1 & 7) != 0) ||
            ((reinterpret_cast<size_t>(d->x1) | reinterpret_cast<size_t>(d->x2) | reinterpret_cast<size_t>(d->y1)) & 7)) {
            scm_set_error("scm_conv3x3_wino: the 16-channel form takes ungated inputs, one plain output tensor and frames "
                          "scm_conv3x3_wino16_serves accepts (%dx%d)", d->H, d->W);
            return SCM_EUNSUPPORTED;
        }
        ConvKArgs k16;
        k16.x1 = d->x1; k16.x2 = d->x2; k16.c1 = d->c1; k16.c2 = d->c2;
        k16.gate1 = k16.gate2 = nullptr; k16.gate_act = 0;
        k16.wpack = d->wpack; k16.cout = d->cout;
        k16.scale = d->scale; k16.bias = d->bias; k16.sbias = d->sbias;
        k16.y1 = d->y1; k16.y2 = nullptr; k16.co1 = d->cout;
        k16.B = d->B; k16.H = d->H; k16.W = d->W; k16.pad_mode = d->pad_mode; k16.act = d->act;
        k16.mask_in = nullptr; k16.mask_out = nullptr; k16.acc_y1 = 0; k16.ymask1 = k16.ymask2 = nullptr;
        k16.tiles_per_img = 0; k16.plane = 0;
        return scm_launch_wino16(k16, (hipStream_t)stream);
    }
    if (!scm_conv3x3_wino_serves(d->H, d->W)) {
        scm_set_error("scm_conv3x3_wino: frame %dx%d does not fit the halo staging map (scm_conv3x3_wino_serves)", d->H,
                      d->W);
        return SCM_EUNSUPPORTED;
    }
    ConvKArgs k;
    k.x1 = d->x1; k.x2 = d->x2; k.c1 = d->c1; k.c2 = d->c2;
    k.gate1 = d->gate1; k.gate2 = d->gate2; k.gate_act = d->gate_act;
    k.wpack = d->wpack; k.cout = d->cout;
    k.scale = d->scale; k.bias = d->bias; k.sbias = d->sbias;
    k.y1 = d->y1; k.y2 = d->y2; k.co1 = d->co1;
    k.B = d->B; k.H = d->H; k.W = d->W; k.pad_mode = d->pad_mode; k.act = d->act;
    k.mask_in = d->gate_mask_in;
    k.acc_y1 = d->accumulate_y1;
    k.ymask1 = d->gate_mask_y1; k.ymask2 = d->gate_mask_y2;
    // (either mask: the epilogue indexes gate_mask_y2 by (co - co1) >> 5 with the bit = the row inside the wave's 32 channels)
    SCM_REQUIRE(((!d->gate_mask_y1 && !d->gate_mask_y2) || (d->co1 & 31) == 0 || d->co1 == d->cout) && ((reinterpret_cast<size_t>(d->gate_mask_y1) | reinterpret_cast<size_t>(d->gate_mask_y2)) & 7) == 0,
                "scm_conv3x3_wino: output gates need co1 %% 32 == 0 and 8-byte aligned masks");
    k.mask_out = (d->act == SCM_ACT_LRELU && d->cout > 32) ? d->gate_mask_out : nullptr;
    SCM_REQUIRE(!d->gate_mask_out || k.mask_out, "scm_conv3x3_wino: gate_mask_out needs act == LRELU and cout > 32");
    SCM_REQUIRE(d->c2 == 0 || (d->c1 & 7) == 0, "scm_conv3x3_wino: c1 must be a multiple of 8 when x2 is given (c1=%d)", d->c1);
    SCM_REQUIRE(((reinterpret_cast<size_t>(d->x1) | reinterpret_cast<size_t>(d->x2) | reinterpret_cast<size_t>(d->gate1) |
                  reinterpret_cast<size_t>(d->gate2) | reinterpret_cast<size_t>(d->gate_mask_in) |
                  reinterpret_cast<size_t>(d->y1) | reinterpret_cast<size_t>(d->y2) |
                  reinterpret_cast<size_t>(d->gate_mask_out)) & 7) == 0,
                "scm_conv3x3_wino: tensors must be 8-byte aligned");
    if (d->c1 + d->c2 <= 16 && scm_wino_k16_serves(k)) {
        // at most 16 input channels: the kernel that walks the output channels (conv3x3_wino_k16.hip); its pack lies behind
        // the wide kernel's in the same buffer
        k.wpack = d->wpack + wino_wide_pack_floats(d->c1 + d->c2, d->cout);
        return scm_launch_wino_k16(k, (hipStream_t)stream);
    }
    const int ntiles = ((d->H + 1) / 2) * (d->W / 2);
    // 128 output channels per workgroup (half the transform and staging work per MFMA) where the layer has whole
    // 128-channel groups, one output tensor per group and the patch rows tile 32 patches
    static const int no128 = getenv("SCM_WINO_NO128") ? atoi(getenv("SCM_WINO_NO128")) : 0;
    const int tcw = d->W / 2;
    // (one raw pixel pair per thread up to 512 floats per channel, two up to 1 024: W = 2 with H > 126 fits neither)
    // (32 consecutive patches are whole patch rows -- W = 16, 32, 64 -- or an aligned part of one: W = 128, 192, ...;
    // their raw rows take one or two pixel pairs per thread)
    static const int any128 = getenv("SCM_WINO_128_ANY") ? atoi(getenv("SCM_WINO_128_ANY")) : 1;
    const bool aligned32 = ntiles % 32 == 0 && ((tcw <= 32 && 32 % tcw == 0) || tcw % 32 == 0);
    // (any128: also runs of 32 patches that straddle patch rows and a ragged last workgroup -- 84 x 84, BASELINE configs[3])
    const bool cw128 = !no128 && (d->cout & 127) == 0 && (d->co1 & 31) == 0 && (aligned32 || any128) &&
                       wino_plane(d->H, d->W, 32) <= 2 * 512;
    const int tw = cw128 ? 32 : WINO_TILES;
    k.tiles_per_img = (ntiles + tw - 1) / tw;
    k.plane = wino_plane(d->H, d->W, tw);
    const int groups = cw128 ? d->cout / 128 : (d->cout + WINO_CO - 1) / WINO_CO;
    hipStream_t s = (hipStream_t)stream;
    const int g = d->gate_act == 0 ? 0 : ((d->gate_mask_in && d->gate_act == SCM_ACT_LRELU && d->c2 == 0) ? 2 : 1);
    const bool circ = d->pad_mode == SCM_PAD_CIRCULAR;
#define SCM_WINO_P(NJ_, G_, CW_) (circ ? launch_wino<NJ_, G_, 1, CW_>(k, groups, s) : launch_wino<NJ_, G_, 0, CW_>(k, groups, s))
#define SCM_WINO_G(NJ_, CW_) (g == 2 ? SCM_WINO_P(NJ_, 2, CW_) : (g == 1 ? SCM_WINO_P(NJ_, 1, CW_) : SCM_WINO_P(NJ_, 0, CW_)))
    if (cw128) return k.plane <= 512 ? SCM_WINO_G(1, 128) : SCM_WINO_G(2, 128);
    if (k.plane <= 512) return SCM_WINO_G(1, 64);
    return SCM_WINO_G(2, 64);
#undef SCM_WINO_G
#undef SCM_WINO_P
}
