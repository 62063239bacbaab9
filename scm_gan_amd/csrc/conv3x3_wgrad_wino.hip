// Weight gradient of the 3x3 stride-1 same-size convolution by Winograd F(2x2, 3x3) on the fp32 MFMA pipe.
// Same operator and the same partial-slab / reduce contract as conv3x3_wgrad.hip (reference: the autograd backward of
// nn.Conv2d w.r.t. weight and bias triggered by loss.backward() at /root/reference/main.py:285 for the wide layers of
// models.py:51-56, 129-134), with 16 products per 2x2 output patch and (co, ci) pair instead of 36.
//
// Forward:  Y = A^T [ U .* V ] A,  U = G g G^T,  V = B^T d B  (per 2x2 output patch Y, 4x4 input patch d, filter g).
// Hence     dL/dU = sum over patches of (A dY A^T) .* V     -- 16 independent products, one per position f = 4i+j:
//               dU_f[co][ci] = sum_{b, patch} DY_f[co][patch] * V_f[ci][patch]
//           dL/dg = G^T dU G                                 -- once per (co, ci), in the epilogue.
//   MFMA A operand = DY (lane -> co, k -> patch),  B operand = V (lane -> ci, k -> patch), both transformed on the
//   fly from registers and staged in LDS as [16][2][64][4] (the forward kernel's operand image, conv3x3_wino.hip).
// Workgroup = 4 waves = 64 co x 64 ci for ALL 16 positions (wave (wm, wn): 32 x 32, 256 AGPRs) over a band of patch
// rows of ONE image (a "split"); K loop over chunks of 8 consecutive patches of a patch row.  Thread (channel = tid >> 2,
// patch pair = tid & 3) loads the 4 x 6 input elements of its pair for input channel ci0 + channel and the 2 x 4
// (gated) output-gradient elements for output channel co0 + channel straight into registers -- a quarter-wave reads 64
// contiguous bytes of a row -- transforms them (56 + 24 additions) and writes its two patches of every position with one
// 8-byte LDS store each.  Two LDS stages, one barrier per chunk; the transform of chunk c+1 and the loads of chunk c+2
// are dealt out over chunk c's 64 MFMAs.  The sign of the fourth row / column of A is left out of DY and put back in
// the epilogue, which also applies G^T . G and writes the 3x3 partials part[split][tap][co][ci]; the fixed-order
// reduce of conv3x3_wgrad.hip finishes (bitwise reproducible, no float atomics).  Per-sample channel sums of the gated
// output gradient (bias and action-plane gradients) are summed on the way by the ci-group-0 workgroups.
#include "wgrad_args.h"
#include <type_traits>

#define WW_SLAB 8192      // floats of one operand stage: [16][2][64][4]

typedef unsigned u32x4q __attribute__((ext_vector_type(4)));

// Phase clock / ablation switches for tools/exp/ww_phases.hip (compiled in only with -DWW_TIMING=<workgroup>)
#ifdef WW_TIMING
__device__ unsigned long long ww_ticks[4];
#define WW_TICK(i) do { const unsigned long long t_ = clock64(); t_acc[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define WW_TICK(i)
#endif
#ifndef WW_ABL
#define WW_ABL 0
#endif

// GATED: 0 none, 1 gate tensor (act' from the saved output), 2 1-bit leaky-relu mask.  PADC: circular padding.
// EDGE: some patches / rows / output channels of a chunk may not exist (H odd, patch row not a multiple of 8 patches,
// cout not a multiple of 64): their output gradient is forced to zero.
template <int GATED, int PADC, int EDGE>
__global__ void __launch_bounds__(256, 1) conv3x3_wgrad_wino_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const DYL = smem;                     // 2 stages of DY [16][2][64][4]
    float* const VL = smem + 2 * WW_SLAB;        // 2 stages of V

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int q = tid & 3, ch = tid >> 2;        // this thread's patch pair of a chunk and channel of the group
    const int W = a.W, H = a.H, HW = H * W;      // W % 4 == 0 (host check)
    const int TCW = W >> 1, TRH = (H + 1) >> 1;
    const int nchr = (TCW + 7) >> 3;             // chunks per patch row

    int split, grp;
    scm_xcd_split_group(a.spi, split, grp);
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int tr0 = si * a.rows_per_split;       // (rows_per_split counts PATCH rows here)
    int tr1 = tr0 + a.rows_per_split;
    if (tr1 > TRH) tr1 = TRH;
    const int nchunk = (tr1 > tr0 ? tr1 - tr0 : 0) * nchr;
    const int cog = grp / a.ci_groups, cig = grp - cog * a.ci_groups;
    const int cin = a.c1 + a.c2;
    const int co = cog * 64 + ch, ci = cig * 64 + ch;
    const int coc = co < a.cout ? co : a.cout - 1, cic = ci < cin ? ci : cin - 1;

    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gfp4;
    typedef const __attribute__((address_space(1))) float* gfp;
    typedef const __attribute__((address_space(1))) u32x4q* gup4;
    // Wave-uniform (SGPR) tensor bases of this image + 32-bit per-lane byte offsets of the thread's channel planes: the
    // row offset goes into the scalar base, the column offset into the lane offset -- no 64-bit vector arithmetic
    // per load.  The input channel group lies in x1 or in x2 as a whole (host check); xb is indexed by the
    // concatenated channel either way.
    const size_t plane_b = (size_t)HW * 4;
    const bool in1 = cig * 64 < a.c1;
    gcp xb = in1 ? (gcp)(a.x1) + (size_t)b * a.c1 * plane_b : (gcp)(a.x2) + ((size_t)b * a.c2 - a.c1) * plane_b;
    gcp db = (gcp)(a.dy) + (size_t)b * a.cout * plane_b;
    gcp gb = GATED == 1 ? (gcp)(a.gate) + (size_t)b * a.cout * plane_b : nullptr;
    const int mask_nw = (a.cout + 31) >> 5;
    gcp mb = GATED == 2 ? (gcp)(a.gate_mask) + (size_t)b * mask_nw * plane_b : nullptr;
    const unsigned xch = (unsigned)cic * (unsigned)plane_b, dch = (unsigned)coc * (unsigned)plane_b;
    const unsigned mch = (unsigned)(coc >> 5) * (unsigned)plane_b;
    const unsigned mbit = 1u << (coc & 31);
    const float chv = co < a.cout ? 1.f : 0.f;
    auto vo = [](unsigned o) { return o; };       // (the lane offsets are laundered in place, once per chunk: ld_setup)

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- registers that live across the K loop.  Chunk c's elements live in register set c & 1: while chunk k's
    // MFMAs run, chunk k+1 (set (k+1) & 1, requested a whole iteration ago) is transformed and chunk k+2 is requested
    // into set k & 1 -- a full chunk of MFMAs between a request and its first use (these are gathers: a quarter-wave
    // per row segment, mostly beyond L2).  The K loop is unrolled by two so that every register index is static.
    float xr[2][4][6];      // input rows 2*ty-1 .. 2*ty+2, columns ix0-1 .. ix0+4 of a patch pair
    float dr[2][2][4];      // output-gradient rows 2*ty, 2*ty+1, columns ix0 .. ix0+3
    float gr[2][GATED == 1 ? 2 : 1][4];
    unsigned mr[2][GATED == 2 ? 2 : 1][4];
    float x_e[2][4], x_e2[2][EDGE ? 4 : 1];   // the outer column(s) of each row as loaded (placed by ld_xfix)
    int l_ty[2] = {0, 0};
    float l_em[2] = {1.f, 1.f}, l_em2[2] = {1.f, 1.f};
    unsigned l_xe = 0, l_xe2 = 0, l_xm = 0, l_dm = 0, l_mm = 0;
    float vx[2][16], vd[2][16];
    float aq[2][4], bq[2][4];
    float csum = 0.f;
    const bool q0 = q == 0, q3 = q == 3;
#define IC(n) std::integral_constant<int, (n)>{}
#define SB() __builtin_amdgcn_sched_barrier(0)

    // Chunk k of the band walks DOWN a column of patches: cx = k / ntr (patches 8*cx .. 8*cx+7), patch row ty = tr0 +
    // k % ntr.  Consecutive chunks then share two of their four input rows (rows 2, 3 of chunk k are rows 0, 1 of chunk
    // k+1): those are copied from the other register set instead of being requested again (row_reuse), except at the
    // top of a column.  The loads of a chunk past the end re-read the last chunk (never used).  ld_setup fixes the
    // chunk's lane offsets, ld_xrow / ld_xedge / ld_drow request one row each into register set P.
    const int ntr = tr1 > tr0 ? tr1 - tr0 : 1;
    auto chunk_pos = [&](int k, int& ty, int& cx) {
        const int kc = k < nchunk ? k : (nchunk > 0 ? nchunk - 1 : 0);
        cx = kc / ntr;
        ty = tr0 + (kc - cx * ntr);
    };
    bool l_reuse[2] = {false, false};       // rows 0, 1 of the set came from the previous chunk (already fixed)
    auto ld_setup = [&](int k, auto p_) {
        constexpr int P = decltype(p_)::value;
        int ty, cx;
        chunk_pos(k, ty, cx);
        l_ty[P] = ty;
        // (k >= 1: the chunk before it is in the other set; past the end the clamped chunk repeats: no reuse)
        l_reuse[P] = k >= 1 && k < nchunk && ty > tr0;
        const int ix0 = 16 * cx + 4 * q;                 // first image column of the pair's outputs
        const int ixm = ix0 < W ? ix0 : W - 4;           // (a pair past the row end reads the last one: finite data)
        // Of the pair's six columns the middle four are one aligned 16-byte load; the outer two are the neighbouring
        // pair's edge columns (lanes q-1 / q+1 of the quad: a DPP move each) -- except the chunk's own outer columns,
        // 16*cx - 1 for pair 0 and 16*cx + 16 for pair 3, which those two lanes fetch with ONE 4-byte load per row.
        // (EDGE builds -- a row may end inside a chunk -- load both outer columns of every pair instead)
        int ixe = EDGE ? ixm - 1 : (q3 ? 16 * cx + 16 : 16 * cx - 1);
        const bool ein = ixe >= 0 && ixe < W;
        ixe = ixe < 0 ? (PADC ? W - 1 : 0) : (ixe >= W ? (PADC ? 0 : W - 1) : ixe);
        if constexpr (!PADC) l_em[P] = ein ? 1.f : 0.f;
        if constexpr (EDGE) {
            int ixr = ixm + 4;
            if constexpr (!PADC) l_em2[P] = ixr < W ? 1.f : 0.f;
            ixr = ixr >= W ? (PADC ? 0 : W - 1) : ixr;
            l_xe2 = xch + ixr * 4;
        }
        l_xe = xch + ixe * 4; l_xm = xch + ixm * 4;
        l_dm = dch + ixm * 4; l_mm = mch + ixm * 4;
        // (keeps the zero-extension of the offsets next to the loads: scalar-base addressing without a copy per load)
        asm volatile("" : "+v"(l_xe), "+v"(l_xm), "+v"(l_dm), "+v"(l_mm));
        if constexpr (EDGE) asm volatile("" : "+v"(l_xe2));
    };
    auto xrow_base = [&](int ty, int i) {
        int iy = 2 * ty - 1 + i;
        if (PADC) iy = iy < 0 ? iy + H : (iy >= H ? iy - H : iy);
        else iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        return xb + (size_t)iy * (W * 4);
    };
    auto ld_xrow = [&](auto p_, auto i_) {
        constexpr int P = decltype(p_)::value, I = decltype(i_)::value;
        gcp row = xrow_base(l_ty[P], I);
        const f32x4 m = *(gfp4)(row + vo(l_xm));
        xr[P][I][1] = m[0]; xr[P][I][2] = m[1]; xr[P][I][3] = m[2]; xr[P][I][4] = m[3];
    };
    auto ld_xedge = [&](auto p_, auto i_) {
        constexpr int P = decltype(p_)::value, I = decltype(i_)::value;
        gcp row = xrow_base(l_ty[P], I);
        x_e[P][I] = *(gfp)(row + vo(l_xe));
        if constexpr (EDGE) x_e2[P][I] = *(gfp)(row + vo(l_xe2));
    };
    // after the loads have landed: outer columns from the quad neighbours / the edge load, zero padding applied
    auto ld_xfix = [&](auto p_, auto i_) {
        constexpr int P = decltype(p_)::value, I = decltype(i_)::value;
        float e = x_e[P][I], e2 = EDGE ? x_e2[P][EDGE ? I : 0] : 0.f;
        if constexpr (!PADC) {
            const int iy = 2 * l_ty[P] - 1 + I;
            const float ym = (iy >= 0 && iy < H) ? 1.f : 0.f;
            e *= ym * l_em[P];
            if constexpr (EDGE) e2 *= ym * l_em2[P];
            xr[P][I][1] *= ym; xr[P][I][2] *= ym; xr[P][I][3] *= ym; xr[P][I][4] *= ym;
        }
        if constexpr (EDGE) {
            xr[P][I][0] = e; xr[P][I][5] = e2;
            return;
        }
        // quad_perm [0,0,1,2] = 0x90: lane q reads lane q-1;  [1,2,3,3] = 0xF9: lane q reads lane q+1
        const float fl = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xr[P][I][4]), 0x90, 0xF, 0xF, true));
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xr[P][I][1]), 0xF9, 0xF, 0xF, true));
        xr[P][I][0] = q0 ? e : fl;
        xr[P][I][5] = q3 ? e : fr;
    };
    auto ld_drow = [&](auto p_, auto i_) {
        constexpr int P = decltype(p_)::value, I = decltype(i_)::value;
        int oy = 2 * l_ty[P] + I;
        oy = oy < H ? oy : H - 1;
        const size_t ro = (size_t)oy * (W * 4);
        gcp drow = db + ro;
        const f32x4 d = *(gfp4)(drow + vo(l_dm));
        dr[P][I][0] = d[0]; dr[P][I][1] = d[1]; dr[P][I][2] = d[2]; dr[P][I][3] = d[3];
        if constexpr (GATED == 1) {
            gcp grow = gb + ro;
            const f32x4 g = *(gfp4)(grow + vo(l_dm));
            gr[P][I][0] = g[0]; gr[P][I][1] = g[1]; gr[P][I][2] = g[2]; gr[P][I][3] = g[3];
        }
        if constexpr (GATED == 2) {
            gcp mrow = mb + ro;
            const u32x4q m = *(gup4)(mrow + vo(l_mm));
            mr[P][I][0] = m[0]; mr[P][I][1] = m[1]; mr[P][I][2] = m[2]; mr[P][I][3] = m[3];
        }
    };
    auto load_chunk = [&](int k, auto p_) {          // (prologue: all four rows, no reuse)
        constexpr int P = decltype(p_)::value;
        ld_setup(k, p_);
        l_reuse[P] = false;
        ld_xrow(p_, IC(0)); ld_xrow(p_, IC(1)); ld_xrow(p_, IC(2)); ld_xrow(p_, IC(3));
        ld_xedge(p_, IC(0)); ld_xedge(p_, IC(1)); ld_xedge(p_, IC(2)); ld_xedge(p_, IC(3));
        ld_drow(p_, IC(0)); ld_drow(p_, IC(1));
    };
    auto fix_chunk = [&](auto p_) {
        constexpr int P = decltype(p_)::value;
        if (!l_reuse[P]) { ld_xfix(p_, IC(0)); ld_xfix(p_, IC(1)); }
        ld_xfix(p_, IC(2)); ld_xfix(p_, IC(3));
    };
    // rows 2, 3 of the (fixed, not yet transformed) set Q are rows 0, 1 of the next chunk in set P
    auto row_reuse = [&](auto p_, auto q_) {
        constexpr int P = decltype(p_)::value, Q = decltype(q_)::value;
        if (l_reuse[P]) {
#pragma unroll
            for (int j = 0; j < 6; ++j) { xr[P][0][j] = xr[Q][2][j]; xr[P][1][j] = xr[Q][3][j]; }
        }
    };
    // gate, validity and channel sum of the loaded output gradient (chunk k, register set P)
    auto gate_chunk = [&](int k, auto p_) {
        constexpr int P = decltype(p_)::value;
        const bool real = k < nchunk;
        const float realf = real ? 1.f : 0.f;
        const int kc = real ? k : (nchunk > 0 ? nchunk - 1 : 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dr[P][i][j];
                if constexpr (GATED == 1) v *= scm_act_grad_from_out(gr[P][i][j], a.gate_act);
                if constexpr (GATED == 2) v = (mr[P][i][j] & mbit) ? v : 0.01f * v;
                if constexpr (EDGE) {
                    int ty, cx;
                    chunk_pos(kc, ty, cx);
                    const bool ok = (2 * ty + i < H) && (16 * cx + 4 * q + j < W);
                    v = ok ? v * chv : 0.f;
                }
                v *= realf;
                dr[P][i][j] = v;
                s += v;
            }
        csum += s;
    };
    // A dY A^T without the signs of its fourth row / column (put back in the epilogue), patch T of the pair
    auto dy_xform = [&](auto p_, auto t_) {
        constexpr int P = decltype(p_)::value, T = decltype(t_)::value;
        const float p = dr[P][0][2 * T], qq = dr[P][0][2 * T + 1], r = dr[P][1][2 * T], s = dr[P][1][2 * T + 1];
        const float r1a = p + r, r1b = qq + s, r2a = p - r, r2b = qq - s;
        vd[T][0] = p;   vd[T][1] = p + qq;     vd[T][2] = p - qq;     vd[T][3] = qq;
        vd[T][4] = r1a; vd[T][5] = r1a + r1b;  vd[T][6] = r1a - r1b;  vd[T][7] = r1b;
        vd[T][8] = r2a; vd[T][9] = r2a + r2b;  vd[T][10] = r2a - r2b; vd[T][11] = r2b;
        vd[T][12] = r;  vd[T][13] = r + s;     vd[T][14] = r - s;     vd[T][15] = s;
    };
    // B^T d B: along y for column J of the pair's 6 columns (in place), then along x for patch T
    auto x_vert = [&](auto p_, auto j_) {
        constexpr int P = decltype(p_)::value, J = decltype(j_)::value;
        const float d0 = xr[P][0][J], d1 = xr[P][1][J], d2 = xr[P][2][J], d3 = xr[P][3][J];
        xr[P][0][J] = d0 - d2; xr[P][1][J] = d1 + d2; xr[P][2][J] = d2 - d1; xr[P][3][J] = d1 - d3;
    };
    auto x_horz = [&](auto p_, auto t_, auto i_) {
        constexpr int P = decltype(p_)::value, T = decltype(t_)::value, I = decltype(i_)::value;
        const float c0 = xr[P][I][2 * T], c1 = xr[P][I][2 * T + 1], c2 = xr[P][I][2 * T + 2], c3 = xr[P][I][2 * T + 3];
        vx[T][4 * I + 0] = c0 - c2; vx[T][4 * I + 1] = c1 + c2; vx[T][4 * I + 2] = c2 - c1; vx[T][4 * I + 3] = c1 - c3;
    };
    // the pair's two patches of position F -> operand image [F][half][channel][step]: patch 2*q + t of the chunk is
    // MFMA k index (half = q >> 1, step = 2 * (q & 1) + t)
    const int wofs = ((q >> 1) * 64 + ch) * 4 + (q & 1) * 2;
    auto dy_write = [&](auto st_, auto f_) {
        constexpr int ST = decltype(st_)::value, F = decltype(f_)::value;
        f32x2 o; o[0] = vd[0][F]; o[1] = vd[1][F];
        *reinterpret_cast<f32x2*>(DYL + ST * WW_SLAB + F * 512 + wofs) = o;
    };
    auto x_write = [&](auto st_, auto f_) {
        constexpr int ST = decltype(st_)::value, F = decltype(f_)::value;
        f32x2 o; o[0] = vx[0][F]; o[1] = vx[1][F];
        *reinterpret_cast<f32x2*>(VL + ST * WW_SLAB + F * 512 + wofs) = o;
    };
    auto ab_read = [&](auto st_, auto f_) {
        constexpr int ST = decltype(st_)::value, F = decltype(f_)::value;
        const f32x4 ta = *reinterpret_cast<const f32x4*>(DYL + ST * WW_SLAB + F * 512 + (lh * 64 + wm * 32 + l31) * 4);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(VL + ST * WW_SLAB + F * 512 + (lh * 64 + wn * 32 + l31) * 4);
        aq[F & 1][0] = ta[0]; aq[F & 1][1] = ta[1]; aq[F & 1][2] = ta[2]; aq[F & 1][3] = ta[3];
        bq[F & 1][0] = tb[0]; bq[F & 1][1] = tb[1]; bq[F & 1][2] = tb[2]; bq[F & 1][3] = tb[3];
    };
    // One position F of chunk k (LDS stage and register parity P = k & 1): 4 MFMAs; behind them the transform of
    // chunk k+1 (register set P ^ 1) into stage P ^ 1 and the requests of chunk k+2 into register set P, one per MFMA.
    auto slot = [&](int k, auto p_, auto f_) {
        constexpr int P = decltype(p_)::value, F = decltype(f_)::value;
        auto mf = [&](auto s_) {
            constexpr int S = decltype(s_)::value;
            acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[F & 1][S], bq[F & 1][S], acc[F], 0, 0, 0);
        };
        constexpr bool DO_L = WW_ABL < 1, DO_T = WW_ABL < 2, DO_W = WW_ABL < 3, DO_R = WW_ABL < 4;
        constexpr auto PN = IC(P ^ 1);
        mf(IC(0));
        if constexpr (F < 15 && DO_R) ab_read(p_, IC(F < 15 ? F + 1 : 0));
        if constexpr (F == 0 && DO_L) { ld_setup(k + 2, p_); ld_xrow(p_, IC(2)); }
        if constexpr (F == 1 && DO_L) ld_xedge(p_, IC(2));
        if constexpr (F == 2 && DO_L) ld_drow(p_, IC(0));
        if constexpr (F == 3 && DO_L) { if (!l_reuse[P]) { ld_xrow(p_, IC(0)); ld_xedge(p_, IC(0)); } }
        SB();
        mf(IC(1));
        if constexpr (F == 0 && DO_T) gate_chunk(k + 1, PN);
        if constexpr (F == 1 && DO_T) { dy_xform(PN, IC(0)); dy_xform(PN, IC(1)); fix_chunk(PN); if constexpr (DO_L) row_reuse(p_, PN); }
        if constexpr (F == 2 && DO_T) { x_vert(PN, IC(0)); x_vert(PN, IC(1)); x_vert(PN, IC(2)); }
        if constexpr (F == 3 && DO_T) { x_vert(PN, IC(3)); x_vert(PN, IC(4)); x_vert(PN, IC(5)); }
        if constexpr (F == 4 && DO_T) { x_horz(PN, IC(0), IC(0)); x_horz(PN, IC(0), IC(1)); }
        if constexpr (F == 5 && DO_T) { x_horz(PN, IC(0), IC(2)); x_horz(PN, IC(0), IC(3)); }
        if constexpr (F == 6 && DO_T) { x_horz(PN, IC(1), IC(0)); x_horz(PN, IC(1), IC(1)); }
        if constexpr (F == 7 && DO_T) { x_horz(PN, IC(1), IC(2)); x_horz(PN, IC(1), IC(3)); }
        if constexpr (F >= 8 && DO_W) { dy_write(PN, IC(F >= 8 ? 2 * (F - 8) : 0)); x_write(PN, IC(F >= 8 ? 2 * (F - 8) : 0)); }
        SB();
        mf(IC(2));
        if constexpr (F >= 8 && DO_W) { dy_write(PN, IC(F >= 8 ? 2 * (F - 8) + 1 : 0)); x_write(PN, IC(F >= 8 ? 2 * (F - 8) + 1 : 0)); }
        SB();
        mf(IC(3));
        if constexpr (F == 0 && DO_L) ld_xrow(p_, IC(3));
        if constexpr (F == 1 && DO_L) ld_xedge(p_, IC(3));
        if constexpr (F == 2 && DO_L) ld_drow(p_, IC(1));
        if constexpr (F == 3 && DO_L) { if (!l_reuse[P]) { ld_xrow(p_, IC(1)); ld_xedge(p_, IC(1)); } }
        SB();
    };
    auto chunk_body = [&](int k, auto p_) {
        if (WW_ABL < 4) ab_read(p_, IC(0));
        SB();
        slot(k, p_, IC(0)); slot(k, p_, IC(1)); slot(k, p_, IC(2)); slot(k, p_, IC(3));
        slot(k, p_, IC(4)); slot(k, p_, IC(5)); slot(k, p_, IC(6)); slot(k, p_, IC(7));
        slot(k, p_, IC(8)); slot(k, p_, IC(9)); slot(k, p_, IC(10)); slot(k, p_, IC(11));
        slot(k, p_, IC(12)); slot(k, p_, IC(13)); slot(k, p_, IC(14)); slot(k, p_, IC(15));
        __syncthreads();            // the other stage holds chunk k+1; every wave is done with this one
        SB();
    };

#ifdef WW_TIMING
    unsigned long long t_acc[4] = {0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    // ---- prologue: chunk 0 (register set 0) transformed into stage 0, chunk 1 requested into set 1
    load_chunk(0, IC(0));
    load_chunk(1, IC(1));
    fix_chunk(IC(0));
    gate_chunk(0, IC(0));
    dy_xform(IC(0), IC(0)); dy_xform(IC(0), IC(1));
    x_vert(IC(0), IC(0)); x_vert(IC(0), IC(1)); x_vert(IC(0), IC(2)); x_vert(IC(0), IC(3)); x_vert(IC(0), IC(4)); x_vert(IC(0), IC(5));
    x_horz(IC(0), IC(0), IC(0)); x_horz(IC(0), IC(0), IC(1)); x_horz(IC(0), IC(0), IC(2)); x_horz(IC(0), IC(0), IC(3));
    x_horz(IC(0), IC(1), IC(0)); x_horz(IC(0), IC(1), IC(1)); x_horz(IC(0), IC(1), IC(2)); x_horz(IC(0), IC(1), IC(3));
    dy_write(IC(0), IC(0)); dy_write(IC(0), IC(1)); dy_write(IC(0), IC(2)); dy_write(IC(0), IC(3)); dy_write(IC(0), IC(4)); dy_write(IC(0), IC(5));
    dy_write(IC(0), IC(6)); dy_write(IC(0), IC(7)); dy_write(IC(0), IC(8)); dy_write(IC(0), IC(9)); dy_write(IC(0), IC(10)); dy_write(IC(0), IC(11));
    dy_write(IC(0), IC(12)); dy_write(IC(0), IC(13)); dy_write(IC(0), IC(14)); dy_write(IC(0), IC(15));
    x_write(IC(0), IC(0)); x_write(IC(0), IC(1)); x_write(IC(0), IC(2)); x_write(IC(0), IC(3)); x_write(IC(0), IC(4)); x_write(IC(0), IC(5));
    x_write(IC(0), IC(6)); x_write(IC(0), IC(7)); x_write(IC(0), IC(8)); x_write(IC(0), IC(9)); x_write(IC(0), IC(10)); x_write(IC(0), IC(11));
    x_write(IC(0), IC(12)); x_write(IC(0), IC(13)); x_write(IC(0), IC(14)); x_write(IC(0), IC(15));
    __syncthreads();
    WW_TICK(0);
    // (always whole pairs of chunks -- the accumulators must not cross a conditional: a chunk past the end has its
    // output gradient forced to zero in gate_chunk and adds nothing)
    for (int k = 0; k < nchunk; k += 2) {
        chunk_body(k, IC(0));
        chunk_body(k + 1, IC(1));
    }
#undef IC
#undef SB
    WW_TICK(1);

    // ---- epilogue: signs of A's fourth row / column, dg = G^T dU G, 3x3 partials of this split
    //   G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
    const int cil = cig * 64 + wn * 32 + l31;
    float* pbase = a.part + (size_t)split * 9 * a.cout * cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int col = cog * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float u[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) u[i][j] = ((i == 3) != (j == 3)) ? -acc[4 * i + j][r] : acc[4 * i + j][r];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h1 = 0.5f * (u[1][j] + u[2][j]), h2 = 0.5f * (u[1][j] - u[2][j]);
            t[0][j] = u[0][j] + h1;
            t[1][j] = h2;
            t[2][j] = h1 + u[3][j];
        }
        if (col < a.cout && cil < cin) {
            float* dst = pbase + (size_t)col * cin + cil;
            const size_t ts = (size_t)a.cout * cin;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float h1 = 0.5f * (t[p][1] + t[p][2]), h2 = 0.5f * (t[p][1] - t[p][2]);
                dst[(3 * p + 0) * ts] = t[p][0] + h1;
                dst[(3 * p + 1) * ts] = h2;
                dst[(3 * p + 2) * ts] = h1 + t[p][3];
            }
        }
    }
    if (a.csum_part && cig == 0) {
        // per-split channel sum of the gated output gradient: the four patch-pair lanes of a channel
        float s = csum;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q == 0 && co < a.cout) a.csum_part[(size_t)split * a.cout + co] = s;
    }
#ifdef WW_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    WW_TICK(2);
    if (split == WW_TIMING && grp == 0 && tid == 0)
        for (int i = 0; i < 4; ++i) ww_ticks[i] = t_acc[i];
#endif
}

// ------------------------------------------------------------------ host
template <int GATED, int PADC>
static int launch_ww(const WgradKArgs& a, dim3 grid, bool edge, hipStream_t s) {
    const size_t lds = (size_t)4 * WW_SLAB * sizeof(float);
    auto kern = edge ? conv3x3_wgrad_wino_kernel<GATED, PADC, 1> : conv3x3_wgrad_wino_kernel<GATED, PADC, 0>;
    static std::atomic<uint64_t> attr_done[2];       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done[edge ? 1 : 0], 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_wino_kernel");
    return SCM_OK;
}

// Shapes the kernel takes: 16-byte aligned rows of whole 4-pixel groups, whole 64-channel groups of the first input
// tensor when there is a second one.
bool scm_wgrad_wino_serves(const WgradKArgs& a) {
    const size_t al = reinterpret_cast<size_t>(a.x1) | reinterpret_cast<size_t>(a.x2) | reinterpret_cast<size_t>(a.dy) |
                      reinterpret_cast<size_t>(a.gate) | reinterpret_cast<size_t>(a.gate_mask);
    return a.W >= 4 && (a.W & 3) == 0 && a.H >= 2 && (al & 15) == 0 && (a.c2 == 0 || (a.c1 & 63) == 0);
}

// One workgroup per (split, 64 co x 64 ci group); `max_splits`: what the caller's workspaces were sized for.
int scm_launch_wgrad_wino(const WgradKArgs& k, int max_splits, int* nsplit_out, int* spi_out, hipStream_t s) {
    WgradKArgs a = k;
    const int cin = a.c1 + a.c2;
    const int cog = (a.cout + 63) / 64, cig = (cin + 63) / 64;
    const int trh = (a.H + 1) / 2;
    int spi = (256 + a.B * cog * cig - 1) / (a.B * cog * cig);
    if (spi < 1) spi = 1;
    if (spi > trh) spi = trh;
    if (spi * a.B > max_splits) spi = max_splits / a.B;
    if (spi < 1) {
        scm_set_error("scm_conv3x3_wgrad_wino: workspace sized for %d splits, batch %d", max_splits, a.B);
        return SCM_EINVAL;
    }
    a.rows_per_split = (trh + spi - 1) / spi;           // patch rows per split
    a.spi = (trh + a.rows_per_split - 1) / a.rows_per_split;
    a.ci_groups = cig;
    *nsplit_out = a.B * a.spi;
    *spi_out = a.spi;
    const bool edge = (a.H & 1) || ((a.W >> 1) & 7) || (a.cout & 63);
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int g = a.gate_act == 0 ? 0 : ((a.gate_mask && a.gate_act == SCM_ACT_LRELU) ? 2 : 1);
    dim3 grid(a.B * a.spi, cog * cig);
    if (scm_wgrad_winoz_serves(a)) {        // (ragged last strips included; odd heights / ragged channel groups are not)
        const int rc = scm_launch_wgrad_winoz(a, grid, s);      // (declines bands that are not whole groups of four chunks)
        if (rc != SCM_EUNSUPPORTED) return rc;
    }
#define WW_P(G_) (circ ? launch_ww<G_, 1>(a, grid, edge, s) : launch_ww<G_, 0>(a, grid, edge, s))
    return g == 2 ? WW_P(2) : (g == 1 ? WW_P(1) : WW_P(0));
#undef WW_P
}
