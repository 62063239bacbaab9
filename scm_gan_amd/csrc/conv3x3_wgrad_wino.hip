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

    const int split = blockIdx.x;
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int tr0 = si * a.rows_per_split;       // (rows_per_split counts PATCH rows here)
    int tr1 = tr0 + a.rows_per_split;
    if (tr1 > TRH) tr1 = TRH;
    const int nchunk = (tr1 > tr0 ? tr1 - tr0 : 0) * nchr;
    const int cog = blockIdx.y / a.ci_groups, cig = blockIdx.y - cog * a.ci_groups;
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
    auto vo = [](unsigned o) { asm volatile("" : "+v"(o)); return o; };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- registers that live across the K loop
    float xr[4][6];         // input rows 2*ty-1 .. 2*ty+2, columns ix0-1 .. ix0+4 of the NEXT chunk's patch pair
    float dr[2][4];         // output-gradient rows 2*ty, 2*ty+1, columns ix0 .. ix0+3
    float gr[GATED == 1 ? 2 : 1][4];
    unsigned mr[GATED == 2 ? 2 : 1][4];
    float vx[2][16], vd[2][16];
    float aq[2][4], bq[2][4];
    float csum = 0.f;

    // chunk k of the band: patch row ty = tr0 + k / nchr, patches 8*cx .. 8*cx+7, cx = k % nchr.  The loads of a chunk
    // past the end re-read the last chunk (never used).  ld_setup fixes the chunk's lane offsets, ld_xrow / ld_drow
    // request one input / output-gradient row each.
    int l_ty = 0;
    unsigned l_xl = 0, l_xm = 0, l_xr = 0, l_dm = 0, l_mm = 0;
    float l_lm = 1.f, l_rm = 1.f;
    auto ld_setup = [&](int k) {
        const int kc = k < nchunk ? k : (nchunk > 0 ? nchunk - 1 : 0);
        l_ty = tr0 + kc / nchr;
        const int cx = kc - (kc / nchr) * nchr;
        const int ix0 = 16 * cx + 4 * q;                 // first image column of the pair's outputs
        const int ixm = ix0 < W ? ix0 : W - 4;           // (a pair past the row end reads the last one: finite data)
        int ixl = ixm - 1, ixr = ixm + 4;
        ixl = ixl < 0 ? (PADC ? W - 1 : 0) : ixl;
        ixr = ixr >= W ? (PADC ? 0 : W - 1) : ixr;
        if constexpr (!PADC) {
            l_lm = ixm == 0 ? 0.f : 1.f;
            l_rm = ixm + 4 >= W ? 0.f : 1.f;
        }
        l_xl = xch + ixl * 4; l_xm = xch + ixm * 4; l_xr = xch + ixr * 4;
        l_dm = dch + ixm * 4; l_mm = mch + ixm * 4;
    };
    auto ld_xrow = [&](auto i_) {
        constexpr int I = decltype(i_)::value;
        int iy = 2 * l_ty - 1 + I;
        float ym = 1.f;
        if (PADC) iy = iy < 0 ? iy + H : (iy >= H ? iy - H : iy);
        else { ym = (iy >= 0 && iy < H) ? 1.f : 0.f; iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy); }
        gcp row = xb + (size_t)iy * (W * 4);
        asm volatile("" : "+s"(row));
        const f32x4 m = *(gfp4)(row + vo(l_xm));
        xr[I][0] = *(gfp)(row + vo(l_xl));
        xr[I][1] = m[0]; xr[I][2] = m[1]; xr[I][3] = m[2]; xr[I][4] = m[3];
        xr[I][5] = *(gfp)(row + vo(l_xr));
        if constexpr (!PADC) {
            xr[I][0] *= ym * l_lm; xr[I][5] *= ym * l_rm;
            xr[I][1] *= ym; xr[I][2] *= ym; xr[I][3] *= ym; xr[I][4] *= ym;
        }
    };
    auto ld_drow = [&](auto i_) {
        constexpr int I = decltype(i_)::value;
        int oy = 2 * l_ty + I;
        oy = oy < H ? oy : H - 1;
        const size_t ro = (size_t)oy * (W * 4);
        gcp drow = db + ro;
        asm volatile("" : "+s"(drow));
        const f32x4 d = *(gfp4)(drow + vo(l_dm));
        dr[I][0] = d[0]; dr[I][1] = d[1]; dr[I][2] = d[2]; dr[I][3] = d[3];
        if constexpr (GATED == 1) {
            gcp grow = gb + ro;
            asm volatile("" : "+s"(grow));
            const f32x4 g = *(gfp4)(grow + vo(l_dm));
            gr[I][0] = g[0]; gr[I][1] = g[1]; gr[I][2] = g[2]; gr[I][3] = g[3];
        }
        if constexpr (GATED == 2) {
            gcp mrow = mb + ro;
            asm volatile("" : "+s"(mrow));
            const u32x4q m = *(gup4)(mrow + vo(l_mm));
            mr[I][0] = m[0]; mr[I][1] = m[1]; mr[I][2] = m[2]; mr[I][3] = m[3];
        }
    };
#define IC(n) std::integral_constant<int, (n)>{}
    auto load_chunk = [&](int k) {
        ld_setup(k);
        ld_xrow(IC(0)); ld_xrow(IC(1)); ld_xrow(IC(2)); ld_xrow(IC(3));
        ld_drow(IC(0)); ld_drow(IC(1));
    };
    // gate, validity and channel sum of the loaded output gradient (chunk k)
    auto gate_chunk = [&](int k) {
        const bool real = k < nchunk;
        const int kc = real ? k : (nchunk > 0 ? nchunk - 1 : 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dr[i][j];
                if constexpr (GATED == 1) v *= scm_act_grad_from_out(gr[i][j], a.gate_act);
                if constexpr (GATED == 2) v = (mr[i][j] & mbit) ? v : 0.01f * v;
                if constexpr (EDGE) {
                    const int ty = tr0 + kc / nchr, cx = kc - (kc / nchr) * nchr;
                    const bool ok = (2 * ty + i < H) && (16 * cx + 4 * q + j < W);
                    v = ok ? v * chv : 0.f;
                }
                dr[i][j] = v;
                s += v;
            }
        csum += real ? s : 0.f;
    };
    // A dY A^T without the signs of its fourth row / column (put back in the epilogue), patch T of the pair
    auto dy_xform = [&](auto t_) {
        constexpr int T = decltype(t_)::value;
        const float p = dr[0][2 * T], qq = dr[0][2 * T + 1], r = dr[1][2 * T], s = dr[1][2 * T + 1];
        const float r1a = p + r, r1b = qq + s, r2a = p - r, r2b = qq - s;
        vd[T][0] = p;   vd[T][1] = p + qq;     vd[T][2] = p - qq;     vd[T][3] = qq;
        vd[T][4] = r1a; vd[T][5] = r1a + r1b;  vd[T][6] = r1a - r1b;  vd[T][7] = r1b;
        vd[T][8] = r2a; vd[T][9] = r2a + r2b;  vd[T][10] = r2a - r2b; vd[T][11] = r2b;
        vd[T][12] = r;  vd[T][13] = r + s;     vd[T][14] = r - s;     vd[T][15] = s;
    };
    // B^T d B: along y for column J of the pair's 6 columns (in place), then along x for patch T
    auto x_vert = [&](auto j_) {
        constexpr int J = decltype(j_)::value;
        const float d0 = xr[0][J], d1 = xr[1][J], d2 = xr[2][J], d3 = xr[3][J];
        xr[0][J] = d0 - d2; xr[1][J] = d1 + d2; xr[2][J] = d2 - d1; xr[3][J] = d1 - d3;
    };
    auto x_horz = [&](auto t_, auto i_) {
        constexpr int T = decltype(t_)::value, I = decltype(i_)::value;
        const float c0 = xr[I][2 * T], c1 = xr[I][2 * T + 1], c2 = xr[I][2 * T + 2], c3 = xr[I][2 * T + 3];
        vx[T][4 * I + 0] = c0 - c2; vx[T][4 * I + 1] = c1 + c2; vx[T][4 * I + 2] = c2 - c1; vx[T][4 * I + 3] = c1 - c3;
    };
    // the pair's two patches of position F -> operand image [F][half][channel][step]: patch 2*q + t of the chunk is
    // MFMA k index (half = q >> 1, step = 2 * (q & 1) + t)
    const int wofs = ((q >> 1) * 64 + ch) * 4 + (q & 1) * 2;
    auto dy_write = [&](int st, auto f_) {
        constexpr int F = decltype(f_)::value;
        f32x2 o; o[0] = vd[0][F]; o[1] = vd[1][F];
        *reinterpret_cast<f32x2*>(DYL + st * WW_SLAB + F * 512 + wofs) = o;
    };
    auto x_write = [&](int st, auto f_) {
        constexpr int F = decltype(f_)::value;
        f32x2 o; o[0] = vx[0][F]; o[1] = vx[1][F];
        *reinterpret_cast<f32x2*>(VL + st * WW_SLAB + F * 512 + wofs) = o;
    };
    auto ab_read = [&](int st, auto f_) {
        constexpr int F = decltype(f_)::value;
        const f32x4 ta = *reinterpret_cast<const f32x4*>(DYL + st * WW_SLAB + F * 512 + (lh * 64 + wm * 32 + l31) * 4);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(VL + st * WW_SLAB + F * 512 + (lh * 64 + wn * 32 + l31) * 4);
        aq[F & 1][0] = ta[0]; aq[F & 1][1] = ta[1]; aq[F & 1][2] = ta[2]; aq[F & 1][3] = ta[3];
        bq[F & 1][0] = tb[0]; bq[F & 1][1] = tb[1]; bq[F & 1][2] = tb[2]; bq[F & 1][3] = tb[3];
    };
#define SB() __builtin_amdgcn_sched_barrier(0)
    // One position F of chunk k (stage cur): 4 MFMAs; the transform of chunk k+1 (in registers) and the loads of chunk
    // k+2 are dealt out behind them.
    auto slot = [&](int k, int cur, auto f_) {
        constexpr int F = decltype(f_)::value;
        auto mf = [&](auto s_) {
            constexpr int S = decltype(s_)::value;
            acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[F & 1][S], bq[F & 1][S], acc[F], 0, 0, 0);
        };
        mf(IC(0));
        if constexpr (F < 15) ab_read(cur, IC(F < 15 ? F + 1 : 0));
        SB();
        mf(IC(1));
        if constexpr (F == 0) gate_chunk(k + 1);
        if constexpr (F == 1) { dy_xform(IC(0)); dy_xform(IC(1)); }
        if constexpr (F == 2) { x_vert(IC(0)); x_vert(IC(1)); x_vert(IC(2)); }
        if constexpr (F == 3) { x_vert(IC(3)); x_vert(IC(4)); x_vert(IC(5)); }
        if constexpr (F == 4) { x_horz(IC(0), IC(0)); x_horz(IC(0), IC(1)); }
        if constexpr (F == 5) { x_horz(IC(0), IC(2)); x_horz(IC(0), IC(3)); }
        if constexpr (F == 6) { x_horz(IC(1), IC(0)); x_horz(IC(1), IC(1)); }
        if constexpr (F == 7) { x_horz(IC(1), IC(2)); x_horz(IC(1), IC(3)); }
        if constexpr (F >= 8) { dy_write(cur ^ 1, IC(F >= 8 ? 2 * (F - 8) : 0)); x_write(cur ^ 1, IC(F >= 8 ? 2 * (F - 8) : 0)); }
        SB();
        mf(IC(2));
        if constexpr (F >= 8) { dy_write(cur ^ 1, IC(F >= 8 ? 2 * (F - 8) + 1 : 0)); x_write(cur ^ 1, IC(F >= 8 ? 2 * (F - 8) + 1 : 0)); }
        SB();
        mf(IC(3));
        // the registers of chunk k+1 are consumed by position 7: chunk k+2 is requested into them, a row per position
        if constexpr (F == 8) { ld_setup(k + 2); ld_xrow(IC(0)); }
        if constexpr (F == 9) ld_xrow(IC(1));
        if constexpr (F == 10) ld_xrow(IC(2));
        if constexpr (F == 11) ld_xrow(IC(3));
        if constexpr (F == 12) ld_drow(IC(0));
        if constexpr (F == 13) ld_drow(IC(1));
        SB();
    };

    // ---- prologue: chunk 0 transformed into stage 0, chunk 1 in the registers
    load_chunk(0);
    gate_chunk(0);
    dy_xform(IC(0)); dy_xform(IC(1));
    x_vert(IC(0)); x_vert(IC(1)); x_vert(IC(2)); x_vert(IC(3)); x_vert(IC(4)); x_vert(IC(5));
    x_horz(IC(0), IC(0)); x_horz(IC(0), IC(1)); x_horz(IC(0), IC(2)); x_horz(IC(0), IC(3));
    x_horz(IC(1), IC(0)); x_horz(IC(1), IC(1)); x_horz(IC(1), IC(2)); x_horz(IC(1), IC(3));
    load_chunk(1);
    dy_write(0, IC(0)); dy_write(0, IC(1)); dy_write(0, IC(2)); dy_write(0, IC(3)); dy_write(0, IC(4)); dy_write(0, IC(5));
    dy_write(0, IC(6)); dy_write(0, IC(7)); dy_write(0, IC(8)); dy_write(0, IC(9)); dy_write(0, IC(10)); dy_write(0, IC(11));
    dy_write(0, IC(12)); dy_write(0, IC(13)); dy_write(0, IC(14)); dy_write(0, IC(15));
    x_write(0, IC(0)); x_write(0, IC(1)); x_write(0, IC(2)); x_write(0, IC(3)); x_write(0, IC(4)); x_write(0, IC(5));
    x_write(0, IC(6)); x_write(0, IC(7)); x_write(0, IC(8)); x_write(0, IC(9)); x_write(0, IC(10)); x_write(0, IC(11));
    x_write(0, IC(12)); x_write(0, IC(13)); x_write(0, IC(14)); x_write(0, IC(15));
    __syncthreads();
    for (int k = 0; k < nchunk; ++k) {
        const int cur = k & 1;
        ab_read(cur, IC(0));
        SB();
        slot(k, cur, IC(0)); slot(k, cur, IC(1)); slot(k, cur, IC(2)); slot(k, cur, IC(3));
        slot(k, cur, IC(4)); slot(k, cur, IC(5)); slot(k, cur, IC(6)); slot(k, cur, IC(7));
        slot(k, cur, IC(8)); slot(k, cur, IC(9)); slot(k, cur, IC(10)); slot(k, cur, IC(11));
        slot(k, cur, IC(12)); slot(k, cur, IC(13)); slot(k, cur, IC(14)); slot(k, cur, IC(15));
        __syncthreads();            // stage cur ^ 1 holds chunk k+1; every wave is done with stage cur
        SB();
    }
#undef IC
#undef SB

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
#define WW_P(G_) (circ ? launch_ww<G_, 1>(a, grid, edge, s) : launch_ww<G_, 0>(a, grid, edge, s))
    return g == 2 ? WW_P(2) : (g == 1 ? WW_P(1) : WW_P(0));
#undef WW_P
}
