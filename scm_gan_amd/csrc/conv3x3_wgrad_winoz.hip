// Weight gradient by Winograd F(2x2, 3x3) on the fp32 MFMA pipe, second form ("z": the chunk walk zigzags over two
// neighbouring column strips).  Same operator, tiles, partial-slab contract and epilogue as conv3x3_wgrad_wino.hip
// (reference: the autograd backward of nn.Conv2d w.r.t. weight and bias behind loss.backward() at
// /root/reference/main.py:285 for the wide layers of models.py:51-56, 129-134); what changed is what a chunk costs
// besides its 64 MFMAs.  On gfx950 an f32 MFMA wave gets no free VALU slots (every other
// instruction adds its issue time to the 64 cycles per MFMA), and the first form spent ~165 VALU instructions per
// chunk, 56 of them v_mov_b32 that assembled (patch 0, patch 1) register pairs for its 8-byte LDS stores.  Here:
//
//   * the transforms are written around the pairs the LOADS deliver.  A thread's 16-byte row segment is the column
//     pairs A = (c1, c2), B = (c3, c4); the two outer columns E = (c0, c5) come from two 4-byte loads (cache hits:
//     the neighbouring lanes' segments).  B^T d B along y is three packed ops per output row, along x
//     (c1 + c2, c2 - c1) is ONE v_pk_add_f32 with operand selects per patch; A dY A^T likewise ((p + q, p - q) from
//     the pair (p, q)).  36 + 12 VALU instructions per chunk instead of ~110 + 56 moves;
//   * the operand image in LDS is [row i of the 4x4 position grid][patch t of the thread's pair][patch pair q]
//     [channel ^ 2q][4 positions of row i, in the order 1, 2, 0, 3], so that what a packed op produces -- two
//     positions of ONE patch -- is half of one 16-byte store, and one ds_read_b128 feeds four MFMAs on four DIFFERENT
//     accumulator tiles.  The xor puts the four pair-lanes of a channel on four different bank groups: a plain
//     [q][channel] image made every ds_write_b128 a four-way bank conflict (+450 ticks per chunk, wz_phases);
//   * nothing is transformed in place: the two input rows a chunk shares with the chunk below it stay in their
//     registers, the roles of the two register halves of a strip swap every other chunk (K loop unrolled by four),
//     no copies;
//   * the walk visits the two 64-byte halves of every 128-byte line back to back (strip 2p, strip 2p+1, next patch
//     row, ...) instead of a whole column of one half first: the first form fetched every line about three times
//     (PMC: 849 MB per 128 -> 128 launch against 268 MB algorithmic);
//   * chunk positions, row bases and lane offsets advance incrementally on the scalar unit (no divisions in the loop).
//
// Takes what the first form's EDGE = 0 instantiation takes with an even number of 8-patch strips per row
// (H even, W % 32 == 0, cout % 64 == 0); everything else stays on conv3x3_wgrad_wino.hip.
#include "wgrad_args.h"
#include <type_traits>

#define WZ_SLAB 8192      // floats of one operand stage: [4 i][2 t][4 q][64 channels][4]

typedef unsigned u32x4z __attribute__((ext_vector_type(4)));

#ifdef WZ_TIMING
__device__ unsigned long long wz_ticks[4];
#define WZ_TICK(i) do { const unsigned long long t_ = clock64(); t_acc[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define WZ_TICK(i)
#endif
#ifndef WZ_ABL
#define WZ_ABL 0          // ablation builds (tools/exp): 1 no loads, 2 no loads / transforms / stores, 3 no LDS stores only,
                          // 4 no operand reads only, 5 loads and stores but no transform arithmetic
#endif

// packed fp32 adds with their operand selects spelled out (the compiler scalarises these forms or builds them from moves)
__device__ __forceinline__ f32x2 wz_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wz_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (p.x + p.y, p.x - p.y)
__device__ __forceinline__ f32x2 wz_sumdiff(f32x2 p) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}
// (p.x + p.y, p.y - p.x)
__device__ __forceinline__ f32x2 wz_sumrdiff(f32x2 p) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}
// a - b as ONE scalar instruction (left to the compiler, pairs of these become a packed op fed by four moves)
__device__ __forceinline__ float wz_sub1(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wz_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// GATED: 0 none, 1 gate tensor (act' from the saved output), 2 1-bit leaky-relu mask.  PADC: circular padding.
// EDGE: the last 8-patch strip of a patch row is ragged (W % 32 != 0, W % 4 == 0: 84 x 84, BASELINE configs[3]): a patch
// pair past the row end stages zeros as its output gradient (and reads the row's last real pair: finite data).
template <int GATED, int PADC, int EDGE>
__global__ void __launch_bounds__(256, 1) conv3x3_wgrad_winoz_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const DYL = smem;                     // 2 stages of the DY image
    float* const VL = smem + 2 * WZ_SLAB;        // 2 stages of the V image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int q = tid & 3, ch = tid >> 2;        // this thread's patch pair of a chunk and channel of the group
    const int W = a.W, H = a.H, HW = H * W;
    const int TCW = W >> 1, TRH = H >> 1;        // (H even, W % 4 == 0: host check)
    const int nchr = (TCW + 7) >> 3;             // 8-patch strips per patch row (even: host check)

    int split, grp;
    scm_xcd_split_group(a.spi, split, grp);
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int tr0 = si * a.rows_per_split;       // (rows_per_split counts PATCH rows)
    int tr1 = tr0 + a.rows_per_split;
    if (tr1 > TRH) tr1 = TRH;
    const int ntr = tr1 > tr0 ? tr1 - tr0 : 0;
    const int nchunk = ntr * nchr;
    const int cog = grp / a.ci_groups, cig = grp - cog * a.ci_groups;
    const int cin = a.c1 + a.c2;
    const int co = cog * 64 + ch, ci = cig * 64 + ch;
    const int cic = ci < cin ? ci : cin - 1;     // (cout % 64 == 0; a ragged last ci group loads a valid plane, never stored)

    typedef const __attribute__((address_space(1))) char* gcp;
    typedef const __attribute__((address_space(1))) f32x4* gfp4;
    typedef const __attribute__((address_space(1))) float* gfp;
    typedef const __attribute__((address_space(1))) u32x4z* gup4;
    const size_t plane_b = (size_t)HW * 4;
    const unsigned row_b = (unsigned)W * 4;
    const bool in1 = cig * 64 < a.c1;
    gcp xb = in1 ? (gcp)(a.x1) + (size_t)b * a.c1 * plane_b : (gcp)(a.x2) + ((size_t)b * a.c2 - a.c1) * plane_b;
    gcp db = (gcp)(a.dy) + (size_t)b * a.cout * plane_b;
    gcp gb = GATED == 1 ? (gcp)(a.gate) + (size_t)b * a.cout * plane_b : nullptr;
    const int mask_nw = (a.cout + 31) >> 5;
    gcp mb = GATED == 2 ? (gcp)(a.gate_mask) + (size_t)b * mask_nw * plane_b : nullptr;
    const unsigned xch = (unsigned)cic * (unsigned)plane_b, dch = (unsigned)co * (unsigned)plane_b;
    const unsigned mch = (unsigned)(co >> 5) * (unsigned)plane_b;
    const unsigned mbit = 1u << (co & 31);

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

#define IC(n) std::integral_constant<int, (n)>{}
#define SB() __builtin_amdgcn_sched_barrier(0)

    // ---- registers that live across the K loop.  Strip S = k & 1 owns two register halves of two rows each; chunk k
    // (role R = (k >> 1) & 1) keeps its input rows 0, 1 in half R and rows 2, 3 in half R ^ 1.  The chunk below it in
    // the same strip (k + 2, role R ^ 1) finds its rows 0, 1 = this chunk's rows 2, 3 where they are and loads its
    // own rows 2, 3 into half R.
    f32x4 xm[2][2][2];       // [strip][half][row of the half]: columns c1 .. c4 (A = .xy, B = .zw)
    f32x2 xe[2][2][2];       // (c0, c5)
    f32x4 dr[2][2];          // [set = strip][row]: output-gradient rows 2*ty, 2*ty + 1, columns ix0 .. ix0 + 3
    f32x4 gr[2][GATED == 1 ? 2 : 1];
    u32x4z mr[2][GATED == 2 ? 2 : 1];
    unsigned o_xm[2], o_x0[2], o_x5[2], o_dm[2], o_mm[2];     // per-strip lane offsets (they change with the strip pair only)
    f32x2 emask[2];                                            // zero padding: (c0 exists, c5 exists)
    f32x2 dmask[2];                                            // EDGE: this thread's patch pair of the strip exists (1, 1) or not (0, 0)
    int c_ty[2] = {0, 0};                                      // patch row of the chunk held in each strip's registers
    bool c_fresh[2] = {true, true};                            // ... all four rows were loaded (top of a column)
    bool c_sum[2] = {false, false};                            // ... its output gradient enters the channel sums
    const bool want_sum = a.csum_part != nullptr && cig == 0;
    f32x2 csum2 = {0.f, 0.f};
    f32x4 aq[2], bq[2];

    // the load stream's position: chunk kl = strip pair pl, position kkl inside it (2 * ntr chunks per pair)
    int kl = 0, pl = 0, kkl = 0;
    auto strip_offsets = [&](auto s_, int cx) {
        constexpr int S = decltype(s_)::value;
        int ix0 = 16 * cx + 4 * q;                       // first image column of the pair's outputs
        if constexpr (EDGE) {
            const bool real = ix0 < W;                   // (W % 4 == 0: a pair exists as a whole or not at all)
            dmask[S][0] = real ? 1.f : 0.f; dmask[S][1] = dmask[S][0];
            ix0 = real ? ix0 : W - 4;
        }
        int i0 = ix0 - 1, i5 = ix0 + 4;
        const bool in0 = i0 >= 0, in5 = i5 < W;
        i0 = in0 ? i0 : (PADC ? W - 1 : 0);
        i5 = in5 ? i5 : (PADC ? 0 : W - 1);
        if constexpr (!PADC) { emask[S][0] = in0 ? 1.f : 0.f; emask[S][1] = in5 ? 1.f : 0.f; }
        o_xm[S] = xch + ix0 * 4; o_x0[S] = xch + i0 * 4; o_x5[S] = xch + i5 * 4;
        o_dm[S] = dch + ix0 * 4; o_mm[S] = mch + ix0 * 4;
    };
    // One input row (byte offset `ro` into the channel planes) into (strip S, half HF, row RW).
    auto ld_xrow = [&](auto s_, auto h_, auto r_, unsigned ro) {
        constexpr int S = decltype(s_)::value, HF = decltype(h_)::value, RW = decltype(r_)::value;
        gcp row = xb + ro;
        xm[S][HF][RW] = *(gfp4)(row + o_xm[S]);
        xe[S][HF][RW][0] = *(gfp)(row + o_x0[S]);
        xe[S][HF][RW][1] = *(gfp)(row + o_x5[S]);
    };
    auto ld_drow = [&](auto s_, auto r_, unsigned ro) {
        constexpr int S = decltype(s_)::value, RW = decltype(r_)::value;
        dr[S][RW] = *(gfp4)(db + ro + o_dm[S]);
        if constexpr (GATED == 1) gr[S][RW] = *(gfp4)(gb + ro + o_dm[S]);
        if constexpr (GATED == 2) mr[S][RW] = *(gup4)(mb + ro + o_mm[S]);
    };
    // Requests of the NEXT chunk of the load stream into strip S (= kl & 1), whose role is R2: rows 2, 3 into half
    // R2 ^ 1, rows 0, 1 into half R2 when the chunk starts a column.  Dealt out in four parts (PART) behind MFMAs.
    // A chunk past the end re-reads the last one (finite data; its output gradient is zeroed).
    int l_ty = 0;
    unsigned l_ro = 0;              // byte offset of image row 2 * l_ty (output-gradient row 0 of the chunk)
    bool l_fresh = true;
    auto ld_begin = [&](auto s_) {
        constexpr int S = decltype(s_)::value;
        const bool real = kl < nchunk;
        if (real) {
            l_ty = tr0 + (kkl >> 1);
            l_ro = (unsigned)(2 * l_ty) * row_b;
            l_fresh = kkl < 2;
            if (l_fresh) strip_offsets(s_, 2 * pl + S);
            ++kkl;
            if (kkl == 2 * ntr) { kkl = 0; ++pl; }
        } else {
            l_fresh = true;              // (roles keep alternating: reload everything, from the last position)
        }
        c_sum[S] = real && want_sum;     // (the chunk transformed past the end repeats the last one)
        c_ty[S] = l_ty;
        c_fresh[S] = l_fresh;
        ++kl;
        // (keeps the zero-extension of the offsets next to the loads: scalar-base addressing without a copy per load)
        unsigned &l0 = o_xm[S], &l1 = o_x0[S], &l2 = o_x5[S], &l3 = o_dm[S], &l4 = o_mm[S];
        asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4));
    };
    auto ld_part = [&](auto s_, auto r2_, auto part_) {
        constexpr int S = decltype(s_)::value, R2 = decltype(r2_)::value, PART = decltype(part_)::value;
        // image rows of the 4x4 patches: 2 ty - 1 .. 2 ty + 2 (wrapped / clamped at the image border; a clamped row
        // is zeroed in fix_chunk); output-gradient rows 2 ty, 2 ty + 1
        if constexpr (PART == 0) { ld_begin(s_); ld_xrow(s_, IC(R2 ^ 1), IC(0), l_ro + row_b); }
        if constexpr (PART == 1) {
            const unsigned r3 = l_ty == TRH - 1 ? (PADC ? 0u : l_ro + row_b) : l_ro + 2 * row_b;
            ld_xrow(s_, IC(R2 ^ 1), IC(1), r3);
        }
        if constexpr (PART == 2) { ld_drow(s_, IC(0), l_ro); ld_drow(s_, IC(1), l_ro + row_b); }
        if constexpr (PART == 3) {
            if (l_fresh) {
                const unsigned r0 = l_ty == 0 ? (PADC ? (unsigned)(H - 1) * row_b : 0u) : l_ro - row_b;
                ld_xrow(s_, IC(R2), IC(0), r0);
                ld_xrow(s_, IC(R2), IC(1), l_ro);
            }
        }
    };

    // ---- transform of the chunk held in strip S (role R) into LDS stage S
    // after its loads have landed: zero padding (rows outside the image, the two outer columns), validity, gate
    auto fix_chunk = [&](auto s_, auto r_) {
        constexpr int S = decltype(s_)::value, R = decltype(r_)::value;
        if constexpr (!PADC) {
            const f32x2 em = emask[S];
            xe[S][R ^ 1][0] = wz_mul(xe[S][R ^ 1][0], em);
            xe[S][R ^ 1][1] = wz_mul(xe[S][R ^ 1][1], em);
            if (c_fresh[S]) {
                xe[S][R][0] = wz_mul(xe[S][R][0], em);
                xe[S][R][1] = wz_mul(xe[S][R][1], em);
                if (c_ty[S] == 0) { xm[S][R][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; xe[S][R][0] = (f32x2){0.f, 0.f}; }     // image row -1
            }
            if (c_ty[S] == TRH - 1) { xm[S][R ^ 1][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; xe[S][R ^ 1][1] = (f32x2){0.f, 0.f}; }   // row H
        }
        if constexpr (EDGE) {
            const f32x2 dm = dmask[S];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x2 lo = wz_mul((f32x2){dr[S][i][0], dr[S][i][1]}, dm), hi = wz_mul((f32x2){dr[S][i][2], dr[S][i][3]}, dm);
                dr[S][i] = (f32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        }
        if constexpr (GATED == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dr[S][i][j] *= scm_act_grad_from_out(gr[S][i][j], a.gate_act);
        }
        if constexpr (GATED == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dr[S][i][j] = (mr[S][i][j] & mbit) ? dr[S][i][j] : 0.01f * dr[S][i][j];
        }
    };
    // Image offset of this thread's two patches.  Patch t of pair q is MFMA k index (k half = q >> 1, k step =
    // 2 * (q & 1) + t).  A ds_write_b128 is served in groups of 8 lanes = 2 channels x 4 pairs: channel ^ 2q spreads
    // them over all 32 banks; a ds_read_b128 in groups of 16 lanes of one q: 16 channels that differ in their low
    // four bits (MI355X_MICROARCH.md, LDS) -- the xor with a constant keeps them apart.
    const int wofs = q * 256 + (ch ^ (q << 1)) * 4;
    // (Two ds_write_b64, spelled out: merged into one ds_write_b128 -- what the compiler makes of two adjacent 8-byte
    // stores -- each store held the wave for ~28 cycles, +450 ticks per chunk (wz_phases -DWZ_ABL=5;
    // MI355X_MICROARCH.md: a 16-byte store of a single wave runs at half rate).)
    typedef __attribute__((address_space(3))) float* ldsp;
    const unsigned w_dy = (unsigned)(size_t)(ldsp)(DYL + wofs), w_v = (unsigned)(size_t)(ldsp)(VL + wofs);
    auto put = [&](unsigned base, auto ofs_, f32x2 p12, f32x2 p03) {
        constexpr int OFS = decltype(ofs_)::value;           // bytes: stage, grid row i, patch t
        if constexpr (WZ_ABL != 3) {
            asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"
                         :: "v"(base), "v"(p12), "v"(p03), "i"(OFS), "i"(OFS + 8) : "memory");
        } else {
            asm volatile("" :: "v"(p12), "v"(p03));
        }
    };
#define WZ_OFS(S_, I_, T_) IC(((S_) * WZ_SLAB + (I_) * 2048 + (T_) * 1024) * 4)
    // A dY A^T without the signs of its fourth row / column (put back in the epilogue).  Rows of the position grid:
    // 0: dy row 0, 1: row 0 + row 1, 2: row 0 - row 1, 3: row 1; along x (p, p + q, p - q, q) of the patch's (p, q).
    f32x2 dsum[2], ddif[2];
    auto dy_vert = [&](auto s_) {
        constexpr int S = decltype(s_)::value;
        const f32x2 r0a = {dr[S][0][0], dr[S][0][1]}, r0b = {dr[S][0][2], dr[S][0][3]};
        const f32x2 r1a = {dr[S][1][0], dr[S][1][1]}, r1b = {dr[S][1][2], dr[S][1][3]};
        if constexpr (WZ_ABL == 5) { dsum[0] = r0a; dsum[1] = r0b; ddif[0] = r1a; ddif[1] = r1b; return; }
        dsum[0] = wz_add(r0a, r1a); dsum[1] = wz_add(r0b, r1b);
        ddif[0] = wz_sub(r0a, r1a); ddif[1] = wz_sub(r0b, r1b);
        if (c_sum[S]) {
            csum2 = wz_add(csum2, dsum[0]);
            csum2 = wz_add(csum2, dsum[1]);
        }
    };
    // What a slice computes is stored by a LATER slice (pend[]: positions (1, 2) and (0, 3) of the pair's two patches):
    // an LDS store that reads the result of the VALU instruction in front of it holds the wave until that result is
    // written back -- with the stores right behind their arithmetic the loop took 5 379 ticks per chunk instead of 4 7xx.
    f32x2 pend[4];
    auto pend_store = [&](unsigned base, auto ofs0_, auto ofs1_) {
        put(base, ofs0_, pend[0], pend[1]);
        put(base, ofs1_, pend[2], pend[3]);
    };
    auto dy_calc = [&](auto s_, auto i_) {
        constexpr int S = decltype(s_)::value, I = decltype(i_)::value;
        f32x2 pq0, pq1;
        if constexpr (I == 0) { pq0 = (f32x2){dr[S][0][0], dr[S][0][1]}; pq1 = (f32x2){dr[S][0][2], dr[S][0][3]}; }
        else if constexpr (I == 1) { pq0 = dsum[0]; pq1 = dsum[1]; }
        else if constexpr (I == 2) { pq0 = ddif[0]; pq1 = ddif[1]; }
        else { pq0 = (f32x2){dr[S][1][0], dr[S][1][1]}; pq1 = (f32x2){dr[S][1][2], dr[S][1][3]}; }
        pend[1] = pq0; pend[3] = pq1;
        if constexpr (WZ_ABL == 5) { pend[0] = pq0; pend[2] = pq1; }          // (ablation: the stores without the arithmetic)
        else { pend[0] = wz_sumdiff(pq0); pend[2] = wz_sumdiff(pq1); }
    };
    auto dy_store = [&](auto s_, auto i_) {
        constexpr int S = decltype(s_)::value, I = decltype(i_)::value;
        pend_store(w_dy, WZ_OFS(S, I, 0), WZ_OFS(S, I, 1));
    };
    // B^T d B: output row I along y from the four input rows (three packed ops), then along x for both patches
    auto x_calc = [&](auto s_, auto r_, auto i_) {
        constexpr int S = decltype(s_)::value, R = decltype(r_)::value, I = decltype(i_)::value;
        // rows of the 4x4 patch: 0, 1 in half R, 2, 3 in half R ^ 1
        const f32x4 m0 = xm[S][R][0], m1 = xm[S][R][1], m2 = xm[S][R ^ 1][0], m3 = xm[S][R ^ 1][1];
        const f32x2 e0 = xe[S][R][0], e1 = xe[S][R][1], e2 = xe[S][R ^ 1][0], e3 = xe[S][R ^ 1][1];
        auto lo = [](f32x4 v) { return (f32x2){v[0], v[1]}; };
        auto hi = [](f32x4 v) { return (f32x2){v[2], v[3]}; };
        f32x2 A, B, E;
        if constexpr (WZ_ABL == 5) {
            pend[0] = lo(I & 1 ? m1 : m0); pend[1] = I & 1 ? e1 : e0;
            pend[2] = hi(I & 1 ? m3 : m2); pend[3] = I & 1 ? e3 : e2;
            return;
        }
        if constexpr (I == 0) { A = wz_sub(lo(m0), lo(m2)); B = wz_sub(hi(m0), hi(m2)); E = wz_sub(e0, e2); }
        if constexpr (I == 1) { A = wz_add(lo(m1), lo(m2)); B = wz_add(hi(m1), hi(m2)); E = wz_add(e1, e2); }
        if constexpr (I == 2) { A = wz_sub(lo(m2), lo(m1)); B = wz_sub(hi(m2), hi(m1)); E = wz_sub(e2, e1); }
        if constexpr (I == 3) { A = wz_sub(lo(m1), lo(m3)); B = wz_sub(hi(m1), hi(m3)); E = wz_sub(e1, e3); }
        // columns: E = (c0, c5), A = (c1, c2), B = (c3, c4).  Patch 0 (c0 .. c3), patch 1 (c2 .. c5):
        //   positions 1, 2 = (k1 + k2, k2 - k1), positions 0, 3 = (k0 - k2, k1 - k3)
        f32x2 z0, z1;
        z0[0] = wz_sub1(E[0], A[1]); z0[1] = wz_sub1(A[0], B[0]);
        z1[0] = wz_sub1(A[1], B[1]); z1[1] = wz_sub1(B[0], E[1]);
        pend[0] = wz_sumrdiff(A); pend[1] = z0;
        pend[2] = wz_sumrdiff(B); pend[3] = z1;
    };
    auto x_store = [&](auto s_, auto i_) {
        constexpr int S = decltype(s_)::value, I = decltype(i_)::value;
        pend_store(w_v, WZ_OFS(S, I, 0), WZ_OFS(S, I, 1));
    };

    // ---- MFMA side: group G = 4 * i + s of the chunk in LDS stage S: row i of the position grid, k step s
    // k step s of k half lh is patch t = s & 1 of pair q = 2 * lh + (s >> 1)
    int rofs_a[2], rofs_b[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        const int qr = 2 * lh + sh;
        rofs_a[sh] = qr * 256 + ((wm * 32 + l31) ^ (qr << 1)) * 4;
        rofs_b[sh] = qr * 256 + ((wn * 32 + l31) ^ (qr << 1)) * 4;
    }
    auto ab_read = [&](auto s_, auto g_) {
        constexpr int S = decltype(s_)::value, G = decltype(g_)::value;
        if constexpr (WZ_ABL != 4) {
            aq[G & 1] = *reinterpret_cast<const f32x4*>(DYL + S * WZ_SLAB + (G >> 2) * 2048 + (G & 1) * 1024 + rofs_a[(G >> 1) & 1]);
            bq[G & 1] = *reinterpret_cast<const f32x4*>(VL + S * WZ_SLAB + (G >> 2) * 2048 + (G & 1) * 1024 + rofs_b[(G >> 1) & 1]);
        }
    };
    // One group of chunk k (strip / stage S, role R): 4 MFMAs on the four accumulator tiles of grid row i; behind
    // them a slice of the transform of chunk k+1 (strip S ^ 1, role R1) and of the requests of chunk k+2 (strip S,
    // role R ^ 1).
    auto group = [&](auto s_, auto r_, auto r1_, auto g_) {
        constexpr int S = decltype(s_)::value, R = decltype(r_)::value, R1 = decltype(r1_)::value, G = decltype(g_)::value;
        constexpr int I = G >> 2;
        constexpr auto SN = IC(S ^ 1);
        constexpr bool DO_L = WZ_ABL < 1 || WZ_ABL == 5, DO_T = WZ_ABL < 2 || WZ_ABL == 3 || WZ_ABL == 5;
        auto mf = [&](auto j_) {
            constexpr int J = decltype(j_)::value;            // image order of the positions of a row: 1, 2, 0, 3
            constexpr int F = 4 * I + (J == 0 ? 1 : (J == 1 ? 2 : (J == 2 ? 0 : 3)));
            acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[G & 1][J], bq[G & 1][J], acc[F], 0, 0, 0);
        };
        mf(IC(0));
        if constexpr (G < 15) ab_read(s_, IC(G < 15 ? G + 1 : 0));
        if constexpr (G < 4 && DO_L) ld_part(s_, IC(R ^ 1), IC(G < 4 ? G : 0));
        SB();
        mf(IC(1));
        if constexpr (DO_T) {
            if constexpr (G == 0) { fix_chunk(SN, r1_); dy_vert(SN); }
            if constexpr (G == 1) dy_calc(SN, IC(0));
            if constexpr (G == 2) { dy_store(SN, IC(0)); dy_calc(SN, IC(1)); }
            if constexpr (G == 3) { dy_store(SN, IC(1)); dy_calc(SN, IC(2)); }
            if constexpr (G == 4) { dy_store(SN, IC(2)); dy_calc(SN, IC(3)); }
            if constexpr (G == 5) { dy_store(SN, IC(3)); x_calc(SN, r1_, IC(0)); }
            if constexpr (G == 6) x_store(SN, IC(0));
            if constexpr (G == 7) x_calc(SN, r1_, IC(1));
            if constexpr (G == 8) x_store(SN, IC(1));
            if constexpr (G == 9) x_calc(SN, r1_, IC(2));
            if constexpr (G == 10) x_store(SN, IC(2));
            if constexpr (G == 11) x_calc(SN, r1_, IC(3));
            if constexpr (G == 12) x_store(SN, IC(3));
        }
        SB();
        mf(IC(2));
        SB();
        mf(IC(3));
        SB();
    };
    auto chunk_body = [&](auto s_, auto r_, auto r1_) {
        ab_read(s_, IC(0));
        SB();
        group(s_, r_, r1_, IC(0)); group(s_, r_, r1_, IC(1)); group(s_, r_, r1_, IC(2)); group(s_, r_, r1_, IC(3));
        group(s_, r_, r1_, IC(4)); group(s_, r_, r1_, IC(5)); group(s_, r_, r1_, IC(6)); group(s_, r_, r1_, IC(7));
        group(s_, r_, r1_, IC(8)); group(s_, r_, r1_, IC(9)); group(s_, r_, r1_, IC(10)); group(s_, r_, r1_, IC(11));
        group(s_, r_, r1_, IC(12)); group(s_, r_, r1_, IC(13)); group(s_, r_, r1_, IC(14)); group(s_, r_, r1_, IC(15));
        // (the image stores are inline asm: the compiler does not count them -- the explicit wait makes the barrier's
        // guarantee independent of where the last operand-read wait happens to sit)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();            // the other stage holds chunk k+1; every wave is done with this one
        SB();
    };

#ifdef WZ_TIMING
    unsigned long long t_acc[4] = {0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    strip_offsets(IC(0), 0);        // (valid lane offsets whatever the band holds)
    strip_offsets(IC(1), 1);
    // ---- prologue: chunk 0 (strip 0, role 0) requested and transformed into stage 0, chunk 1 (strip 1, role 0) requested
    ld_part(IC(0), IC(0), IC(0)); ld_part(IC(0), IC(0), IC(1)); ld_part(IC(0), IC(0), IC(2)); ld_part(IC(0), IC(0), IC(3));
    ld_part(IC(1), IC(0), IC(0)); ld_part(IC(1), IC(0), IC(1)); ld_part(IC(1), IC(0), IC(2)); ld_part(IC(1), IC(0), IC(3));
    fix_chunk(IC(0), IC(0));
    dy_vert(IC(0));
    dy_calc(IC(0), IC(0)); dy_store(IC(0), IC(0)); dy_calc(IC(0), IC(1)); dy_store(IC(0), IC(1));
    dy_calc(IC(0), IC(2)); dy_store(IC(0), IC(2)); dy_calc(IC(0), IC(3)); dy_store(IC(0), IC(3));
    x_calc(IC(0), IC(0), IC(0)); x_store(IC(0), IC(0)); x_calc(IC(0), IC(0), IC(1)); x_store(IC(0), IC(1));
    x_calc(IC(0), IC(0), IC(2)); x_store(IC(0), IC(2)); x_calc(IC(0), IC(0), IC(3)); x_store(IC(0), IC(3));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    WZ_TICK(0);
    // Chunk k: strip k & 1, role (k >> 1) & 1; the chunk it transforms, k + 1, has role ((k + 1) >> 1) & 1.  The
    // host sends bands of whole groups of four chunks only (nchunk % 4 == 0: scm_launch_wgrad_winoz): the accumulators
    // must not cross a conditional, and nothing has to be zeroed -- the two chunks requested (and the one transformed)
    // past the end are never multiplied.
    for (int k = 0; k < nchunk; k += 4) {
        chunk_body(IC(0), IC(0), IC(0));
        chunk_body(IC(1), IC(0), IC(1));
        chunk_body(IC(0), IC(1), IC(1));
        chunk_body(IC(1), IC(1), IC(0));
    }
#undef WZ_OFS
#undef IC
#undef SB
    WZ_TICK(1);

    // ---- epilogue (as conv3x3_wgrad_wino.hip): signs of A's fourth row / column, dg = G^T dU G, 3x3 partials of this split
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
        float s = csum2[0] + csum2[1];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q == 0) a.csum_part[(size_t)split * a.cout + co] = s;
    }
#ifdef WZ_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    WZ_TICK(2);
    if (split == WZ_TIMING && grp == 0 && tid == 0)
        for (int i = 0; i < 4; ++i) wz_ticks[i] = t_acc[i];
#endif
}

// ------------------------------------------------------------------ host
template <int GATED, int PADC, int EDGE>
static int launch_wz_e(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)4 * WZ_SLAB * sizeof(float);
    auto kern = conv3x3_wgrad_winoz_kernel<GATED, PADC, EDGE>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_winoz_kernel");
    return SCM_OK;
}

template <int GATED, int PADC>
static int launch_wz(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    return (a.W & 31) ? launch_wz_e<GATED, PADC, 1>(a, grid, s) : launch_wz_e<GATED, PADC, 0>(a, grid, s);
}

// What this form takes of what scm_wgrad_wino_serves() accepts: even heights, an even number of 8-patch strips per patch
// row (the last one may be ragged: W % 4 == 0), whole 64-channel output groups.
bool scm_wgrad_winoz_serves(const WgradKArgs& a) {
    static const int off = getenv("SCM_NO_WGRAD_WINOZ") ? atoi(getenv("SCM_NO_WGRAD_WINOZ")) : 0;
    const int nchr = (a.W / 2 + 7) / 8;
    return !off && (a.H & 1) == 0 && (a.W & 3) == 0 && a.W >= 32 && (nchr & 1) == 0 && (a.cout & 63) == 0;
}

// `a` arrives with the split of scm_launch_wgrad_wino (spi, rows_per_split, ci_groups) filled in.  Returns
// SCM_EUNSUPPORTED (nothing launched, no error text) when a band would not be whole groups of four chunks.
int scm_launch_wgrad_winoz(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    const int trh = a.H / 2, nchr = (a.W / 2 + 7) / 8;
    const int last = trh - (a.spi - 1) * a.rows_per_split;          // patch rows of an image's last band
    if ((a.rows_per_split * nchr) % 4 != 0 || (last * nchr) % 4 != 0 || last < 1) return SCM_EUNSUPPORTED;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int g = a.gate_act == 0 ? 0 : ((a.gate_mask && a.gate_act == SCM_ACT_LRELU) ? 2 : 1);
#define WZ_P(G_) (circ ? launch_wz<G_, 1>(a, grid, s) : launch_wz<G_, 0>(a, grid, s))
    return g == 2 ? WZ_P(2) : (g == 1 ? WZ_P(1) : WZ_P(0));
#undef WZ_P
}
