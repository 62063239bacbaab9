// Weight gradient of the 3x3 same-size convolution in fp32 arithmetic on the bf16 MFMA pipe: the companion of
// conv3x3_x3.hip (which explains the exact three-way bf16 split and the six partial products per product).
//
//   dWc[co][ci][tap] = sum_{b,p} dyp[b,co,p] * Xpad[b,ci,p+tap],  dyp = dy * act'(y)
//
// Same work split, fp32 partial slabs part[split][tap][co][ci], fixed-order reduction and per-sample channel sums
// as conv3x3_wgrad.hip / conv3x3_wgrad_h16.hip.  GEMM view as in the 16-bit kernel: M = co, N = ci, K = 8
// consecutive pixels per lane fragment, one accumulator per tap.  Differences from the 16-bit kernel:
//   * both operands are staged as three LDS images (hi / mid / lo), split while staging;
//   * the x row is stored ONCE (plus a halo unit on each side), not as three pre-shifted images -- three components
//     of three shifted images would not fit the LDS; the kx = 0 / 2 fragments are built in registers from the unit and
//     its neighbours with five v_alignbit_b32 per (row, component);
//   * 128 co x 32 ci per workgroup, 4 waves (one per SIMD, the 9 tap accumulators in AGPRs), dy single-buffered;
//     its own row split (twice the groups of the 128 x 64 plan: half the splits, wgrad_impl) and the ci groups of one
//     split on one XCD (they read the same dy rows);
//   * whole-group row segments (VEC): validity as bit masks, the 16-pixel step loop unrolled with the next row's
//     loads dealt out over its steps, LDS operands read a step ahead.
#include "wgrad_args.h"
#include <type_traits>

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 b16x2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 wmfma_bf16(u32x4w a, u32x4w b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8w, a), __builtin_bit_cast(b16x8w, b), c, 0, 0, 0);
}
__device__ __forceinline__ void wsplit3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    b16x2w h = {(__bf16)x0, (__bf16)x1};
    const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
    b16x2w m = {(__bf16)r0, (__bf16)r1};
    const float s0 = r0 - (float)m[0], s1 = r1 - (float)m[1];
    b16x2w l = {(__bf16)s0, (__bf16)s1};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

#define X3W_NG 8                       // 8-pixel groups per row segment (PS <= 64)
#define X3W_XP 33                      // ci pitch of an x image (units)
#define X3W_XG (X3W_NG + 2)            // groups incl. one halo unit on each side
#define X3W_XCOMP (X3W_XG * X3W_XP)    // units per component of an x row slot
#define X3W_XSLOT (3 * X3W_XCOMP)
// The shifted fragments need the last dword of the previous unit and the first dword of the next one.  Read out of the
// 16-byte units those are stride-4 dword reads (4-way bank conflicts, 46 % of the kernel's LDS cycles); they are kept a
// second time in dense dword arrays instead: [slot][comp][first | last][group][ci pitch 33].
#define X3W_ECOMP (2 * X3W_XG * X3W_XP)        // dwords per component of a slot
#define X3W_ESLOT (3 * X3W_ECOMP)
#define X3W_DP 129                     // co pitch of the dyp image
#define X3W_DCOMP (X3W_NG * X3W_DP)

// Phase clock for tools/exp/x3w_phases.hip (compiled in only with -DX3W_TIMING): shader-clock ticks per phase of wave 0
#ifdef X3W_TIMING
__device__ unsigned long long x3w_ticks[8];
#define X3W_TICK(i)                                     \
    do {                                                \
        const unsigned long long t_ = clock64();        \
        t_acc[i] += t_ - t_last;                        \
        t_last = t_;                                    \
    } while (0)
#else
#define X3W_TICK(i)
#endif

// VEC: 16-byte loads and whole 8-pixel groups (see store_x); NKS: 16-pixel MFMA steps per row segment, known at
// compile time for the VEC kernels (the step loop is unrolled so that the next row's loads can be dealt out over the
// steps without conditional register merges), 0 = run-time
template <bool VEC, int NKS>
__global__ void __launch_bounds__(256, 1) conv3x3_wgrad_x3_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4w wsx[];
    u32x4w* xs = wsx;                          // [4 slots][3 comps][X3W_XG][X3W_XP]
    u32x4w* dsm = wsx + 4 * X3W_XSLOT;         // [3 comps][X3W_NG][X3W_DP]
    unsigned* xe = reinterpret_cast<unsigned*>(dsm + 3 * X3W_DCOMP);      // edge dwords, see X3W_ECOMP

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave;                       // co tile of this wave
    const int W = a.W, H = a.H, HW = H * W;
    const int PS = a.PS;

    // grid: x = (co group, ci group), y = split.  The ci groups of one split read the same dy rows and should share an
    // L2, i.e. an XCD: workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), so dispatch slot L runs
    // on XCD L % 8.  Consecutive slots OF ONE XCD (L, L + 8, ..) are therefore given the groups of one split
    // (PMC: 778 MB of HBM traffic per 128 -> 128 launch with the plain mapping, dy fetched once per XCD that needs it).
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = L >> 3;
        bx = slot % gridDim.x;
        by = (slot / gridDim.x) * 8 + (L & 7);
    }
    const int cig = bx % a.ci_groups;
    const int cog = bx / a.ci_groups;
    const int split = by;
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int y_begin = si * a.rows_per_split;
    int y_end = y_begin + a.rows_per_split;
    if (y_end > H) y_end = H;
    const int co_base = cog * 128, ci_base = cig * 32;
    const int cin = a.c1 + a.c2;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int gate_act = a.gate_act;
    const bool gated = gate_act != 0;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- staging roles (as the 4-wave form of the 16-bit kernel)
    const int xc = tid >> 3, xg = tid & 7;
    const int xci = ci_base + xc;
    const bool xvalid = xci < cin;
    const int xcc = xvalid ? xci : 0;
    const float* xsrc = xcc < a.c1 ? a.x1 + ((size_t)b * a.c1 + xcc) * HW : a.x2 + ((size_t)b * a.c2 + (xcc - a.c1)) * HW;
    constexpr int DT = 4, DSTEP = 32;
    const int dg = tid & 7;
    const float* dsrc[DT];
    const float* gsrc[DT];
    bool dvalid[DT];
    int dbit[DT];
    const bool use_mask = a.gate_mask != nullptr && gate_act == SCM_ACT_LRELU;
    const int mask_nw = (a.cout + 31) >> 5;
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        const int co = co_base + (tid >> 3) + DSTEP * j;
        dvalid[j] = co < a.cout;
        const size_t o = ((size_t)b * a.cout + (dvalid[j] ? co : 0)) * HW;
        dsrc[j] = a.dy + o;
        gsrc[j] = (gated ? a.gate : a.dy) + o;
        dbit[j] = co & 31;
        if (use_mask) {
            const int w = (dvalid[j] ? co : 0) >> 5;
            gsrc[j] = reinterpret_cast<const float*>(a.gate_mask + ((size_t)b * mask_nw + w) * HW);
        }
    }
    float csum[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) csum[j] = 0.f;
    const bool do_csum = a.csum_part != nullptr && cig == 0;

    float px[10], pe = 0.f;
    float pd[DT][8], pgv[DT][8];

#ifdef X3W_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = clock64();
#endif
    for (int seg = 0; seg < a.nseg; ++seg) {
        const int x0 = seg * PS;
        int ps = W - x0;
        if (ps > PS) ps = PS;
        const int nks = (ps + 15) >> 4;

        auto load8 = [&](const float* row, int c0, float* v) {
            if constexpr (VEC) {
                const int cc = c0 + 7 < W ? c0 : W - 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(row + cc);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(row + cc + 4);
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = c0 + i;
                    v[i] = row[c < W ? c : W - 1];
                }
            }
        };
        const int xc0 = x0 + 8 * xg;
        const int dc0 = x0 + 8 * dg;

        auto fetch_x = [&](int r) {
            int ri = r;
            if (circ) ri = r < 0 ? r + H : (r >= H ? r - H : r);
            ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
            const float* row = xsrc + (size_t)ri * W;
            load8(row, xc0, px + 1);
            int cl = xc0 - 1, cr = xc0 + 8;
            if (circ) {
                cl = cl < 0 ? cl + W : cl;
                cr = cr >= W ? cr - W : cr;
            }
            cl = cl < 0 ? 0 : (cl >= W ? W - 1 : cl);
            cr = cr >= W ? W - 1 : cr;
            px[0] = row[cl];
            px[9] = row[cr];
            // the pixel one past the segment (the kx = 2 tap of its last pixel): wherever it falls in this thread's
            // ten, it is the wrapped column under circular padding -- load8 clamps instead of wrapping
            if constexpr (!VEC) {
                int ce = x0 + ps;
                if (circ) ce = ce >= W ? ce - W : ce;
                pe = row[ce < W ? ce : W - 1];
            }
        };
        // VEC kernels (W == nseg * PS, PS % 8 == 0): an 8-pixel group lies wholly inside or wholly outside the segment,
        // and the pixel one past the segment is px[9] of its last group (fetch_x wraps / clamps it).  Validity is then
        // three per-thread bit masks (whole group, left halo pixel, right halo pixel) and one wave-uniform row mask.
        const unsigned xm_all = (xvalid && 8 * xg < ps) ? ~0u : 0u;
        const unsigned xm_l = (circ || xc0 > 0) ? xm_all : 0u;
        const unsigned xm_r = (circ || xc0 + 8 < W) ? xm_all : 0u;
        auto store_x = [&](int r) {
            float v[10];
            if constexpr (VEC) {
                const unsigned rowm = (circ || (r >= 0 && r < H)) ? ~0u : 0u;
                const unsigned ma = xm_all & rowm;
                v[0] = __uint_as_float(__float_as_uint(px[0]) & (xm_l & rowm));
#pragma unroll
                for (int i = 1; i < 9; ++i) v[i] = __uint_as_float(__float_as_uint(px[i]) & ma);
                v[9] = __uint_as_float(__float_as_uint(px[9]) & (xm_r & rowm));
            } else {
                bool rok = xvalid;
                if (!circ) rok = rok && r >= 0 && r < H;
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int c = xc0 - 1 + i;
                    const bool ok = rok && (circ || (c >= 0 && c < W)) && (c < x0 + ps + 1);
                    v[i] = ok ? (c == x0 + ps ? pe : px[i]) : 0.f;
                }
            }
            unsigned hh[5], mm[5], ll[5];           // pairs (v0,v1) (v2.. no: (v1,v2),(v3,v4),(v5,v6),(v7,v8) + halo pair (v0,v9)
            wsplit3_pair(v[1], v[2], hh[0], mm[0], ll[0]);
            wsplit3_pair(v[3], v[4], hh[1], mm[1], ll[1]);
            wsplit3_pair(v[5], v[6], hh[2], mm[2], ll[2]);
            wsplit3_pair(v[7], v[8], hh[3], mm[3], ll[3]);
            wsplit3_pair(v[9], v[0], hh[4], mm[4], ll[4]);      // low half = p0+8 (right halo), high half = p0-1 (left halo)
            u32x4w* slot = xs + ((r + 1) & 3) * X3W_XSLOT + (xg + 1) * X3W_XP + xc;
            unsigned* es = xe + ((r + 1) & 3) * X3W_ESLOT + (xg + 1) * X3W_XP + xc;
            u32x4w u;
            u[0] = hh[0]; u[1] = hh[1]; u[2] = hh[2]; u[3] = hh[3];
            slot[0] = u;
            // VEC: the group after the segment's last one is all zeros here and must not overwrite the first-dword
            // entries its neighbour (which holds pixel x0 + ps as its right halo) writes for it below
            const bool last_group = VEC ? 8 * xg + 8 == ps : xg == X3W_NG - 1;
            const bool keep_first = !(VEC && 8 * xg == ps);
            if (keep_first) { es[0] = hh[0]; es[X3W_ECOMP] = mm[0]; es[2 * X3W_ECOMP] = ll[0]; }
            es[X3W_XG * X3W_XP] = hh[3];
            u[0] = mm[0]; u[1] = mm[1]; u[2] = mm[2]; u[3] = mm[3];
            slot[X3W_XCOMP] = u;
            es[X3W_ECOMP + X3W_XG * X3W_XP] = mm[3];
            u[0] = ll[0]; u[1] = ll[1]; u[2] = ll[2]; u[3] = ll[3];
            slot[2 * X3W_XCOMP] = u;
            es[2 * X3W_ECOMP + X3W_XG * X3W_XP] = ll[3];
            if (xg == 0) {                              // left halo unit: its last dword = (.., pixel x0 - 1)
                unsigned* hs = es - X3W_XP + X3W_XG * X3W_XP;
                hs[0] = hh[4] & 0xffff0000u;
                hs[X3W_ECOMP] = mm[4] & 0xffff0000u;
                hs[2 * X3W_ECOMP] = ll[4] & 0xffff0000u;
            }
            if (last_group) {                           // the unit after the segment: its first dword = (pixel x0 + ps, ..)
                unsigned* hs = es + X3W_XP;
                hs[0] = hh[4] & 0xffffu;
                hs[X3W_ECOMP] = mm[4] & 0xffffu;
                hs[2 * X3W_ECOMP] = ll[4] & 0xffffu;
            }
        };
        auto fetch_d_part = [&](int y, int j) {
            load8(dsrc[j] + (size_t)y * W, dc0, pd[j]);
            if (gated) load8(gsrc[j] + (size_t)y * W, dc0, pgv[j]);
        };
        auto fetch_d = [&](int y) {
#pragma unroll
            for (int j = 0; j < DT; ++j) fetch_d_part(y, j);
        };
        unsigned dm[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) dm[j] = (dvalid[j] && 8 * dg < ps) ? ~0u : 0u;
        // MODE 0: dyp = dy; 1: dy * act'(gate tensor); 2: dy * leaky-relu slope from the 1-bit gate mask.  One branch per
        // row on the (wave-uniform) mode, none per element.
        auto store_d_mode = [&](auto modec) {
            constexpr int MODE = decltype(modec)::value;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                float v[8];
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float t = pd[j][i];
                    if constexpr (MODE == 2) t *= ((__float_as_uint(pgv[j][i]) >> dbit[j]) & 1u) ? 1.f : 0.01f;
                    if constexpr (MODE == 1) t *= scm_act_grad_from_out(pgv[j][i], gate_act);
                    if constexpr (VEC) v[i] = __uint_as_float(__float_as_uint(t) & dm[j]);
                    else v[i] = (dvalid[j] && dc0 + i < x0 + ps) ? t : 0.f;
                    s += v[i];
                }
                csum[j] += s;
                u32x4w uh, um, ul;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned hi, mid, lo;
                    wsplit3_pair(v[2 * q], v[2 * q + 1], hi, mid, lo);
                    uh[q] = hi; um[q] = mid; ul[q] = lo;
                }
                u32x4w* dd = dsm + dg * X3W_DP + (tid >> 3) + DSTEP * j;
                dd[0] = uh;
                dd[X3W_DCOMP] = um;
                dd[2 * X3W_DCOMP] = ul;
            }
        };
        auto store_d = [&]() {
            if (use_mask) store_d_mode(std::integral_constant<int, 2>{});
            else if (gated) store_d_mode(std::integral_constant<int, 1>{});
            else store_d_mode(std::integral_constant<int, 0>{});
        };

        __syncthreads();                         // the previous segment's MFMAs are done with the images
        fetch_x(y_begin - 1);
        store_x(y_begin - 1);
        fetch_x(y_begin);
        store_x(y_begin);
        fetch_x(y_begin + 1);
        fetch_d(y_begin);
        X3W_TICK(0);

        for (int y = y_begin; y < y_end; ++y) {
            store_x(y + 1);
            X3W_TICK(7);
            store_d();
            X3W_TICK(1);
            __syncthreads();
            X3W_TICK(2);
            // MFMA steps (ks, ky).  The LDS operands of a step are read while the previous step's 54 MFMAs run (one
            // wave per SIMD: nothing else covers the latency), and the next row's global loads go out a few per ks
            // instead of in one burst that stalls on the memory pipeline's queues (tools/exp/x3w_phases.hip).
            const bool more = y + 1 < y_end;
            const u32x4w* dst = dsm + wm * 32 + l31;
            struct XFrag { u32x4w u[3]; unsigned pw[3], nx[3]; };
            auto load_xf = [&](XFrag& f, int ks, int ky) {
                const int g = 2 * ks + lh;
                const u32x4w* xr = xs + ((y + ky) & 3) * X3W_XSLOT + (g + 1) * X3W_XP + l31;
                const unsigned* er = xe + ((y + ky) & 3) * X3W_ESLOT + (g + 1) * X3W_XP + l31;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f.u[c] = xr[c * X3W_XCOMP];
                    f.pw[c] = er[c * X3W_ECOMP + X3W_XG * X3W_XP - X3W_XP];     // previous unit, pixels 6..7
                    f.nx[c] = er[c * X3W_ECOMP + X3W_XP];                        // next unit, pixels 0..1
                }
            };
            auto load_av = [&](u32x4w* av, int ks) {
                const int g = 2 * ks + lh;
#pragma unroll
                for (int c = 0; c < 3; ++c) av[c] = dst[c * X3W_DCOMP + g * X3W_DP];
            };
            auto products = [&](const XFrag& f, const u32x4w* av, int ky) {
                u32x4w f0[3], f1[3], f2[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const u32x4w u = f.u[c];
                    const unsigned s01 = __builtin_amdgcn_alignbit(u[1], u[0], 16);
                    const unsigned s12 = __builtin_amdgcn_alignbit(u[2], u[1], 16);
                    const unsigned s23 = __builtin_amdgcn_alignbit(u[3], u[2], 16);
                    f1[c] = u;
                    f0[c][0] = __builtin_amdgcn_alignbit(u[0], f.pw[c], 16);        // pixels p-1 .. p+6
                    f0[c][1] = s01; f0[c][2] = s12; f0[c][3] = s23;
                    f2[c][0] = s01; f2[c][1] = s12; f2[c][2] = s23;                 // pixels p+1 .. p+8
                    f2[c][3] = __builtin_amdgcn_alignbit(f.nx[c], u[3], 16);
                }
                // fixed order per accumulator: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; consecutive MFMAs go to
                // different accumulators
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[ky * 3 + 0] = wmfma_bf16(av[PA[q]], f0[PB[q]], acc[ky * 3 + 0]);
                    acc[ky * 3 + 1] = wmfma_bf16(av[PA[q]], f1[PB[q]], acc[ky * 3 + 1]);
                    acc[ky * 3 + 2] = wmfma_bf16(av[PA[q]], f2[PB[q]], acc[ky * 3 + 2]);
                }
            };
            XFrag fa, fb, fc;
            u32x4w av[3], avn[3];
            if constexpr (NKS == 0) {
                if (more) {
                    fetch_x(y + 2);
                    fetch_d(y + 1);
                }
            }
            load_av(av, 0);
            load_xf(fa, 0, 0);
            auto step = [&](int ks, int nk) {
                load_xf(fb, ks, 1);
                if constexpr (NKS > 0) {
                    // this step's share of the next row's loads; after the last row the same rows are read again
                    // (unconditional loads: a branch here would make the compiler wait for them on the spot)
                    if (ks == 0) fetch_x(more ? y + 2 : y + 1);
#pragma unroll
                    for (int j = 0; j < DT; ++j)
                        if (j >= ks * DT / NKS && j < (ks + 1) * DT / NKS) fetch_d_part(more ? y + 1 : y, j);
                }
                products(fa, av, 0);
                load_xf(fc, ks, 2);
                products(fb, av, 1);
                const int kn = ks + 1 < nk ? ks + 1 : ks;      // (the last prefetch is a harmless re-read)
                load_av(avn, kn);
                load_xf(fa, kn, 0);
                products(fc, av, 2);
#pragma unroll
                for (int c = 0; c < 3; ++c) av[c] = avn[c];
            };
            if constexpr (NKS > 0) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) step(ks, NKS);
            } else {
                for (int ks = 0; ks < nks; ++ks) step(ks, nks);
            }
            X3W_TICK(4);
            __syncthreads();                     // dyp image (single stage) free for the next row
            X3W_TICK(5);
        }
    }

#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci_base + l31;
            if (co < a.cout && ci < cin) a.part[(((size_t)split * 9 + t) * a.cout + co) * cin + ci] = acc[t][r];
        }
    }
    if (do_csum) {
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            float s = csum[j];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            const int co = co_base + (tid >> 3) + DSTEP * j;
            if (dg == 0 && co < a.cout) a.csum_part[(size_t)split * a.cout + co] = s;
        }
    }
    X3W_TICK(6);
#ifdef X3W_TIMING
    if (bx == 1 && by == X3W_TIMING && tid == 0)
        for (int i = 0; i < 8; ++i) x3w_ticks[i] = t_acc[i];
#endif
}

int scm_launch_wgrad_x3(const WgradKArgs& k, const WgradPlan& p, hipStream_t s) {
    if (k.PS > 64 || k.W < 8) {
        scm_set_error("scm_conv3x3_wgrad_x3: row segment of %d pixels (W = %d) unsupported", k.PS, k.W);
        return SCM_EUNSUPPORTED;
    }
    WgradKArgs a = k;
    const int cin = a.c1 + a.c2;
    a.ci_groups = (cin + 31) / 32;
    const bool vec = (a.W % 4 == 0) && (a.PS % 8 == 0) && (a.W == a.nseg * a.PS) &&
                     ((reinterpret_cast<size_t>(a.x1) | reinterpret_cast<size_t>(a.x2) | reinterpret_cast<size_t>(a.dy) |
                       reinterpret_cast<size_t>(a.gate) | reinterpret_cast<size_t>(a.gate_mask)) % 16 == 0);
    const size_t lds = ((size_t)4 * X3W_XSLOT + 3 * X3W_DCOMP) * 16 + (size_t)4 * X3W_ESLOT * 4;
    const dim3 grid(p.co_groups * a.ci_groups, p.nsplit);
    static std::atomic<uint64_t> attr_done[5];
    auto launch = [&](auto kern, int slot) -> int {
        if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done[slot], 160 * 1024)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return SCM_OK;
    };
    int rc;
    if (!vec) rc = launch(conv3x3_wgrad_x3_kernel<false, 0>, 0);
    else if (a.PS == 64) rc = launch(conv3x3_wgrad_x3_kernel<true, 4>, 4);
    else if (a.PS == 48) rc = launch(conv3x3_wgrad_x3_kernel<true, 3>, 3);
    else if (a.PS == 32) rc = launch(conv3x3_wgrad_x3_kernel<true, 2>, 2);
    else if (a.PS == 16) rc = launch(conv3x3_wgrad_x3_kernel<true, 1>, 1);
    else rc = launch(conv3x3_wgrad_x3_kernel<false, 0>, 0);      // 8-, 24-, 40-, 56-pixel segments: the general kernel
    if (rc) return rc;
    SCM_LAUNCH_CHECK("conv3x3_wgrad_x3_kernel");
    return SCM_OK;
}
