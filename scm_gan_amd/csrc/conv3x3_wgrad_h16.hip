// Weight gradient of the 3x3 same-size convolution on the 16-bit MFMA
// (opt-in variant of conv3x3_wgrad.hip; same work split, same fp32 partial slabs
// part[split][tap][co][ci], same fixed-order reduction, same per-sample channel sums).
//
//   dWc[co][ci][tap] = sum_{b,p} dyp[b,co,p] * Xpad[b,ci,p+tap],  dyp = dy * act'(y)
//
// GEMM view: M = co, N = ci (one 32x32 tile per tap), K = pixels, on
// v_mfma_f32_32x32x16_bf16: a lane's fragment is 8 CONSECUTIVE k = 8 consecutive
// pixels of one image row for one channel (16 bytes).  NCHW rows give exactly that
// for the centre tap; the kx = 0 / 2 taps need the row shifted by one pixel, which
// would be a misaligned 16-byte LDS read (replayed at 64 cycles).  So the staging
// pass, which holds pixels p0-1 .. p0+8 of its channel row in registers anyway,
// writes THREE images of every x row (shifted by -1, 0, +1 pixel): every B fragment
// of every tap is then one aligned ds_read_b128, and circular / zero padding is
// resolved once, while staging.
//
// Workgroup = 8 waves (two per SIMD), wave (wm, wn) owns co tile wm of 4 and ci tile
// wn of 2 with its 9 tap accumulators (144 registers); it walks a band of rows of one
// image: x rows in 4 rolling LDS slots (one new row per step), the dyp row double
// buffered, one barrier per row.  The kernel is HBM-bound (x + dy + gate are read
// once per ci group), the MFMAs take about a third of its time.
#include <type_traits>
#include "wgrad_args.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));

template <int PREC>
__device__ __forceinline__ unsigned wpack2(float lo, float hi) {
    if constexpr (PREC == SCM_PREC_F16) {
        h16x2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(unsigned, v);
    } else {
        b16x2 v = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
}

template <int PREC>
__device__ __forceinline__ f32x16 wmfma(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (PREC == SCM_PREC_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
}

#define WG_NG 8                    // 8-pixel groups per row segment (PS <= 64)
#define WG_DP 129                  // co pitch of the dyp image (units): 8 lanes x stride 129 units hit 8 different bank quads
#define WG_DSTAGE (WG_NG * WG_DP)  // units per dyp stage: [g][co]

// WN = ci tiles (of 32) per workgroup: 2 -> 8 waves, 128 co x 64 ci; 1 -> 4 waves, 128 co x 32 ci
// (layers with cin <= 32).  VEC: W % 4 == 0 and PS % 8 == 0 -> every 8-pixel group is two aligned
// float4 loads.
template <int PREC, bool VEC, int WN>
__global__ void __launch_bounds__(256 * WN, 1) conv3x3_wgrad_h16_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 wsm[];
    constexpr int WG_XP = WN * 32 + 1;             // ci pitch of an x image
    constexpr int WG_XSLOT = 3 * WG_NG * WG_XP;    // units per x row slot: [kx][g][ci]
    constexpr int DT = 4 / WN;                     // dyp tasks per thread (128 co x 8 groups over 256*WN threads)
    constexpr int DSTEP = 32 * WN;                 // co stride between a thread's dyp tasks
    u32x4* xs = wsm;                         // [4 slots][3][WG_NG][WG_XP]
    u32x4* dsm = wsm + 4 * WG_XSLOT;         // [2][WG_NG][WG_DP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.W, H = a.H, HW = H * W;
    const int PS = a.PS;

    const int split = blockIdx.x;
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int y_begin = si * a.rows_per_split;
    int y_end = y_begin + a.rows_per_split;
    if (y_end > H) y_end = H;
    const int cig = blockIdx.y % a.ci_groups;
    const int cog = blockIdx.y / a.ci_groups;
    const int co_base = cog * 128, ci_base = cig * (WN * 32);
    const int cin = a.c1 + a.c2;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int gate_act = a.gate_act;
    const bool gated = gate_act != 0;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- staging roles
    // x: one (channel, group) task per thread: channel xc = tid >> 3, group xg = tid & 7
    const int xc = tid >> 3, xg = tid & 7;
    const int xci = ci_base + xc;
    const bool xvalid = xci < cin;
    const int xcc = xvalid ? xci : 0;
    const float* xsrc = xcc < a.c1 ? a.x1 + ((size_t)b * a.c1 + xcc) * HW : a.x2 + ((size_t)b * a.c2 + (xcc - a.c1)) * HW;
    // x_half: the activations are IEEE-half tensors (scm_conv3x3_desc.y_half): same element offsets, two bytes each
    const bool xh = a.x_half != 0;
    const _Float16* xsrch = xcc < a.c1 ? reinterpret_cast<const _Float16*>(a.x1) + ((size_t)b * a.c1 + xcc) * HW
                                       : reinterpret_cast<const _Float16*>(a.x2) + ((size_t)b * a.c2 + (xcc - a.c1)) * HW;
    // dyp: DT tasks per thread: channel dc = (tid >> 3) + DSTEP j, group dg = tid & 7
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
        // 1-bit gate: the mask words of this channel's 32-group travel through the same float
        // registers (loads do not touch the bits) and are decoded when the row is staged
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

    float px[10], pe = 0.f;       // pixels p0-1 .. p0+8 of the x row; the pixel one past the segment
    float pd[DT][8], pgv[DT][8];

    for (int seg = 0; seg < a.nseg; ++seg) {
        const int x0 = seg * PS;
        int ps = W - x0;
        if (ps > PS) ps = PS;
        const int nks = (ps + 15) >> 4;          // 16-pixel K steps of a row segment

        // 8 consecutive pixels starting at image column c0 of row pointer `row`:
        // unconditional loads on clamped addresses; masks are applied by the callers.
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
        const int xc0 = x0 + 8 * xg;             // first image column of this thread's x group
        const int dc0 = x0 + 8 * dg;

        // the same eight pixels of a half tensor: one 16-byte load (VEC) or eight 2-byte ones
        auto load8h = [&](const _Float16* row, int c0, float* v) {
            if constexpr (VEC) {
                const int cc = c0 + 7 < W ? c0 : W - 8;
                const h16x8 t = *reinterpret_cast<const h16x8*>(row + cc);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = c0 + i;
                    v[i] = (float)row[c < W ? c : W - 1];
                }
            }
        };
        auto fetch_x = [&](int r) {              // image row r (may be -1 or H)
            int ri = r;
            if (circ) ri = r < 0 ? r + H : (r >= H ? r - H : r);
            ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
            if (xh) {
                const _Float16* row = xsrch + (size_t)ri * W;
                load8h(row, xc0, px + 1);
                int cl = xc0 - 1, cr = xc0 + 8;
                if (circ) {
                    cl = cl < 0 ? cl + W : cl;
                    cr = cr >= W ? cr - W : cr;
                }
                cl = cl < 0 ? 0 : (cl >= W ? W - 1 : cl);
                cr = cr >= W ? W - 1 : cr;
                px[0] = (float)row[cl];
                px[9] = (float)row[cr];
                if constexpr (!VEC) {
                    int ce = x0 + ps;
                    if (circ) ce = ce >= W ? ce - W : ce;
                    pe = (float)row[ce < W ? ce : W - 1];
                }
                return;
            }
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
        // three per-thread bit masks (whole group, left halo pixel, right halo pixel) and one wave-uniform row mask:
        // no per-pixel branches or selects.
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
                    const int c = xc0 - 1 + i;        // image column (before wrapping)
                    const bool ok = rok && (circ || (c >= 0 && c < W)) && (c < x0 + ps + 1);
                    v[i] = ok ? (c == x0 + ps ? pe : px[i]) : 0.f;
                }
            }
            u32x4* slot = xs + ((r + 1) & 3) * WG_XSLOT + xg * WG_XP + xc;
            const unsigned m12 = wpack2<PREC>(v[1], v[2]), m34 = wpack2<PREC>(v[3], v[4]);
            const unsigned m56 = wpack2<PREC>(v[5], v[6]);
            u32x4 u;
            u[0] = wpack2<PREC>(v[0], v[1]); u[1] = wpack2<PREC>(v[2], v[3]);
            u[2] = wpack2<PREC>(v[4], v[5]); u[3] = wpack2<PREC>(v[6], v[7]);
            slot[0 * WG_NG * WG_XP] = u;                         // kx = 0: pixels p-1
            u[0] = m12; u[1] = m34; u[2] = m56; u[3] = wpack2<PREC>(v[7], v[8]);
            slot[1 * WG_NG * WG_XP] = u;                         // kx = 1: pixels p
            u[0] = wpack2<PREC>(v[2], v[3]); u[1] = wpack2<PREC>(v[4], v[5]);
            u[2] = wpack2<PREC>(v[6], v[7]); u[3] = wpack2<PREC>(v[8], v[9]);
            slot[2 * WG_NG * WG_XP] = u;                         // kx = 2: pixels p+1
        };
        auto fetch_d = [&](int y) {
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                load8(dsrc[j] + (size_t)y * W, dc0, pd[j]);
                if (gated) load8(gsrc[j] + (size_t)y * W, dc0, pgv[j]);
            }
        };
        unsigned dm[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) dm[j] = (dvalid[j] && 8 * dg < ps) ? ~0u : 0u;
        // MODE 0: dyp = dy; 1: dy * act'(gate tensor); 2: dy * leaky-relu slope from the 1-bit gate mask.  One branch per
        // row on the (wave-uniform) mode, none per element.
        auto store_d_mode = [&](int y, auto modec) {
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
                u32x4 u;
                u[0] = wpack2<PREC>(v[0], v[1]); u[1] = wpack2<PREC>(v[2], v[3]);
                u[2] = wpack2<PREC>(v[4], v[5]); u[3] = wpack2<PREC>(v[6], v[7]);
                dsm[(y & 1) * WG_DSTAGE + dg * WG_DP + (tid >> 3) + DSTEP * j] = u;
            }
        };
        auto store_d = [&](int y) {
            if (use_mask) store_d_mode(y, std::integral_constant<int, 2>{});
            else if (gated) store_d_mode(y, std::integral_constant<int, 1>{});
            else store_d_mode(y, std::integral_constant<int, 0>{});
        };

        // prologue: x rows y_begin-1 and y_begin
        // (issuing the loads two row steps ahead from a second register set was measured: no gain
        // with 4 waves, and with 8 waves the second set spills.  What bounds the kernel is that a
        // row step reads one 256-byte image row from each of ~300 channel planes.)
        __syncthreads();                         // the previous segment's MFMAs are done with the slots
        fetch_x(y_begin - 1);
        store_x(y_begin - 1);
        fetch_x(y_begin);
        store_x(y_begin);
        fetch_x(y_begin + 1);
        fetch_d(y_begin);

        for (int y = y_begin; y < y_end; ++y) {
            store_x(y + 1);
            store_d(y);
            __syncthreads();
            // the next row's loads, in flight under this row's MFMAs; after the last row they re-read it (unconditional:
            // under a branch the compiler waits for the loads on the spot to merge the registers)
            const bool more = y + 1 < y_end;
            fetch_x(more ? y + 2 : y + 1);
            fetch_d(more ? y + 1 : y);
            const u32x4* dst = dsm + (y & 1) * WG_DSTAGE + wm * 32 + l31;
            for (int ks = 0; ks < nks; ++ks) {
                const int g = 2 * ks + lh;
                const u32x4 av = dst[g * WG_DP];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const u32x4* xr = xs + ((y + ky) & 3) * WG_XSLOT + g * WG_XP + wn * 32 + l31;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = wmfma<PREC>(av, xr[kx * WG_NG * WG_XP], acc[ky * 3 + kx]);
                }
            }
        }
    }

    // ---- partial slab part[split][tap][co][ci]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci_base + wn * 32 + l31;
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
}

template <int PREC, bool VEC, int WN>
static int launch_wgrad_h16(const WgradKArgs& a, const WgradPlan& p, hipStream_t s) {
    const size_t lds = ((size_t)4 * 3 * WG_NG * (WN * 32 + 1) + 2 * WG_DSTAGE) * 16;
    auto kern = conv3x3_wgrad_h16_kernel<PREC, VEC, WN>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nsplit, p.co_groups * p.ci_groups), dim3(256 * WN), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_h16_kernel");
    return SCM_OK;
}

int scm_launch_wgrad_h16(const WgradKArgs& a, const WgradPlan& p, int precision, hipStream_t s) {
    if (a.PS > 64 || a.W < 8) {
        scm_set_error("scm_conv3x3_wgrad_h16: row segment of %d pixels (W = %d) unsupported", a.PS, a.W);
        return SCM_EUNSUPPORTED;
    }
    const bool vec = (a.W % 4 == 0) && (a.PS % 8 == 0) && (a.W == a.nseg * a.PS) &&
                     ((reinterpret_cast<size_t>(a.x1) | reinterpret_cast<size_t>(a.x2) | reinterpret_cast<size_t>(a.dy) |
                       reinterpret_cast<size_t>(a.gate) | reinterpret_cast<size_t>(a.gate_mask)) % 16 == 0);
    const bool narrow = p.ci_groups == 1 && a.c1 + a.c2 <= 32;      // one ci tile: 4 waves
    if (precision == SCM_PREC_F16) {
        if (narrow) return vec ? launch_wgrad_h16<SCM_PREC_F16, true, 1>(a, p, s) : launch_wgrad_h16<SCM_PREC_F16, false, 1>(a, p, s);
        return vec ? launch_wgrad_h16<SCM_PREC_F16, true, 2>(a, p, s) : launch_wgrad_h16<SCM_PREC_F16, false, 2>(a, p, s);
    }
    if (narrow) return vec ? launch_wgrad_h16<SCM_PREC_BF16, true, 1>(a, p, s) : launch_wgrad_h16<SCM_PREC_BF16, false, 1>(a, p, s);
    return vec ? launch_wgrad_h16<SCM_PREC_BF16, true, 2>(a, p, s) : launch_wgrad_h16<SCM_PREC_BF16, false, 2>(a, p, s);
}
