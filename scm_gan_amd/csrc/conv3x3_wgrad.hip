// Weight gradient of the 3x3 stride-1 same-size convolution on the fp32 MFMA
// pipe.  Replaces the autograd backward of nn.Conv2d / nn.ConvTranspose2d
// w.r.t. weight and bias triggered by loss.backward() at main.py:285 for the
// layers of models.py:51-56, 129-134, 260-266.
//
//   dWc[co][ci][tap] = sum_{b,p} dyp[b,co,p] * Xpad[b,ci,p+tap],  dyp = dy * act'(y)
//
// GEMM view: M = co, N = ci (one 32x32 tile per tap), K = pixels.
//   MFMA A operand = dyp  (lane -> co, k -> pixel)
//   MFMA B operand = Xpad (lane -> ci, k -> pixel), the same halo rows read at
//   9 shifted addresses: the im2col is an LDS address offset.
// A workgroup (4 waves as MT x NT tiles of 32 co x 32 ci) owns a band of image
// rows of ONE image (a "split") and walks it row by row: the x halo rows live
// in a rolling 3-row LDS buffer (one new row per step), the dyp row tile is
// restaged each step.  Every wave keeps its 9 tap accumulators (144 VGPRs) in
// registers for the whole band and writes them once, to a per-split partial
// slab part[split][tap][co][ci]; a second kernel sums the slabs in a fixed
// order (bitwise reproducible, no float atomics).  Per-sample channel sums of
// dyp (bias gradient, action-plane gradients) fall out of the A operands.
#include "wgrad_args.h"
#include <type_traits>


// WM x WN waves (8 = two per SIMD for the big layers); wave (wm, wn) owns co tile
// wm and NTW ci tiles, i.e. 9*NTW accumulator tiles of 16 registers.  NTW = 1 keeps
// the 144 accumulators inside the AGPR half of the register file: at NTW = 2 (288)
// hipcc shuttles tiles between AGPRs and VGPRs inside the k loop.  Software
// pipeline per image row y of the band:
//   barrier | registers -> LDS (x row y+1 into the rolling slot, dyp row y) |
//   barrier | issue global loads of row y+1's data into registers | 288 MFMAs
// so the HBM/L2 latency of the next row hides under this row's MFMAs.
// MASK: the leaky-relu gate comes from the 1-bit mask (one word per pixel for the wave's DR
// channels) instead of DR gate floats per pixel.
template <int WM, int WN, int NTW, bool MASK>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN) / 4) conv3x3_wgrad_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    constexpr int NW = WM * WN;
    constexpr int MT = WM;                 // co tiles per workgroup
    constexpr int NT = WN * NTW;           // ci tiles per workgroup
    constexpr int DR = MT * 32 / NW;       // dy rows staged per wave
    constexpr int XR = NT * 32 / NW;       // x channels staged per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: keeps bases in SGPRs
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.W, H = a.H, HW = H * W;
    const int PS = a.PS;
    const int PSP = PS + 4;                 // row pitch of the x tile (cols -1 .. PS+2)
    const int XPL = (4 * PSP) | 1;          // 4 rolling row slots; odd plane pitch: conflict-free lane->ci reads
    const int DPL = (PS + 2) | 1;           // odd row pitch of the dy tile: lane->co reads
    float* xl = smem;                       // [NT*32][XPL]
    float* dyl = smem + NT * 32 * XPL;      // [2][MT*32][DPL]  (double-buffered)
    const int DYB = MT * 32 * DPL;

    const int split = blockIdx.x;
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int y_begin = si * a.rows_per_split;
    int y_end = y_begin + a.rows_per_split;
    if (y_end > H) y_end = H;
    const int cig = blockIdx.y % a.ci_groups;
    const int cog = blockIdx.y / a.ci_groups;
    const int co_base = cog * MT * 32, ci_base = cig * NT * 32;
    const int cin = a.c1 + a.c2;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int gate_act = a.gate_act;

    f32x16 acc[NTW][9];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][t][r] = 0.f;
    float csum = 0.f;
    const bool do_csum = a.csum_part != nullptr && cig == 0 && wn == 0;

    const float* ap = dyl + (wm * 32 + l31) * DPL + lh;
    const float* bp = xl + (wn * NTW * 32 + l31) * XPL + lh;
    bool tile_live[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
        tile_live[n] = (ci_base + (wn * NTW + n) * 32 < cin) && (co_base + wm * 32 < a.cout);
    bool all_live = true;
#pragma unroll
    for (int n = 0; n < NTW; ++n) all_live = all_live && tile_live[n];

    // Every global load below is UNCONDITIONAL on a clamped, always-valid address and
    // masked afterwards with a select: a load under a per-lane condition makes hipcc
    // branch around each load and serialise them (cdna guide, "three .s-level traps" (c)).
    // wave-uniform base of x channel c of this workgroup's ci range (x1 when beyond cin)
    // x_half: the activations are IEEE-half tensors (scm_conv3x3_desc.y_half); xbase() then returns the plane's address
    // counted in HALF elements from a.x1 / a.x2 reinterpreted (same element offsets, two bytes each): ld_x() reads it
    const bool xh = a.x_half != 0;
    auto xbase = [&](int c, bool& valid) -> const float* {
        const int ci = ci_base + c;
        valid = ci < cin;
        const int cc = valid ? ci : 0;
        const size_t o = cc < a.c1 ? ((size_t)b * a.c1 + cc) * HW : ((size_t)b * a.c2 + (cc - a.c1)) * HW;
        const float* base = cc < a.c1 ? a.x1 : a.x2;
        return xh ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(base) + o) : base + o;
    };
    auto ld_x = [&](const float* plane, int idx) -> float {
        return xh ? (float)reinterpret_cast<const _Float16*>(plane)[idx] : plane[idx];
    };
    // the one halo element of this thread: channel hc, side hs (0: left col, 1: right col)
    const int hc = tid >> 1, hs = tid & 1;
    bool hvalid = false;
    const float* hsrc = xbase(hc < NT * 32 ? hc : 0, hvalid);
    hvalid = hvalid && hc < NT * 32;

    float pdy[DR], pg[MASK ? 1 : DR], px[XR], phalo = 0.f;
    unsigned pmask = 0u;
    static_assert(!MASK || 32 % DR == 0, "a wave's dy channels must not straddle mask words");
    const int mask_nw = (a.cout + 31) >> 5;

    for (int seg = 0; seg < a.nseg; ++seg) {
        const int x0 = seg * PS;
        int ps = W - x0;
        if (ps > PS) ps = PS;
        const int pse = (ps + 1) & ~1;

        // image column of LDS column cc (cc = 0 is x0-1); ok = inside the image (or wrapped)
        auto img_col = [&](int cc, bool& ok) -> int {
            int xi = x0 + cc - 1;
            if (circ) {
                ok = true;
                xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);
            } else {
                ok = (xi >= 0 && xi < W);
            }
            return xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
        };
        bool ok_main, ok_halo;
        const int xi_main = img_col(lane + 1, ok_main);     // this lane's interior column
        ok_main = ok_main && (lane < pse + 1);
        const int hcc = hs ? pse + 1 : 0;
        const int xi_halo = img_col(hcc, ok_halo);
        ok_halo = ok_halo && hvalid;

        // fetch_* only ISSUE loads (raw values stay in registers, nothing consumes them
        // before the MFMA loop); store_* apply the zero-fill masks and write LDS.
        auto fetch_xrow = [&](int r) {      // image row r (may be -1 or H) -> px[], phalo
            int ri = r;
            if (circ) ri = r < 0 ? r + H : (r >= H ? r - H : r);
            ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
            const int ro = ri * W;
#pragma unroll
            for (int k = 0; k < XR; ++k) {
                bool cv;
                const float* src = xbase(wave * XR + k, cv);
                px[k] = ld_x(src, ro + xi_main);
            }
            phalo = ld_x(hsrc, ro + xi_halo);
        };
        auto store_xrow = [&](int r) {
            const int slot = (r + 4) & 3;
            const bool rvalid = circ || (r >= 0 && r < H);
            if (lane < pse + 1) {
#pragma unroll
                for (int k = 0; k < XR; ++k) {
                    const bool cv = ci_base + wave * XR + k < cin;
                    xl[(wave * XR + k) * XPL + slot * PSP + lane + 1] = (cv && rvalid && ok_main) ? px[k] : 0.f;
                }
            }
            if (hc < NT * 32) xl[hc * XPL + slot * PSP + hcc] = (rvalid && ok_halo) ? phalo : 0.f;
        };
        const int lane_c = lane < ps ? lane : ps - 1;
        auto fetch_dy = [&](int y) {
#pragma unroll
            for (int k = 0; k < DR; ++k) {
                const int co = co_base + wave * DR + k;
                const int coc = co < a.cout ? co : a.cout - 1;
                const size_t rowo = ((size_t)b * a.cout + coc) * HW + (size_t)y * W + x0;   // wave-uniform
                pdy[k] = a.dy[rowo + lane_c];
                if constexpr (!MASK) pg[k] = gate_act ? a.gate[rowo + lane_c] : 1.f;
            }
            if constexpr (MASK) {      // the wave's DR channels share one mask word per pixel (DR divides 32)
                const int w = (co_base + wave * DR) >> 5;
                pmask = a.gate_mask[((size_t)b * mask_nw + (w < mask_nw ? w : 0)) * HW + (size_t)y * W + x0 + lane_c];
            }
        };
        auto store_dy = [&](int buf) {      // mask + act' multiply happen here, after the MFMAs that hid the loads
            float* dyb = dyl + buf * DYB;
            if (lane < pse) {
#pragma unroll
                for (int k = 0; k < DR; ++k) {
                    const bool ok = co_base + wave * DR + k < a.cout && lane < ps;
                    float v;
                    if constexpr (MASK) v = pdy[k] * (((pmask >> ((wave * DR + k) & 31)) & 1u) ? 1.f : 0.01f);
                    else v = gate_act ? pdy[k] * scm_act_grad_from_out(pg[k], gate_act) : pdy[k];
                    dyb[(wave * DR + k) * DPL + lane] = ok ? v : 0.f;
                }
            }
        };

        // ---- prologue of the band: rows y_begin-1 .. y_begin+1 and dy row y_begin go straight to LDS
        __syncthreads();
        fetch_xrow(y_begin - 1);
        store_xrow(y_begin - 1);
        fetch_xrow(y_begin);
        store_xrow(y_begin);
        fetch_xrow(y_begin + 1);
        store_xrow(y_begin + 1);
        fetch_dy(y_begin);
        store_dy(0);
        if (y_begin + 1 < y_end) {
            fetch_xrow(y_begin + 2);
            fetch_dy(y_begin + 1);
        }
        __syncthreads();

        // One barrier per image row: the dy tile is double-buffered and the x rows live in
        // FOUR rolling slots, so row y+1's data (prefetched into registers a row ago) is
        // written to LDS in the middle of row y's MFMAs, then row y+2's loads are issued.
        for (int y = y_begin; y < y_end; ++y) {
            const int cur = (y - y_begin) & 1;
            const float* apc = ap + cur * DYB;
            const int ro0 = ((y + 3) & 3) * PSP, ro1 = ((y + 4) & 3) * PSP, ro2 = ((y + 5) & 3) * PSP;
            const int nks = pse >> 1;
            const int nhalf = (nks / 2 + 1) & ~1;        // even split point (the k loop steps by 2)
            auto lds_k = [&](int ks, float& av, float (&bv)[NTW][9]) {
                av = apc[2 * ks];
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const float* q = bp + n * 32 * XPL + 2 * ks;
                    bv[n][0] = q[ro0]; bv[n][1] = q[ro0 + 1]; bv[n][2] = q[ro0 + 2];
                    bv[n][3] = q[ro1]; bv[n][4] = q[ro1 + 1]; bv[n][5] = q[ro1 + 2];
                    bv[n][6] = q[ro2]; bv[n][7] = q[ro2 + 1]; bv[n][8] = q[ro2 + 2];
                }
            };
            auto mfma_k = [&](float av, const float (&bv)[NTW][9], auto checked) {
                if (do_csum) csum += av;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    // wave-uniform: ci tiles beyond cin (co tiles beyond cout) do no MFMA work
                    if (!decltype(checked)::value || tile_live[n]) {
#pragma unroll
                        for (int t = 0; t < 9; ++t)
                            acc[n][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[n][t], acc[n][t], 0, 0, 0);
                    }
                }
            };
            // k-steps [k0, k1) with an explicit register double-buffer of the LDS operands
            auto row_loop = [&](int k0, int k1, auto checked) {
                if (k0 >= k1) return;
                float av0, av1, bv0[NTW][9], bv1[NTW][9];
                lds_k(k0, av0, bv0);
#pragma unroll 1
                for (int ks = k0; ks < k1; ks += 2) {
                    const int ka = ks + 1 < k1 ? ks + 1 : k1 - 1;
                    lds_k(ka, av1, bv1);
                    mfma_k(av0, bv0, checked);
                    const int kb = ks + 2 < k1 ? ks + 2 : k1 - 1;
                    lds_k(kb, av0, bv0);
                    if (ks + 1 < k1) mfma_k(av1, bv1, checked);
                }
            };
            if (all_live) row_loop(0, nhalf, std::false_type{});
            else row_loop(0, nhalf, std::true_type{});
            if (y + 1 < y_end) {             // row y+1's tiles -> the idle dy buffer and the free x slot
                store_xrow(y + 2);
                store_dy(cur ^ 1);
            }
            if (y + 2 < y_end) {             // row y+2's loads fly under the rest of this row and the next
                fetch_xrow(y + 3);
                fetch_dy(y + 2);
            }
            if (all_live) row_loop(nhalf, nks, std::false_type{});
            else row_loop(nhalf, nks, std::true_type{});
            __syncthreads();
        }
    }

    // ---- partial slab part[split][tap][co][ci]
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int ci = ci_base + (wn * NTW + n) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.cout && ci < cin)
                    a.part[(((size_t)split * 9 + t) * a.cout + co) * cin + ci] = acc[n][t][r];
            }
        }
    }
    if (a.csum_part != nullptr && cig == 0 && wn == 0) {
        csum += __shfl_xor(csum, 32);
        const int co = co_base + wm * 32 + l31;
        if (lh == 0 && co < a.cout) a.csum_part[(size_t)split * a.cout + co] = csum;
    }
}

// ---------------------------------------------------------------- cout <= 16
// Same algorithm on v_mfma_f32_16x16x4_f32 for the 16-channel-latent layers
// (Transition.conv6, Encoder.conv4, the Decoder's factor-summed conv2): a 32-row
// tile would be half (or 29/32) padding.  A = dyp[co (lane&15)][4 pixels (lane>>4)],
// B = x[4 pixels][ci (lane&15)]; 8 waves (two per SIMD), wave w owns ci tile w of the
// workgroup's 128 input channels and its 9 tap accumulators (36 registers).
template <int NCO, int NCI, bool MASK>
__global__ void __launch_bounds__(512, 2) conv3x3_wgrad_m16_kernel(const WgradKArgs a) {
    static_assert(NCO * NCI == 8, "8 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = 8, NCIC = NCI * 16, NCOC = NCO * 16, XR = NCIC / NW, DR = NCOC / NW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int W = a.W, H = a.H, HW = H * W;
    const int PS = a.PS;
    const int PSP = PS + 6;                 // cols -1 .. pse+1 with pse <= PS+3
    const int XPL = (3 * PSP) | 1;
    const int DPL = (PS + 4) | 1;
    float* xl = smem;                  // [NCI*16][XPL]
    float* dyl = smem + NCIC * XPL;    // [NCO*16][DPL]

    const int split = blockIdx.x;
    const int b = split / a.spi;
    const int si = split - b * a.spi;
    const int y_begin = si * a.rows_per_split;
    int y_end = y_begin + a.rows_per_split;
    if (y_end > H) y_end = H;
    const int cig = blockIdx.y % a.ci_groups;
    const int cog = blockIdx.y / a.ci_groups;
    const int ci_base = cig * NCIC, co_base = cog * NCOC;
    const int wco = wave / NCI, wci = wave % NCI;
    const int cin = a.c1 + a.c2;
    const bool circ = a.pad_mode == SCM_PAD_CIRCULAR;
    const int gate_act = a.gate_act;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    const bool do_csum = a.csum_part != nullptr && cig == 0 && wci == 0 && co_base + wco * 16 < a.cout;
    const bool live = ci_base + wci * 16 < cin && co_base + wco * 16 < a.cout;

    const float* ap = dyl + (wco * 16 + l15) * DPL + lq;
    const float* bp = xl + (wci * 16 + l15) * XPL + lq;

    // x_half: the activations are IEEE-half tensors (scm_conv3x3_desc.y_half); xbase() then returns the plane's address
    // counted in HALF elements from a.x1 / a.x2 reinterpreted (same element offsets, two bytes each): ld_x() reads it
    const bool xh = a.x_half != 0;
    auto xbase = [&](int c, bool& valid) -> const float* {
        const int ci = ci_base + c;
        valid = ci < cin;
        const int cc = valid ? ci : 0;
        const size_t o = cc < a.c1 ? ((size_t)b * a.c1 + cc) * HW : ((size_t)b * a.c2 + (cc - a.c1)) * HW;
        const float* base = cc < a.c1 ? a.x1 : a.x2;
        return xh ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(base) + o) : base + o;
    };
    auto ld_x = [&](const float* plane, int idx) -> float {
        return xh ? (float)reinterpret_cast<const _Float16*>(plane)[idx] : plane[idx];
    };
    const int hc = tid >> 1, hs = tid & 1;       // halo element of this thread (threads 0..255)
    bool hvalid = false;
    const float* hsrc = xbase(hc < NCIC ? hc : 0, hvalid);
    hvalid = hvalid && hc < NCIC;

    float pdy[DR], pg[MASK ? 1 : DR], px[XR], phalo = 0.f;
    unsigned pmask = 0u;
    static_assert(!MASK || 32 % DR == 0, "a wave's dy channels must not straddle mask words");
    const int mask_nw = (a.cout + 31) >> 5;

    for (int seg = 0; seg < a.nseg; ++seg) {
        const int x0 = seg * PS;
        int ps = W - x0;
        if (ps > PS) ps = PS;
        const int pse = (ps + 3) & ~3;           // k-steps of 4 pixels
        auto img_col = [&](int cc, bool& ok) -> int {
            int xi = x0 + cc - 1;
            if (circ) {
                ok = true;
                xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);
            } else {
                ok = (xi >= 0 && xi < W);
            }
            return xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
        };
        bool ok_main, ok_halo;
        const int xi_main = img_col(lane + 1, ok_main);
        ok_main = ok_main && (lane < pse + 1);
        const int hcc = hs ? pse + 1 : 0;
        const int xi_halo = img_col(hcc, ok_halo);
        ok_halo = ok_halo && hvalid;
        const int lane_c = lane < ps ? lane : ps - 1;

        auto fetch_xrow = [&](int r) {
            int ri = r;
            if (circ) ri = r < 0 ? r + H : (r >= H ? r - H : r);
            ri = ri < 0 ? 0 : (ri >= H ? H - 1 : ri);
            const int ro = ri * W;
#pragma unroll
            for (int k = 0; k < XR; ++k) {
                bool cv;
                const float* src = xbase(wave * XR + k, cv);
                px[k] = ld_x(src, ro + xi_main);
            }
            phalo = ld_x(hsrc, ro + xi_halo);
        };
        auto store_xrow = [&](int r) {
            const int slot = (r + 3) % 3;
            const bool rvalid = circ || (r >= 0 && r < H);
            if (lane < pse + 1) {
#pragma unroll
                for (int k = 0; k < XR; ++k) {
                    const bool cv = ci_base + wave * XR + k < cin;
                    xl[(wave * XR + k) * XPL + slot * PSP + lane + 1] = (cv && rvalid && ok_main) ? px[k] : 0.f;
                }
            }
            if (hc < NCIC) xl[hc * XPL + slot * PSP + hcc] = (rvalid && ok_halo) ? phalo : 0.f;
        };
        auto fetch_dy = [&](int y) {
#pragma unroll
            for (int k = 0; k < DR; ++k) {
                const int co = co_base + wave * DR + k;
                const int coc = co < a.cout ? co : a.cout - 1;
                const size_t rowo = ((size_t)b * a.cout + coc) * HW + (size_t)y * W + x0;
                pdy[k] = a.dy[rowo + lane_c];
                if constexpr (!MASK) pg[k] = gate_act ? a.gate[rowo + lane_c] : 1.f;
            }
            if constexpr (MASK) {
                const int w = (co_base + wave * DR) >> 5;
                pmask = a.gate_mask[((size_t)b * mask_nw + (w < mask_nw ? w : 0)) * HW + (size_t)y * W + x0 + lane_c];
            }
        };
        auto store_dy = [&]() {
            if (lane < pse) {
#pragma unroll
                for (int k = 0; k < DR; ++k) {
                    const bool ok = co_base + wave * DR + k < a.cout && lane < ps;
                    float v;
                    if constexpr (MASK) v = pdy[k] * (((pmask >> ((wave * DR + k) & 31)) & 1u) ? 1.f : 0.01f);
                    else v = gate_act ? pdy[k] * scm_act_grad_from_out(pg[k], gate_act) : pdy[k];
                    dyl[(wave * DR + k) * DPL + lane] = ok ? v : 0.f;
                }
            }
        };

        __syncthreads();
        fetch_xrow(y_begin - 1);
        store_xrow(y_begin - 1);
        fetch_xrow(y_begin);
        store_xrow(y_begin);
        fetch_xrow(y_begin + 1);
        fetch_dy(y_begin);

        for (int y = y_begin; y < y_end; ++y) {
            __syncthreads();
            store_xrow(y + 1);
            store_dy();
            __syncthreads();
            if (y + 1 < y_end) {
                fetch_xrow(y + 2);
                fetch_dy(y + 1);
            }
            const int ro0 = ((y + 2) % 3) * PSP, ro1 = (y % 3) * PSP, ro2 = ((y + 1) % 3) * PSP;
            const int nks = pse >> 2;
            if (live || do_csum) {
#pragma unroll 2
                for (int ks = 0; ks < nks; ++ks) {
                    const float av = ap[4 * ks];
                    if (do_csum) csum += av;
                    const float* q = bp + 4 * ks;
                    float bv[9];
                    bv[0] = q[ro0]; bv[1] = q[ro0 + 1]; bv[2] = q[ro0 + 2];
                    bv[3] = q[ro1]; bv[4] = q[ro1 + 1]; bv[5] = q[ro1 + 2];
                    bv[6] = q[ro2]; bv[7] = q[ro2 + 1]; bv[8] = q[ro2 + 2];
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
                }
            }
        }
    }

    const int ci = ci_base + wci * 16 + l15;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co_base + wco * 16 + 4 * lq + r;
            if (co < a.cout && ci < cin) a.part[(((size_t)split * 9 + t) * a.cout + co) * cin + ci] = acc[t][r];
        }
    }
    if (do_csum) {
        csum += __shfl_xor(csum, 16);
        csum += __shfl_xor(csum, 32);
        const int co = co_base + wco * 16 + l15;
        if (lq == 0 && co < a.cout) a.csum_part[(size_t)split * a.cout + co] = csum;
    }
}

// Sum the per-split slabs in split order and scatter into the torch layout.
//   dw[co][ci][3][3]                    (transposed_dst == 0, nn.Conv2d)
//   dw[ci][co][3][3] with taps flipped  (transposed_dst == 1, nn.ConvTranspose2d)
// The last gridDim.x - nblk_w workgroups of the same launch reduce the per-sample channel sums (bias and action-plane
// gradients): csum[b][co] = sum over the image's splits, db[co] = sum_b csum[b][co]; 64 co x 4 batch lanes each.
struct CsumReduceArgs {
    const float* csum_part; int B; int spi; int accumulate; float* csum; float* db;
};
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int cout,
                                                           int cin, int transposed_dst, int accumulate, int dct,
                                                           int dc0, float* __restrict__ dw,
                                                           const float* __restrict__ dot_w,
                                                           float* __restrict__ dot_part, int nblk_w,
                                                           const CsumReduceArgs cs) {
    __shared__ float red[4];
    if ((int)blockIdx.x >= nblk_w) {
        __shared__ float redc[4][64];
        const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
        const int co = (blockIdx.x - nblk_w) * 64 + cl;
        float tot = 0.f;
        if (co < cout) {
            for (int b = bl; b < cs.B; b += 4) {
                float sb = 0.f;
                for (int k = 0; k < cs.spi; ++k) sb += cs.csum_part[((size_t)b * cs.spi + k) * cout + co];
                if (cs.csum) cs.csum[(size_t)b * cout + co] = sb;
                tot += sb;
            }
        }
        redc[bl][cl] = tot;
        __syncthreads();
        if (bl == 0 && co < cout && cs.db) {
            const float t = (redc[0][cl] + redc[1][cl]) + (redc[2][cl] + redc[3][cl]);
            cs.db[co] = cs.accumulate ? cs.db[co] + t : t;
        }
        return;
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_tap = cout * cin;
    float prod = 0.f;
    if (idx < 9 * per_tap) {
    const int t = idx / per_tap;
    const int rem = idx - t * per_tap;
    const int co = rem / cin, ci = rem - co * cin;
    // 8 independent partial sums keep 8 slab loads in flight; fixed order -> reproducible
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)9 * per_tap;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] += part[(size_t)(k + j) * stride + idx];
    }
    for (; k < nsplit; ++k) s8[0] += part[(size_t)k * stride + idx];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    const size_t o = transposed_dst ? ((size_t)(dc0 + ci) * cout + co) * 9 + (8 - t)
                                    : ((size_t)co * dct + dc0 + ci) * 9 + t;
    dw[o] = accumulate ? dw[o] + s : s;
    if (dot_w) prod = s * dot_w[o];
    }
    if (dot_part) {      // <dW_eff, W_bar> partial of this block, for the SpectralNorm fold
        prod = wave_sum(prod);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = prod;
        __syncthreads();
        if (threadIdx.x == 0) dot_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ------------------------------------------------------------------ host

static int wgrad_plan(int B, int H, int W, int cin, int cout, WgradPlan* p) {
    const int cot = (cout + 31) / 32, cit = (cin + 31) / 32;
    p->m16 = cout <= 16 ? 1 : (cin <= 16 ? 2 : 0);     // 1: 16 co x 128 ci per WG; 2: 128 co x 16 ci per WG
    if (cot >= 3) { p->mt = 4; p->nt = cit >= 2 ? 2 : 1; }     // 128 co x 64 (32) ci per workgroup
    else if (cot == 2) { p->mt = 2; p->nt = cit >= 3 ? 4 : 2; } //  64 co x 128 (64) ci
    else { p->mt = 1; p->nt = 4; }                             //  32 co x 128 ci
    p->co_groups = (cot + p->mt - 1) / p->mt;
    p->ci_groups = (cit + p->nt - 1) / p->nt;
    if (p->m16 == 1) { p->co_groups = 1; p->ci_groups = (cin + 127) / 128; }
    if (p->m16 == 2) { p->co_groups = (cout + 127) / 128; p->ci_groups = 1; }
    p->nseg = (W + 63) / 64;
    p->PS = (W + p->nseg - 1) / p->nseg;
    const int groups = p->co_groups * p->ci_groups;
    // one workgroup per CU fills the chip (8 waves = 2 per SIMD): aim at ~256 workgroups
    int spi = (256 + B * groups - 1) / (B * groups);
    if (spi < 1) spi = 1;
    if (spi > H) spi = H;
    p->rows_per_split = (H + spi - 1) / spi;
    p->spi = (H + p->rows_per_split - 1) / p->rows_per_split;
    p->nsplit = B * p->spi;
    p->ws_splits = p->nsplit;
    if (p->m16 && B * ((W + 31) / 32) > p->ws_splits) p->ws_splits = B * ((W + 31) / 32);
    return SCM_OK;
}

extern "C" int scm_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout, size_t* part_floats,
                                           size_t* csum_floats) {
    SCM_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, "scm_conv3x3_wgrad_workspace: bad shape");
    WgradPlan p;
    wgrad_plan(B, H, W, cin, cout, &p);
    if (part_floats) *part_floats = (size_t)p.ws_splits * 9 * cout * cin;
    if (csum_floats) *csum_floats = (size_t)p.ws_splits * cout;
    return SCM_OK;
}

template <int WM, int WN, int NTW, bool MASK>
static int launch_wgrad_m(const WgradKArgs& a, const WgradPlan& p, hipStream_t s) {
    constexpr int MT = WM, NT = WN * NTW;
    const int PSP = a.PS + 4, XPL = (4 * PSP) | 1, DPL = (a.PS + 2) | 1;
    const size_t lds = ((size_t)NT * 32 * XPL + (size_t)2 * MT * 32 * DPL) * sizeof(float);
    auto kern = conv3x3_wgrad_kernel<WM, WN, NTW, MASK>;
    static std::atomic<uint64_t> attr_done{0};       // per (instantiation, device)
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done, 160 * 1024)) return rc;
    if (lds > 160 * 1024) {
        // e.g. 33..64 output channels from > 64 input channels on 64-wide maps (174 080 B): no shipped layer has
        // that shape; fail with a message instead of an opaque launch error (the forward conv checks the same way)
        scm_set_error("scm_conv3x3_wgrad: the <%d,%d> tile needs %zu bytes of LDS at row segment %d (limit 160 KB)", WM,
                      NT, lds, a.PS);
        return SCM_EUNSUPPORTED;
    }
    dim3 grid(p.nsplit, p.co_groups * p.ci_groups);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_kernel");
    return SCM_OK;
}

template <int WM, int WN, int NTW>
static int launch_wgrad(const WgradKArgs& a, const WgradPlan& p, hipStream_t s) {
    if (a.gate_mask != nullptr && a.gate_act == SCM_ACT_LRELU) return launch_wgrad_m<WM, WN, NTW, true>(a, p, s);
    return launch_wgrad_m<WM, WN, NTW, false>(a, p, s);
}

static int wgrad_impl(const scm_conv3x3_wgrad_desc* d, int precision, void* stream) {
    SCM_REQUIRE(d != nullptr, "scm_conv3x3_wgrad: null descriptor");
    SCM_REQUIRE(d->x1 && d->c1 > 0, "scm_conv3x3_wgrad: x1/c1 required");
    SCM_REQUIRE((d->x2 != nullptr) == (d->c2 > 0) && d->c2 >= 0, "scm_conv3x3_wgrad: x2/c2 mismatch");
    SCM_REQUIRE(d->dy && d->cout > 0, "scm_conv3x3_wgrad: dy/cout required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "scm_conv3x3_wgrad: bad shape");
    SCM_REQUIRE(d->part_ws && d->dw, "scm_conv3x3_wgrad: part_ws and dw required");
    SCM_REQUIRE(d->pad_mode == SCM_PAD_ZERO || d->pad_mode == SCM_PAD_CIRCULAR, "scm_conv3x3_wgrad: bad pad_mode");
    SCM_REQUIRE(d->gate_act >= 0 && d->gate_act <= 2 && (d->gate_act == 0 || d->gate), "scm_conv3x3_wgrad: bad gate");
    const bool want_cs = d->db != nullptr || d->csum != nullptr;
    SCM_REQUIRE(!want_cs || d->csum_ws, "scm_conv3x3_wgrad: csum_ws required for db/csum");
    const int cin = d->c1 + d->c2;
    WgradPlan p;
    wgrad_plan(d->B, d->H, d->W, cin, d->cout, &p);
    if (precision == SCM_PREC_BF16X3 && d->cout > 64 && p.m16 == 0 && p.mt == 4) {
        // the split kernel's tiles are 128 co x 32 ci (twice the groups of the plan above): fewer, longer row splits fill
        // the chip just as well, with half the partial slabs to write and reduce and half the 3-row prologues.  Never
        // more splits than the plan above, which sized the workspaces.
        const int groups = p.co_groups * ((cin + 31) / 32);
        int spi = (256 + d->B * groups - 1) / (d->B * groups);
        if (spi < 1) spi = 1;
        if (spi > p.spi) spi = p.spi;
        p.rows_per_split = (d->H + spi - 1) / spi;       // (56.3-56.6 vs 56.7 ms per configs[1] iteration)
        p.spi = (d->H + p.rows_per_split - 1) / p.rows_per_split;
        p.nsplit = d->B * p.spi;
    }
    WgradKArgs a;
    a.x1 = d->x1; a.x2 = d->x2; a.c1 = d->c1; a.c2 = d->c2;
    a.dy = d->dy; a.gate = d->gate; a.gate_act = d->gate_act; a.cout = d->cout;
    a.part = d->part_ws; a.csum_part = want_cs ? d->csum_ws : nullptr;
    a.B = d->B; a.H = d->H; a.W = d->W; a.pad_mode = d->pad_mode;
    a.spi = p.spi; a.rows_per_split = p.rows_per_split; a.nseg = p.nseg; a.PS = p.PS;
    a.ci_groups = p.ci_groups;
    a.gate_mask = d->gate_mask;
    a.x_half = d->x_half != 0;
    // IEEE-half activations: read by the 16-bit weight-gradient kernel and the direct 16-channel kernels only
    const bool h16_wide = precision != SCM_PREC_F32 && precision != SCM_PREC_BF16X3 && precision != SCM_ALGO_WINO &&
                          d->cout > 64 && p.m16 == 0 && p.mt == 4;
    SCM_REQUIRE(!a.x_half || h16_wide || (p.m16 && precision != SCM_ALGO_WINO),
                "scm_conv3x3_wgrad: x_half is taken by the 16-bit weight-gradient kernel (scm_conv3x3_wgrad_h16, more than 64 "
                "output channels) and the direct 16-channel-side kernels only");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (precision == SCM_ALGO_WINO && d->cout >= 48 && cin >= 48 && scm_wgrad_wino_serves(a)) {
        // Winograd F(2x2,3x3) on the f32 MFMA pipe (conv3x3_wgrad_wino.hip); same slabs, its own (coarser) split
        rc = scm_launch_wgrad_wino(a, p.nsplit, &p.nsplit, &p.spi, s);
    } else if (precision == SCM_ALGO_WINO && p.m16 && scm_wgrad_wino16_serves(a, p.ws_splits)) {
        // the same algorithm for the 16-channel-side layers (conv3x3_wgrad_wino16.hip)
        rc = scm_launch_wgrad_wino16(a, p.ws_splits, &p.nsplit, &p.spi, s);
    } else if (precision == SCM_ALGO_WINO) {
        return wgrad_impl(d, SCM_PREC_F32, stream);          // shapes it does not take: the direct kernels
    } else if (precision == SCM_PREC_BF16X3 && d->cout > 64 && p.m16 == 0 && p.mt == 4) {
        // fp32 arithmetic on the bf16 MFMA pipe (conv3x3_wgrad_x3.hip); same split, same slabs
        rc = scm_launch_wgrad_x3(a, p, s);
    } else if (precision != SCM_PREC_F32 && precision != SCM_PREC_BF16X3 && d->cout > 64 && p.m16 == 0 && p.mt == 4) {
        // 128 co x 64 (32) ci tiles on the 16-bit MFMA (conv3x3_wgrad_h16.hip); same split, same slabs.
        // cin <= 16 (Transition.conv1) stays on the fp32 kernel: that launch is bound by reading dy,
        // 16-bit operands gain nothing there (106 vs 109 us) and cost gradient accuracy.
        rc = scm_launch_wgrad_h16(a, p, precision, s);
    } else if (p.m16) {
        const int nci = p.m16 == 1 ? 128 : 16, nco = p.m16 == 1 ? 16 : 128;
        const int PSP = a.PS + 6, XPL = (3 * PSP) | 1, DPL = (a.PS + 4) | 1;
        const size_t lds = ((size_t)nci * XPL + (size_t)nco * DPL) * sizeof(float);
        // the 1-bit gate serves the 128-co form (a wave's 16 channels sit in one mask word)
        const bool mk = p.m16 == 2 && a.gate_mask != nullptr && a.gate_act == SCM_ACT_LRELU;
        auto kern = p.m16 == 1 ? conv3x3_wgrad_m16_kernel<1, 8, false>
                               : (mk ? conv3x3_wgrad_m16_kernel<8, 1, true> : conv3x3_wgrad_m16_kernel<8, 1, false>);
        static std::atomic<uint64_t> attr_done[3];        // per (instantiation, device)
        const int ai = p.m16 == 1 ? 0 : (mk ? 2 : 1);
        if ((rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(kern), attr_done[ai], 160 * 1024))) return rc;
        if (lds > 160 * 1024) {
            scm_set_error("scm_conv3x3_wgrad: %zu bytes of LDS needed for %d -> %d channels at width %d (limit 160 KB)",
                          lds, cin, d->cout, d->W);
            return SCM_EUNSUPPORTED;
        }
        hipLaunchKernelGGL(kern, dim3(p.nsplit, p.co_groups * p.ci_groups), dim3(512), lds, s, a);
        SCM_LAUNCH_CHECK("conv3x3_wgrad_m16_kernel");
        rc = SCM_OK;
    } else if (p.mt == 4) rc = p.nt == 2 ? launch_wgrad<4, 2, 1>(a, p, s) : launch_wgrad<4, 1, 1>(a, p, s);
    else if (p.mt == 2) rc = p.nt == 4 ? launch_wgrad<2, 4, 1>(a, p, s) : launch_wgrad<2, 2, 1>(a, p, s);
    else rc = launch_wgrad<1, 4, 1>(a, p, s);
    if (rc) return rc;
    const int total = 9 * d->cout * cin;
    const int dct = d->dw_cin_total ? d->dw_cin_total : cin;
    SCM_REQUIRE(d->dw_ci0 >= 0 && d->dw_ci0 + cin <= dct, "scm_conv3x3_wgrad: bad dw channel slice");
    SCM_REQUIRE(!d->dot_w || (d->dot_part && !(d->accumulate & 1)), "scm_conv3x3_wgrad: dot_w needs dot_part and no dw accumulation");
    const int nblk_w = (total + 255) / 256;
    CsumReduceArgs cs;
    cs.csum_part = d->csum_ws; cs.B = d->B; cs.spi = p.spi; cs.accumulate = (d->accumulate >> 1) & 1; cs.csum = d->csum; cs.db = d->db;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk_w + (want_cs ? (d->cout + 63) / 64 : 0)), dim3(256), 0, s, d->part_ws,
                       p.nsplit, d->cout, cin, d->transposed_dst, d->accumulate & 1, dct, d->dw_ci0, d->dw, d->dot_w,
                       d->dot_w ? d->dot_part : nullptr, nblk_w, cs);
    SCM_LAUNCH_CHECK("wgrad_reduce_kernel");
    return SCM_OK;
}

extern "C" int scm_conv3x3_wgrad(const scm_conv3x3_wgrad_desc* d, void* stream) {
    return wgrad_impl(d, SCM_PREC_F32, stream);
}

// The same gradient by Winograd F(2x2,3x3) for layers with at least 48 channels on both sides and 4-pixel-aligned rows
// (others fall through to the direct kernels): fp32 operands and accumulation, 4/9 of the products.
extern "C" int scm_conv3x3_wgrad_wino(const scm_conv3x3_wgrad_desc* d, void* stream) {
    return wgrad_impl(d, SCM_ALGO_WINO, stream);
}

// Opt-in: the 128 co x 64 ci workgroup tiles run on bf16/fp16 operands (fp32 accumulate, fp32
// partial slabs and reduction); every other shape uses the fp32 kernels above.
extern "C" int scm_conv3x3_wgrad_h16(const scm_conv3x3_wgrad_desc* d, int precision, void* stream) {
    SCM_REQUIRE(precision == SCM_PREC_F16 || precision == SCM_PREC_BF16 || precision == SCM_PREC_BF16X3,
                "scm_conv3x3_wgrad_h16: bad precision %d", precision);
    return wgrad_impl(d, precision, stream);
}
