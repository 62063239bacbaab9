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
#include "common.h"

struct WgradKArgs {
    const float* x1; const float* x2; int c1; int c2;
    const float* dy; const float* gate; int gate_act; int cout;
    float* part; float* csum_part;
    int B; int H; int W; int pad_mode;
    int spi; int rows_per_split; int nseg; int PS;
    int ci_groups;
};

template <int MT, int NT>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_kernel(const WgradKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NT, wn = wave % NT;
    const int W = a.W, H = a.H, HW = H * W;
    const int PS = a.PS;
    const int PSP = PS + 4;                 // row pitch of the x tile (cols -1 .. PS+2)
    const int XPL = (3 * PSP) | 1;          // odd plane pitch: conflict-free lane->ci reads
    const int DPL = (PS + 2) | 1;           // odd row pitch of the dy tile: lane->co reads
    float* xl = smem;                       // [NT*32][XPL]
    float* dyl = smem + NT * 32 * XPL;      // [MT*32][DPL]

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

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float csum = 0.f;
    const bool do_csum = a.csum_part != nullptr && cig == 0 && wn == 0;

    const float* ap = dyl + (wm * 32 + l31) * DPL + lh;
    const float* bp = xl + (wn * 32 + l31) * XPL + lh;

    for (int seg = 0; seg < a.nseg; ++seg) {
        const int x0 = seg * PS;
        int ps = W - x0;
        if (ps > PS) ps = PS;
        const int pse = (ps + 1) & ~1;
        for (int y = y_begin; y < y_end; ++y) {
            __syncthreads();
            // ---- x halo rows: all three on the first row of a band, then only row y+1
            for (int rr = (y == y_begin ? -1 : 1); rr <= 1; ++rr) {
                const int r = y + rr;
                const int slot = (r + 3) % 3;
                int ri = r;
                bool rvalid = true;
                if (circ) ri = r < 0 ? r + H : (r >= H ? r - H : r);
                else rvalid = (r >= 0 && r < H);
                for (int c = wave; c < NT * 32; c += 4) {
                    const int ci = ci_base + c;
                    const float* src = nullptr;
                    if (ci < cin && rvalid)
                        src = ci < a.c1 ? a.x1 + ((size_t)b * a.c1 + ci) * HW + (size_t)ri * W
                                        : a.x2 + ((size_t)b * a.c2 + (ci - a.c1)) * HW + (size_t)ri * W;
                    float* dst = xl + c * XPL + slot * PSP;
                    for (int cc = lane; cc < pse + 2; cc += 64) {
                        int xi = x0 + cc - 1;
                        bool v = src != nullptr;
                        if (circ) xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);
                        else v = v && xi >= 0 && xi < W;
                        dst[cc] = v ? src[xi] : 0.f;
                    }
                }
            }
            // ---- dyp row tile
            for (int c = wave; c < MT * 32; c += 4) {
                const int co = co_base + c;
                float v = 0.f;
                if (co < a.cout && lane < ps) {
                    const size_t o = ((size_t)b * a.cout + co) * HW + (size_t)y * W + x0 + lane;
                    v = a.dy[o];
                    if (a.gate_act) v *= scm_act_grad_from_out(a.gate[o], a.gate_act);
                }
                if (lane < pse) dyl[c * DPL + lane] = v;
            }
            __syncthreads();
            // ---- MFMA: k = pixel pairs of this row segment
            const int ro0 = ((y + 2) % 3) * PSP, ro1 = (y % 3) * PSP, ro2 = ((y + 1) % 3) * PSP;
            const int nks = pse >> 1;
            for (int ks = 0; ks < nks; ++ks) {
                const float av = ap[2 * ks];
                if (do_csum) csum += av;
                const float* q = bp + 2 * ks;
                float bv[9];
                bv[0] = q[ro0]; bv[1] = q[ro0 + 1]; bv[2] = q[ro0 + 2];
                bv[3] = q[ro1]; bv[4] = q[ro1 + 1]; bv[5] = q[ro1 + 2];
                bv[6] = q[ro2]; bv[7] = q[ro2 + 1]; bv[8] = q[ro2 + 2];
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
            }
        }
    }

    // ---- partial slab part[split][tap][co][ci]
    const int ci = ci_base + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < a.cout && ci < cin)
                a.part[(((size_t)split * 9 + t) * a.cout + co) * cin + ci] = acc[t][r];
        }
    }
    if (a.csum_part != nullptr && cig == 0 && wn == 0) {
        csum += __shfl_xor(csum, 32);
        const int co = co_base + wm * 32 + l31;
        if (lh == 0 && co < a.cout) a.csum_part[(size_t)split * a.cout + co] = csum;
    }
}

// Sum the per-split slabs in split order and scatter into the torch layout.
//   dw[co][ci][3][3]                    (transposed_dst == 0, nn.Conv2d)
//   dw[ci][co][3][3] with taps flipped  (transposed_dst == 1, nn.ConvTranspose2d)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int cout, int cin,
                                    int transposed_dst, int accumulate, int dct, int dc0, float* __restrict__ dw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_tap = cout * cin;
    if (idx >= 9 * per_tap) return;
    const int t = idx / per_tap;
    const int rem = idx - t * per_tap;
    const int co = rem / cin, ci = rem - co * cin;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * 9 * per_tap + idx];
    const size_t o = transposed_dst ? ((size_t)(dc0 + ci) * cout + co) * 9 + (8 - t)
                                    : ((size_t)co * dct + dc0 + ci) * 9 + t;
    dw[o] = accumulate ? dw[o] + s : s;
}

// csum[b][co] = sum over the image's splits; db[co] = sum_b csum[b][co]
__global__ void csum_reduce_kernel(const float* __restrict__ csum_part, int B, int spi, int cout,
                                   int accumulate, float* __restrict__ csum, float* __restrict__ db) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= cout) return;
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int k = 0; k < spi; ++k) s += csum_part[((size_t)b * spi + k) * cout + co];
        if (csum) csum[(size_t)b * cout + co] = s;
        tot += s;
    }
    if (db) db[co] = accumulate ? db[co] + tot : tot;
}

// ------------------------------------------------------------------ host
struct WgradPlan {
    int mt, nt, spi, rows_per_split, nseg, PS, co_groups, ci_groups, nsplit;
};

static int wgrad_plan(int B, int H, int W, int cin, int cout, WgradPlan* p) {
    const int cot = (cout + 31) / 32, cit = (cin + 31) / 32;
    if (cot >= 4 || cit == 1) { p->mt = 4; p->nt = 1; }
    else if (cot >= 2) { p->mt = 2; p->nt = 2; }
    else { p->mt = 1; p->nt = 4; }
    p->co_groups = (cot + p->mt - 1) / p->mt;
    p->ci_groups = (cit + p->nt - 1) / p->nt;
    p->nseg = (W + 63) / 64;
    p->PS = (W + p->nseg - 1) / p->nseg;
    const int groups = p->co_groups * p->ci_groups;
    int spi = (768 + B * groups - 1) / (B * groups);
    if (spi < 1) spi = 1;
    if (spi > H) spi = H;
    p->rows_per_split = (H + spi - 1) / spi;
    p->spi = (H + p->rows_per_split - 1) / p->rows_per_split;
    p->nsplit = B * p->spi;
    return SCM_OK;
}

extern "C" int scm_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout, size_t* part_floats,
                                           size_t* csum_floats) {
    SCM_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, "scm_conv3x3_wgrad_workspace: bad shape");
    WgradPlan p;
    wgrad_plan(B, H, W, cin, cout, &p);
    if (part_floats) *part_floats = (size_t)p.nsplit * 9 * cout * cin;
    if (csum_floats) *csum_floats = (size_t)p.nsplit * cout;
    return SCM_OK;
}

template <int MT, int NT>
static int launch_wgrad(const WgradKArgs& a, const WgradPlan& p, hipStream_t s) {
    const int PSP = a.PS + 4, XPL = (3 * PSP) | 1, DPL = (a.PS + 2) | 1;
    const size_t lds = ((size_t)NT * 32 * XPL + (size_t)MT * 32 * DPL) * sizeof(float);
    auto kern = conv3x3_wgrad_kernel<MT, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = scm_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                               "hipFuncSetAttribute");
        if (rc) return rc;
        attr_set = true;
    }
    dim3 grid(p.nsplit, p.co_groups * p.ci_groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("conv3x3_wgrad_kernel");
    return SCM_OK;
}

extern "C" int scm_conv3x3_wgrad(const scm_conv3x3_wgrad_desc* d, void* stream) {
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
    WgradKArgs a;
    a.x1 = d->x1; a.x2 = d->x2; a.c1 = d->c1; a.c2 = d->c2;
    a.dy = d->dy; a.gate = d->gate; a.gate_act = d->gate_act; a.cout = d->cout;
    a.part = d->part_ws; a.csum_part = want_cs ? d->csum_ws : nullptr;
    a.B = d->B; a.H = d->H; a.W = d->W; a.pad_mode = d->pad_mode;
    a.spi = p.spi; a.rows_per_split = p.rows_per_split; a.nseg = p.nseg; a.PS = p.PS;
    a.ci_groups = p.ci_groups;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (p.mt == 4) rc = launch_wgrad<4, 1>(a, p, s);
    else if (p.mt == 2) rc = launch_wgrad<2, 2>(a, p, s);
    else rc = launch_wgrad<1, 4>(a, p, s);
    if (rc) return rc;
    const int total = 9 * d->cout * cin;
    const int dct = d->dw_cin_total ? d->dw_cin_total : cin;
    SCM_REQUIRE(d->dw_ci0 >= 0 && d->dw_ci0 + cin <= dct, "scm_conv3x3_wgrad: bad dw channel slice");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, d->part_ws, p.nsplit, d->cout,
                       cin, d->transposed_dst, d->accumulate, dct, d->dw_ci0, d->dw);
    SCM_LAUNCH_CHECK("wgrad_reduce_kernel");
    if (want_cs) {
        hipLaunchKernelGGL(csum_reduce_kernel, dim3((d->cout + 63) / 64), dim3(64), 0, s, d->csum_ws, d->B, p.spi,
                           d->cout, d->accumulate, d->csum, d->db);
        SCM_LAUNCH_CHECK("csum_reduce_kernel");
    }
    return SCM_OK;
}
