// Generic direct 2-D convolution (any square kernel, stride, zero padding) on the
// vector ALU, for the small layers of the path that do not fit the 3x3 same-size
// MFMA kernel: RewardPredictor conv1 (16->32, k3 s1 p0) and conv2 (32->3R, k3 s2
// p0), models.py:230-232; Discriminator conv1-3 (k3 s2 p0), models.py:199-203;
// CoordConv2d's arbitrary nn.Conv2d arguments, coordconv.py:8.  Together < 1 %
// of the path's FLOPs (Dz).
//
// forward : thread = one output pixel x CB consecutive output channels
// dgrad   : thread = one input pixel x CB consecutive input channels
// wgrad   : workgroup = one (co, ci) pair, threads stride the (b, oy, ox) reduction
#include "common.h"

#define CB 8

struct SmallConvArgs {
    const float* x; const float* w; const float* bias; const float* scale; float* y;
    const float* gate; int gate_act;
    int B, cin, H, W, cout, OH, OW, K, stride, pad, act;
};

template <int KT>
__global__ void __launch_bounds__(256) smallconv_fwd_kernel(const SmallConvArgs a) {
    const int K = KT ? KT : a.K;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int co0 = blockIdx.y * CB;
    const int b = blockIdx.z;
    if (p >= a.OH * a.OW) return;
    const int oy = p / a.OW, ox = p - oy * a.OW;
    float acc[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) acc[k] = 0.f;
    const int KK = K * K;
    for (int ci = 0; ci < a.cin; ++ci) {
        const float* xp = a.x + ((size_t)b * a.cin + ci) * a.H * a.W;
        // all K*K loads of this input plane are issued unconditionally (clamped, masked) so
        // they fly together instead of one dependent load per tap
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
            const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * a.stride - a.pad + kx;
                const int ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
                const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const float xr = xp[iyc * a.W + ixc];
                const float xv = ok ? xr : 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int co = co0 + k < a.cout ? co0 + k : a.cout - 1;
                    acc[k] = fmaf(a.w[((size_t)co * a.cin + ci) * KK + ky * K + kx], xv, acc[k]);
                }
            }
        }
    }
    const float sc = a.scale ? *a.scale : 1.f;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        const int co = co0 + k;
        if (co < a.cout) {
            const float v = acc[k] * sc + (a.bias ? a.bias[co] : 0.f);
            a.y[((size_t)b * a.cout + co) * a.OH * a.OW + p] = scm_act(v, a.act);
        }
    }
}

// dx[b,ci,iy,ix] = scale * sum_{co,ky,kx} dyp[b,co,oy,ox] w[co,ci,ky,kx], oy*stride-pad+ky == iy
template <int KT>
__global__ void __launch_bounds__(256) smallconv_dgrad_kernel(const SmallConvArgs a, const float* __restrict__ dy,
                                                              float* __restrict__ dx) {
    const int K = KT ? KT : a.K;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int ci0 = blockIdx.y * CB;
    const int b = blockIdx.z;
    if (p >= a.H * a.W) return;
    const int iy = p / a.W, ix = p - iy * a.W;
    float acc[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) acc[k] = 0.f;
    const int KK = K * K;
    const int OHW = a.OH * a.OW;
    for (int co = 0; co < a.cout; ++co) {
        const size_t ob = ((size_t)b * a.cout + co) * OHW;
        // unconditional clamped loads of the (at most K*K) output pixels that read this input pixel
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ty = iy + a.pad - ky;
            const int oy = ty / a.stride;
            const bool oky = ty >= 0 && ty - oy * a.stride == 0 && oy < a.OH;
            const int oyc = oy < 0 ? 0 : (oy >= a.OH ? a.OH - 1 : oy);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int tx = ix + a.pad - kx;
                const int ox = tx / a.stride;
                const bool ok = oky && tx >= 0 && tx - ox * a.stride == 0 && ox < a.OW;
                const int oxc = ox < 0 ? 0 : (ox >= a.OW ? a.OW - 1 : ox);
                const size_t o = ob + oyc * a.OW + oxc;
                float g = dy[o];
                if (a.gate_act) g *= scm_act_grad_from_out(a.gate[o], a.gate_act);
                g = ok ? g : 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int ci = ci0 + k < a.cin ? ci0 + k : a.cin - 1;
                    acc[k] = fmaf(a.w[((size_t)co * a.cin + ci) * KK + ky * K + kx], g, acc[k]);
                }
            }
        }
    }
    const float sc = a.scale ? *a.scale : 1.f;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
        const int ci = ci0 + k;
        if (ci < a.cin) dx[((size_t)b * a.cin + ci) * a.H * a.W + p] = acc[k] * sc;
    }
}

// dw[co,ci,ky,kx] = sum_{b,oy,ox} dyp[b,co,oy,ox] x[b,ci,oy*s-pad+ky,ox*s-pad+kx];  db[co] = sum dyp
// Two deterministic phases: workgroup (co, ci, split) reduces its slice of the batch
// into ws[split][co][ci][k] (+ ws_db[split][co]); a second kernel sums the splits.
#define SW_MAXKK 25
template <int KT>
__global__ void __launch_bounds__(256) smallconv_wgrad_kernel(const SmallConvArgs a, const float* __restrict__ dy,
                                                              float* __restrict__ ws, float* __restrict__ ws_db,
                                                              int bps) {
    __shared__ float red[4];
    const int co = blockIdx.x, ci = blockIdx.y, sp = blockIdx.z;
    const int K = KT ? KT : a.K;
    const int KK = K * K;
    float acc[SW_MAXKK];
#pragma unroll
    for (int k = 0; k < SW_MAXKK; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    const int OHW = a.OH * a.OW;
    const int b0 = sp * bps;
    int b1 = b0 + bps;
    if (b1 > a.B) b1 = a.B;
    for (int b = b0; b < b1; ++b) {
        const float* xp = a.x + ((size_t)b * a.cin + ci) * a.H * a.W;
        const size_t ob = ((size_t)b * a.cout + co) * OHW;
        for (int p = threadIdx.x; p < OHW; p += 256) {
            const int oy = p / a.OW, ox = p - oy * a.OW;
            float g = dy[ob + p];
            if (a.gate_act) g *= scm_act_grad_from_out(a.gate[ob + p], a.gate_act);
            bsum += g;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
#pragma unroll
            for (int k = 0; k < SW_MAXKK; ++k) {
                if (k < KK) {
                    const int ky = k / K, kx = k - ky * K;
                    const int iy = iy0 + ky, ix = ix0 + kx;
                    const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
                    const int ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
                    const float xv = xp[iyc * a.W + ixc];
                    acc[k] = fmaf(g, ok ? xv : 0.f, acc[k]);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wsp = ws + (((size_t)sp * a.cout + co) * a.cin + ci) * KK;
#pragma unroll
    for (int k = 0; k < SW_MAXKK; ++k) {
        if (k < KK) {
            float v = wave_sum(acc[k]);
            __syncthreads();
            if (lane == 0) red[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) wsp[k] = red[0] + red[1] + red[2] + red[3];
        }
    }
    if (ws_db != nullptr && ci == 0) {
        float v = wave_sum(bsum);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) ws_db[(size_t)sp * a.cout + co] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void smallconv_wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ ws_db,
                                              int nsplit, int nw, int cout, int accumulate, float* __restrict__ dw,
                                              float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += ws[(size_t)k * nw + i];
        dw[i] = accumulate ? dw[i] + s : s;
    }
    if (db != nullptr && i < cout) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += ws_db[(size_t)k * cout + i];
        db[i] = accumulate ? db[i] + s : s;
    }
}

static int smallconv_splits(int B, int cout, int cin) {
    int ns = 4096 / (cout * cin);
    if (ns < 1) ns = 1;
    if (ns > B) ns = B;
    return ns;
}

extern "C" size_t scm_conv2d_bwd_workspace(int B, int cin, int cout, int K) {
    if (B <= 0 || cin <= 0 || cout <= 0 || K <= 0) return 0;
    const int ns = smallconv_splits(B, cout, cin);
    return (size_t)ns * ((size_t)cout * cin * K * K + cout);
}

static int fill_args(const scm_conv2d_desc* d, SmallConvArgs* a, const char* who) {
    SCM_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SCM_REQUIRE(d->B > 0 && d->cin > 0 && d->cout > 0 && d->H > 0 && d->W > 0, "%s: bad shape", who);
    SCM_REQUIRE(d->K >= 1 && d->K * d->K <= SW_MAXKK && d->stride >= 1 && d->pad > -d->K, "%s: bad K/stride/pad", who);
    const int OH = (d->H + 2 * d->pad - d->K) / d->stride + 1, OW = (d->W + 2 * d->pad - d->K) / d->stride + 1;
    SCM_REQUIRE(OH > 0 && OW > 0, "%s: empty output", who);
    SCM_REQUIRE(d->act >= 0 && d->act <= 2 && d->gate_act >= 0 && d->gate_act <= 2, "%s: bad act", who);
    a->x = d->x; a->w = d->w; a->bias = d->bias; a->scale = d->scale; a->y = d->y;
    a->gate = d->gate; a->gate_act = d->gate_act;
    a->B = d->B; a->cin = d->cin; a->H = d->H; a->W = d->W; a->cout = d->cout; a->OH = OH; a->OW = OW;
    a->K = d->K; a->stride = d->stride; a->pad = d->pad; a->act = d->act;
    return SCM_OK;
}

extern "C" int scm_conv2d_fwd(const scm_conv2d_desc* d, void* stream) {
    SmallConvArgs a;
    int rc = fill_args(d, &a, "scm_conv2d_fwd");
    if (rc) return rc;
    SCM_REQUIRE(d->x && d->w && d->y, "scm_conv2d_fwd: x, w, y required");
    dim3 grid((a.OH * a.OW + 255) / 256, (a.cout + CB - 1) / CB, a.B);
    if (a.K == 3) hipLaunchKernelGGL(smallconv_fwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(smallconv_fwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
    SCM_LAUNCH_CHECK("smallconv_fwd_kernel");
    return SCM_OK;
}

extern "C" int scm_conv2d_bwd(const scm_conv2d_desc* d, const float* dy, float* dx, float* dw, float* db,
                              int accumulate, float* ws, void* stream) {
    SmallConvArgs a;
    int rc = fill_args(d, &a, "scm_conv2d_bwd");
    if (rc) return rc;
    SCM_REQUIRE(dy && d->w, "scm_conv2d_bwd: dy and w required");
    SCM_REQUIRE(!d->gate_act || d->gate, "scm_conv2d_bwd: gate tensor required");
    if (dx) {
        dim3 grid((a.H * a.W + 255) / 256, (a.cin + CB - 1) / CB, a.B);
        if (a.K == 3) hipLaunchKernelGGL(smallconv_dgrad_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a, dy, dx);
        else hipLaunchKernelGGL(smallconv_dgrad_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a, dy, dx);
        SCM_LAUNCH_CHECK("smallconv_dgrad_kernel");
    }
    if (dw) {
        SCM_REQUIRE(d->x && ws, "scm_conv2d_bwd: x and the workspace are required for dw");
        const int ns = smallconv_splits(a.B, a.cout, a.cin);
        const int bps = (a.B + ns - 1) / ns;
        const int nw = a.cout * a.cin * a.K * a.K;
        float* ws_db = ws + (size_t)ns * nw;
        dim3 grid(a.cout, a.cin, ns);
        if (a.K == 3)
            hipLaunchKernelGGL(smallconv_wgrad_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a, dy, ws,
                               db ? ws_db : nullptr, bps);
        else
            hipLaunchKernelGGL(smallconv_wgrad_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a, dy, ws,
                               db ? ws_db : nullptr, bps);
        SCM_LAUNCH_CHECK("smallconv_wgrad_kernel");
        hipLaunchKernelGGL(smallconv_wgrad_reduce_kernel, dim3((nw + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws,
                           ws_db, ns, nw, a.cout, accumulate, dw, db);
        SCM_LAUNCH_CHECK("smallconv_wgrad_reduce_kernel");
    }
    return SCM_OK;
}
