// Bandwidth-bound pieces of the path: latent sampling, decoder factor-sum,
// pixel losses, reward-head softmax tail, action-plane bias, small linear layer,
// coordinate planes.  Each kernel cites the reference lines it replaces.
#include "common.h"

// ----------------------------------------------------------------- Philox
// Philox4x32-10 counter RNG (Salmon et al. 2011), used when the caller does not
// inject the uniform noise for the Bernoulli latent sampler.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// z' = DifferentiableBernoulliSampler / eval threshold, models.py:30-40,107-113.
//   mode 0: (p > 0.5)      mode 1: (noise < p), noise given      mode 2: (philox < p)
__global__ void bernoulli_kernel(const float* __restrict__ p, const float* __restrict__ noise, float* __restrict__ out,
                                 size_t n, int mode, uint64_t seed, uint64_t offset,
                                 const uint64_t* __restrict__ offset_dev) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    float u[4] = {0.5f, 0.5f, 0.5f, 0.5f};
    if (mode == 2) {
        const uint64_t ctr = offset + (offset_dev ? *offset_dev : 0) + i4 / 4;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = u01(c[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = i4 + k;
        if (i < n) {
            if (mode == 1) u[k] = noise[i];
            const float pv = p[i];
            out[i] = (mode == 0) ? (pv > 0.5f ? 1.f : 0.f) : (u[k] < pv ? 1.f : 0.f);
        }
    }
}

extern "C" int scm_bernoulli(const float* p, const float* noise, float* out, size_t n, int mode, uint64_t seed,
                             uint64_t offset, const uint64_t* offset_dev, void* stream) {
    SCM_REQUIRE(p && out && n > 0 && mode >= 0 && mode <= 2, "scm_bernoulli: bad arguments");
    SCM_REQUIRE(mode != 1 || noise, "scm_bernoulli: mode 1 needs the noise tensor");
    const size_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(bernoulli_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                       noise, out, n, mode, seed, offset, offset_dev);
    SCM_LAUNCH_CHECK("bernoulli_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------- factor sum
// Decoder: x.view(B, L, C, H, W).sum(dim=1), models.py:280-287; backward is a
// broadcast of dy over the L factors.
__global__ void factor_sum_kernel(const float* __restrict__ x, float* __restrict__ y, int L, int C, int HW,
                                  size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*HW
    if (i >= total) return;
    const size_t chw = (size_t)C * HW;
    const size_t b = i / chw, r = i - b * chw;
    const float* xp = x + b * L * chw + r;
    float s = 0.f;
    for (int f = 0; f < L; ++f) s += xp[(size_t)f * chw];
    y[i] = s;
}
__global__ void factor_bcast_kernel(const float* __restrict__ dy, float* __restrict__ dx, int L, int C, int HW,
                                    size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*L*C*HW
    if (i >= total) return;
    const size_t chw = (size_t)C * HW;
    const size_t b = i / (chw * L), r = i % chw;
    dx[i] = dy[b * chw + r];
}
extern "C" int scm_factor_sum(const float* x, float* y, int B, int L, int C, int HW, void* stream) {
    SCM_REQUIRE(x && y && B > 0 && L > 0 && C > 0 && HW > 0, "scm_factor_sum: bad arguments");
    const size_t total = (size_t)B * C * HW;
    hipLaunchKernelGGL(factor_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       y, L, C, HW, total);
    SCM_LAUNCH_CHECK("factor_sum_kernel");
    return SCM_OK;
}
extern "C" int scm_factor_bcast(const float* dy, float* dx, int B, int L, int C, int HW, void* stream) {
    SCM_REQUIRE(dy && dx && B > 0 && L > 0 && C > 0 && HW > 0, "scm_factor_bcast: bad arguments");
    const size_t total = (size_t)B * L * C * HW;
    hipLaunchKernelGGL(factor_bcast_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, dx, L, C, HW, total);
    SCM_LAUNCH_CHECK("factor_bcast_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------ pixel losses
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// per_sample[b] = mean_{c,h,w} loss(sigmoid(logit), target)
//   kind 0: binary cross entropy with torch's log clamp at -100
//           (torch.sigmoid + decoder_pixel_loss, main.py:189-190,310-312)
//   kind 1: squared error (measure_prediction_mse, main.py:814-816)
// prob (optional) receives sigmoid(logit).
__global__ void __launch_bounds__(1024) pixel_loss_kernel(const float* __restrict__ logit,
                                                          const float* __restrict__ target, int n, int kind,
                                                          float* __restrict__ per_sample, float* __restrict__ prob) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const float* lp = logit + (size_t)b * n;
    const float* tp = target + (size_t)b * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float p = 1.f / (1.f + expf(-lp[i]));
        const float t = tp[i];
        if (prob) prob[(size_t)b * n + i] = p;
        if (kind == 0) {
            const float l1 = fmaxf(logf(p), -100.f), l0 = fmaxf(log1pf(-p), -100.f);
            s -= t * l1 + (1.f - t) * l0;
        } else {
            const float d = t - p;
            s = fmaf(d, d, s);
        }
    }
    // 16 waves: wave sums meet in LDS, summed in a fixed order
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        per_sample[b] = t / (float)n;
    }
}
// dlogit = g[b]/n * dloss/dp * p(1-p); BCE: torch's binary_cross_entropy_backward
// (p - t) / max((1-p) p, 1e-12); MSE: 2 (p - t)
__global__ void pixel_loss_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ target,
                                      const float* __restrict__ g, int n, int kind, size_t total,
                                      float* __restrict__ dlogit) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / n;
    const float p = 1.f / (1.f + expf(-logit[i]));
    const float t = target[i];
    const float pq = p * (1.f - p);
    const float dp = kind == 0 ? (p - t) / fmaxf(pq, 1e-12f) : 2.f * (p - t);
    dlogit[i] = g[b] / (float)n * dp * pq;
}
extern "C" int scm_pixel_loss(const float* logit, const float* target, int B, int n, int kind, float* per_sample,
                              float* prob, void* stream) {
    SCM_REQUIRE(logit && target && per_sample && B > 0 && n > 0 && (kind == 0 || kind == 1), "scm_pixel_loss: bad arguments");
    hipLaunchKernelGGL(pixel_loss_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logit, target, n, kind, per_sample,
                       prob);
    SCM_LAUNCH_CHECK("pixel_loss_kernel");
    return SCM_OK;
}
extern "C" int scm_pixel_loss_bwd(const float* logit, const float* target, const float* g, int B, int n, int kind,
                                  float* dlogit, void* stream) {
    SCM_REQUIRE(logit && target && g && dlogit && B > 0 && n > 0 && (kind == 0 || kind == 1), "scm_pixel_loss_bwd: bad arguments");
    const size_t total = (size_t)B * n;
    hipLaunchKernelGGL(pixel_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logit, target, g, n, kind, total, dlogit);
    SCM_LAUNCH_CHECK("pixel_loss_bwd_kernel");
    return SCM_OK;
}

// ------------------------------------------------------- reward-head tail
// RewardPredictor tail, models.py:241-250: x.view(B,3,R,h,w) -> softmax over the
// 3 classes -> p0 - p2 -> sum over h,w.  One workgroup per (b, r).
__global__ void __launch_bounds__(256) reward_tail_kernel(const float* __restrict__ x, int R, int hw,
                                                          float* __restrict__ r_out, float* __restrict__ map_out) {
    __shared__ float red[4];
    const int b = blockIdx.x / R, r = blockIdx.x - b * R;
    const float* x0 = x + ((size_t)b * 3 * R + r) * hw;
    const float* x1 = x0 + (size_t)R * hw;
    const float* x2 = x1 + (size_t)R * hw;
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float a0 = x0[i], a1 = x1[i], a2 = x2[i];
        const float m = fmaxf(a0, fmaxf(a1, a2));
        const float e0 = expf(a0 - m), e1 = expf(a1 - m), e2 = expf(a2 - m);
        const float v = (e0 - e2) / (e0 + e1 + e2);
        if (map_out) map_out[((size_t)b * R + r) * hw + i] = v;
        s += v;
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) r_out[(size_t)b * R + r] = s;
}
// y = p0 - p2;  dx_k = (g_r[b,r] + g_map[b,r,i]) * p_k * (s_k - y),  s = (1, 0, -1)
__global__ void reward_tail_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g_r,
                                       const float* __restrict__ g_map, int R, int hw, size_t total,
                                       float* __restrict__ dx) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*R*hw
    if (idx >= total) return;
    const size_t br = idx / hw;
    const int i = (int)(idx - br * hw);
    const size_t b = br / R;
    const int r = (int)(br - b * R);
    const size_t o0 = ((size_t)b * 3 * R + r) * hw + i, o1 = o0 + (size_t)R * hw, o2 = o1 + (size_t)R * hw;
    const float a0 = x[o0], a1 = x[o1], a2 = x[o2];
    const float m = fmaxf(a0, fmaxf(a1, a2));
    const float e0 = expf(a0 - m), e1 = expf(a1 - m), e2 = expf(a2 - m);
    const float inv = 1.f / (e0 + e1 + e2);
    const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
    const float y = p0 - p2;
    float g = 0.f;
    if (g_r) g += g_r[br];
    if (g_map) g += g_map[idx];
    dx[o0] = g * p0 * (1.f - y);
    dx[o1] = g * p1 * (0.f - y);
    dx[o2] = g * p2 * (-1.f - y);
}
extern "C" int scm_reward_tail(const float* x, int B, int R, int hw, float* r_out, float* map_out, void* stream) {
    SCM_REQUIRE(x && r_out && B > 0 && R > 0 && hw > 0, "scm_reward_tail: bad arguments");
    hipLaunchKernelGGL(reward_tail_kernel, dim3(B * R), dim3(256), 0, (hipStream_t)stream, x, R, hw, r_out, map_out);
    SCM_LAUNCH_CHECK("reward_tail_kernel");
    return SCM_OK;
}
extern "C" int scm_reward_tail_bwd(const float* x, const float* g_r, const float* g_map, int B, int R, int hw,
                                   float* dx, void* stream) {
    SCM_REQUIRE(x && dx && (g_r || g_map) && B > 0 && R > 0 && hw > 0, "scm_reward_tail_bwd: bad arguments");
    const size_t total = (size_t)B * R * hw;
    hipLaunchKernelGGL(reward_tail_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, g_r, g_map, R, hw, total, dx);
    SCM_LAUNCH_CHECK("reward_tail_bwd_kernel");
    return SCM_OK;
}

// ------------------------------------------------------ action-plane bias
// Transition.forward broadcasts the action vector over the map and concatenates
// it to z (models.py:68-73).  Under circular padding every output pixel sees
// all 9 taps of those constant planes, so their contribution to conv1 is a
// per-sample bias:  sbias[b][co] = scale * sum_a a[b][a] * sum_tap W[co][c0+a][tap].
__global__ void action_bias_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                   const float* __restrict__ scale, int A, int cout, int cin, int c0, int total,
                                   float* __restrict__ sbias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*cout
    if (idx >= total) return;
    const int b = idx / cout, co = idx - b * cout;
    float s = 0.f;
    for (int a = 0; a < A; ++a) {
        const float* wp = w + ((size_t)co * cin + c0 + a) * 9;
        float ws = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) ws += wp[t];
        s = fmaf(act[(size_t)b * A + a], ws, s);
    }
    sbias[idx] = s * (scale ? *scale : 1.f);
}
// csum[b][co] = d sbias.  d_act[b][a] = scale * sum_co csum[b][co] * Wsum[co][a];
// dW_eff[co][c0+a][tap] = sum_b act[b][a] * csum[b][co]  (same value for the 9 taps)
__global__ void __launch_bounds__(64) action_bias_bwd_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ csum, int B, int A, int cout,
                                                             int cin, int c0, float* __restrict__ d_act,
                                                             float* __restrict__ dw_eff) {
    // one wave per output element: blocks [0, B*A) -> d_act, blocks [B*A, B*A + cout*A) -> dw_eff
    const int lane = threadIdx.x;
    const float sc = scale ? *scale : 1.f;
    int idx = blockIdx.x;
    if (idx < B * A) {
        if (!d_act) return;
        const int b = idx / A, a = idx - b * A;
        float s = 0.f;
        for (int co = lane; co < cout; co += 64) {
            const float* wp = w + ((size_t)co * cin + c0 + a) * 9;
            float ws = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) ws += wp[t];
            s = fmaf(csum[(size_t)b * cout + co], ws, s);
        }
        s = wave_sum(s);
        if (lane == 0) d_act[idx] = s * sc;
        return;
    }
    idx -= B * A;
    if (!dw_eff || idx >= cout * A) return;
    const int co = idx / A, a = idx - co * A;
    float s = 0.f;
    for (int b = lane; b < B; b += 64) s = fmaf(act[(size_t)b * A + a], csum[(size_t)b * cout + co], s);
    s = wave_sum(s);
    if (lane < 9) dw_eff[((size_t)co * cin + c0 + a) * 9 + lane] = s;
}
extern "C" int scm_action_bias(const float* act, const float* w, const float* scale, int B, int A, int cout, int cin,
                               int c0, float* sbias, void* stream) {
    SCM_REQUIRE(act && w && sbias && B > 0 && A > 0 && cout > 0 && c0 >= 0 && c0 + A <= cin, "scm_action_bias: bad arguments");
    const int total = B * cout;
    hipLaunchKernelGGL(action_bias_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, act, w, scale,
                       A, cout, cin, c0, total, sbias);
    SCM_LAUNCH_CHECK("action_bias_kernel");
    return SCM_OK;
}
extern "C" int scm_action_bias_bwd(const float* act, const float* w, const float* scale, const float* csum, int B,
                                   int A, int cout, int cin, int c0, float* d_act, float* dw_eff, void* stream) {
    SCM_REQUIRE(act && w && csum && (d_act || dw_eff) && B > 0 && A > 0 && cout > 0 && c0 >= 0 && c0 + A <= cin,
                "scm_action_bias_bwd: bad arguments");
    hipLaunchKernelGGL(action_bias_bwd_kernel, dim3(B * A + cout * A), dim3(64), 0, (hipStream_t)stream, act, w,
                       scale, csum, B, A, cout, cin, c0, d_act, dw_eff);
    SCM_LAUNCH_CHECK("action_bias_bwd_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------ small linear
// y[b][o] = act( sum_i x[b][i] w[o][i] + bias[o] ): Discriminator.fc1 + leaky_relu,
// models.py:205,221-222.  One workgroup per (b, o).
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int nin, int nout, int act,
                                                     float* __restrict__ y) {
    __shared__ float red[4];
    const int b = blockIdx.x / nout, o = blockIdx.x - b * nout;
    float s = 0.f;
    for (int i = threadIdx.x; i < nin; i += 256) s = fmaf(x[(size_t)b * nin + i], w[(size_t)o * nin + i], s);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) y[blockIdx.x] = scm_act(s + (bias ? bias[o] : 0.f), act);
}
// gp = gy * act'(y);  dx[b][i] = sum_o gp[b][o] w[o][i];  dw[o][i] = sum_b gp[b][o] x[b][i];  db[o] = sum_b gp
__global__ void linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ y, const float* __restrict__ gy, int B, int nin, int nout,
                                  int act, float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (dx && idx < B * nin) {
        const int b = idx / nin, i = idx - b * nin;
        float s = 0.f;
        for (int o = 0; o < nout; ++o) {
            const float gp = gy[b * nout + o] * scm_act_grad_from_out(y[b * nout + o], act);
            s = fmaf(gp, w[(size_t)o * nin + i], s);
        }
        dx[idx] = s;
    }
    if (dw && idx < nout * nin) {
        const int o = idx / nin, i = idx - o * nin;
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float gp = gy[b * nout + o] * scm_act_grad_from_out(y[b * nout + o], act);
            s = fmaf(gp, x[(size_t)b * nin + i], s);
        }
        dw[idx] = s;
    }
    if (db && idx < nout) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += gy[b * nout + idx] * scm_act_grad_from_out(y[b * nout + idx], act);
        db[idx] = s;
    }
}
extern "C" int scm_linear(const float* x, const float* w, const float* bias, int B, int nin, int nout, int act,
                          float* y, void* stream) {
    SCM_REQUIRE(x && w && y && B > 0 && nin > 0 && nout > 0, "scm_linear: bad arguments");
    hipLaunchKernelGGL(linear_kernel, dim3(B * nout), dim3(256), 0, (hipStream_t)stream, x, w, bias, nin, nout, act, y);
    SCM_LAUNCH_CHECK("linear_kernel");
    return SCM_OK;
}
extern "C" int scm_linear_bwd(const float* x, const float* w, const float* y, const float* gy, int B, int nin, int nout,
                              int act, float* dx, float* dw, float* db, void* stream) {
    SCM_REQUIRE(x && w && y && gy && B > 0 && nin > 0 && nout > 0, "scm_linear_bwd: bad arguments");
    int total = B * nin;
    if (nout * nin > total) total = nout * nin;
    hipLaunchKernelGGL(linear_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, y, gy, B,
                       nin, nout, act, dx, dw, db);
    SCM_LAUNCH_CHECK("linear_bwd_kernel");
    return SCM_OK;
}

// ------------------------------------------------------- coordinate planes
// CoordConv2d, coordconv.py:12-14: out = cat([x, coord_x, coord_y], 1) with
// coord_x[i][j] = -1 + 2j/W, coord_y[i][j] = -1 + 2i/H (half-open arange).
__global__ void coord_cat_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                 size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*(C+2)*H*W
    if (i >= total) return;
    const int HW = H * W;
    const size_t bc = i / HW;
    const int p = (int)(i - bc * HW);
    const size_t b = bc / (C + 2);
    const int c = (int)(bc - b * (C + 2));
    float v;
    if (c < C) v = x[(b * C + c) * HW + p];
    else if (c == C) v = (float)(-1.0 + (double)(p % W) * (2.0 / (double)W));   // torch.arange evaluates in double
    else v = (float)(-1.0 + (double)(p / W) * (2.0 / (double)H));
    out[i] = v;
}
extern "C" int scm_coord_cat(const float* x, float* out, int B, int C, int H, int W, void* stream) {
    SCM_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "scm_coord_cat: bad arguments");
    const size_t total = (size_t)B * (C + 2) * H * W;
    hipLaunchKernelGGL(coord_cat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, C, H, W, total);
    SCM_LAUNCH_CHECK("coord_cat_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------ per-step loss
// The loss bookkeeping of one unrolled step, main.py:178-197, in one launch:
//   mask_out[b] = mask_in[b] * (1 - done[b])
//   rd   = mean_b( mean_r((rhat - r)^2) * mask_out )          ("Rd Loss t=..")
//   rec  = mean_b( bce[b] * mask_out )                        ("Reconstruction t=..")
//   loss_out = loss_in + coef * rd + rec                      (coef = theta * REWARD_COEF)
// r and done are column t of (B, T, R) / (B, T) tensors: element strides are passed.
__global__ void __launch_bounds__(256) step_loss_kernel(const float* __restrict__ rhat, const float* __restrict__ r,
                                                        long r_stride, const float* __restrict__ done,
                                                        long done_stride, const float* __restrict__ bce,
                                                        const float* __restrict__ mask_in, const float* loss_in,
                                                        int B, int R, float coef_v, const float* coef_dev,
                                                        float* __restrict__ mask_out,
                                                        float* __restrict__ loss_out, float* __restrict__ rd_rec) {
    __shared__ float red[4];
    const float coef = coef_dev ? *coef_dev : coef_v;
    float srd = 0.f, srec = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float m = mask_in[b] * (1.f - done[(size_t)b * done_stride]);
        mask_out[b] = m;
        float s = 0.f;
        for (int k = 0; k < R; ++k) {
            const float d = rhat[(size_t)b * R + k] - r[(size_t)b * r_stride + k];
            s = fmaf(d, d, s);
        }
        srd += s / (float)R * m;
        srec += bce[b] * m;
    }
    srd = block_sum_256(srd, red);
    srec = block_sum_256(srec, red);
    if (threadIdx.x == 0) {
        const float rd = srd / (float)B, rec = srec / (float)B;
        loss_out[0] = (loss_in ? *loss_in : 0.f) + coef * rd + rec;
        rd_rec[0] = rd;
        rd_rec[1] = rec;
    }
}
// d rhat[b][k] = g * coef * 2 (rhat - r) / (R B) * mask;  d bce[b] = g * mask / B
__global__ void step_loss_bwd_kernel(const float* __restrict__ rhat, const float* __restrict__ r, long r_stride,
                                     const float* __restrict__ mask, const float* __restrict__ g, int B, int R,
                                     float coef_v, const float* coef_dev, float* __restrict__ d_rhat,
                                     float* __restrict__ d_bce) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * R) return;
    const float coef = coef_dev ? *coef_dev : coef_v;
    const int b = idx / R, k = idx - b * R;
    const float gg = *g, m = mask[b];
    d_rhat[idx] = gg * coef * 2.f * (rhat[idx] - r[(size_t)b * r_stride + k]) / (float)(R * B) * m;
    if (k == 0) d_bce[b] = gg * m / (float)B;
}
extern "C" int scm_step_loss(const float* rhat, const float* r, long r_stride, const float* done, long done_stride,
                             const float* bce, const float* mask_in, const float* loss_in, int B, int R, float coef,
                             const float* coef_dev, float* mask_out, float* loss_out, float* rd_rec, void* stream) {
    SCM_REQUIRE(rhat && r && done && bce && mask_in && mask_out && loss_out && rd_rec && B > 0 && R > 0,
                "scm_step_loss: bad arguments");
    hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rhat, r, r_stride, done,
                       done_stride, bce, mask_in, loss_in, B, R, coef, coef_dev, mask_out, loss_out, rd_rec);
    SCM_LAUNCH_CHECK("step_loss_kernel");
    return SCM_OK;
}
extern "C" int scm_step_loss_bwd(const float* rhat, const float* r, long r_stride, const float* mask, const float* g,
                                 int B, int R, float coef, const float* coef_dev, float* d_rhat, float* d_bce,
                                 void* stream) {
    SCM_REQUIRE(rhat && r && mask && g && d_rhat && d_bce && B > 0 && R > 0, "scm_step_loss_bwd: bad arguments");
    hipLaunchKernelGGL(step_loss_bwd_kernel, dim3((B * R + 255) / 256), dim3(256), 0, (hipStream_t)stream, rhat, r,
                       r_stride, mask, g, B, R, coef, coef_dev, d_rhat, d_bce);
    SCM_LAUNCH_CHECK("step_loss_bwd_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------------ one-hot
// onehot(a_idx, num_actions) of main.py:447-452 on the device: out[i][a] = (idx[i] == a).
__global__ void onehot_kernel(const long long* __restrict__ idx, long n, int A, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * A) {
        const long row = i / A;
        const int a = (int)(i - row * A);
        out[i] = idx[row] == a ? 1.f : 0.f;
    }
}

extern "C" int scm_onehot(const long long* idx, long n, int num_actions, float* out, void* stream) {
    SCM_REQUIRE(idx && out && n > 0 && num_actions > 0, "scm_onehot: bad arguments");
    const long total = n * num_actions;
    hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n,
                       num_actions, out);
    SCM_LAUNCH_CHECK("onehot_kernel");
    return SCM_OK;
}
