// Spectral normalisation: power iteration and its gradient fold.
// Replaces spectral_normalization.py:23-35 (SpectralNorm._update_u_v) and the
// autograd backward through sigma = u . (W_bar v), W = W_bar / sigma.
//
// One workgroup of 1024 threads per layer (the matrices are 32..128 rows by
// 27..2304 columns, 3 KB .. 1.2 MB): the two mat-vecs are latency-bound, not
// bandwidth-bound, so the layer's whole iteration -- v <- normalise(W^T u),
// u <- normalise(W v), sigma = u . (W v) -- runs in ONE launch, with
// wavefront-shuffle (DPP) reductions; several layers share a launch
// (blockIdx.x = layer).  u and v are updated in place in caller memory, as the
// reference does with u.data / v.data.
#include "common.h"

#define SN_THREADS 1024
#define SN_MAX_LAYERS 8

struct SnLayer {
    const float* w; float* u; float* v; float* sigma; float* inv_sigma;
    int rows; int cols;
};
struct SnBatch {
    SnLayer l[SN_MAX_LAYERS];
    int steps;      // power-iteration calls to run back to back (>=1)
};

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    // red: 16 floats of LDS
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = (lane < (SN_THREADS / 64)) ? red[lane] : 0.f;
    t = wave_sum(t);
    return t;
}

__global__ void __launch_bounds__(SN_THREADS) sn_power_iter_kernel(const SnBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SnLayer L = batch.l[blockIdx.x];
    const int rows = L.rows, cols = L.cols;
    float* us = smem;              // [rows]
    float* vs = smem + rows;       // [cols]
    float* ss = vs + cols;         // [rows]  W v
    float* red = ss + rows;        // [16]  (then the partial vectors of the float4 path: [ng][cols])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < rows; i += SN_THREADS) us[i] = L.u[i];
    __syncthreads();
    // Rows of whole 16-byte groups (every 3x3 layer with cin % 4 == 0: 1152, 2304, 288 columns): both mat-vecs read W
    // as float4.  W^T u: thread (column group cq, row group rg) sums rows rg, rg + ng, ... of its four columns -- 16-byte
    // lanes, consecutive threads on consecutive addresses, many independent loads in flight per thread -- and the ng
    // partial vectors meet in LDS.  (Before: one column per thread, 4-byte loads, one row after the other: 70-200 us
    // per iteration of a 128 x 1152 layer, a third of a batch-8 training iteration.)
    const bool vec4 = (cols & 3) == 0 && (rows & 3) == 0 && ((reinterpret_cast<size_t>(L.w) & 15) == 0);
    const int c4 = cols >> 2;
    int ng = vec4 ? SN_THREADS / c4 : 1;           // row groups (1 when a row has 1024 or more column groups)
    ng = ng < 1 ? 1 : (ng > rows ? rows : ng);
    float* pv = red + 16;          // [ng][cols] partial W^T u (vec4 path)
    for (int step = 0; step < batch.steps; ++step) {
        float part = 0.f;
        if (vec4) {
            const f32x4* w4 = reinterpret_cast<const f32x4*>(L.w);
            const int rg = c4 >= SN_THREADS ? 0 : tid / c4;
            for (int cq = c4 >= SN_THREADS ? tid : (rg < ng ? tid - rg * c4 : c4); cq < c4; cq += SN_THREADS) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int r = rg; r < rows; r += ng) {
                    const f32x4 w = w4[(size_t)r * c4 + cq];
                    const float ur = us[r];
                    acc[0] = fmaf(w[0], ur, acc[0]); acc[1] = fmaf(w[1], ur, acc[1]);
                    acc[2] = fmaf(w[2], ur, acc[2]); acc[3] = fmaf(w[3], ur, acc[3]);
                }
                *reinterpret_cast<f32x4*>(pv + (size_t)rg * cols + 4 * cq) = acc;
            }
            __syncthreads();
            for (int j = tid; j < cols; j += SN_THREADS) {
                float acc = 0.f;
                for (int g = 0; g < ng; ++g) acc += pv[(size_t)g * cols + j];
                vs[j] = acc;
                part = fmaf(acc, acc, part);
            }
        } else {
            // ---- v = l2normalize(W^T u): thread j owns columns j, j+1024, ...
            for (int j = tid; j < cols; j += SN_THREADS) {
                float acc = 0.f;
                for (int r = 0; r < rows; ++r) acc = fmaf(L.w[(size_t)r * cols + j], us[r], acc);
                vs[j] = acc;
                part = fmaf(acc, acc, part);
            }
        }
        float nrm = sqrtf(block_sum_1024(part, red));
        float inv = 1.f / (nrm + 1e-12f);
        for (int j = tid; j < cols; j += SN_THREADS) vs[j] *= inv;
        __syncthreads();
        // ---- s = W v: one wave per row, lanes stride the columns (16-byte lanes, four rows in flight, where rows allow)
        if (vec4) {
            const f32x4* w4 = reinterpret_cast<const f32x4*>(L.w);
            const f32x4* v4 = reinterpret_cast<const f32x4*>(vs);
            constexpr int NW = SN_THREADS / 64;
            for (int r0 = wave; r0 < rows; r0 += 4 * NW) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int j = lane; j < c4; j += 64) {
                    const f32x4 v = v4[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = r0 + k * NW;
                        if (r < rows) {
                            const f32x4 w = w4[(size_t)r * c4 + j];
                            acc[k] = fmaf(w[0], v[0], fmaf(w[1], v[1], fmaf(w[2], v[2], fmaf(w[3], v[3], acc[k]))));
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = wave_sum(acc[k]);
                    if (lane == 0 && r0 + k * NW < rows) ss[r0 + k * NW] = t;
                }
            }
        } else {
            for (int r = wave; r < rows; r += SN_THREADS / 64) {
                float acc = 0.f;
                for (int j = lane; j < cols; j += 64) acc = fmaf(L.w[(size_t)r * cols + j], vs[j], acc);
                acc = wave_sum(acc);
                if (lane == 0) ss[r] = acc;
            }
        }
        __syncthreads();
        part = 0.f;
        for (int r = tid; r < rows; r += SN_THREADS) part = fmaf(ss[r], ss[r], part);
        nrm = sqrtf(block_sum_1024(part, red));
        inv = 1.f / (nrm + 1e-12f);
        // ---- u = s / (|s| + eps); sigma = u . s
        part = 0.f;
        for (int r = tid; r < rows; r += SN_THREADS) {
            const float un = ss[r] * inv;
            us[r] = un;
            part = fmaf(un, ss[r], part);
        }
        const float sigma = block_sum_1024(part, red);
        if (tid == 0) {
            if (L.sigma) L.sigma[step] = sigma;
            if (L.inv_sigma) L.inv_sigma[step] = 1.f / sigma;
        }
        __syncthreads();
    }
    for (int i = tid; i < rows; i += SN_THREADS) L.u[i] = us[i];
    for (int j = tid; j < cols; j += SN_THREADS) L.v[j] = vs[j];
}

extern "C" int scm_sn_power_iter(int nlayers, const scm_sn_layer* layers, int steps, void* stream) {
    SCM_REQUIRE(nlayers > 0 && nlayers <= SN_MAX_LAYERS && layers, "scm_sn_power_iter: 1..%d layers", SN_MAX_LAYERS);
    SCM_REQUIRE(steps >= 1, "scm_sn_power_iter: steps >= 1");
    SnBatch b;
    size_t lds = 0;
    for (int i = 0; i < nlayers; ++i) {
        const scm_sn_layer& s = layers[i];
        SCM_REQUIRE(s.w_bar && s.u && s.v && s.rows > 0 && s.cols > 0, "scm_sn_power_iter: layer %d invalid", i);
        b.l[i].w = s.w_bar; b.l[i].u = s.u; b.l[i].v = s.v;
        b.l[i].sigma = s.sigma; b.l[i].inv_sigma = s.inv_sigma;
        b.l[i].rows = s.rows; b.l[i].cols = s.cols;
        // (us, ss, vs, red, + the partial W^T u vectors of the float4 path: at most 1024 / (cols / 4) row groups)
        size_t ng = (s.cols & 3) == 0 ? (size_t)(SN_THREADS / (s.cols / 4) > 0 ? SN_THREADS / (s.cols / 4) : 1) : 0;
        if (ng > (size_t)s.rows) ng = s.rows;
        const size_t need = ((size_t)2 * s.rows + s.cols + 16 + ng * s.cols + 4) * sizeof(float);
        if (need > lds) lds = need;
    }
    SCM_REQUIRE(lds <= 64 * 1024, "scm_sn_power_iter: matrix too large for the LDS vectors");
    b.steps = steps;
    hipLaunchKernelGGL(sn_power_iter_kernel, dim3(nlayers), dim3(SN_THREADS), lds, (hipStream_t)stream, b);
    SCM_LAUNCH_CHECK("sn_power_iter_kernel");
    return SCM_OK;
}

// ---------------------------------------------------------------- backward
// Given g = dL/dW_eff (gradient w.r.t. the normalised weight W_bar/sigma):
//   dwbar (=|+=) g * inv_sigma - (<g, W_bar> * inv_sigma^2) * u v^T
// u, v are read NOW (the reference's saved u/v alias the live parameters, so
// its backward sees the latest power-iteration state; oracle sn_weight()).
__global__ void __launch_bounds__(SN_THREADS) sn_fold_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ wbar,
                                                             const float* __restrict__ u,
                                                             const float* __restrict__ v,
                                                             const float* __restrict__ inv_sigma, int rows, int cols,
                                                             int accumulate, float* __restrict__ dwbar) {
    __shared__ float red[16];
    const int n = rows * cols;
    float part = 0.f;
    for (int i = threadIdx.x; i < n; i += SN_THREADS) part = fmaf(g[i], wbar[i], part);
    const float dot = block_sum_1024(part, red);
    const float is = *inv_sigma;
    const float c = dot * is * is;
    // (every workgroup of the launch computed the same dot, in the same order; each applies the fold to its slice)
    for (int i = blockIdx.x * SN_THREADS + threadIdx.x; i < n; i += gridDim.x * SN_THREADS) {
        const int r = i / cols, j = i - r * cols;
        const float val = g[i] * is - c * u[r] * v[j];
        dwbar[i] = accumulate ? dwbar[i] + val : val;
    }
}

// Same fold with <g, W_bar> given as ndot partial sums (wgrad_reduce_kernel): every
// workgroup re-reduces the (<= few thousand) partials in the same fixed order, then
// handles its slice of the matrix -- no single-workgroup pass over the weights.
__global__ void __launch_bounds__(256) sn_fold_parts_kernel(const float* __restrict__ g,
                                                            const float* __restrict__ u, const float* __restrict__ v,
                                                            const float* __restrict__ inv_sigma,
                                                            const float* __restrict__ dot_part, int ndot, int rows,
                                                            int cols, int accumulate, float* __restrict__ dwbar) {
    __shared__ float red[4];
    float part = 0.f;
    for (int i = threadIdx.x; i < ndot; i += 256) part += dot_part[i];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    const float is = *inv_sigma;
    const float c = dot * is * is;
    const int n = rows * cols;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int r = i / cols, j = i - r * cols;
        const float val = g[i] * is - c * u[r] * v[j];
        dwbar[i] = accumulate ? dwbar[i] + val : val;
    }
}

extern "C" int scm_sn_fold_grad(const float* g, const float* w_bar, const float* u, const float* v,
                                const float* inv_sigma, const float* dot_part, int ndot, int rows, int cols,
                                int accumulate, float* dw_bar, void* stream) {
    SCM_REQUIRE(g && w_bar && u && v && inv_sigma && dw_bar && rows > 0 && cols > 0, "scm_sn_fold_grad: bad arguments");
    if (dot_part != nullptr && ndot > 0) {
        int blocks = (rows * cols + 256 * 4 - 1) / (256 * 4);
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(sn_fold_parts_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, u, v, inv_sigma,
                           dot_part, ndot, rows, cols, accumulate, dw_bar);
        SCM_LAUNCH_CHECK("sn_fold_parts_kernel");
        return SCM_OK;
    }
    // (up to 16 workgroups: each re-reads g and W_bar for the dot -- they sit in L2 -- and writes 1/16 of the matrix)
    int nb = (rows * cols + 4 * SN_THREADS - 1) / (4 * SN_THREADS);
    nb = nb < 1 ? 1 : (nb > 16 ? 16 : nb);
    if (dw_bar == g || dw_bar == w_bar) nb = 1;       // (in place: one workgroup, which finishes the dot before it writes)
    hipLaunchKernelGGL(sn_fold_kernel, dim3(nb), dim3(SN_THREADS), 0, (hipStream_t)stream, g, w_bar, u, v, inv_sigma,
                       rows, cols, accumulate, dw_bar);
    SCM_LAUNCH_CHECK("sn_fold_kernel");
    return SCM_OK;
}
