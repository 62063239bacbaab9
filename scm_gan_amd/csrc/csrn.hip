// CSRN (convolutional spatial-recurrent network) directional sweep, forward and
// BPTT backward.  Replaces spatial_recurrent.py:55-114: per image row (or column)
// a one-step bias-free GRU over the row's tokens, whose output is the context row
// and, through Conv1d(k=3, pad 1) + tanh, the hidden state of the next row.
//
// The recurrence is strictly sequential in the sweep direction (H or W dependent
// steps), so one workgroup owns one image of one sweep and walks it with
// __syncthreads() between the GRU phase and the conv phase; the token x channel
// plane of the row lives in LDS.  Parallelism = batch x 4 directions x (tokens x
// channels) per step: this operator is latency-bound by construction (SURVEY.md
// section 7, "hard parts").  The reference never instantiates CSRN on the timed
// path; it ships as an operator with parity tests.
#include "common.h"

struct CsrnArgs {
    const float* x;        // [B][ch][H][W]
    float* ctx;            // [B][ch][H][W]   GRU output of every step
    float* hprev;          // [steps][B][ntok][ch] hidden state fed to every step (saved for backward)
    const float* w_ih;     // [3ch][ch]  gates r, z, n
    const float* w_hh;     // [3ch][ch]
    const float* wc;       // [ch][ch][3]
    const float* bc;       // [ch]
    int B, ch, H, W;
    int along_rows;        // 1: steps over rows, tokens = columns; 0: steps over columns, tokens = rows
    int reverse;           // sweep from the last row/column
};

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ int csrn_pos(const CsrnArgs& a, int step, int tok) {
    return a.along_rows ? step * a.W + tok : tok * a.W + step;
}

__global__ void __launch_bounds__(256) csrn_fwd_kernel(const CsrnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int ch = a.ch, HW = a.H * a.W;
    const int ntok = a.along_rows ? a.W : a.H;
    const int nsteps = a.along_rows ? a.H : a.W;
    float* h = smem;                 // [ntok][ch] state entering the step
    float* hn = smem + ntok * ch;    // [ntok][ch] GRU output
    for (int i = threadIdx.x; i < ntok * ch; i += 256) h[i] = 0.f;
    __syncthreads();
    for (int si = 0; si < nsteps; ++si) {
        const int s = a.reverse ? nsteps - 1 - si : si;
        for (int idx = threadIdx.x; idx < ntok * ch; idx += 256) {
            const int k = idx / ch, c = idx - k * ch;
            const float* xp = a.x + (size_t)b * ch * HW + csrn_pos(a, s, k);
            float ir = 0.f, iz = 0.f, in_ = 0.f, hr = 0.f, hz = 0.f, hnn = 0.f;
            for (int j = 0; j < ch; ++j) {
                const float xv = xp[(size_t)j * HW];
                const float hv = h[k * ch + j];
                ir = fmaf(a.w_ih[(size_t)c * ch + j], xv, ir);
                iz = fmaf(a.w_ih[(size_t)(ch + c) * ch + j], xv, iz);
                in_ = fmaf(a.w_ih[(size_t)(2 * ch + c) * ch + j], xv, in_);
                hr = fmaf(a.w_hh[(size_t)c * ch + j], hv, hr);
                hz = fmaf(a.w_hh[(size_t)(ch + c) * ch + j], hv, hz);
                hnn = fmaf(a.w_hh[(size_t)(2 * ch + c) * ch + j], hv, hnn);
            }
            const float r = sigm(ir + hr), z = sigm(iz + hz);
            const float n = tanhf(in_ + r * hnn);
            const float hp = h[idx];
            const float out = (1.f - z) * n + z * hp;
            hn[idx] = out;
            a.ctx[((size_t)b * ch + c) * HW + csrn_pos(a, s, k)] = out;
            a.hprev[(((size_t)si * a.B + b) * ntok + k) * ch + c] = hp;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < ntok * ch; idx += 256) {
            const int k = idx / ch, co = idx - k * ch;
            float acc = a.bc[co];
            for (int ci = 0; ci < ch; ++ci) {
                const float* wp = a.wc + ((size_t)co * ch + ci) * 3;
                if (k > 0) acc = fmaf(wp[0], hn[(k - 1) * ch + ci], acc);
                acc = fmaf(wp[1], hn[k * ch + ci], acc);
                if (k + 1 < ntok) acc = fmaf(wp[2], hn[(k + 1) * ch + ci], acc);
            }
            h[idx] = tanhf(acc);
        }
        __syncthreads();
    }
}

struct CsrnBwdArgs {
    CsrnArgs f;
    const float* dctx;     // [B][ch][H][W] gradient w.r.t. the context map
    float* dx;             // [B][ch][H][W] (accumulated: caller zero-fills, the 4 sweeps add)
    float* slab;           // [B][3ch*ch + 3ch*ch + 3ch*ch + ch] per-image parameter gradients
};

// BPTT: steps in reverse sweep order.  LDS: h (state entering the step), hn (its GRU
// output), dh_in (grad w.r.t. the state entering the NEXT step = this step's conv
// output), dhn (grad w.r.t. hn), gate pre-activation grads gr/gz/gn, and r for W_hn.
__global__ void __launch_bounds__(256) csrn_bwd_kernel(const CsrnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const CsrnArgs& a = p.f;
    const int b = blockIdx.x;
    const int ch = a.ch, HW = a.H * a.W;
    const int ntok = a.along_rows ? a.W : a.H;
    const int nsteps = a.along_rows ? a.H : a.W;
    const int n = ntok * ch;
    float* h = smem;            // [ntok][ch]
    float* hn = h + n;
    float* dh_in = hn + n;      // grad w.r.t. state entering step si+1 (zero for the last step)
    float* dpre = dh_in + n;    // conv pre-activation grad of this step
    float* dhn = dpre + n;
    float* gr = dhn + n;
    float* gz = gr + n;
    float* gn = gz + n;
    float* rr = gn + n;         // r gate, and r * (W_hn h) helper
    float* dh_out = rr + n;     // grad w.r.t. state entering THIS step
    const int nih = 3 * ch * ch;
    float* s_ih = p.slab + (size_t)b * (3 * nih + ch);
    float* s_hh = s_ih + nih;
    float* s_wc = s_hh + nih;
    float* s_bc = s_wc + nih;
    for (int i = threadIdx.x; i < 3 * nih + ch; i += 256) s_ih[i] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) dh_in[i] = 0.f;
    __syncthreads();
    for (int si = nsteps - 1; si >= 0; --si) {
        const int s = a.reverse ? nsteps - 1 - si : si;
        // load h (saved), hn (= ctx of this step)
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int k = idx / ch, c = idx - k * ch;
            h[idx] = a.hprev[(((size_t)si * a.B + b) * ntok + k) * ch + c];
            hn[idx] = a.ctx[((size_t)b * ch + c) * HW + csrn_pos(a, s, k)];
        }
        __syncthreads();
        // conv pre-activation grad: dpre = dh_in * (1 - conv_out^2); conv_out = state of step si+1
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            float g = 0.f;
            if (si + 1 < nsteps) {
                const int k = idx / ch, c = idx - k * ch;
                const float co = a.hprev[(((size_t)(si + 1) * a.B + b) * ntok + k) * ch + c];
                g = dh_in[idx] * (1.f - co * co);
            }
            dpre[idx] = g;
        }
        __syncthreads();
        // dhn = dctx + conv^T(dpre);  conv weight / bias grads
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int k = idx / ch, ci = idx - k * ch;
            float g = p.dctx[((size_t)b * ch + ci) * HW + csrn_pos(a, s, k)];
            for (int co = 0; co < ch; ++co) {
                const float* wp = a.wc + ((size_t)co * ch + ci) * 3;
                // out[k'] uses hn[k'+j-1] with tap j: hn[k] feeds out[k+1] (j=0), out[k] (j=1), out[k-1] (j=2)
                if (k + 1 < ntok) g = fmaf(wp[0], dpre[(k + 1) * ch + co], g);
                g = fmaf(wp[1], dpre[k * ch + co], g);
                if (k > 0) g = fmaf(wp[2], dpre[(k - 1) * ch + co], g);
            }
            dhn[idx] = g;
        }
        for (int e = threadIdx.x; e < nih; e += 256) {          // dWc[co][ci][j]
            const int j = e % 3, ci = (e / 3) % ch, co = e / (3 * ch);
            float acc = 0.f;
            for (int k = 0; k < ntok; ++k) {
                const int kk = k + j - 1;
                if (kk >= 0 && kk < ntok) acc = fmaf(dpre[k * ch + co], hn[kk * ch + ci], acc);
            }
            s_wc[e] += acc;
        }
        for (int co = threadIdx.x; co < ch; co += 256) {
            float acc = 0.f;
            for (int k = 0; k < ntok; ++k) acc += dpre[k * ch + co];
            s_bc[co] += acc;
        }
        __syncthreads();
        // GRU backward, elementwise part (gates recomputed from x and h)
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int k = idx / ch, c = idx - k * ch;
            const float* xp = a.x + (size_t)b * ch * HW + csrn_pos(a, s, k);
            float ir = 0.f, iz = 0.f, in_ = 0.f, hr = 0.f, hz = 0.f, hnn = 0.f;
            for (int j = 0; j < ch; ++j) {
                const float xv = xp[(size_t)j * HW];
                const float hv = h[k * ch + j];
                ir = fmaf(a.w_ih[(size_t)c * ch + j], xv, ir);
                iz = fmaf(a.w_ih[(size_t)(ch + c) * ch + j], xv, iz);
                in_ = fmaf(a.w_ih[(size_t)(2 * ch + c) * ch + j], xv, in_);
                hr = fmaf(a.w_hh[(size_t)c * ch + j], hv, hr);
                hz = fmaf(a.w_hh[(size_t)(ch + c) * ch + j], hv, hz);
                hnn = fmaf(a.w_hh[(size_t)(2 * ch + c) * ch + j], hv, hnn);
            }
            const float r = sigm(ir + hr), z = sigm(iz + hz);
            const float nn = tanhf(in_ + r * hnn);
            const float g = dhn[idx];
            const float dn = g * (1.f - z);
            const float dz = g * (h[idx] - nn);
            const float dpn = dn * (1.f - nn * nn);
            gn[idx] = dpn;
            gz[idx] = dz * z * (1.f - z);
            gr[idx] = dpn * hnn * r * (1.f - r);
            rr[idx] = r;
            dh_out[idx] = g * z;          // direct path h -> h'
        }
        __syncthreads();
        // input / state grads through the mat-vecs
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int k = idx / ch, j = idx - k * ch;
            float dxv = 0.f, dhv = dh_out[idx];
            for (int c = 0; c < ch; ++c) {
                const float vr = gr[k * ch + c], vz = gz[k * ch + c], vn = gn[k * ch + c];
                dxv = fmaf(a.w_ih[(size_t)c * ch + j], vr, dxv);
                dxv = fmaf(a.w_ih[(size_t)(ch + c) * ch + j], vz, dxv);
                dxv = fmaf(a.w_ih[(size_t)(2 * ch + c) * ch + j], vn, dxv);
                dhv = fmaf(a.w_hh[(size_t)c * ch + j], vr, dhv);
                dhv = fmaf(a.w_hh[(size_t)(ch + c) * ch + j], vz, dhv);
                dhv = fmaf(a.w_hh[(size_t)(2 * ch + c) * ch + j], vn * rr[k * ch + c], dhv);
            }
            p.dx[((size_t)b * ch + j) * HW + csrn_pos(a, s, k)] += dxv;
            dh_in[idx] = dhv;            // becomes the conv-output grad of step si-1
        }
        // GRU weight grads: dW_ih[g*ch+c][j] += sum_k G[k][c] x[k][j];  dW_hh likewise with h (n gate: G*r)
        for (int e = threadIdx.x; e < nih; e += 256) {
            const int j = e % ch, c = (e / ch) % ch, gate = e / (ch * ch);
            const float* G = gate == 0 ? gr : (gate == 1 ? gz : gn);
            float ai = 0.f, ah = 0.f;
            for (int k = 0; k < ntok; ++k) {
                const float gv = G[k * ch + c];
                const float xv = a.x[((size_t)b * ch + j) * HW + csrn_pos(a, s, k)];
                ai = fmaf(gv, xv, ai);
                ah = fmaf(gate == 2 ? gv * rr[k * ch + c] : gv, h[k * ch + j], ah);
            }
            s_ih[e] += ai;
            s_hh[e] += ah;
        }
        __syncthreads();
    }
}

// parameter gradients: sum the per-image slabs (fixed order)
__global__ void csrn_param_reduce_kernel(const float* __restrict__ slab, int B, int nparam, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nparam) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += slab[(size_t)b * nparam + i];
    out[i] = s;
}

static int csrn_fill(const scm_csrn_desc* d, CsrnArgs* a, const char* who) {
    SCM_REQUIRE(d && d->x && d->ctx && d->hprev && d->w_ih && d->w_hh && d->wc && d->bc, "%s: null pointer", who);
    SCM_REQUIRE(d->B > 0 && d->ch > 0 && d->H > 0 && d->W > 0, "%s: bad shape", who);
    a->x = d->x; a->ctx = d->ctx; a->hprev = d->hprev; a->w_ih = d->w_ih; a->w_hh = d->w_hh; a->wc = d->wc;
    a->bc = d->bc; a->B = d->B; a->ch = d->ch; a->H = d->H; a->W = d->W;
    a->along_rows = d->along_rows; a->reverse = d->reverse;
    return SCM_OK;
}

extern "C" int scm_csrn_sweep_fwd(const scm_csrn_desc* d, void* stream) {
    CsrnArgs a;
    int rc = csrn_fill(d, &a, "scm_csrn_sweep_fwd");
    if (rc) return rc;
    const int ntok = a.along_rows ? a.W : a.H;
    const size_t lds = (size_t)2 * ntok * a.ch * sizeof(float);
    SCM_REQUIRE(lds <= 64 * 1024, "scm_csrn_sweep_fwd: tokens x channels = %d does not fit the LDS plane", ntok * a.ch);
    hipLaunchKernelGGL(csrn_fwd_kernel, dim3(a.B), dim3(256), lds, (hipStream_t)stream, a);
    SCM_LAUNCH_CHECK("csrn_fwd_kernel");
    return SCM_OK;
}

extern "C" int scm_csrn_sweep_bwd(const scm_csrn_desc* d, const float* dctx, float* dx, float* slab, float* dparams,
                                  void* stream) {
    CsrnBwdArgs p;
    int rc = csrn_fill(d, &p.f, "scm_csrn_sweep_bwd");
    if (rc) return rc;
    SCM_REQUIRE(dctx && dx && slab && dparams, "scm_csrn_sweep_bwd: null pointer");
    p.dctx = dctx; p.dx = dx; p.slab = slab;
    const int ntok = p.f.along_rows ? p.f.W : p.f.H;
    const size_t lds = (size_t)10 * ntok * p.f.ch * sizeof(float);
    SCM_REQUIRE(lds <= 64 * 1024, "scm_csrn_sweep_bwd: tokens x channels = %d does not fit the LDS planes",
                ntok * p.f.ch);
    hipLaunchKernelGGL(csrn_bwd_kernel, dim3(p.f.B), dim3(256), lds, (hipStream_t)stream, p);
    SCM_LAUNCH_CHECK("csrn_bwd_kernel");
    const int nparam = 9 * p.f.ch * p.f.ch + p.f.ch;
    hipLaunchKernelGGL(csrn_param_reduce_kernel, dim3((nparam + 255) / 256), dim3(256), 0, (hipStream_t)stream, slab,
                       p.f.B, nparam, dparams);
    SCM_LAUNCH_CHECK("csrn_param_reduce_kernel");
    return SCM_OK;
}
