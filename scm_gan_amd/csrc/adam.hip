// clip_grad_value_ + Adam for many tensors in one launch.
// Replaces torch.nn.utils.clip_grad_value_(params, 0.1) (main.py:287-290) followed
// by torch.optim.Adam.step() with default betas/eps and no weight decay
// (main.py:125-129,292-296).  The host passes a table of up to 48 tensors by
// value; blockIdx.y selects the tensor, blockIdx.x strides its elements.
#include "common.h"

#define ADAM_MAX_TENSORS 48

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    int n[ADAM_MAX_TENSORS];
    int count;
};

__global__ void __launch_bounds__(256) adam_clip_kernel(const AdamTable t, float lr, float b1, float b2, float eps,
                                                        int step_v, const int* __restrict__ step_dev, float clip,
                                                        float grad_scale) {
    // bias corrections from the step number, by value or from device memory (a captured iteration keeps its
    // step counter on the device); computed here in both cases so the two launch forms give the same bits
    const float stepf = (float)(step_dev ? *step_dev : step_v);
    const float bc1 = 1.f - powf(b1, stepf);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, stepf));
    const int k = blockIdx.y;
    const int n = t.n[k];
    float* p = t.p[k]; float* g = t.g[k]; float* m = t.m[k]; float* v = t.v[k];
    const float step_size = lr / bc1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        if (clip > 0.f || grad_scale != 1.f) g[i] = gi;    // clip_grad_value_ clamps the stored gradient
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

extern "C" int scm_adam_clip_multi(int count, float* const* params, float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, const int* numel, float lr, float beta1, float beta2,
                                   float eps, int step, const int* step_dev, float clip_value, float grad_scale,
                                   void* stream) {
    SCM_REQUIRE(count >= 0 && params && grads && exp_avg && exp_avg_sq && numel, "scm_adam_clip_multi: bad arguments");
    SCM_REQUIRE(step_dev || step >= 1, "scm_adam_clip_multi: step counts from 1");
    for (int base = 0; base < count; base += ADAM_MAX_TENSORS) {
        AdamTable t;
        int c = count - base;
        if (c > ADAM_MAX_TENSORS) c = ADAM_MAX_TENSORS;
        int maxn = 0;
        for (int i = 0; i < c; ++i) {
            SCM_REQUIRE(params[base + i] && grads[base + i] && exp_avg[base + i] && exp_avg_sq[base + i] &&
                            numel[base + i] > 0,
                        "scm_adam_clip_multi: tensor %d invalid", base + i);
            t.p[i] = params[base + i]; t.g[i] = grads[base + i];
            t.m[i] = exp_avg[base + i]; t.v[i] = exp_avg_sq[base + i];
            t.n[i] = numel[base + i];
            if (numel[base + i] > maxn) maxn = numel[base + i];
        }
        t.count = c;
        int bx = (maxn + 255) / 256;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(adam_clip_kernel, dim3(bx, c), dim3(256), 0, (hipStream_t)stream, t, lr, beta1, beta2, eps,
                           step, step_dev, clip_value, grad_scale);
        SCM_LAUNCH_CHECK("adam_clip_kernel");
    }
    return SCM_OK;
}
