// Shared helpers for libscmgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/scm_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// thread-local error slot, defined in api.cpp
void scm_set_error(const char* fmt, ...);
int scm_check_hip(hipError_t e, const char* what);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device function attribute: set it once per
// (kernel, device), from any host thread.  `done` is one bit per device ordinal of the calling
// thread's current device (a static std::atomic<uint64_t> next to each kernel instantiation).
static inline int scm_ensure_lds_attr(const void* kernel, std::atomic<uint64_t>& done, int bytes) {
    int dev = 0;
    int rc = scm_check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    rc = scm_check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
                       "hipFuncSetAttribute");
    if (rc) return rc;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

#define SCM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            scm_set_error(__VA_ARGS__);        \
            return SCM_EINVAL;                 \
        }                                      \
    } while (0)

#define SCM_LAUNCH_CHECK(what)                             \
    do {                                                   \
        int _rc = scm_check_hip(hipGetLastError(), what);  \
        if (_rc) return _rc;                               \
    } while (0)

__device__ __forceinline__ float scm_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ float scm_act(float v, int act) {
    if (act == SCM_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
    if (act == SCM_ACT_SIGMOID) return scm_sigmoid(v);
    return v;
}

// The conv epilogues apply the activation to hundreds of accumulator values per lane; with `act` a run-time value
// scm_act() costs two scalar branches per value (the sigmoid arm is too long to be if-converted), which made the
// epilogue 14 % of a 128 -> 128 convolution.  Epilogues therefore branch ONCE on sigmoid and use this select for the
// other two: slope 1 (none) or 0.01 (leaky_relu); 1.f * v == v bitwise, so both agree with scm_act().
__device__ __forceinline__ float scm_act_slope(int act) { return act == SCM_ACT_LRELU ? 0.01f : 1.f; }
__device__ __forceinline__ float scm_act_sel(float v, float slope) { return v > 0.f ? v : slope * v; }

// derivative of the activation expressed through the saved OUTPUT y = act(x)
// (leaky_relu keeps the sign of x; sigmoid' = y(1-y)).
__device__ __forceinline__ float scm_act_grad_from_out(float y, int act) {
    if (act == SCM_ACT_LRELU) return y > 0.f ? 1.f : 0.01f;
    if (act == SCM_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// Packed fp32 forms with their operand selects spelled out (left to the compiler they are scalarised or built from
// moves; next to f32 MFMAs every VALU instruction is paid for)
__device__ __forceinline__ f32x2 scm_pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 scm_pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 scm_pk_sumdiff(f32x2 p) {            // (p.x + p.y, p.x - p.y)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 scm_pk_addsub(f32x2 t, f32x2 p) {     // (t.x + p.x, t.y - p.y)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(t), "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 scm_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 scm_pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
