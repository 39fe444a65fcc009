// Device helpers shared by the fp32 (kernels.hip) and bf16 (kernels_bf16.hip) conv-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace hipvoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// sin(t)^2 for SnakeBeta.  Quadrant reduction n = rint(t*2/pi), r = t - n*pi/2 by a three-constant
// Cody-Waite fma chain (exact cancellation under fma; good to |t| ~ 1e5), then the sin or cos
// minimax polynomial on [-pi/4, pi/4]; the sign is irrelevant because the result is squared.
// Max error of sin/cos here ~1 ulp, i.e. the same class as libm's sinf the CPU oracle uses.
__device__ __forceinline__ float sin_sq(float t) {
    float n = __builtin_rintf(t * 0.63661977236758134308f);
    float r = __builtin_fmaf(n, -1.57079625129699707031e+00f, t);
    r = __builtin_fmaf(n, -7.54978941586159635335e-08f, r);
    r = __builtin_fmaf(n, -5.39030252995776476554e-15f, r);
    float r2 = r * r;
    // sin(r) = r + r^3 * S(r^2),  cos(r) = 1 - r^2/2 + r^4 * C(r^2)
    float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.86567956e-6f, r2, -1.98559923e-4f), r2, 8.33338592e-3f), r2, -1.66666672e-1f);
    float sv = __builtin_fmaf(r * r2, sp, r);
    float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.44677067e-5f, r2, -1.38877297e-3f), r2, 4.16666567e-2f), r2, -0.5f);
    float cv = __builtin_fmaf(r2, cp, 1.0f);
    int q = (int)n;
    float v = (q & 1) ? cv : sv;
    return v * v;
}

// Makes a (wave-uniform) pointer opaque at this point of the program, so loads through it cannot be
// hoisted above it (hipcc otherwise lifts every epilogue constant load over the whole K loop and
// keeps dozens of registers live).
// (The pointer itself stays a kernel argument, so the loads stay global_load; only a zero offset
// goes through the asm.)
template <typename T>
__device__ __forceinline__ T *launder(T *p) {
    long z = 0;
    asm volatile("" : "+s"(z));
    return p ? p + z : p;
}
// Same, and additionally not before `dep` has been computed (pins the point in program order).
template <typename T>
__device__ __forceinline__ T *launder_after(T *p, float dep) {
    long z = 0;
    asm volatile("" : "+s"(z) : "v"(dep));
    return p ? p + z : p;
}

__device__ __forceinline__ float snake_beta(float x, float ea, float ib) { return x + ib * sin_sq(x * ea); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

}  // namespace hipvoc
