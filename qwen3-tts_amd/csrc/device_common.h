// Device helpers shared by the fp32 (kernels.hip) and bf16 (kernels_bf16.hip) conv-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

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


// Epilogue of one 32-row accumulator tile (fp32): bias, GELU, per-channel scale, residual, clamp, optional
// SnakeBeta (to y, or to y2 beside the raw y), store.  Register r of the tile is row nbase + (r&3) + 8*(r>>2)
// (nbase already holds the lane half's +4h), column m.
// Every per-channel vector and the residual are fetched as BATCHES of eight unconditional loads under
// wave-uniform `if`s: a per-element `ptr ? ptr[i] : 0` makes hipcc branch around each load and wait for it,
// i.e. 3-4 dependent memory round trips for each of the 16*TN elements of a lane (measured: 5 % of the fp32
// conv7 kernels, 35 % of the 1x1 convs, 38 % of a bf16-mode pass).  Vectors are zero-padded by the loader, so
// padded rows index them safely; an out-of-range element reads residual offset 0 and is never stored.
__device__ __forceinline__ void conv_epilogue_tile(const ConvGemmArgs &a, const f32x16 &acc, int nbase, int m, float *yu,
                                                   const float *ru, float *y2u, const float *bp, const float *scp,
                                                   const float *pa, const float *pb) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
        int co[8];
        long o[8];
        bool ok[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = half * 8 + q;
            const int n = nbase + (r & 3) + 8 * (r >> 2);
            ok[q] = n < a.N && m < a.M;
            int c = n, ph = 0;
            if (a.S > 1) { c = n / a.S; ph = n - c * a.S; }
            co[q] = c;
            o[q] = (long)c * a.y_cs + (long)m * a.S + ph;
        }
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = acc[half * 8 + q];
        if (bp) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = bp[co[q]];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] += t[q];
        }
        if (a.act == ACT_GELU) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = gelu_erf(v[q]);
        }
        if (scp) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = scp[co[q]];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] *= t[q];
        }
        if (ru) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = ru[ok[q] ? o[q] : 0];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = t[q] + v[q];
        }
        if (a.act == ACT_CLAMP1) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = fminf(fmaxf(v[q], -1.0f), 1.0f);
        }
        if (pa) {
            float ta[8], tb[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { ta[q] = pa[co[q]]; tb[q] = pb[co[q]]; }
            if (y2u) {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (ok[q]) { yu[o[q]] = v[q]; y2u[o[q]] = snake_beta(v[q], ta[q], tb[q]); }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (ok[q]) yu[o[q]] = snake_beta(v[q], ta[q], tb[q]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (ok[q]) yu[o[q]] = v[q];
        }
    }
}

// Epilogue of one output tile of the fused residual unit: y = res + (acc2 + b2) [then SnakeBeta]; resv holds the
// residual prefetched before the second product.
__device__ __forceinline__ void fused_epilogue_tile(const ConvGemmArgs &a, const f32x16 &acc2, const float *resv, int nbase,
                                                    int m, float *yu, const float *b2p, const float *pa, const float *pb) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
        int n[8];
        float v[8], t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int r = half * 8 + q; n[q] = nbase + (r & 3) + 8 * (r >> 2); }
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = b2p[n[q]];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = resv[half * 8 + q] + (acc2[half * 8 + q] + t[q]);
        if (pa) {
            float ta[8], tb[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { ta[q] = pa[n[q]]; tb[q] = pb[n[q]]; }
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = snake_beta(v[q], ta[q], tb[q]);
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (n[q] < a.N && m < a.M) yu[(long)n[q] * a.y_cs + m] = v[q];
    }
}

}  // namespace hipvoc
