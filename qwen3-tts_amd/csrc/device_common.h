// Device helpers shared by the fp32 (kernels.hip) and bf16 (kernels_bf16.hip) conv-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace hipvoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

// sin^2(r)/r^2 - 1 = C1 z + ... + C5 z^5 on z = r^2 in [0, (pi/2)^2] (weighted least squares on Chebyshev nodes,
// tools/fit_sinsq.py; 9e-9 in exact arithmetic)
#define SINSQ_C1 -0.3333328366279602f
#define SINSQ_C2 0.04444241523742676f
#define SINSQ_C3 -0.003171588061377406f
#define SINSQ_C4 0.0001389977551298216f
#define SINSQ_C5 -3.5771574857790256e-06f

// sin(t)^2 for SnakeBeta, on TWO values at a time (every multiply / fma below is a v_pk_*_f32, one instruction
// for both lanes of the pair; component-wise IEEE, so the scalar form gives bit-identical results).
// sin^2 has period pi: n = rint(t/pi), r = t - n*pi by a three-constant Cody-Waite fma chain (pi = 3.140625 +
// 9.67502594e-4 + 1.50995803e-7: the first two have 8 and 11 significant bits, so n*c is exact for |n| < 2^13 and
// the reduction is good to |t| ~ 2.5e4), then sin^2(r) = z*P(z), z = r^2 in [0, (pi/2)^2], P the degree-5 minimax
// fit with P(0) = 1 (so sin^2 ~ r^2 keeps its relative accuracy near the zeros).  Max abs error 1.7e-7 against
// float64 (libm's sinf squared in fp32: 1.2e-7); 7.5 VALU instructions per value instead of the 23 of a
// quadrant-select sin/cos pair.
__device__ __forceinline__ f32x2 mk2(float a, float b) { f32x2 v = {a, b}; return v; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#ifndef HIPVOC_SNAKE_SCALAR
#define HIPVOC_SNAKE_SCALAR 0     // 1: the same chain on each component with scalar fp32 instructions (component-wise IEEE: same bits)
#endif
__device__ __forceinline__ float sin_sq1(float t) {
    float n = __builtin_rintf(t * 0.31830988618379067154f);
    float r = __builtin_fmaf(n, -3.140625f, t);
    r = __builtin_fmaf(n, -9.675025939941406e-4f, r);
    r = __builtin_fmaf(n, -1.5099580252808664e-7f, r);
    float z = r * r;
    float p = __builtin_fmaf(SINSQ_C5, z, SINSQ_C4);
    p = __builtin_fmaf(p, z, SINSQ_C3);
    p = __builtin_fmaf(p, z, SINSQ_C2);
    p = __builtin_fmaf(p, z, SINSQ_C1);
    p = __builtin_fmaf(p, z, 1.0f);
    return z * p;
}
__device__ __forceinline__ f32x2 sin_sq2(f32x2 t) {
#if HIPVOC_SNAKE_SCALAR
    return mk2(sin_sq1(t.x), sin_sq1(t.y));
#else
    f32x2 n = __builtin_elementwise_rint(t * 0.31830988618379067154f);
    f32x2 r = fma2(n, (f32x2)(-3.140625f), t);
    r = fma2(n, (f32x2)(-9.675025939941406e-4f), r);
    r = fma2(n, (f32x2)(-1.5099580252808664e-7f), r);
    f32x2 z = r * r;
    f32x2 p = fma2((f32x2)(SINSQ_C5), z, (f32x2)(SINSQ_C4));
    p = fma2(p, z, (f32x2)(SINSQ_C3));
    p = fma2(p, z, (f32x2)(SINSQ_C2));
    p = fma2(p, z, (f32x2)(SINSQ_C1));
    p = fma2(p, z, (f32x2)(1.0f));
    return z * p;
#endif
}
__device__ __forceinline__ float sin_sq(float t) {
    f32x2 v = sin_sq2((f32x2)(t));
    return v.x;
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

// SnakeBeta: x + 1/(e^beta + 1e-9) * sin(x e^alpha)^2.  The pair form is what the kernels use; both round identically.
__device__ __forceinline__ f32x2 snake_beta2(f32x2 x, f32x2 ea, f32x2 ib) {
#if HIPVOC_SNAKE_SCALAR
    return mk2(__builtin_fmaf(ib.x, sin_sq1(x.x * ea.x), x.x), __builtin_fmaf(ib.y, sin_sq1(x.y * ea.y), x.y));
#else
    return fma2(ib, sin_sq2(x * ea), x);
#endif
}
// bf16 conv-GEMM mode only (FAST): sin^2 by the hardware's v_sin_f32 (argument in revolutions, after v_fract_f32) instead of the
// polynomial: 4 packed + 4 scalar instructions per PAIR of values against 14 packed ones.  Its sin^2 is good to 7e-6 for |t| < 100
// (tools/ubench/vsin_err.hip) -- three orders of magnitude under the mode's own bf16 rounding of every operand (2^-9 relative) and
// of no consequence for its stated tolerance, but NOT good enough for the fp32 gates: the fp32 and bf16x3 modes keep the polynomial.
// Every kernel of the bf16 mode uses this one form (staging-side, fused and epilogue activations alike), so batch == single and the
// bitwise tests between kernel shapes hold as before.  HIPVOC_SNAKE_HW=0 at compile time restores the polynomial everywhere.
#ifndef HIPVOC_SNAKE_HW
#define HIPVOC_SNAKE_HW 1
#endif
__device__ __forceinline__ f32x2 snake_beta2_hw(f32x2 x, f32x2 ea, f32x2 ib) {
    const f32x2 rev = (x * ea) * 0.15915494309189535f;
    const float s0 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.x)), s1 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.y));
    return fma2(ib, mk2(s0 * s0, s1 * s1), x);
}
template <bool FAST>
__device__ __forceinline__ f32x2 snake_beta2_t(f32x2 x, f32x2 ea, f32x2 ib) {
    if constexpr (FAST && HIPVOC_SNAKE_HW) return snake_beta2_hw(x, ea, ib);
    else return snake_beta2(x, ea, ib);
}
__device__ __forceinline__ float snake_beta(float x, float ea, float ib) {
    f32x2 v = snake_beta2((f32x2)(x), (f32x2)(ea), (f32x2)(ib));
    return v.x;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// encoder path: ELU(alpha = 1) as speech_tokenizer.rs:84-92 writes it, and candle's tanh-form GELU (:438)
__device__ __forceinline__ float elu1(float v) { return fmaxf(v, 0.0f) + (expf(fminf(v, 0.0f)) - 1.0f); }
__device__ __forceinline__ float gelu_tanh(float v) {
    return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * v * (1.0f + 0.044715f * v * v)));
}


// Per-row parameter table in LDS, filled once at kernel start and read by the epilogues (and, in the fused
// residual unit, by the in-register activation).  Reading bias / scale / SnakeBeta parameters from global memory
// inside the epilogue costs a dependent L2 round trip per batch -- dozens per workgroup -- while the table costs
// one load per row, issued before the main loop and long complete when it is needed; indexing it by tile ROW also
// removes the n / S division of transposed convs from the epilogue.
//   unfused: [4][BN]  bias | rscale | psn_a | psn_ib      (entries of absent vectors are never read)
//   fused:   [6][BN]  bias1 | sn2_a | sn2_ib | bias2 | psn_a | psn_ib
// All source vectors are zero-padded by the loader, so padded rows read zeros.
// Octet-interleaved rows of a transposed conv (ConvGemmArgs::pm == 2; stride S = 4 or 8, Cout % 8 == 0).  With phase-major rows a
// workgroup's rows are ONE phase of its channels: every 4- or 8-byte store lands in a different cache line, whose other phases come
// from other workgroups at other times (bf16 mode, b2's trans-conv: 7.9 GB written for 3.0 GB of tensors).  Here the rows are ordered
// so that the accumulator layout itself is the memory layout: 8 S consecutive rows are all phases of 8 channels, and in the 32-row
// tile ti of that group the lane half h holds, in registers r = 0..15, channel r >> 1 at phase (S/2) h + 2 ti + (r & 1) -- the
// fp32 tensor gets 8-byte stores that tile whole lines across the wave (lane m, half h), the packed bf16 tensor whole 16-byte
// elements (8 channels of one position), two adjacent ones per lane.  No cross-lane traffic; the loader packs the weights so
// (kernels.h: tconv_oi_n, the inverse map).
// row n -> (channel, phase):
__device__ __forceinline__ void tconv_oi_row(int n, int S, int &c, int &ph) {
    const int sup = n / (8 * S), t = n - sup * 8 * S, ti = t >> 5, rho = t & 31;
    const int h = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    c = 8 * sup + (r >> 1);
    ph = (S >> 1) * h + 2 * ti + (r & 1);
}
template <int BN, bool FUSE>
__device__ __forceinline__ void fill_param_table(const ConvGemmArgs &a, float *Pm, int n0, int tid, int nthreads) {
    for (int row = tid; row < BN; row += nthreads) {
        const int n = n0 + row;
        int co = a.S > 1 ? (a.pm ? n % (a.N / a.S) : n / a.S) : n;
        if (a.S > 1 && a.pm == 2) { int ph_; tconv_oi_row(n, a.S, co, ph_); }
        if (a.S > 1 && a.pm == 3) { const int t = n % (32 * a.S); co = (n / (32 * a.S)) * 32 + (t & 31); }      // phase-tile rows, see tconv_pt_epilogue
        if constexpr (FUSE) {
            Pm[0 * BN + row] = a.bias[co];
            Pm[1 * BN + row] = a.sn2_a[co];
            Pm[2 * BN + row] = a.sn2_ib[co];
            Pm[3 * BN + row] = a.bias2[co];
            if (a.psn_a) { Pm[4 * BN + row] = a.psn_a[co]; Pm[5 * BN + row] = a.psn_ib[co]; }
        } else {
            if (a.bias) Pm[0 * BN + row] = a.bias[co];
            if (a.rscale) Pm[1 * BN + row] = a.rscale[co];
            if (a.psn_a) { Pm[2 * BN + row] = a.psn_a[co]; Pm[3 * BN + row] = a.psn_ib[co]; }
        }
    }
}

// Epilogue of one 32-row accumulator tile (fp32): bias, GELU, per-channel scale, residual, clamp, optional
// SnakeBeta (to y, or to y2 beside the raw y), store.  Register r of the tile is tile row rbase + (r&3) + 8*(r>>2)
// (rbase = 32*i + 4h), output row n0 + that, column m.  Parameters come from the LDS table; the residual is
// fetched as one batch of eight unconditional loads under a wave-uniform `if` (a per-element `ptr ? ptr[i] : 0`
// makes hipcc branch around each load and wait for it; measured: 5 % of the fp32 conv7 kernels, 35 % of the 1x1
// convs, 38 % of a bf16-mode pass).  An out-of-range element reads residual offset 0 and is never stored.
__device__ __forceinline__ unsigned short bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    return make_uint2((unsigned)bf16_bits(a) | ((unsigned)bf16_bits(b) << 16), (unsigned)bf16_bits(c) | ((unsigned)bf16_bits(d) << 16));
}

__device__ __forceinline__ float bf16_round(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ void unpack4_bf16(uint2 u, float *o) {
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
}
// ConvGemmArgs::rbf == 2: the packed bf16 tensor behind a float-typed field, at the unit the kernel's fp32 pointer arithmetic chose
// (pu = field + unit * y_us in floats; the packed tensor counts the same strides in 16-byte elements)
__device__ __forceinline__ unsigned short *rbf_packed(const float *field, const float *pu) {
    return reinterpret_cast<unsigned short *>(const_cast<float *>(field)) + ((pu - field) << 3);
}

// ybu != null (bf16 packed output, see ConvGemmArgs::yb): yu (may be null) gets the raw value, ybu the activated one
// as bf16 in channel-group-major form.  For S == 1 a lane's four rows (r&3) are half of a channel group: one 8-byte
// store; the same holds for transposed convs with phase-major rows (ConvGemmArgs::pm); with channel-major rows
// (channel, phase) pairs the bf16 values go out one by one.
// Epilogue of one 32-row tile with octet-interleaved rows (pm == 2; the decoder blocks' transposed convs: bias, optional SnakeBeta
// to y / y2 / the packed tensor -- no residual, scale or activation, which the host checks).
template <int BN, bool FAST = false>
__device__ __forceinline__ void tconv_oi_epilogue_tile(const ConvGemmArgs &a, const f32x16 &acc, const float *Pm, int n0, int rbase, int m,
                                                       float *yu, float *y2u, unsigned short *ybu) {
    const int S = a.S, h = (rbase >> 2) & 1, nb = n0 + (rbase & ~31);
    const int sup = nb / (8 * S), ti = (nb - sup * 8 * S) >> 5;
    const int c0 = 8 * sup, ph0 = (S >> 1) * h + 2 * ti;
    if (c0 >= a.N / S || m >= a.M) return;                      // (rows are whole groups of 8 channels: N % (8 S) == 0)
    float v[16], w[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        v[r] = a.bias ? acc[r] + Pm[0 * BN + row] : acc[r];
        w[r] = v[r];
    }
    if (a.psn_a) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                       // registers r, r + 1: one channel, two phases
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            const f32x2 sv = snake_beta2_t<FAST>(mk2(v[r], v[r + 1]), mk2(Pm[2 * BN + row], Pm[2 * BN + row + 1]), mk2(Pm[3 * BN + row], Pm[3 * BN + row + 1]));
            w[r] = sv.x; w[r + 1] = sv.y;
        }
    }
    const long pos = (long)m * S + ph0;                          // even: 8-byte aligned in rows of 4 k floats
    if (a.rbf == 1) {                                            // (the raw values are the block's residual stream, kept in bf16)
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = bf16_round(v[r]);
    }
    if (ybu) {
        if (yu && a.rbf == 2) {
            uint4 *e = reinterpret_cast<uint4 *>(rbf_packed(a.y, yu) + (((long)(c0 >> 3) * a.y_cs + pos) << 3));
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const uint2 lo = pack4_bf16(v[p], v[2 + p], v[4 + p], v[6 + p]), hi = pack4_bf16(v[8 + p], v[10 + p], v[12 + p], v[14 + p]);
                e[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else if (yu) {
#pragma unroll
            for (int j = 0; j < 8; j++) *reinterpret_cast<float2 *>(yu + (long)(c0 + j) * a.y_cs + pos) = make_float2(v[2 * j], v[2 * j + 1]);
        }
        uint4 *e = reinterpret_cast<uint4 *>(ybu + (((long)(c0 >> 3) * a.y_cs + pos) << 3));
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint2 lo = pack4_bf16(w[p], w[2 + p], w[4 + p], w[6 + p]), hi = pack4_bf16(w[8 + p], w[10 + p], w[12 + p], w[14 + p]);
            e[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        return;
    }
    if (a.psn_a && y2u) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            *reinterpret_cast<float2 *>(yu + (long)(c0 + j) * a.y_cs + pos) = make_float2(v[2 * j], v[2 * j + 1]);
            *reinterpret_cast<float2 *>(y2u + (long)(c0 + j) * a.y_cs + pos) = make_float2(w[2 * j], w[2 * j + 1]);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) *reinterpret_cast<float2 *>(yu + (long)(c0 + j) * a.y_cs + pos) = make_float2(w[2 * j], w[2 * j + 1]);
}

// Phase-tile rows (ConvGemmArgs::pm == 3; odd strides S = 3, 5; Cout % 32 == 0): the workgroup's S row tiles are the S phases of
// the same 32 channels (row n = (c / 32) 32 S + phase 32 + c % 32), so the lane that holds channel c at input step m in tile 0
// holds its other phases in the same register of the other tiles: S consecutive floats of y (4-byte stores that tile whole lines
// across the wave and the S store instructions), and for the packed tensor the 8-byte halves (four channels) of S adjacent
// elements.  Needs the launch's tile height to be S (detail::conv_gemm_launch_tn).
template <int BN, int S, bool FAST = false>
__device__ __forceinline__ void tconv_pt_epilogue(const ConvGemmArgs &a, const f32x16 (&acc)[S], const float *Pm, int n0, int h, int m,
                                                  float *yu, float *y2u, unsigned short *ybu) {
    static_assert(BN == 32 * S, "one row tile per phase");
    const int cbase = (n0 / BN) * 32;
    if (m >= a.M) return;
    const long pos = (long)m * S;
#pragma unroll
    for (int g = 0; g < 4; g++) {                               // registers 4g .. 4g+3: channels c .. c+3
        const int c = cbase + 8 * g + 4 * h;
        if (c >= a.N / S) continue;
        float v[S][4], w[S][4];
#pragma unroll
        for (int ph = 0; ph < S; ph++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = ph * 32 + 8 * g + 4 * h + j;
                v[ph][j] = a.bias ? acc[ph][4 * g + j] + Pm[0 * BN + row] : acc[ph][4 * g + j];
                w[ph][j] = v[ph][j];
            }
        if (a.psn_a) {
#pragma unroll
            for (int ph = 0; ph < S; ph++)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int row = ph * 32 + 8 * g + 4 * h + j;
                    const f32x2 sv = snake_beta2_t<FAST>(mk2(v[ph][j], v[ph][j + 1]), mk2(Pm[2 * BN + row], Pm[2 * BN + row + 1]), mk2(Pm[3 * BN + row], Pm[3 * BN + row + 1]));
                    w[ph][j] = sv.x; w[ph][j + 1] = sv.y;
                }
        }
        const bool raw_y = ybu || (a.psn_a && y2u);             // y keeps the raw value when the activated one has its own tensor
        if (a.rbf == 1 && raw_y) {
#pragma unroll
            for (int ph = 0; ph < S; ph++)
#pragma unroll
                for (int j = 0; j < 4; j++) v[ph][j] = bf16_round(v[ph][j]);
        }
        if (yu && a.rbf == 2) {
            unsigned short *yrb = rbf_packed(a.y, yu);
#pragma unroll
            for (int ph = 0; ph < S; ph++)
                *reinterpret_cast<uint2 *>(yrb + ((((long)(c >> 3) * a.y_cs + pos + ph) << 3) + (c & 4))) = pack4_bf16(v[ph][0], v[ph][1], v[ph][2], v[ph][3]);
        } else if (yu) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ph = 0; ph < S; ph++) yu[(long)(c + j) * a.y_cs + pos + ph] = raw_y ? v[ph][j] : w[ph][j];
        }
        if (!ybu && a.psn_a && y2u) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ph = 0; ph < S; ph++) y2u[(long)(c + j) * a.y_cs + pos + ph] = w[ph][j];
        }
        if (ybu) {
#pragma unroll
            for (int ph = 0; ph < S; ph++)
                *reinterpret_cast<uint2 *>(ybu + ((((long)(c >> 3) * a.y_cs + pos + ph) << 3) + (c & 4))) = pack4_bf16(w[ph][0], w[ph][1], w[ph][2], w[ph][3]);
        }
    }
}

struct ResPacked { uint2 q[4]; };      // four 8-byte pieces of the packed bf16 stream: a 32-row tile's residual, requested ahead of its use

// OI: the kernel instance for octet-interleaved transposed convs (pm == 2).  A compile-time choice: as a run-time branch the second
// epilogue cost the 128-register bf16 kernels spills in every layer (measured: conv7 layers 8x slower).
template <int BN, bool ENC = false, bool OI = false, bool FAST = false>
__device__ __forceinline__ void conv_epilogue_tile(const ConvGemmArgs &a, const f32x16 &acc, const float *Pm, int n0, int rbase,
                                                   int m, float *yu, const float *ru, float *y2u, unsigned short *ybu = nullptr) {
    if constexpr (OI) {
        tconv_oi_epilogue_tile<BN, FAST>(a, acc, Pm, n0, rbase, m, yu, y2u, ybu);
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        long o[8];
        bool ok[8];
        int row[8];
        int cg[2] = {0, 0}, pg[2] = {0, 0};      // (channel, phase) of rows row[0] and row[4] (grouped forms)
        if (a.S > 1 && a.pm) {                   // phase-major rows: one division per half tile; a lane's four
            const int Co = a.N / a.S;            // consecutive rows are four channels of ONE phase (Co % 8 == 0)
            const int n = n0 + rbase + 16 * half;
            pg[0] = n / Co; cg[0] = n - pg[0] * Co;
            cg[1] = cg[0] + 8; pg[1] = pg[0];
            if (cg[1] >= Co) { cg[1] -= Co; pg[1]++; }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = half * 8 + q;
            row[q] = rbase + (r & 3) + 8 * (r >> 2);
            const int n = n0 + row[q];
            ok[q] = n < a.N && m < a.M;
            int c = n, ph = 0;
            if (a.S > 1) {
                if (a.pm) { c = cg[q >> 2] + (q & 3); ph = pg[q >> 2]; }
                else { c = n / a.S; ph = n - c * a.S; }
            }
            o[q] = (long)c * a.y_cs + (long)m * a.S + ph;
        }
        if (a.S == 1) { cg[0] = n0 + row[0]; cg[1] = n0 + row[4]; }
        long ro[8];                               // residual offsets (differ from o[] only with s2d)
#pragma unroll
        for (int q = 0; q < 8; q++) ro[q] = o[q];
        if constexpr (ENC) {
            if (a.s2d > 1) {
                const int qd = m / a.s2d, j = m - qd * a.s2d;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const long c = n0 + row[q];
                    ro[q] = c * a.x_cs + m;
                    o[q] = (c * a.s2d + j) * a.y_cs + qd;
                }
            }
        }
        float v[8], t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = acc[half * 8 + q];
#ifndef EPI_ABL
#define EPI_ABL 0          // timing-only experiments: 1 = no residual loads (zeros), 2 = no fp32 stores beside a packed output
#endif
        if (ru && a.rbi == 2) {                   // packed bf16 residual (S == 1 or phase-major rows: host)
            const unsigned short *rrb = rbf_packed(a.res, ru);
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int c = cg[g];
                const uint2 u = *reinterpret_cast<const uint2 *>(rrb + (ok[4 * g] ? ((((long)(c >> 3) * a.y_cs + (long)m * a.S + pg[g]) << 3) + (c & 4)) : 0));
                unpack4_bf16(u, t + 4 * g);
            }
        } else if (ru) {
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = (EPI_ABL & 1) ? 0.0f : ru[ok[q] ? ro[q] : 0];
        }
        if (a.bias) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] += Pm[0 * BN + row[q]];
        }
        if (a.act == ACT_GELU) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = gelu_erf(v[q]);
        }
        if (a.rscale) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] *= Pm[1 * BN + row[q]];
        }
        if (ru) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = t[q] + v[q];
        }
        if (a.act == ACT_CLAMP1) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = fminf(fmaxf(v[q], -1.0f), 1.0f);
        }
        if constexpr (ENC) {
            if (a.act == ACT_ELU) {
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = elu1(v[q]);
            } else if (a.act == ACT_GELU_TANH) {
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = gelu_tanh(v[q]);
            }
        }
        if (ybu) {
            float w[8];
#pragma unroll
            for (int q = 0; q < 8; q++) w[q] = v[q];
            if (a.psn_a) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2 sv = snake_beta2_t<FAST>(mk2(v[q], v[q + 1]), mk2(Pm[2 * BN + row[q]], Pm[2 * BN + row[q + 1]]),
                                           mk2(Pm[3 * BN + row[q]], Pm[3 * BN + row[q + 1]]));
                    w[q] = sv.x; w[q + 1] = sv.y;
                }
            }
            if (yu && a.rbf == 2) {
                unsigned short *yrb = rbf_packed(a.y, yu);
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const int c = cg[g];
                    if (ok[4 * g])
                        *reinterpret_cast<uint2 *>(yrb + ((((long)(c >> 3) * a.y_cs + (long)m * a.S + pg[g]) << 3) + (c & 4))) =
                            pack4_bf16(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                }
            } else if (yu && !(EPI_ABL & 2)) {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (ok[q]) yu[o[q]] = a.rbf ? bf16_round(v[q]) : v[q];
            }
            if (a.S == 1 || a.pm) {
#pragma unroll
                for (int g = 0; g < 2; g++) {      // rows row[4g..4g+3] = channels 4 consecutive, aligned to 4
                    const int c = cg[g];
                    if (ok[4 * g])                  // ((N/S) % 8 == 0: a half group is valid as a whole)
                        *reinterpret_cast<uint2 *>(ybu + ((((long)(c >> 3) * a.y_cs + (long)m * a.S + pg[g]) << 3) + (c & 4))) =
                            pack4_bf16(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int n = n0 + row[q], c = n / a.S, ph = n - c * a.S;
                    if (ok[q]) ybu[(((long)(c >> 3) * a.y_cs + (long)m * a.S + ph) << 3) + (c & 7)] = bf16_bits(w[q]);
                }
            }
        } else if (a.psn_a) {
            if (y2u) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2 sv = snake_beta2_t<FAST>(mk2(v[q], v[q + 1]), mk2(Pm[2 * BN + row[q]], Pm[2 * BN + row[q + 1]]),
                                           mk2(Pm[3 * BN + row[q]], Pm[3 * BN + row[q + 1]]));
                    if (ok[q]) { yu[o[q]] = a.rbf ? bf16_round(v[q]) : v[q]; y2u[o[q]] = sv.x; }
                    if (ok[q + 1]) { yu[o[q + 1]] = a.rbf ? bf16_round(v[q + 1]) : v[q + 1]; y2u[o[q + 1]] = sv.y; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2 sv = snake_beta2_t<FAST>(mk2(v[q], v[q + 1]), mk2(Pm[2 * BN + row[q]], Pm[2 * BN + row[q + 1]]),
                                           mk2(Pm[3 * BN + row[q]], Pm[3 * BN + row[q + 1]]));
                    if (ok[q]) yu[o[q]] = sv.x;
                    if (ok[q + 1]) yu[o[q + 1]] = sv.y;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (ok[q]) yu[o[q]] = a.rbf ? bf16_round(v[q]) : v[q];
        }
    }
}

// Epilogue of one output tile of the fused residual unit: y = res + (acc2 + b2) [then SnakeBeta]; resv holds the
// residual prefetched before the second product; parameters from the LDS table (rows rbase + ...).
template <int BN, bool FAST = false>
__device__ __forceinline__ void fused_epilogue_tile(const ConvGemmArgs &a, const f32x16 &acc2, const float *resv, const float *Pm,
                                                    int rbase, int m, float *yu, unsigned short *ybu = nullptr) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
        int n[8];
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int r = half * 8 + q; n[q] = rbase + (r & 3) + 8 * (r >> 2); }
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = resv[half * 8 + q] + (acc2[half * 8 + q] + Pm[3 * BN + n[q]]);
        if (ybu) {     // raw fp32 to yu (if wanted), activated bf16 to the packed tensor
            if (yu && a.rbf == 2) {
                unsigned short *yrb = rbf_packed(a.y, yu);
#pragma unroll
                for (int g = 0; g < 2; g++)
                    if (n[4 * g] < a.N && m < a.M)
                        *reinterpret_cast<uint2 *>(yrb + ((((long)(n[4 * g] >> 3) * a.y_cs + m) << 3) + (n[4 * g] & 4))) =
                            pack4_bf16(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            } else if (yu) {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (n[q] < a.N && m < a.M) yu[(long)n[q] * a.y_cs + m] = a.rbf ? bf16_round(v[q]) : v[q];
            }
            if (a.psn_a) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2 sv = snake_beta2_t<FAST>(mk2(v[q], v[q + 1]), mk2(Pm[4 * BN + n[q]], Pm[4 * BN + n[q + 1]]),
                                           mk2(Pm[5 * BN + n[q]], Pm[5 * BN + n[q + 1]]));
                    v[q] = sv.x; v[q + 1] = sv.y;
                }
            }
#pragma unroll
            for (int g = 0; g < 2; g++)
                if (n[4 * g] < a.N && m < a.M)
                    *reinterpret_cast<uint2 *>(ybu + ((((long)(n[4 * g] >> 3) * a.y_cs + m) << 3) + (n[4 * g] & 4))) =
                        pack4_bf16(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            continue;
        }
        if (a.psn_a) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                f32x2 sv = snake_beta2_t<FAST>(mk2(v[q], v[q + 1]), mk2(Pm[4 * BN + n[q]], Pm[4 * BN + n[q + 1]]),
                                       mk2(Pm[5 * BN + n[q]], Pm[5 * BN + n[q + 1]]));
                v[q] = sv.x; v[q + 1] = sv.y;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (n[q] < a.N && m < a.M) yu[(long)n[q] * a.y_cs + m] = (a.rbf && !a.psn_a) ? bf16_round(v[q]) : v[q];
    }
}

// The residual of one 32-row output tile of a fused unit, as the kernels request it ahead of the 1x1 product: four 8-byte pieces of
// the packed bf16 stream (ConvGemmArgs::rbi == 2; resu = a.res + unit * y_us) -- register 4g + j of the tile is channel
// rbase + 8g + j (rbase = 32 o + 4 h) -- unpacked where the epilogue needs the values.
__device__ __forceinline__ void fused_fetch_res_packed(const ConvGemmArgs &a, const float *resu, int rbase, int m, bool ok, ResPacked &d) {     // ok: m < M; N % 8 == 0
    const unsigned short *rrb = rbf_packed(a.res, resu);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int c = rbase + 8 * g;
        d.q[g] = *reinterpret_cast<const uint2 *>(rrb + ((ok && c < a.N) ? ((((long)(c >> 3) * a.y_cs + m) << 3) + (c & 4)) : 0));
    }
}
__device__ __forceinline__ void fused_unpack_res(const ResPacked &d, float *resv) {
#pragma unroll
    for (int g = 0; g < 4; g++) unpack4_bf16(d.q[g], resv + 4 * g);
}

}  // namespace hipvoc
