// gfx950 (CDNA4 / MI355X) kernels of the speech-tokenizer ENCODER path that are not conv-GEMMs
// (/root/reference/src/speech_tokenizer.rs).  The convolutions, projections and the attention run
// on the vocoder's kernels (kernels.hip: conv_gemm_kernel<..., ENC>, attention_kernel<16>).
#include "device_common.h"
#include "kernels.h"

namespace hipvoc {

// ---------------------------------------------------------------------------------------------------
// Space-to-depth (+ optional ELU): y[(c*r + j)][u][q] = act(x[c][u][q*r + j]), q < Lq = L / r.
// A CausalConv1d of kernel 2r, stride r, left pad r (speech_tokenizer.rs:65-79, 204-211, 595-603) over x is then an
// ordinary 2-tap conv (shifts -1, 0) over the c*r channels of y, which is what conv_gemm runs.
// HBM-bound copy: a thread reads r consecutive inputs of one channel (lanes: consecutive q -> one contiguous
// span per wave) and writes r rows.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s2d_kernel(const float *x, float *y, int C, int G, int L, int Lp, int r, int Lq,
                                                   int Lqp, int elu) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, u = blockIdx.z;
    if (q >= Lq) return;
    const float *xr = x + ((long)c * G + u) * Lp + (long)q * r;
    float *yr = y + ((long)c * r * G + u) * Lqp + q;
    const long ys = (long)G * Lqp;
    for (int j = 0; j < r; j++) {
        float v = xr[j];
        if (elu) v = elu1(v);
        yr[j * ys] = v;
    }
}

hipError_t launch_s2d(const float *x, float *y, int C, int G, int L, int Lp, int r, int Lq, int Lqp, int elu, hipStream_t s) {
    if (C <= 0 || G <= 0 || Lq <= 0) return hipSuccess;
    if (r < 1 || (long)Lq * r > L || C > 65535 || G > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(s2d_kernel, dim3((Lq + 255) / 256, C, G), dim3(256), 0, s, x, y, C, G, L, Lp, r, Lq, Lqp, elu);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm with bias over channels of x[C][P] (speech_tokenizer.rs:394-403): mean, biased variance of the centred
// values, (x - mean) / sqrt(var + eps) * w + b.  Workgroup = 32 positions x 8 channel slices, slices reduced through
// LDS in a fixed order (same shape as rmsnorm_kernel).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, float *y, const float *w, const float *b, int C,
                                                         long P, float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    long p = (long)blockIdx.x * 32 + lp;
    bool ok = p < P;
    float s = 0.0f;
    if (ok)
        for (int c = sl; c < C; c += 8) s += x[(long)c * P + p];
    red[sl][lp] = s;
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mean += red[i][lp];
    mean /= (float)C;
    __syncthreads();
    float v = 0.0f;
    if (ok)
        for (int c = sl; c < C; c += 8) {
            float d = x[(long)c * P + p] - mean;
            v += d * d;
        }
    red[sl][lp] = v;
    __syncthreads();
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) var += red[i][lp];
    float den = sqrtf(var / (float)C + eps);
    if (ok)
        for (int c = sl; c < C; c += 8) y[(long)c * P + p] = (x[(long)c * P + p] - mean) / den * w[c] + b[c];
}

hipError_t launch_layernorm(const float *x, float *y, const float *w, const float *b, int C, long P, float eps, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, x, y, w, b, C, P, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// RVQ encode (speech_tokenizer.rs:494-509, 536-556, 581-585): one workgroup per frame.
//   proj = input_proj * h (1x1 conv, no bias), then per layer: nearest codeword of the residual, residual -= codeword.
// BIT-EXACT CONTRACT (oracle/enc_oracle.c: linear, nearest, rvq_encode): every dot product is a k-ascending fmaf
// chain from 0; |x|^2 and |e|^2 are k-ascending sums of ROUNDED squares; distance = (|x|^2 + |e|^2) - 2*<x,e>;
// argmin = the first minimum.  Contraction is ruled out by spelling every operation as an intrinsic.
// Thread c walks codewords c, c+256, ... of the transposed codebook (coalesced); the residual lives in LDS.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rvq_encode_kernel(RvqEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hs = smem;              // [dim]
    float *xs = hs + a.dim;        // [D] residual
    __shared__ float bd[256];
    __shared__ int bi[256];
    const int tid = threadIdx.x;
    const int u = blockIdx.x / a.T, t = blockIdx.x - u * a.T;
    const long cs = (long)a.G * a.Lp, base = (long)u * a.Lp + t;
    for (int k = tid; k < a.dim; k += 256) hs[k] = a.h[(long)k * cs + base];
    __syncthreads();
    long long *cp = a.codes + ((long)u * a.T + t) * (a.n_sem + a.n_ac);
    for (int group = 0; group < 2; group++) {
        const int nl = group ? a.n_ac : a.n_sem;
        if (!nl) continue;
        const float *wp = a.wp_t + (long)group * a.dim * a.D;
        for (int d = tid; d < a.D; d += 256) {
            float p = 0.0f;
            for (int k = 0; k < a.dim; k++) p = __builtin_fmaf(wp[(long)k * a.D + d], hs[k], p);
            xs[d] = p;
            if (a.proj) a.proj[(long)(group * a.D + d) * cs + base] = p;
        }
        __syncthreads();
        for (int l = 0; l < nl; l++) {
            const int q = group ? a.n_sem + l : l;
            const float *et = a.emb_t + (long)q * a.D * a.cb_size;
            const float *esq = a.e_sq + (long)q * a.cb_size;
            float x_sq = 0.0f;
            for (int d = 0; d < a.D; d++) x_sq = __fadd_rn(x_sq, __fmul_rn(xs[d], xs[d]));
            float best = INFINITY;
            int besti = 0x7fffffff;
            for (int c = tid; c < a.cb_size; c += 256) {
                float xe = 0.0f;
                for (int d = 0; d < a.D; d++) xe = __builtin_fmaf(xs[d], et[(long)d * a.cb_size + c], xe);
                float dist = __fsub_rn(__fadd_rn(x_sq, esq[c]), __fmul_rn(xe, 2.0f));
                if (dist < best) { best = dist; besti = c; }      // ascending c: the first minimum of this thread
            }
            bd[tid] = best;
            bi[tid] = besti;
            __syncthreads();
            for (int sft = 128; sft > 0; sft >>= 1) {
                if (tid < sft) {
                    float od = bd[tid + sft];
                    int oi = bi[tid + sft];
                    if (od < bd[tid] || (od == bd[tid] && oi < bi[tid])) { bd[tid] = od; bi[tid] = oi; }
                }
                __syncthreads();
            }
            int code = bi[0];
            if (code == 0x7fffffff) code = 0;                      // every distance NaN / +inf: argmin of candle returns 0
            if (tid == 0) cp[q] = code;
            __syncthreads();                                       // bd / bi are rewritten by the next layer
            for (int d = tid; d < a.D; d += 256) xs[d] = __fsub_rn(xs[d], et[(long)d * a.cb_size + code]);
            __syncthreads();
        }
    }
}

hipError_t launch_rvq_encode(const RvqEncArgs &a, hipStream_t s) {
    if (a.G <= 0 || a.T <= 0) return hipSuccess;
    size_t lds = sizeof(float) * ((size_t)a.dim + a.D);
    if (lds > 60 * 1024 || a.cb_size < 1 || a.D < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)(a.G * a.T)), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace hipvoc
