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
// Tap stack of a single-channel signal: y[j][u][t] = x[u][t + j - (K-1)] (0 before the start), j < K.  The encoder's
// first conv (1 -> 64 channels, K = 7, causal) is then a 1-tap GEMM over K "channels": one K stage instead of K stages
// whose 16-channel blocks hold one real channel each.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tap_stack_kernel(const float *x, float *y, int K, int G, int L, int Lp) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (t >= L) return;
    const float *xr = x + (long)u * Lp;
    for (int j = 0; j < K; j++) {
        const int sidx = t + j - (K - 1);
        y[((long)j * G + u) * Lp + t] = sidx >= 0 ? xr[sidx] : 0.0f;
    }
}

hipError_t launch_tap_stack(const float *x, float *y, int K, int G, int L, int Lp, hipStream_t s) {
    if (G <= 0 || L <= 0) return hipSuccess;
    if (K < 1 || G > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tap_stack_kernel, dim3((L + 255) / 256, G), dim3(256), 0, s, x, y, K, G, L, Lp);
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

// Register-resident form: same slices and summation orders (the same bits), x read once, all loads in flight together.
template <int NPT>
__global__ __launch_bounds__(256) void layernorm_reg_kernel(const float *x, float *y, const float *w, const float *b,
                                                             int C, long P, float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    long p = (long)blockIdx.x * 32 + lp;
    bool ok = p < P;
    long pc = ok ? p : P - 1;
    float v[NPT];
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        int c = sl + 8 * j;
        v[j] = c < C ? x[(long)c * P + pc] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NPT; j++) s += v[j];
    red[sl][lp] = s;
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mean += red[i][lp];
    mean /= (float)C;
    __syncthreads();
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        float d = v[j] - mean;
        q += (sl + 8 * j < C) ? d * d : 0.0f;
    }
    red[sl][lp] = q;
    __syncthreads();
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) var += red[i][lp];
    float den = sqrtf(var / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        int c = sl + 8 * j;
        if (ok && c < C) y[(long)c * P + p] = (v[j] - mean) / den * w[c] + b[c];
    }
}

const char *layernorm_kernel_name(int C) {
    return C <= 64 ? "layernorm_reg_kernel<8>" : C <= 512 ? "layernorm_reg_kernel<64>" : C <= 1024 ? "layernorm_reg_kernel<128>" : "layernorm_kernel";
}

hipError_t launch_layernorm(const float *x, float *y, const float *w, const float *b, int C, long P, float eps, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    {
        const dim3 grid((unsigned)((P + 31) / 32));
#define LN(N_) { hipLaunchKernelGGL(layernorm_reg_kernel<N_>, grid, dim3(256), 0, s, x, y, w, b, C, P, eps); return hipGetLastError(); }
        if (C <= 64) LN(8)
        if (C <= 512) LN(64)
        if (C <= 1024) LN(128)
#undef LN
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, x, y, w, b, C, P, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// RVQ encode (speech_tokenizer.rs:494-509, 536-556, 581-585): a workgroup takes FB frames through every layer.
//   proj = input_proj * h (1x1 conv, no bias), then per layer: nearest codeword of the residual, residual -= codeword.
// BIT-EXACT CONTRACT (the encoder test oracle: linear, nearest, rvq_encode): every dot product is a k-ascending
// fmaf chain from 0; |x|^2 and |e|^2 are k-ascending sums of ROUNDED squares; distance = (|x|^2 + |e|^2) - 2*<x,e>;
// argmin = the first minimum.  Contraction is ruled out by spelling every operation as an intrinsic.
// The chains fix the order along k only, so the search is a register-tiled VALU GEMM: a thread owns 4 consecutive
// codewords x FB frames (32 independent chains); per k it loads its 4 codeword values once (transposed codebook:
// 16 contiguous bytes per thread) and the FB residual values as two broadcast ds_read_b128.  (MFMA is not an option
// here: its internal summation is not an fmaf chain.)
// ---------------------------------------------------------------------------------------------------
#ifndef HIPVOC_RVQ_FB
#define HIPVOC_RVQ_FB 8
#endif
static constexpr int FB = HIPVOC_RVQ_FB;    // frames per workgroup
static constexpr int CPT = 4;   // consecutive codewords per thread
static constexpr int PF = 4;    // k-steps of codeword prefetch (register ring)
static constexpr int NT = 512;  // threads: 2048 codewords / CPT.  (256 threads x 8 codewords left 1000 equal, long-running
                                // workgroups for 768 slots: a second round at 30 % fill; 512 threads give ~2 even rounds)

__global__ __launch_bounds__(NT, 4) void rvq_encode_kernel(RvqEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hs = smem;                  // [dim][FB]
    float *xs = hs + a.dim * FB;       // [D][FB] residual
    __shared__ float bd[FB][NT];
    __shared__ int bi[FB][NT];
    __shared__ float xsq_s[FB];
    __shared__ int code_s[FB];
    const int tid = threadIdx.x;
    const long nframes = (long)a.G * a.T, f0 = (long)blockIdx.x * FB;
    const long cs = (long)a.G * a.Lp;
    const int nq = a.n_sem + a.n_ac;
    for (int i = tid; i < a.dim * FB; i += NT) {
        const int k = i / FB, f = i - k * FB;
        const long g = f0 + f;
        float v = 0.0f;
        if (g < nframes) {
            const long u = g / a.T, t = g - u * a.T;
            v = a.h[(long)k * cs + u * a.Lp + t];
        }
        hs[i] = v;
    }
    __syncthreads();
    for (int group = 0; group < 2; group++) {
        const int nl = group ? a.n_ac : a.n_sem;
        if (!nl) continue;
        const float *wp = a.wp_t + (long)group * a.dim * a.D;
        for (int d = tid; d < a.D; d += NT) {
            float p[FB];
#pragma unroll
            for (int f = 0; f < FB; f++) p[f] = 0.0f;
            for (int k = 0; k < a.dim; k++) {
                const float w = wp[(long)k * a.D + d];
                float xv[FB];
#pragma unroll
                for (int f4 = 0; f4 < FB / 4; f4++) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(hs + k * FB + 4 * f4);
                    xv[4 * f4] = x4.x; xv[4 * f4 + 1] = x4.y; xv[4 * f4 + 2] = x4.z; xv[4 * f4 + 3] = x4.w;
                }
#pragma unroll
                for (int f = 0; f < FB; f++) p[f] = __builtin_fmaf(w, xv[f], p[f]);
            }
#pragma unroll
            for (int f = 0; f < FB; f++) {
                xs[d * FB + f] = p[f];
                const long g = f0 + f;
                if (a.proj && g < nframes) {
                    const long u = g / a.T, t = g - u * a.T;
                    a.proj[(long)(group * a.D + d) * cs + u * a.Lp + t] = p[f];
                }
            }
        }
        __syncthreads();
        for (int l = 0; l < nl; l++) {
            const int q = group ? a.n_sem + l : l;
            const float *et = a.emb_t + (long)q * a.D * a.cb_size;
            const float *esq = a.e_sq + (long)q * a.cb_size;
            if (tid < FB) {
                float s = 0.0f;
                for (int d = 0; d < a.D; d++) { const float v = xs[d * FB + tid]; s = __fadd_rn(s, __fmul_rn(v, v)); }
                xsq_s[tid] = s;
            }
            __syncthreads();
            float best[FB];
            int besti[FB];
#pragma unroll
            for (int f = 0; f < FB; f++) { best[f] = INFINITY; besti[f] = 0x7fffffff; }
            const bool vec = (a.cb_size & 3) == 0;
            for (int c0 = tid * CPT; c0 < a.cb_size; c0 += NT * CPT) {      // ascending chunks: first minimum is kept
                float acc[CPT][FB];
#pragma unroll
                for (int j = 0; j < CPT; j++)
#pragma unroll
                    for (int f = 0; f < FB; f++) acc[j][f] = 0.0f;
                auto step = [&](int d, const float *e) {
                    float xv[FB];
#pragma unroll
                    for (int f4 = 0; f4 < FB / 4; f4++) {
                        const float4 x4 = *reinterpret_cast<const float4 *>(xs + d * FB + 4 * f4);
                        xv[4 * f4] = x4.x; xv[4 * f4 + 1] = x4.y; xv[4 * f4 + 2] = x4.z; xv[4 * f4 + 3] = x4.w;
                    }
#pragma unroll
                    for (int j = 0; j < CPT; j++)
#pragma unroll
                        for (int f = 0; f < FB; f++) acc[j][f] = __builtin_fmaf(xv[f], e[j], acc[j][f]);
                };
                if (vec && (a.D % PF) == 0) {
                    // codeword values travel PF k-steps ahead of their use in a register ring: one k-step's 32 FMAs are
                    // ~50 ns, an L2 hit ~600 ns, and four waves per SIMD alone do not cover the difference
                    static_assert(CPT == 4, "one float4 per k-step");
                    float4 ring[PF];
                    const float *ep = et + c0;
#pragma unroll
                    for (int i = 0; i < PF; i++) ring[i] = *reinterpret_cast<const float4 *>(ep + (long)i * a.cb_size);
                    for (int d0 = 0; d0 < a.D; d0 += PF) {
#pragma unroll
                        for (int i = 0; i < PF; i++) {
                            const float4 cur = ring[i];
                            const int dn = d0 + i + PF;
                            if (dn < a.D) ring[i] = *reinterpret_cast<const float4 *>(ep + (long)dn * a.cb_size);
                            const float e[CPT] = {cur.x, cur.y, cur.z, cur.w};
                            step(d0 + i, e);
                        }
                    }
                } else {
                    for (int d = 0; d < a.D; d++) {
                        float e[CPT];
                        const float *ep = et + (long)d * a.cb_size + c0;
#pragma unroll
                        for (int j = 0; j < CPT; j++) e[j] = c0 + j < a.cb_size ? ep[j] : 0.0f;
                        step(d, e);
                    }
                }
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    if (c0 + j < a.cb_size) {
                        const float es = esq[c0 + j];
#pragma unroll
                        for (int f = 0; f < FB; f++) {
                            const float dist = __fsub_rn(__fadd_rn(xsq_s[f], es), __fmul_rn(acc[j][f], 2.0f));
                            if (dist < best[f]) { best[f] = dist; besti[f] = c0 + j; }
                        }
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < FB; f++) { bd[f][tid] = best[f]; bi[f][tid] = besti[f]; }
            __syncthreads();
            for (int sft = NT / 2; sft > 0; sft >>= 1) {
                if (tid < sft) {
#pragma unroll
                    for (int f = 0; f < FB; f++) {
                        const float od = bd[f][tid + sft];
                        const int oi = bi[f][tid + sft];
                        if (od < bd[f][tid] || (od == bd[f][tid] && oi < bi[f][tid])) { bd[f][tid] = od; bi[f][tid] = oi; }
                    }
                }
                __syncthreads();
            }
            if (tid < FB) {
                int code = bi[tid][0];
                if (code == 0x7fffffff) code = 0;                  // every distance NaN / +inf: argmin of candle returns 0
                code_s[tid] = code;
                const long g = f0 + tid;
                if (g < nframes) a.codes[g * nq + q] = code;       // codes[u][t][q], frames are (u, t) flattened
            }
            __syncthreads();
            for (int d = tid; d < a.D; d += NT)
#pragma unroll
                for (int f = 0; f < FB; f++)
                    xs[d * FB + f] = __fsub_rn(xs[d * FB + f], et[(long)d * a.cb_size + code_s[f]]);
            __syncthreads();
        }
    }
}

hipError_t launch_rvq_encode(const RvqEncArgs &a, hipStream_t s) {
    if (a.G <= 0 || a.T <= 0) return hipSuccess;
    size_t lds = sizeof(float) * FB * ((size_t)a.dim + a.D);
    if (lds > 64 * 1024 || a.cb_size < 1 || a.D < 1 || ((uintptr_t)a.emb_t & 15)) return hipErrorInvalidValue;
    const long nframes = (long)a.G * a.T;
    hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)((nframes + FB - 1) / FB)), dim3(NT), lds, s, a);
    return hipGetLastError();
}

}  // namespace hipvoc
