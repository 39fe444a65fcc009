// gfx950 (CDNA4 / MI355X) kernels of the vocoder path.  Hand-written HIP, wave64, fp32 MFMA.
//
// Orientation of every GEMM-shaped op: the A operand of v_mfma_f32_32x32x2_f32 holds WEIGHTS
// (row i = output channel), the B operand holds ACTIVATIONS (column j = time position), so the
// accumulator tile has time on the lane and output channels in the 16 registers.  With the
// channel-first / time-contiguous activation layout this makes every global load and store of an
// activation a run of consecutive floats across a wavefront, lets the 7 taps of a causal conv
// re-read ONE LDS halo tile at shifted offsets, and evaluates SnakeBeta once per staged element.
#include "kernels.h"

namespace hipvoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BM = 256;   // time positions per workgroup (4 waves x 64)
static constexpr int BK = 32;    // input channels per K stage

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// -------------------------------------------------------------------------------------------------
// conv_gemm: LDS-tiled implicit GEMM on v_mfma_f32_32x32x2_f32.
//   workgroup tile: BN = 32*TN rows (output channels x phases)  x  BM = 256 time positions
//   wave w: all BN rows x positions [64w, 64w+64)  -> TN x 2 accumulator tiles of 32x32
//   K loop: for each block of 32 input channels: stage the halo tile Xs[32][XW] once (SnakeBeta
//   applied on the way in), then for each tap stage Ws[32][BN] (double-buffered, next tap
//   prefetched into registers during the MFMAs) and run 16 k-steps reading Xs at a shifted offset.
// -------------------------------------------------------------------------------------------------
template <int TN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit,
                                                            int n_tiles) {
    constexpr int BN = TN * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                    // [BK][XW]
    float *Ws = smem + BK * XW;          // [2][BK][BN]

    // XCD-aware remap (bijective): blocks that share an activation tile (same position tile,
    // different row tiles) get consecutive ids on one XCD, so the tile is fetched into one L2.
    int nwg = gridDim.x, orig = blockIdx.x;
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * BM;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;

    const float *xu = a.x + (long)unit * a.x_us;
    const int ncb = (a.Cin + BK - 1) / BK;
    const int CinP = ncb * BK;
    const int nstage = ncb * a.ntaps;

    f32x16 acc[TN][2];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // W staging: tile [BK][BN] = 8*BN float4, 256 threads -> BN/32 = TN float4 each (named
    // registers: an indexed float4 array here ends up in scratch)
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0, w3 = w0;
    const int wk0 = tid / (BN / 4), wc0 = (tid - wk0 * (BN / 4)) * 4;                    // idx = tid
    const int wk1 = (tid + 256) / (BN / 4), wc1 = (tid + 256 - wk1 * (BN / 4)) * 4;      // idx = tid + 256
    const int wk2 = (tid + 512) / (BN / 4), wc2 = (tid + 512 - wk2 * (BN / 4)) * 4;
    const int wk3 = (tid + 768) / (BN / 4), wc3 = (tid + 768 - wk3 * (BN / 4)) * 4;
#define W_LOAD(stage_)                                                                             \
    {                                                                                              \
        int cb_ = (stage_) / a.ntaps, tap_ = (stage_)-cb_ * a.ntaps;                               \
        const float *wp_ = a.w + ((long)tap_ * CinP + (long)cb_ * BK) * a.NP + n0;                 \
        w0 = *reinterpret_cast<const float4 *>(wp_ + (long)wk0 * a.NP + wc0);                      \
        if constexpr (TN > 1) w1 = *reinterpret_cast<const float4 *>(wp_ + (long)wk1 * a.NP + wc1); \
        if constexpr (TN > 2) w2 = *reinterpret_cast<const float4 *>(wp_ + (long)wk2 * a.NP + wc2); \
        if constexpr (TN > 3) w3 = *reinterpret_cast<const float4 *>(wp_ + (long)wk3 * a.NP + wc3); \
    }
#define W_STORE(buf_)                                                                              \
    {                                                                                              \
        float *ws_ = Ws + (buf_) * (BK * BN) + tid * 4;                                            \
        *reinterpret_cast<float4 *>(ws_) = w0;                                                     \
        if constexpr (TN > 1) *reinterpret_cast<float4 *>(ws_ + 1024) = w1;                        \
        if constexpr (TN > 2) *reinterpret_cast<float4 *>(ws_ + 2048) = w2;                        \
        if constexpr (TN > 3) *reinterpret_cast<float4 *>(ws_ + 3072) = w3;                        \
    }
    // X staging: rows k = wave, wave+4, ...; lanes sweep the row in float4 steps.
    auto x_stage = [&](int cb) {
        const int xw4 = XW >> 2;
        for (int k = wave; k < BK; k += 4) {
            int ci = cb * BK + k;
            bool cok = ci < a.Cin;
            const float *xr = xu + (long)ci * a.x_cs;
            float sa = 0.0f, sib = 0.0f;
            if (a.sn_a && cok) { sa = a.sn_a[ci]; sib = a.sn_ib[ci]; }
            for (int c4 = lane; c4 < xw4; c4 += 64) {
                int pos = m0 - HL + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cok) {
                    if (pos >= 0 && pos + 4 <= a.Lin) {
                        v = *reinterpret_cast<const float4 *>(xr + pos);
                    } else {
                        if (pos + 0 >= 0 && pos + 0 < a.Lin) v.x = xr[pos + 0];
                        if (pos + 1 >= 0 && pos + 1 < a.Lin) v.y = xr[pos + 1];
                        if (pos + 2 >= 0 && pos + 2 < a.Lin) v.z = xr[pos + 2];
                        if (pos + 3 >= 0 && pos + 3 < a.Lin) v.w = xr[pos + 3];
                    }
                    if (a.sn_a) {   // SnakeBeta: x + 1/(e^beta+1e-9) * sin(x e^alpha)^2 ; snake(0) == 0
                        float s;
                        s = sinf(v.x * sa); v.x = v.x + sib * (s * s);
                        s = sinf(v.y * sa); v.y = v.y + sib * (s * s);
                        s = sinf(v.z * sa); v.z = v.z + sib * (s * s);
                        s = sinf(v.w * sa); v.w = v.w + sib * (s * s);
                    }
                }
                *reinterpret_cast<float4 *>(Xs + k * XW + c4 * 4) = v;
            }
        }
    };

    W_LOAD(0);
    W_STORE(0);
    for (int st = 0; st < nstage; st++) {
        int cb = st / a.ntaps, tap = st - cb * a.ntaps;
        if (tap == 0) {
            __syncthreads();          // every wave is done reading the previous Xs
            x_stage(cb);
        }
        if (st + 1 < nstage) W_LOAD(st + 1);
        __syncthreads();              // Ws[st&1] and Xs are visible
        {
            const float *ws = Ws + (st & 1) * (BK * BN) + l31;
            const float *xs = Xs + wave * 64 + l31 + HL + a.shift0 + tap * a.shift_step;
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) {
                int k = 2 * kk + h;
                float av[TN], bv[2];
#pragma unroll
                for (int i = 0; i < TN; i++) av[i] = ws[k * BN + i * 32];
#pragma unroll
                for (int j = 0; j < 2; j++) bv[j] = xs[k * XW + j * 32];
#pragma unroll
                for (int i = 0; i < TN; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if (st + 1 < nstage) W_STORE((st + 1) & 1);
    }

    // epilogue: bias, activation, per-channel scale, residual, store (time on the lane: coalesced)
    float *yu = a.y + (long)unit * a.y_us;
    const float *ru = a.res ? a.res + (long)unit * a.y_us : nullptr;
#pragma unroll
    for (int i = 0; i < TN; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (n >= a.N) continue;
            int co = n, ph = 0;
            if (a.S > 1) { co = n / a.S; ph = n - co * a.S; }
            float bb = a.bias ? a.bias[co] : 0.0f;
            float sc = a.rscale ? a.rscale[co] : 1.0f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                int m = m0 + wave * 64 + j * 32 + l31;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bb;
                if (a.act == ACT_GELU) v = gelu_erf(v);
                if (a.rscale) v = v * sc;
                long o = (long)co * a.y_cs + (long)m * a.S + ph;
                if (ru) v = ru[o] + v;
                if (a.act == ACT_CLAMP1) v = fminf(fmaxf(v, -1.0f), 1.0f);
                yu[o] = v;
            }
        }
    }
}

static void conv_gemm_geometry(const ConvGemmArgs &a, int &HL, int &XW) {
    int smin = a.shift0, smax = a.shift0;
    int last = a.shift0 + (a.ntaps - 1) * a.shift_step;
    if (last < smin) smin = last;
    if (last > smax) smax = last;
    int hl = smin < 0 ? -smin : 0, hr = smax > 0 ? smax : 0;
    HL = (hl + 3) & ~3;
    XW = BM + HL + ((hr + 3) & ~3);
}

size_t conv_gemm_lds_bytes(const ConvGemmArgs &a) {
    int HL, XW;
    conv_gemm_geometry(a, HL, XW);
    return sizeof(float) * ((size_t)BK * XW + 2 * (size_t)BK * 32 * a.TN);
}

hipError_t launch_conv_gemm(const ConvGemmArgs &a, hipStream_t s) {
    int HL, XW;
    conv_gemm_geometry(a, HL, XW);
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    // host-side shape checks: what the kernel's vector accesses assume
    if ((a.x_cs & 3) || (a.x_us & 3) || (a.NP % (32 * a.TN)) || a.NP < a.N || a.TN < 1 || a.TN > 4 ||
        ((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || a.ntaps < 1 || a.S < 1)
        return hipErrorInvalidValue;
    int tiles_per_unit = (a.M + BM - 1) / BM;
    int n_tiles = a.NP / (32 * a.TN);
    long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    size_t lds = conv_gemm_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    {   // dynamic LDS above 64 KiB must be enabled per function and device
        static bool done[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !done[dev]) {
            const int cap = 160 * 1024;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e != hipSuccess) return e;
            done[dev] = true;
        }
    }
    dim3 g((unsigned)grid), b(256);
    switch (a.TN) {
        case 1: hipLaunchKernelGGL(conv_gemm_kernel<1>, g, b, lds, s, a, HL, XW, tiles_per_unit, n_tiles); break;
        case 2: hipLaunchKernelGGL(conv_gemm_kernel<2>, g, b, lds, s, a, HL, XW, tiles_per_unit, n_tiles); break;
        case 3: hipLaunchKernelGGL(conv_gemm_kernel<3>, g, b, lds, s, a, HL, XW, tiles_per_unit, n_tiles); break;
        default: hipLaunchKernelGGL(conv_gemm_kernel<4>, g, b, lds, s, a, HL, XW, tiles_per_unit, n_tiles); break;
    }
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// RVQ dequantisation: gather + ordered sum (bit-exact contract), then the two 1x1 projections as
// k-ascending fmaf chains.  One workgroup per frame.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rvq_kernel(RvqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s0 = smem, *s1 = smem + a.D;
    int u = blockIdx.x / a.T, t = blockIdx.x - u * a.T;
    const long long *cp = a.codes + ((long)u * a.T + t) * a.nq;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        float v0 = 0.0f, v1 = 0.0f;
        for (int q = 0; q < a.nq; q++) {
            long long c = cp[q];
            if (c < 0 || c >= a.cb_size) {
                if (d == 0) *a.err = 1;
                c = 0;
            }
            float e = a.embed[((long)q * a.cb_size + c) * a.D + d];
            if (q == 0) v0 = 0.0f + e;
            else v1 = v1 + e;
        }
        s0[d] = v0;
        s1[d] = v1;
        if (a.sums) {
            a.sums[(long)d * a.G * a.Lp + (long)u * a.Lp + t] = v0;
            a.sums[(long)(a.D + d) * a.G * a.Lp + (long)u * a.Lp + t] = v1;
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < a.Q; n += blockDim.x) {
        float p = 0.0f, r = 0.0f;
        for (int k = 0; k < a.D; k++) p = __builtin_fmaf(a.wf_t[(long)k * a.Q + n], s0[k], p);
        for (int k = 0; k < a.D; k++) r = __builtin_fmaf(a.wr_t[(long)k * a.Q + n], s1[k], r);
        a.out[(long)n * a.G * a.Lp + (long)u * a.Lp + t] = p + r;
    }
}

hipError_t launch_rvq(const RvqArgs &a, hipStream_t s) {
    if (a.G <= 0 || a.T <= 0) return hipSuccess;
    hipLaunchKernelGGL(rvq_kernel, dim3(a.G * a.T), dim3(256), sizeof(float) * 2 * a.D, s, a);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// RMSNorm over channels; one thread per position, lanes on consecutive positions.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float *x, float *y, const float *w, int C, long P,
                                                       float eps) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float ss = 0.0f;
    for (int c = 0; c < C; c++) {
        float v = x[(long)c * P + p];
        ss += v * v;
    }
    float r = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = 0; c < C; c++) y[(long)c * P + p] = w[c] * (x[(long)c * P + p] * r);
}

hipError_t launch_rmsnorm(const float *x, float *y, const float *w, int C, long P, float eps, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, s, x, y, w, C, P, eps);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Attention: one workgroup per (unit, head, block of 32 queries).  RoPE (rotate-half) is applied
// while q and k are staged into LDS.  Mask: kv <= q && kv > q - window.  fp32 softmax.
// -------------------------------------------------------------------------------------------------
static constexpr int QB = 32;

__global__ __launch_bounds__(256) void attention_kernel(const float *qkv, float *out, const float *cos_t,
                                                         const float *sin_t, int nh, int hd, int G, int T, int Lp,
                                                         int window, int NKP) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Qs = smem;                 // [hd][QB]
    float *Ks = Qs + hd * QB;         // [hd][NKP]
    float *Vs = Ks + hd * NKP;        // [hd][NKP]
    float *Ps = Vs + hd * NKP;        // [QB][NKP]
    int u = blockIdx.x, hh = blockIdx.y, q0 = blockIdx.z * QB;
    int nq = min(QB, T - q0);
    int k_lo = max(0, q0 - window + 1);
    int nk = q0 + nq - k_lo;
    long cs = (long)G * Lp;
    long base = (long)u * Lp;
    const float *qp = qkv + (long)(hh * hd) * cs + base;
    const float *kp = qkv + (long)((nh + hh) * hd) * cs + base;
    const float *vp = qkv + (long)((2 * nh + hh) * hd) * cs + base;
    int half = hd >> 1;
    int tid = threadIdx.x;
    // stage q (rope), k (rope), v
    for (int i = tid; i < half * QB; i += 256) {
        int d = i / QB, tq = i - d * QB;
        float lo = 0.f, hi = 0.f;
        if (tq < nq) {
            int t = q0 + tq;
            float a = qp[(long)d * cs + t], b = qp[(long)(d + half) * cs + t];
            float c = cos_t[t * half + d], s = sin_t[t * half + d];
            lo = a * c + (-b) * s;
            hi = b * c + a * s;
        }
        Qs[d * QB + tq] = lo;
        Qs[(d + half) * QB + tq] = hi;
    }
    for (int i = tid; i < half * nk; i += 256) {
        int d = i / nk, tk = i - d * nk;
        int t = k_lo + tk;
        float a = kp[(long)d * cs + t], b = kp[(long)(d + half) * cs + t];
        float c = cos_t[t * half + d], s = sin_t[t * half + d];
        Ks[d * NKP + tk] = a * c + (-b) * s;
        Ks[(d + half) * NKP + tk] = b * c + a * s;
    }
    for (int i = tid; i < hd * nk; i += 256) {
        int d = i / nk, tk = i - d * nk;
        Vs[d * NKP + tk] = vp[(long)d * cs + k_lo + tk];
    }
    __syncthreads();
    // scores + softmax: 8 threads per query row
    int tq = tid >> 3, l8 = tid & 7;
    float scaling = 1.0f / sqrtf((float)hd);
    int t_abs = q0 + tq;
    float mx = -INFINITY;
    if (tq < nq) {
        for (int tk = l8; tk < nk; tk += 8) {
            int kt = k_lo + tk;
            bool ok = kt <= t_abs && kt > t_abs - window;
            float sc = -INFINITY;
            if (ok) {
                float acc = 0.0f;
                for (int d = 0; d < hd; d++) acc += Qs[d * QB + tq] * Ks[d * NKP + tk];
                sc = acc * scaling;
            }
            Ps[tq * NKP + tk] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    float den = 0.0f;
    if (tq < nq) {
        for (int tk = l8; tk < nk; tk += 8) {
            float sc = Ps[tq * NKP + tk];
            float e = sc == -INFINITY ? 0.0f : expf(sc - mx);
            Ps[tq * NKP + tk] = e;
            den += e;
        }
    }
    den += __shfl_xor(den, 1);
    den += __shfl_xor(den, 2);
    den += __shfl_xor(den, 4);
    __syncthreads();
    if (tq < nq) {
        float inv = 1.0f / den;
        for (int d = l8; d < hd; d += 8) {
            float acc = 0.0f;
            for (int tk = 0; tk < nk; tk++) acc += (Ps[tq * NKP + tk] * inv) * Vs[d * NKP + tk];
            out[(long)(hh * hd + d) * cs + base + t_abs] = acc;
        }
    }
}

hipError_t launch_attention(const float *qkv, float *out, const float *cos_t, const float *sin_t, int nh, int hd,
                            int G, int T, int Lp, int window, hipStream_t s) {
    if (G <= 0) return hipSuccess;
    int nkmax = QB + (window > 1 ? window - 1 : 0);
    if (nkmax > T) nkmax = T;
    int NKP = nkmax | 1;    // odd stride: conflict-free column walks
    size_t lds = sizeof(float) * ((size_t)hd * QB + 2 * (size_t)hd * NKP + (size_t)QB * NKP);
    if (lds > 160 * 1024 || (hd & 1)) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    dim3 g(G, nh, (T + QB - 1) / QB);
    hipLaunchKernelGGL(attention_kernel, g, dim3(256), lds, s, qkv, out, cos_t, sin_t, nh, hd, G, T, Lp, window, NKP);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_kernel(float *g, const float *u, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        float4 a = *reinterpret_cast<float4 *>(g + i);
        float4 b = *reinterpret_cast<const float4 *>(u + i);
        a.x = a.x / (1.0f + expf(-a.x)) * b.x;
        a.y = a.y / (1.0f + expf(-a.y)) * b.y;
        a.z = a.z / (1.0f + expf(-a.z)) * b.z;
        a.w = a.w / (1.0f + expf(-a.w)) * b.w;
        *reinterpret_cast<float4 *>(g + i) = a;
    } else {
        for (; i < n; i++) g[i] = g[i] / (1.0f + expf(-g[i])) * u[i];
    }
}

hipError_t launch_swiglu(float *g, const float *u, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    long nth = (n + 3) / 4;
    hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s, g, u, n);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// ConvNeXt front half: depthwise causal conv (k taps, left pad k-1) + LayerNorm over channels.
// One thread per (unit, position); lanes on consecutive positions.  y doubles as scratch for the
// conv output between the passes (each thread re-reads only what it wrote).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dwconv_ln_kernel(const float *x, float *y, const float *dw_w,
                                                        const float *dw_b, const float *ln_w, const float *ln_b,
                                                        int C, int K, int G, int L, int Lp, float eps) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int u = blockIdx.y;
    if (t >= L) return;
    long cs = (long)G * Lp, base = (long)u * Lp + t;
    float mean = 0.0f;
    for (int c = 0; c < C; c++) {
        float acc = dw_b[c];
        const float *xr = x + (long)c * cs + (long)u * Lp;
        for (int j = 0; j < K; j++) {
            int sidx = t - (K - 1 - j);
            if (sidx >= 0) acc += dw_w[c * K + j] * xr[sidx];
        }
        y[(long)c * cs + base] = acc;
        mean += acc;
    }
    mean /= (float)C;
    float var = 0.0f;
    for (int c = 0; c < C; c++) {
        float d = y[(long)c * cs + base] - mean;
        var += d * d;
    }
    float r = 1.0f / sqrtf(var / (float)C + eps);
    for (int c = 0; c < C; c++) {
        float v = y[(long)c * cs + base];
        y[(long)c * cs + base] = (v - mean) * r * ln_w[c] + ln_b[c];
    }
}

hipError_t launch_dwconv_ln(const float *x, float *y, const float *dw_w, const float *dw_b, const float *ln_w,
                            const float *ln_b, int C, int K, int G, int L, int Lp, float eps, hipStream_t s) {
    if (G <= 0 || L <= 0) return hipSuccess;
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3((L + 63) / 64, G), dim3(64), 0, s, x, y, dw_w, dw_b, ln_w, ln_b, C, K,
                       G, L, Lp, eps);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcm16_kernel(const float *x, int16_t *y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i] * 32767.0f;
    int16_t o = 0;
    if (v == v) {                       // NaN -> 0 (Rust `as i16`)
        v = fminf(fmaxf(v, -32768.0f), 32767.0f);
        o = (int16_t)(int)v;            // truncation toward zero
    }
    y[i] = o;
}

hipError_t launch_pcm16(const float *x, int16_t *y, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

}  // namespace hipvoc
