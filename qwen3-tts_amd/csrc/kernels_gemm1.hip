// gfx950 (CDNA4 / MI355X): the 1-tap GEMMs of the path (transformer projections, ConvNeXt pointwise convs) when the
// launch is SMALL -- a single utterance (BASELINE configs[1]) or a few units.
//
// conv_gemm_kernel (kernels.hip) stages 16 input channels per barrier and prefetches one stage ahead: right when
// several workgroups share a CU and hide each other's memory latency, wrong when the whole launch is a few dozen
// workgroups -- then every one of the K/16 stages waits a full L2/HBM round trip (1.4 us measured; o_proj with
// K = 1024 took 90 us at ANY batch size up to 64 units).  This kernel makes the same products with 64 input channels
// per stage (a quarter of the round trips, 32 MFMAs per tile and wave between barriers) and 128 positions per
// workgroup (4 waves: twice the workgroups on a small launch).
//
// Same v_mfma_f32_32x32x2_f32 chain per output element, k ascending in the same (2kk + h) pairs, the same epilogue
// (device_common.h): the results are bit-identical to conv_gemm_kernel's, so which of the two runs may depend on the
// launch size without breaking batch == single (tests/test_gpu_parity.py, test_gemm1_bitwise).
#include "kernels.h"
#include "device_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

static constexpr int G1_BK = 64;     // input channels per stage
static constexpr int G1_BM = 128;    // positions per workgroup (4 waves x 32)
static constexpr int G1_NTH = 256;

template <int TN, bool ENC>
__global__ __launch_bounds__(G1_NTH, 2) void gemm1_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    constexpr int BN = TN * 32;
    constexpr int XP = G1_BK * (G1_BM / 4) / G1_NTH;   // float4 of X per thread and stage: 8
    constexpr int WP = G1_BK * (BN / 4) / G1_NTH;      // float4 of W per thread and stage: 2 TN
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                               // [4][BN] parameter table
    float *Xs = smem_all + 4 * BN;                      // [2][64][128]
    float *Ws = Xs + 2 * G1_BK * G1_BM;                 // [2][64][BN]

    int nwg = gridDim.x, orig = blockIdx.x;             // XCD-aware remap, as conv_gemm_kernel
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * G1_BM;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int CinP = ((a.Cin + 31) / 32) * 32;          // rows of the packed weight tensor
    const int nch = (a.Cin + G1_BK - 1) / G1_BK;

    fill_param_table<BN, false>(a, Pm, n0, tid, G1_NTH);

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // X piece j of this thread: row (tid >> 5) + 8 j of the stage, float4 column tid & 31.  Loads are unconditional
    // from clamped addresses; rows >= Cin and positions >= Lin are zeroed at commit.
    const int c4 = tid & 31, xr0 = tid >> 5;
    const int xlim = ((a.Lin + 3) & ~3) - 4;
    const int pos = m0 + 4 * c4;
    const int posc = pos > xlim ? xlim : pos;
    // W piece j: float4 index tid + 256 j of the [64][BN] tile
    constexpr int WC4 = BN / 4;
    float4 xr[XP], wr[WP];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int j = 0; j < XP; j++) {
            int row = ch * G1_BK + xr0 + 8 * j;
            row = row < a.Cin ? row : 0;
            xr[j] = *reinterpret_cast<const float4 *>(xu + (long)row * a.x_cs + posc);
        }
#pragma unroll
        for (int j = 0; j < WP; j++) {
            const int q = tid + G1_NTH * j;
            int row = ch * G1_BK + q / WC4;
            row = row < CinP ? row : 0;
            wr[j] = *reinterpret_cast<const float4 *>(a.w + (long)row * a.NP + n0 + (q % WC4) * 4);
        }
    };
    auto commit = [&](int ch, int buf) {
#pragma unroll
        for (int j = 0; j < XP; j++) {
            const int row = ch * G1_BK + xr0 + 8 * j;
            const bool ok = row < a.Cin;
            float4 v = xr[j];
            v.x = ok && pos + 0 < a.Lin ? v.x : 0.0f;
            v.y = ok && pos + 1 < a.Lin ? v.y : 0.0f;
            v.z = ok && pos + 2 < a.Lin ? v.z : 0.0f;
            v.w = ok && pos + 3 < a.Lin ? v.w : 0.0f;
            *reinterpret_cast<float4 *>(Xs + (buf * G1_BK + xr0 + 8 * j) * G1_BM + c4 * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < WP; j++) {
            const int q = tid + G1_NTH * j;
            const bool ok = ch * G1_BK + q / WC4 < CinP;
            float4 v = wr[j];
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(Ws + buf * G1_BK * BN + q * 4) = v;
        }
    };

    fetch(0);
    commit(0, 0);
    for (int ch = 0; ch < nch; ch++) {
        // unconditional (the last stage re-fetches itself into the idle buffer): see conv_gemm_kernel
        const int nx = ch + 1 < nch ? ch + 1 : ch;
        fetch(nx);
        __syncthreads();                                 // stage ch visible; the readers of the other buffer are past
        {
            // LDS reads run one batch of eight k-pairs ahead of the MFMAs (a small launch has ONE wave per SIMD: nothing
            // else hides the ds_read latency), and the commit below must not be hoisted into the MFMAs -- it waits for
            // the global loads of the next stage, whose latency the MFMAs are there to cover.
            const float *ws = Ws + (ch & 1) * G1_BK * BN + l31;
            const float *xs = Xs + (ch & 1) * G1_BK * G1_BM + wave * 32 + l31;
            constexpr int KB = 8, NB = G1_BK / 2 / KB;
            float av[2][KB][TN], bv[2][KB];
            auto lds_batch = [&](int b, int sl) {
#pragma unroll
                for (int q = 0; q < KB; q++) {
                    const int k = 2 * (b * KB + q) + h;
                    bv[sl][q] = xs[k * G1_BM];
#pragma unroll
                    for (int i = 0; i < TN; i++) av[sl][q][i] = ws[k * BN + i * 32];
                }
            };
            lds_batch(0, 0);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b + 1 < NB) lds_batch(b + 1, (b + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < KB; q++)
#pragma unroll
                    for (int i = 0; i < TN; i++)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][q][i], bv[b & 1][q], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        commit(nx, (ch + 1) & 1);
    }

    float *yu = a.y + (long)unit * a.y_us;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, ENC>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u);
}

// ---------------------------------------------------------------------------------------------------
static bool g1_enc(const ConvGemmArgs &a) { return a.act == ACT_ELU || a.act == ACT_GELU_TANH; }

// what the kernel supports: fp32, one tap at shift 0 (plain or transposed: rows = channels x phases), no input activation, no fusion, no s2d
static bool g1_supported(const ConvGemmArgs &a) {
    return a.w && !a.wb && !a.w2 && !a.wb2 && !a.xb && !a.yb && a.ntaps == 1 && a.shift0 == 0 && a.S >= 1 && !a.sn_a &&
           !a.in_elu && a.s2d <= 1 && !(a.x_cs & 3) && !(a.x_us & 3) && !((uintptr_t)a.x & 15) && !((uintptr_t)a.w & 15) &&
           a.TN >= 1 && a.TN <= 4 && !(a.NP % (32 * a.TN)) && a.NP >= a.N && a.Lin >= 1 && (!a.y2 || (a.psn_a && a.psn_ib));
}

static int g1_tn(const ConvGemmArgs &a) {
    const long tiles_m = (long)((a.M + G1_BM - 1) / G1_BM) * a.units;
    int tn = a.TN;
    while (tn > 1 && tiles_m * (a.NP / (32 * tn)) < 512) {
        int t2 = tn == 4 ? 2 : 1;
        if (a.NP % (32 * t2)) break;
        tn = t2;
    }
    return tn;
}

// Used when conv_gemm_kernel's own grid would be at most HIPVOC_GEMM1_WGS workgroups (0 disables the kernel).
bool gemm1_selected(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_GEMM1_WGS");      // read per call: a test toggles it between two handles
    const int max_wgs = e ? atoi(e) : 384;
    if (max_wgs <= 0 || !g1_supported(a)) return false;
    // layers the LDS-DMA kernels take (activated input, Cin % 16 == 0, stride 1) are theirs at every size: gemm_dmas_kernel on small
    // launches, gemm_dma_kernel above (the transformer's projections at 64 units: 15-35 % faster there than here).  HIPVOC_GEMM1_DMA=1
    // keeps them here.
    if (gemmd_selected(a) && !getenv("HIPVOC_GEMM1_DMA")) return false;
    const long tiles_m = (long)((a.M + 255) / 256) * a.units;
    return tiles_m * (a.NP / (32 * detail::conv_gemm_launch_tn(a))) <= max_wgs;
}

const char *gemm1_kernel_name(const ConvGemmArgs &a) {
    static char names[2][8][32];
    const int tn = g1_tn(a), enc = g1_enc(a) ? 1 : 0;
    char *n = names[enc][tn & 7];
    if (!n[0]) snprintf(n, 32, "gemm1_kernel<%d,%s>", tn, enc ? "true" : "false");
    return n;
}

template <int TN, bool ENC>
static hipError_t gemm1_launch_one(const ConvGemmArgs &a, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm1_kernel<TN, ENC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((gemm1_kernel<TN, ENC>), dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_gemm1(const ConvGemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    if (!g1_supported(a)) return hipErrorInvalidValue;
    const int tn = g1_tn(a);
    const int tiles_per_unit = (a.M + G1_BM - 1) / G1_BM;
    const int n_tiles = a.NP / (32 * tn);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (4 * 32 * (size_t)tn + 2 * G1_BK * (G1_BM + 32 * (size_t)tn));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define G1(TN_)                                                                                         \
    return g1_enc(a) ? gemm1_launch_one<TN_, true>(a, tiles_per_unit, n_tiles, grid, lds, s)            \
                     : gemm1_launch_one<TN_, false>(a, tiles_per_unit, n_tiles, grid, lds, s)
    switch (tn) {
        case 1: G1(1);
        case 2: G1(2);
        case 3: G1(3);
        default: G1(4);
    }
#undef G1
}

// ===================================================================================================
// Multi-tap convs (conv7 of the decoder blocks, transposed convs, pre_conv) on a small launch.  conv_gemm_kernel
// stages ONE tap's 16 x BN weight tile per barrier, one stage ahead: 8 TN MFMAs (0.2 TN us) against a 1.4 us memory
// round trip -- fine under the other workgroups of a busy CU, 470 us for b0's conv7 (336 stages) when the launch is
// a few hundred workgroups.  Here a stage is a channel block of 16 with ALL its taps (conv7: 56 TN MFMAs per barrier),
// 128 positions per workgroup, and the products are made in conv_gemm_kernel's order (channel block, tap, k pair),
// so the results are bit-identical.  (The CB = 4 form for the 2-tap transposed convs was measured: 1.1-2.1x SLOWER than
// conv_gemm_kernel on a single utterance -- their N = channels x stride makes many row tiles re-read X -- and is not
// instantiated.)
// ===================================================================================================
template <int TN, int CB>
__global__ __launch_bounds__(G1_NTH, (CB == 1 && TN == 1) ? 3 : 2) void convs_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    constexpr int BN = TN * 32;
    constexpr int KR = CB * 16;                         // input channels per stage
    constexpr int XPM = CB == 1 ? 4 : 10;               // float4 of X per thread and stage (host checks the fit)
    constexpr int WPM = 4 * TN;                         // float4 of W per thread and stage: CB * ntaps <= 8 slabs of [16][BN]
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;
    float *Xs = smem_all + 4 * BN;                      // [2][KR][XW]
    const int xtile = KR * XW;
    const int slabs = CB * a.ntaps;
    const int wtile = slabs * 16 * BN;
    float *Ws = Xs + 2 * xtile;                         // [2][slabs][16][BN]

    int nwg = gridDim.x, orig = blockIdx.x;
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * G1_BM;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int CinP = ((a.Cin + 31) / 32) * 32;
    const int nst = (a.Cin + KR - 1) / KR;
    const int xw4 = XW >> 2;
    const int xlim = ((a.Lin + 3) & ~3) - 4;

    fill_param_table<BN, false>(a, Pm, n0, tid, G1_NTH);

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // ---- per-thread piece geometry (loop-invariant)
    int xk[XPM], xps[XPM];                              // stage row (or -1: no such piece), unclamped position
    unsigned xoff[XPM];
#pragma unroll
    for (int j = 0; j < XPM; j++) {
        const int p = tid + G1_NTH * j;
        const int row = p / xw4, c4 = p - row * xw4;
        const bool ex = row < KR;
        int pos = m0 - HL + 4 * c4;
        xk[j] = ex ? row : -1;
        xps[j] = pos;
        pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
        xoff[j] = (unsigned)(ex ? row : 0) * (unsigned)a.x_cs + (unsigned)pos;
    }
    int wrow[WPM], wk[WPM];                             // weight row within the tensor at stage 0; channel within the stage (or -1)
#pragma unroll
    for (int j = 0; j < WPM; j++) {
        const int q = tid + G1_NTH * j;
        const int slab = q / (4 * BN), rem = q - slab * (4 * BN);
        const int k = rem / (BN / 4);
        const bool ex = slab < slabs;
        const int cbi = ex ? slab / a.ntaps : 0, tap = ex ? slab - cbi * a.ntaps : 0;
        wk[j] = ex ? cbi * 16 + k : -1;
        wrow[j] = tap * CinP + cbi * 16 + k;
    }
    float4 xr[XPM], wr[WPM];
    float2 sp[XPM];
    auto fetch = [&](int st) {
        const bool full = (st + 1) * KR <= a.Cin;       // wave-uniform
#pragma unroll
        for (int j = 0; j < XPM; j++) {
            int ci = st * KR + (xk[j] < 0 ? 0 : xk[j]);
            if (full) {
                xr[j] = *reinterpret_cast<const float4 *>(xu + (long)st * KR * a.x_cs + xoff[j]);
            } else {
                ci = ci < a.Cin ? ci : 0;
                int pos = xps[j];
                pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
                xr[j] = *reinterpret_cast<const float4 *>(xu + (long)ci * a.x_cs + pos);
            }
            ci = ci < a.Cin ? ci : 0;
            sp[j] = a.sn_a ? make_float2(a.sn_a[ci], a.sn_ib[ci]) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < WPM; j++) {
            const int q = tid + G1_NTH * j;
            const int rem = q % (4 * BN);
            const bool ok = wk[j] >= 0 && st * KR + wk[j] < CinP;
            const long row = ok ? (long)wrow[j] + (long)st * KR : 0;
            wr[j] = *reinterpret_cast<const float4 *>(a.w + row * a.NP + n0 + (rem % (BN / 4)) * 4);
        }
    };
    auto commit = [&](int st, int buf) {
#pragma unroll
        for (int j = 0; j < XPM; j++) {
            if (xk[j] < 0) continue;
            const int pos = xps[j];
            const bool ok = st * KR + xk[j] < a.Cin && pos >= 0;     // pos is a multiple of 4
            float4 v = xr[j];
            v.x = ok && pos + 0 < a.Lin ? v.x : 0.0f;
            v.y = ok && pos + 1 < a.Lin ? v.y : 0.0f;
            v.z = ok && pos + 2 < a.Lin ? v.z : 0.0f;
            v.w = ok && pos + 3 < a.Lin ? v.w : 0.0f;
            if (a.sn_a) {
                f32x2 lo = snake_beta2(mk2(v.x, v.y), (f32x2)(sp[j].x), (f32x2)(sp[j].y));
                f32x2 hi = snake_beta2(mk2(v.z, v.w), (f32x2)(sp[j].x), (f32x2)(sp[j].y));
                v = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            *reinterpret_cast<float4 *>(Xs + buf * xtile + (tid + G1_NTH * j) * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < WPM; j++) {
            if (wk[j] < 0) continue;
            float4 v = wr[j];
            if (st * KR + wk[j] >= CinP) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(Ws + buf * wtile + (tid + G1_NTH * j) * 4) = v;
        }
    };

    fetch(0);
    commit(0, 0);
    for (int st = 0; st < nst; st++) {
        const int nx = st + 1 < nst ? st + 1 : st;
        fetch(nx);
        __syncthreads();
        {
            const float *wsb = Ws + (st & 1) * wtile + l31;
            const float *xsb = Xs + (st & 1) * xtile + wave * 32 + l31 + HL + a.shift0;
            float av[2][8][TN], bv[2][8];
            auto lds_slab = [&](int sl, int set) {     // slab = (channel block, tap): eight k pairs
                const int cbi = sl / a.ntaps, tap = sl - cbi * a.ntaps;
                const float *ws = wsb + sl * 16 * BN;
                const float *xs = xsb + cbi * 16 * XW + tap * a.shift_step;
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const int k = 2 * kk + h;
                    bv[set][kk] = xs[k * XW];
#pragma unroll
                    for (int i = 0; i < TN; i++) av[set][kk][i] = ws[k * BN + i * 32];
                }
            };
            lds_slab(0, 0);
            for (int sl = 0; sl < slabs; sl += 2) {
                lds_slab(sl + 1 < slabs ? sl + 1 : sl, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 8; kk++)
#pragma unroll
                    for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][kk][i], bv[0][kk], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (sl + 1 < slabs) {
                    lds_slab(sl + 2 < slabs ? sl + 2 : sl + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 8; kk++)
#pragma unroll
                        for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][kk][i], bv[1][kk], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        commit(nx, (st + 1) & 1);
    }

    float *yu = a.y + (long)unit * a.y_us;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, false>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u);
}

static int convs_cb(const ConvGemmArgs &a) { return a.ntaps >= 5 ? 1 : 2; }     // 3- and 4-tap convs: two channel blocks per stage
static int convs_tn(const ConvGemmArgs &a) {       // CB = 4 keeps ten float4 of X in flight: one tile of 32 rows only
    if (convs_cb(a) != 1) return 1;
    const int tn = g1_tn(a);
    return tn <= 2 ? tn : (a.NP % 64 ? 1 : 2);
}

static bool convs_supported(const ConvGemmArgs &a) {
    if (!(a.w && !a.wb && !a.w2 && !a.wb2 && !a.xb && !a.yb && a.ntaps >= 3 && a.ntaps <= 8 && a.S >= 1 && !a.in_elu &&
          a.s2d <= 1 && a.act != ACT_ELU && a.act != ACT_GELU_TANH && !(a.x_cs & 3) && !(a.x_us & 3) &&
          !((uintptr_t)a.x & 15) && !((uintptr_t)a.w & 15) && a.TN >= 1 && a.TN <= 4 && !(a.NP % (32 * a.TN)) &&
          a.NP >= a.N && a.Lin >= 1 && (!a.y2 || (a.psn_a && a.psn_ib))))
        return false;
    int HL, XW;
    detail::conv_gemm_geometry(a, HL, XW, G1_BM);
    const int cb = convs_cb(a);
    if (cb * a.ntaps > 8) return false;                                   // W pieces: 4 TN float4 per thread
    if (cb * 16 * (XW / 4) > G1_NTH * (cb == 1 ? 4 : 10)) return false;   // X pieces
    return true;
}

bool convs_selected(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_CONVS_WGS");
    const int max_wgs = e ? atoi(e) : 384;
    if (max_wgs <= 0 || !convs_supported(a)) return false;
    const long tiles_m = (long)((a.M + 255) / 256) * a.units;
    return tiles_m * (a.NP / (32 * detail::conv_gemm_launch_tn(a))) <= max_wgs;
}

const char *convs_kernel_name(const ConvGemmArgs &a) {
    static char names[2][8][32];
    const int tn = convs_tn(a), cb = convs_cb(a);
    char *n = names[cb == 1 ? 0 : 1][tn & 7];
    if (!n[0]) snprintf(n, 32, "convs_kernel<%d,%d>", tn, cb);
    return n;
}

template <int TN, int CB>
static hipError_t convs_launch_one(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(convs_kernel<TN, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((convs_kernel<TN, CB>), dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_convs(const ConvGemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    if (!convs_supported(a)) return hipErrorInvalidValue;
    int HL, XW;
    detail::conv_gemm_geometry(a, HL, XW, G1_BM);
    const int tn = convs_tn(a), cb = convs_cb(a);
    const int tiles_per_unit = (a.M + G1_BM - 1) / G1_BM;
    const int n_tiles = a.NP / (32 * tn);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (4 * 32 * (size_t)tn + 2 * (size_t)cb * 16 * XW + 2 * (size_t)cb * a.ntaps * 16 * 32 * tn);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (cb == 2) return convs_launch_one<1, 2>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s);
    return tn == 1 ? convs_launch_one<1, 1>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
                   : convs_launch_one<2, 1>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s);
}

// ===================================================================================================
// bf16 mode (conv_gemm_bf16_kernel's small-launch form): 128 input channels per stage -- eight
// v_mfma_f32_32x32x16_bf16 per tile between barriers instead of two -- from the same operands in the same order
// (k-groups ascending, lane half h supplies group 2s + h), so again bit-identical.  Activations are the channel-first
// fp32 tensor (gathered eight channels at a time, rounded RNE at commit, as the XB = false form of the big kernel).
// ===================================================================================================
typedef __bf16 g1_bf16x8 __attribute__((ext_vector_type(8)));

// X3: the split-operand mode (ConvGemmArgs::wbm, kernels_bf16.hip): hi and mid planes of both operands in LDS, three
// MFMAs per product in conv_gemm_bf16x3_kernel's order (mid*hi, hi*mid, hi*hi per k-group pair); half the channels
// per stage so that the four tiles still fit.
template <int TN, bool X3>
__device__ __forceinline__ void gemm1_bf16_body(const ConvGemmArgs &a, int tiles_per_unit, int n_tiles) {
    constexpr int G1B_KG = X3 ? 8 : 16;                 // k-groups (of 8 channels) per stage
    constexpr int BN = TN * 32;
    constexpr int XP = G1B_KG * G1_BM / G1_NTH;         // (k-group, position) pairs per thread and stage: 8
    constexpr int WP = G1B_KG * BN / G1_NTH;            // 16-byte weight elements per thread and stage: 2 TN
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_all[];
    float *Pm = reinterpret_cast<float *>(smem_raw_all);                   // [4][BN]
    uint4 *Xs = reinterpret_cast<uint4 *>(smem_raw_all + 16 * BN);         // [2][KG][128]
    uint4 *Ws = Xs + 2 * G1B_KG * G1_BM;                                   // [2][KG][BN]
    uint4 *Xm = Ws + 2 * G1B_KG * BN;                                      // X3: the mid planes, same shapes
    uint4 *Wm = Xm + 2 * G1B_KG * G1_BM;

    int nwg = gridDim.x, orig = blockIdx.x;
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * G1_BM;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int ngr = ((a.Cin + 31) / 32) * 4;            // k-groups of the packed weight tensor
    const int nch = (a.Cin + 8 * G1B_KG - 1) / (8 * G1B_KG);

    fill_param_table<BN, false>(a, Pm, n0, tid, G1_NTH);

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // pair j of this thread: k-group (tid >> 7) + 2 j of the stage, position tid & 127
    const int t = tid & 127, kg0 = tid >> 7;
    const int pos = m0 + t;
    const bool pos_ok = pos < a.Lin;
    const int posc = pos_ok ? pos : a.Lin - 1;
    const uint4 *wg = reinterpret_cast<const uint4 *>(a.wb);
    const uint4 *wgm = reinterpret_cast<const uint4 *>(a.wbm);
    float xr[XP][8];
    uint4 wr[WP], wrm[X3 ? WP : 1];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int j = 0; j < XP; j++) {
            const int c0 = (ch * G1B_KG + kg0 + 2 * j) * 8;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                int row = c0 + c;
                row = row < a.Cin ? row : 0;
                xr[j][c] = xu[(long)row * a.x_cs + posc];
            }
        }
#pragma unroll
        for (int j = 0; j < WP; j++) {
            const int q = tid + G1_NTH * j;
            int g = ch * G1B_KG + q / BN;
            g = g < ngr ? g : 0;
            wr[j] = wg[(long)g * a.NP + n0 + q % BN];
            if constexpr (X3) wrm[j] = wgm[(long)g * a.NP + n0 + q % BN];
        }
    };
    auto commit = [&](int ch, int buf) {
#pragma unroll
        for (int j = 0; j < XP; j++) {
            const int c0 = (ch * G1B_KG + kg0 + 2 * j) * 8;
            g1_bf16x8 o, om;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float v = (pos_ok && c0 + c < a.Cin) ? xr[j][c] : 0.0f;
                o[c] = (__bf16)v;
                om[c] = (__bf16)(v - (float)o[c]);
            }
            Xs[(buf * G1B_KG + kg0 + 2 * j) * G1_BM + t] = __builtin_bit_cast(uint4, o);
            if constexpr (X3) Xm[(buf * G1B_KG + kg0 + 2 * j) * G1_BM + t] = __builtin_bit_cast(uint4, om);
        }
#pragma unroll
        for (int j = 0; j < WP; j++) {
            const int q = tid + G1_NTH * j;
            const bool ok = ch * G1B_KG + q / BN < ngr;
            Ws[buf * G1B_KG * BN + q] = ok ? wr[j] : make_uint4(0, 0, 0, 0);
            if constexpr (X3) Wm[buf * G1B_KG * BN + q] = ok ? wrm[j] : make_uint4(0, 0, 0, 0);
        }
    };

    fetch(0);
    commit(0, 0);
    for (int ch = 0; ch < nch; ch++) {
        const int nx = ch + 1 < nch ? ch + 1 : ch;
        fetch(nx);
        __syncthreads();
        {
            const g1_bf16x8 *wsb = reinterpret_cast<const g1_bf16x8 *>(Ws) + (ch & 1) * G1B_KG * BN + l31;
            const g1_bf16x8 *xsb = reinterpret_cast<const g1_bf16x8 *>(Xs) + (ch & 1) * G1B_KG * G1_BM + wave * 32 + l31;
            g1_bf16x8 av[G1B_KG / 2][TN], bv[G1B_KG / 2];
            g1_bf16x8 am[X3 ? G1B_KG / 2 : 1][TN], bm[X3 ? G1B_KG / 2 : 1];
            const g1_bf16x8 *wmb = wsb + (reinterpret_cast<const g1_bf16x8 *>(Wm) - reinterpret_cast<const g1_bf16x8 *>(Ws));
            const g1_bf16x8 *xmb = xsb + (reinterpret_cast<const g1_bf16x8 *>(Xm) - reinterpret_cast<const g1_bf16x8 *>(Xs));
#pragma unroll
            for (int sp = 0; sp < G1B_KG / 2; sp++) {
                const int kq = 2 * sp + h;
                bv[sp] = xsb[kq * G1_BM];
                if constexpr (X3) bm[sp] = xmb[kq * G1_BM];
#pragma unroll
                for (int i = 0; i < TN; i++) {
                    av[sp][i] = wsb[kq * BN + i * 32];
                    if constexpr (X3) am[sp][i] = wmb[kq * BN + i * 32];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sp = 0; sp < G1B_KG / 2; sp++) {
                if constexpr (X3) {
#pragma unroll
                    for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[sp][i], bv[sp], acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[sp][i], bm[sp], acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[sp][i], bv[sp], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        commit(nx, (ch + 1) & 1);
    }

    float *yu = a.y + (long)unit * a.y_us;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, false, false, !X3>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u);
}

template <int TN>
__global__ __launch_bounds__(G1_NTH, 2) void gemm1_bf16_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    gemm1_bf16_body<TN, false>(a, tiles_per_unit, n_tiles);
}
template <int TN>
__global__ __launch_bounds__(G1_NTH, 2) void gemm1_bf16x3_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    gemm1_bf16_body<TN, true>(a, tiles_per_unit, n_tiles);
}

static bool g1b_supported(const ConvGemmArgs &a) {
    return a.wb && (!a.wbm || !((uintptr_t)a.wbm & 15)) && !a.wb2 && !a.w2 && !a.xb && !a.yb && a.y && a.ntaps == 1 && a.shift0 == 0 && a.S >= 1 && !a.sn_a &&
           !a.in_elu && a.s2d <= 1 && !((uintptr_t)a.wb & 15) && a.TN >= 1 && a.TN <= 4 && !(a.NP % (32 * a.TN)) &&
           a.NP >= a.N && a.Lin >= 1 && a.act != ACT_ELU && a.act != ACT_GELU_TANH && (!a.y2 || (a.psn_a && a.psn_ib));
}

bool gemm1_bf16_selected(const ConvGemmArgs &a) {
    // (a lower bar than the fp32 form: the eight-channel gather of the fp32 activations is instruction-bound once the
    // launch fills the chip -- at 64 units qkv took 1.0 ms here against 0.45 ms in conv_gemm_bf16_kernel)
    const char *e = getenv("HIPVOC_GEMM1_WGS");
    const int max_wgs = e ? atoi(e) : 128;
    if (max_wgs <= 0 || !g1b_supported(a)) return false;
    const long tiles_m = (long)((a.M + 255) / 256) * a.units;
    return tiles_m * (a.NP / (32 * detail::conv_gemm_launch_tn(a))) <= max_wgs;
}

// tiles of at most 64 rows: the whole stage's operands sit in registers ahead of the MFMAs
static int g1b_tn(const ConvGemmArgs &a) {
    const int tn = g1_tn(a);
    return tn <= 2 ? tn : (a.NP % 64 ? 1 : 2);
}

const char *gemm1_bf16_kernel_name(const ConvGemmArgs &a) {
    static char names[8][32];
    const int tn = g1b_tn(a);
    char *n = names[tn & 7];
    if (a.wbm) {
        static char n3[8][32];
        char *m = n3[tn & 7];
        if (!m[0]) snprintf(m, 32, "gemm1_bf16x3_kernel<%d>", tn);
        return m;
    }
    if (!n[0]) snprintf(n, 32, "gemm1_bf16_kernel<%d>", tn);
    return n;
}

template <int TN, bool X3>
static hipError_t gemm1_bf16_launch_one(const ConvGemmArgs &a, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = X3 ? hipFuncSetAttribute(reinterpret_cast<const void *>(gemm1_bf16x3_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
               : hipFuncSetAttribute(reinterpret_cast<const void *>(gemm1_bf16_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    if constexpr (X3) hipLaunchKernelGGL((gemm1_bf16x3_kernel<TN>), dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, tiles_per_unit, n_tiles);
    else hipLaunchKernelGGL((gemm1_bf16_kernel<TN>), dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_gemm1_bf16(const ConvGemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    if (!g1b_supported(a)) return hipErrorInvalidValue;
    const int tn = g1b_tn(a);
    const int tiles_per_unit = (a.M + G1_BM - 1) / G1_BM;
    const int n_tiles = a.NP / (32 * tn);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    // plain: 16 k-groups per stage, one plane; split-operand: 8 k-groups, two planes -- the same bytes
    const size_t lds = 16 * 32 * (size_t)tn + 16 * 2 * 16 * (G1_BM + 32 * (size_t)tn);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.wbm) return tn == 1 ? gemm1_bf16_launch_one<1, true>(a, tiles_per_unit, n_tiles, grid, lds, s)
                              : gemm1_bf16_launch_one<2, true>(a, tiles_per_unit, n_tiles, grid, lds, s);
    return tn == 1 ? gemm1_bf16_launch_one<1, false>(a, tiles_per_unit, n_tiles, grid, lds, s)
                   : gemm1_bf16_launch_one<2, false>(a, tiles_per_unit, n_tiles, grid, lds, s);
}

// ===================================================================================================
// bf16 mode, conv7 on a small launch: conv_gemm_bf16_kernel's stage is one tap of 32 channels (2 TN MFMAs of 32
// cycles against the same 1.4 us round trip: 107 us for b0's conv7 on one utterance).  Here a stage is 64 channels
// with all their taps (28 MFMAs per tile), single-buffered in 52 KB of LDS so that three workgroups share a CU and
// cover each other's waits; products in conv_gemm_bf16_kernel's order (channel block of 32, tap, k-group pair).
// XB: packed bf16 input (ConvGemmArgs::xb); otherwise channel-first fp32 without an input activation.
// ===================================================================================================
// X3 (fp32 input only): the split-operand mode, both planes of both operands in LDS (104 KB: one workgroup per CU).
template <bool XB, bool X3>
__device__ __forceinline__ void convs_bf16_body(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    static_assert(!(XB && X3), "the split-operand mode exchanges fp32 tensors");
    constexpr int BN = 32, KG = 8;                      // one 32-row tile; k-groups (of 8 channels) per stage
    constexpr int XPM = 6;                              // (k-group, position) pairs per thread and stage (host checks)
    constexpr int WPM = 7;                              // 16-byte weight elements per thread: 2 blocks x ntaps <= 7 x 4 x 32 / 256
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_all[];
    float *Pm = reinterpret_cast<float *>(smem_raw_all);                   // [4][BN]
    uint4 *Xs = reinterpret_cast<uint4 *>(smem_raw_all + 16 * BN);         // [KG][XW]
    uint4 *Ws = Xs + KG * XW;                                              // [2][ntaps][4][BN]
    uint4 *Xm = Ws + 2 * a.ntaps * 4 * BN;                                 // X3: the mid planes, same shapes
    uint4 *Wm = Xm + KG * XW;
    // fp32 input with an input SnakeBeta: exp(alpha) | 1/(exp(beta)+1e-9) per channel, [2][8 * k-groups]
    float *Sn = reinterpret_cast<float *>(X3 ? Wm + 2 * a.ntaps * 4 * BN : Xm);

    int nwg = gridDim.x, orig = blockIdx.x;
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * G1_BM;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int ncb = (a.Cin + 31) / 32;                  // channel blocks of the packed weight tensor
    const int ngr = (a.Cin + 7) >> 3;                   // k-groups of the input
    const int nst = (ncb + 1) / 2;
    const int wtot = 2 * a.ntaps * 4 * BN;              // weight elements per stage

    fill_param_table<BN, false>(a, Pm, n0, tid, G1_NTH);
    if (!XB && a.sn_a) {                                // (read at the first commit, behind the loop's first barrier)
        for (int i = tid; i < ngr * 8; i += G1_NTH) {
            Sn[i] = i < a.Cin ? a.sn_a[i] : 0.0f;
            Sn[ngr * 8 + i] = i < a.Cin ? a.sn_ib[i] : 0.0f;
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    // ---- X pairs: p = tid + 256 j -> k-group p / XW, tile position p % XW
    int xg[XPM];                                        // k-group within the stage, or -1
    long xoff[XPM];                                     // element offset at stage 0 (position clamped)
    unsigned xokm = 0;                                  // bit j: the position is inside [0, Lin)
#pragma unroll
    for (int j = 0; j < XPM; j++) {
        const int p = tid + G1_NTH * j;
        const int kg = p / XW, t = p - kg * XW;
        const bool ex = kg < KG;
        int pos = m0 - HL + t;
        if (pos >= 0 && pos < a.Lin) xokm |= 1u << j;
        pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
        xg[j] = ex ? kg : -1;
        xoff[j] = XB ? (long)unit * a.x_us + (long)(ex ? kg : 0) * a.x_cs + pos
                     : (long)unit * a.x_us + (long)(ex ? kg : 0) * 8 * a.x_cs + pos;
    }
    const uint4 *xq = reinterpret_cast<const uint4 *>(a.xb);
    const uint4 *wg = reinterpret_cast<const uint4 *>(a.wb);
    const uint4 *wgm = reinterpret_cast<const uint4 *>(a.wbm);
    uint4 xr[XPM], wr[WPM], wrm[X3 ? WPM : 1];
    float xf[XB ? 1 : XPM][8];
    auto fetch = [&](int st) {
#pragma unroll
        for (int j = 0; j < XPM; j++) {
            const int g = st * KG + (xg[j] < 0 ? 0 : xg[j]);
            if constexpr (XB) {
                xr[j] = xq[g < ngr ? xoff[j] + (long)st * KG * a.x_cs : xoff[j] - (long)(xg[j] < 0 ? 0 : xg[j]) * a.x_cs];
            } else {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int ch = g * 8 + c;
                    xf[j][c] = a.x[ch < a.Cin ? xoff[j] + ((long)st * KG * 8 + c) * a.x_cs
                                              : xoff[j] - (long)(xg[j] < 0 ? 0 : xg[j]) * 8 * a.x_cs];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < WPM; j++) {
            const int q = tid + G1_NTH * j;
            const int slab = q / (4 * BN), rem = q - slab * (4 * BN);     // slab = (block of the stage, tap)
            const int cbi = slab / a.ntaps, tap = slab - cbi * a.ntaps;
            const int cb = 2 * st + cbi;
            const bool ok = q < wtot && cb < ncb;
            const long wi = ok ? ((long)(tap * ncb + cb) * 4 + rem / BN) * a.NP + n0 + rem % BN : 0;
            wr[j] = wg[wi];
            if constexpr (X3) wrm[j] = wgm[wi];
        }
    };
    auto commit = [&](int st) {
#pragma unroll
        for (int j = 0; j < XPM; j++) {
            if (xg[j] < 0) continue;
            const bool pok = (xokm >> j) & 1;
            uint4 v;
            if constexpr (XB) {
                v = xr[j];
                if (!pok || st * KG + xg[j] >= ngr) v = make_uint4(0, 0, 0, 0);
            } else {
                g1_bf16x8 o, om;
                float f[8];
#pragma unroll
                for (int c = 0; c < 8; c++) f[c] = (pok && (st * KG + xg[j]) * 8 + c < a.Cin) ? xf[j][c] : 0.0f;
                if (a.sn_a) {           // as conv_gemm_bf16_kernel's commit: packed evaluation on channel pairs
                    const int c0 = (st * KG + xg[j]) * 8;
                    const float4 *sa = reinterpret_cast<const float4 *>(Sn + (c0 < ngr * 8 ? c0 : 0));
                    const float4 *sb = reinterpret_cast<const float4 *>(Sn + ngr * 8 + (c0 < ngr * 8 ? c0 : 0));
                    const float4 a0 = sa[0], a1 = sa[1], b0 = sb[0], b1 = sb[1];
                    const f32x2 p0 = snake_beta2_t<!X3>(mk2(f[0], f[1]), mk2(a0.x, a0.y), mk2(b0.x, b0.y));
                    const f32x2 p1 = snake_beta2_t<!X3>(mk2(f[2], f[3]), mk2(a0.z, a0.w), mk2(b0.z, b0.w));
                    const f32x2 p2 = snake_beta2_t<!X3>(mk2(f[4], f[5]), mk2(a1.x, a1.y), mk2(b1.x, b1.y));
                    const f32x2 p3 = snake_beta2_t<!X3>(mk2(f[6], f[7]), mk2(a1.z, a1.w), mk2(b1.z, b1.w));
                    f[0] = p0.x; f[1] = p0.y; f[2] = p1.x; f[3] = p1.y; f[4] = p2.x; f[5] = p2.y; f[6] = p3.x; f[7] = p3.y;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    o[c] = (__bf16)f[c];
                    om[c] = (__bf16)(f[c] - (float)o[c]);
                }
                v = __builtin_bit_cast(uint4, o);
                if constexpr (X3) Xm[tid + G1_NTH * j] = __builtin_bit_cast(uint4, om);
            }
            Xs[tid + G1_NTH * j] = v;
        }
#pragma unroll
        for (int j = 0; j < WPM; j++) {
            const int q = tid + G1_NTH * j;
            if (q >= wtot) continue;
            const int cbi = (q / (4 * BN)) / a.ntaps;
            const bool ok = 2 * st + cbi < ncb;
            Ws[q] = ok ? wr[j] : make_uint4(0, 0, 0, 0);
            if constexpr (X3) Wm[q] = ok ? wrm[j] : make_uint4(0, 0, 0, 0);
        }
    };

    fetch(0);
    for (int st = 0; st < nst; st++) {
        __syncthreads();                                 // the previous stage's readers are past
        commit(st);
        fetch(st + 1 < nst ? st + 1 : st);               // unconditional; in flight during this stage's MFMAs
        __syncthreads();
        const g1_bf16x8 *wsb = reinterpret_cast<const g1_bf16x8 *>(Ws) + l31;
        const g1_bf16x8 *xsb = reinterpret_cast<const g1_bf16x8 *>(Xs) + wave * 32 + l31 + HL + a.shift0;
        for (int cbi = 0; cbi < 2; cbi++)
            for (int tap = 0; tap < a.ntaps; tap++) {
                const g1_bf16x8 *ws = wsb + (cbi * a.ntaps + tap) * 4 * BN;
                const g1_bf16x8 *xs = xsb + cbi * 4 * XW + tap * a.shift_step;
                g1_bf16x8 b0 = xs[h * XW], b1 = xs[(2 + h) * XW];
                g1_bf16x8 a0 = ws[h * BN], a1 = ws[(2 + h) * BN];
                if constexpr (X3) {     // per k-group pair: mid*hi, hi*mid, hi*hi (conv_gemm_bf16x3_kernel's order)
                    const g1_bf16x8 *xm = xs + (reinterpret_cast<const g1_bf16x8 *>(Xm) - reinterpret_cast<const g1_bf16x8 *>(Xs));
                    const g1_bf16x8 *wm = ws + (reinterpret_cast<const g1_bf16x8 *>(Wm) - reinterpret_cast<const g1_bf16x8 *>(Ws));
                    g1_bf16x8 bm0 = xm[h * XW], bm1 = xm[(2 + h) * XW];
                    g1_bf16x8 am0 = wm[h * BN], am1 = wm[(2 + h) * BN];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bm0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bm1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                }
            }
    }

    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    unsigned short *ybu0 = launder_after(a.yb, acc[15]);
    unsigned short *ybu = ybu0 ? ybu0 + (((long)unit * a.y_us) << 3) : nullptr;
    const int m = m0 + wave * 32 + l31;
    conv_epilogue_tile<BN, false, false, !X3>(a, acc, Pm, n0, 4 * h, m, yu, ru, y2u, ybu);
}

template <bool XB>
__global__ __launch_bounds__(G1_NTH, XB ? 3 : 2) void convs_bf16_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    convs_bf16_body<XB, false>(a, HL, XW, tiles_per_unit, n_tiles);
}
__global__ __launch_bounds__(G1_NTH, 2) void convs_bf16x3_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    convs_bf16_body<false, true>(a, HL, XW, tiles_per_unit, n_tiles);
}

static bool convs_bf16_supported(const ConvGemmArgs &a) {
    if (a.wbm && (a.xb || a.yb || ((uintptr_t)a.wbm & 15))) return false;
    if (!(a.wb && !a.wb2 && !a.w2 && a.ntaps >= 5 && a.ntaps <= 7 && a.S >= 1 && !a.in_elu && a.s2d <= 1 && !(a.xb && a.sn_a) &&
          !((uintptr_t)a.wb & 15) && !(a.NP % 32) && a.NP >= a.N && a.Lin >= 1 && a.act != ACT_ELU && a.act != ACT_GELU_TANH &&
          (a.y || a.yb) && (!a.y2 || (a.psn_a && a.psn_ib && !a.yb))))
        return false;
    if (a.xb && ((a.Cin & 7) || ((uintptr_t)a.xb & 15))) return false;
    if (a.yb && (((a.N / a.S) & 7) || (a.N % a.S) || ((uintptr_t)a.yb & 15))) return false;
    int HL, XW;
    detail::conv_gemm_geometry(a, HL, XW, G1_BM);
    return 8 * XW <= G1_NTH * 6;
}

bool convs_bf16_selected(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_CONVS_WGS");
    const int max_wgs = e ? atoi(e) : 384;              // (b1's conv7 on one utterance, 480 workgroups: 22 % slower here)
    if (max_wgs <= 0 || !convs_bf16_supported(a)) return false;
    const long tiles_m = (long)((a.M + 255) / 256) * a.units;
    return tiles_m * (a.NP / (32 * detail::conv_gemm_launch_tn(a))) <= max_wgs;
}

const char *convs_bf16_kernel_name(const ConvGemmArgs &a) {
    if (a.wbm) return "convs_bf16x3_kernel";
    return a.xb ? "convs_bf16_kernel<true>" : "convs_bf16_kernel<false>";
}

static hipError_t convs_bf16x3_launch(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(convs_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL(convs_bf16x3_kernel, dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

template <bool XB>
static hipError_t convs_bf16_launch_one(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(convs_bf16_kernel<XB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((convs_bf16_kernel<XB>), dim3((unsigned)grid), dim3(G1_NTH), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_convs_bf16(const ConvGemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    if (!convs_bf16_supported(a)) return hipErrorInvalidValue;
    int HL, XW;
    detail::conv_gemm_geometry(a, HL, XW, G1_BM);
    const int tiles_per_unit = (a.M + G1_BM - 1) / G1_BM;
    const int n_tiles = a.NP / 32;
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = 16 * 32 + (a.wbm ? 2 : 1) * 16 * ((size_t)8 * XW + (size_t)2 * a.ntaps * 4 * 32) +
                       ((!a.xb && a.sn_a) ? 2 * sizeof(float) * 8 * (size_t)((a.Cin + 7) / 8) : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.wbm) return convs_bf16x3_launch(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s);
    return a.xb ? convs_bf16_launch_one<true>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
                : convs_bf16_launch_one<false>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s);
}

}  // namespace hipvoc
