// gfx950 (CDNA4 / MI355X): the causal k = 7 convolutions of the decoder blocks by Winograd / Cook-Toom F(2,7), fp32.
//
// The fp32 path is bound by v_mfma_f32_32x32x2_f32 (157 TFLOP/s; the direct implicit-GEMM kernel of kernels.hip reaches
// 0.84 of it), so the only way past that roof in true fp32 arithmetic is FEWER multiplications.  F(2,7) computes two
// outputs of a 7-tap filter from 8 products instead of 14:
//     y[p], y[p+d]  =  A^T [ (G g) * (B^T x) ],     x = the eight inputs x[p - 6d + i d], i = 0..7  (d = dilation),
// with the interpolation points 0, +-1, +-2, +-1/2, inf.  Summed over input channels each of the eight "points" is an
// ordinary GEMM,  M_k[co][tile] = sum_ci U_k[co][ci] V_k[ci][tile]  (U = G g, transformed once at load; V = B^T x,
// transformed on the fly), so the matrix cores do 8/14 = 4/7 of the direct kernel's work.  Rounding: the transforms'
// constants are small (|B^T| <= 5.25, |A^T| <= 2); measured on a C = 384 layer the fp32 error is 5x the direct kernel's
// (1.8e-6 of the output rms against 3.6e-7), a factor the whole path's margin to the 1e-3 / 60 dB gates absorbs
// (DESIGN.md section 4b has the measured waveform figures).
//
// Orientation as everywhere: weights are the A operand (rows = output channels), activations the B operand (column =
// TILE on the lane).  A workgroup owns 64 output channels x 128 tiles (256 positions).  Default (RT = 1): 8 waves, two
// per SIMD; waves w and w + 4 own the tiles [32 (w & 3), +32) and one 32-row tile each, so a lane keeps 8 points x 16 =
// 128 accumulator registers.  RT = 2 (HIPVOC_WINO_RT=2): 4 waves, one per SIMD, both row tiles per wave (256 accumulator
// registers).  K loop over blocks of 16 input channels: the RAW activation window Xs[16][XW] and the eight transformed
// weight tiles Ws[8][16][64] are double-buffered in LDS and fetched one block ahead by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no commit pass; the causal zero padding is fetched from a page of zeros, see below); ONE
// barrier per block (64 RT MFMAs per wave between barriers).  A lane reads its eight inputs of channel k from Xs, forms
// the eight V_k (24 VALU instructions) and feeds each to the MFMAs of its point; the loop is software-pipelined by hand
// (LDS reads two channels ahead, transform one ahead; sched_group_barrier interleaves them with the MFMAs).  The input's
// SnakeBeta is normally applied by the PRODUCER of the tensor (dual outputs: the raw residual stream and the activated
// operand, ConvGemmArgs::psn_*), so the window is a plain copy; the SN = true variant applies it on the fly for producers
// that cannot (45 % slower: VALU-bound).
//
// wino_tconv_kernel further down: the decoder blocks' transposed convolutions (kernel = 2 x stride: every output phase is a
// 2-tap filter over the input) by F(3,2): three consecutive input steps' outputs of a phase from 4 products instead of 6.
// Its epilogue transposes each wave's tile through LDS so that every channel's run of q * S + phase outputs is written
// contiguously.
//
// Dilation: tile T of a unit is the output pair (p, p + d) with p = 2d (T / d) + T % d, so consecutive tiles are
// (almost) consecutive positions and both the LDS reads and the stores stay on neighbouring addresses.
#include "kernels.h"
#include "device_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

static constexpr int WN_BK = 16;      // input channels per K stage
static constexpr int WN_NT = 128;     // tiles per workgroup (4 waves x 32)
static constexpr int WN_PTS = 8;
// Compile-time switches of the experiments behind DESIGN.md section 4b (tools/build_variants.sh builds libraries that differ
// in them; the defaults are what was measured best): VALU / DS instructions per sched_group (WN_VG / WN_DG), per-lane masks
// instead of the zero page (WN_MASK / WN_ZP), depth of the input pipeline (WN_DD), the DMA by builtin instead of inline asm
// (WN_ASMDMA), default workgroup shapes, timing-only ablation bits (PS_ABL).
#ifndef WN_VG
#define WN_VG 8
#endif
#ifndef WN_MASK
#define WN_MASK 0
#endif
#ifndef WN_ZP
#define WN_ZP 1
#endif
#ifndef WN_DD
#define WN_DD 3
#endif
#ifndef WN_ASMDMA
#define WN_ASMDMA 1
#endif
#ifndef WN_MODE_DEFAULT
#define WN_MODE_DEFAULT 5
#endif
#ifndef PS_ABL
#define PS_ABL 0   // timing-only ablation bits (results are wrong with any set)
#endif
#ifndef WN_UNIT_WG_DEFAULT
#define WN_UNIT_WG_DEFAULT 8
#endif
// LDS operand reads: volatile keeps hipcc from merging neighbouring reads into ds_read2 forms, whose 8-bit offsets force a
// VALU address add per step (matrix-pipe time); single ds_read_b32 take the whole step offset as a 16-bit immediate
#ifndef WN_LDS_VOLATILE
#define WN_LDS_VOLATILE 1
#endif
#if WN_LDS_VOLATILE
typedef const volatile __attribute__((address_space(3))) float *wn_vlds_t;
#define WN_LDS(p_) (*(wn_vlds_t)(p_))
#else
#define WN_LDS(p_) (*(p_))
#endif
#ifndef GEMMD_NW_DEFAULT
#define GEMMD_NW_DEFAULT 4
#endif
#ifndef WN_PS_OCC
#define WN_PS_OCC 3          // workgroups per CU the 4-wave conv7 form is compiled for: 3 = at most 168 registers per lane (hipcc needs 171
                             // at 2: five dwords of scratch outside the loop) and 42-48 KB of LDS each; conv7 layers 2-5 % faster than at 2
#endif
#ifndef WN_DG
#define WN_DG 2
#endif
static constexpr bool wino_order_pin = false;  // (measured: LDS reads of a step first in program order, then MFMAs: 7.31 ms on b1 against 7.05 interleaved)

// RT = row tiles (32 output channels each) per wave: 2 -> 4 waves per workgroup, one wave per SIMD (256 accumulator
// registers); 1 -> 8 waves, two per SIMD: waves w and w + 4 own the same tiles and one row tile each.
template <int D, int RT, int RH>
struct WinoGeom {
    static constexpr int NTH = 4 * RH * 64;                               // 4 tile groups x RH row groups of waves
    static constexpr int BN = 32 * RT * RH;                               // output channels per workgroup
    static constexpr int XW = (257 + 10 * D + 3) & ~3;                    // raw window: see the bound in launch_wino_conv7
    static constexpr int XP = (WN_BK * (XW / 4) + NTH - 1) / NTH;         // LDS-DMA pieces (16 B per lane) of the window per thread
    static constexpr int XBUF = WN_BK * XW;                               // floats per window buffer (the last piece of a thread may not exist)
    static constexpr int WP = WN_PTS * WN_BK * BN / 4 / NTH;              // pieces of the weight tiles per thread
    static constexpr int WBUF = WN_PTS * WN_BK * BN;                      // floats per weight buffer
};

typedef __attribute__((address_space(1))) const void *wn_gptr_t;
typedef __attribute__((address_space(3))) void *wn_lptr_t;

// 16 bytes per lane from global memory straight into LDS (wave-uniform LDS base in M0 + 16 * lane).  Issued as inline asm,
// not through __builtin_amdgcn_global_load_lds: the compiler books that builtin as a FLAT access to both address spaces
// ("pending flat"), and while one is in flight -- here: always -- every wait for an LDS read becomes lgkmcnt(0) instead of
// the count that lets younger reads stay in flight.  The kernels below wait for their DMA themselves (vmcnt(0) before the
// block barrier) and use M0 for nothing else.
__device__ __forceinline__ void wn_dma16(const float *g, float *lds) {
#if WN_ASMDMA
    const unsigned l = (unsigned)(size_t)(wn_lptr_t)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
#else
    __builtin_amdgcn_global_load_lds((wn_gptr_t)g, (wn_lptr_t)lds, 16, 0, 0);
#endif
}
// ... the same with the address as a wave-uniform base (SGPR pair) + a per-lane byte offset: the per-block address arithmetic
// stays on the scalar unit (a 64-bit VALU add per piece is matrix-pipe time on this chip, see wino_ps_kernel's header)
__device__ __forceinline__ void wn_dma16_s(const float *sbase, unsigned voff_bytes, float *lds) {
    const unsigned l = (unsigned)(size_t)(wn_lptr_t)lds;
    // (the base is wave-uniform by construction; readfirstlane makes the compiler keep it in SGPRs whatever unit computed it)
    const unsigned long long b = (unsigned long long)sbase;
    // (readfirstlane returns int: both halves go through unsigned, or the low half's top bit would sign-extend into the high one)
    const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)), b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned long long bs = ((unsigned long long)b_hi << 32) | (unsigned long long)b_lo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff_bytes), "s"(bs) : "memory");
}

// SN: the conv has a SnakeBeta prologue (a.sn_a / a.sn_ib), applied to the inputs as they are read from LDS
template <int D, bool SN, int RT, int RH>
__global__ __launch_bounds__((WinoGeom<D, RT, RH>::NTH), (RT * RH == 1 ? 2 : 1)) void wino_conv7_kernel(ConvGemmArgs a, int wg_per_unit, int n_tiles) {
    using GEO = WinoGeom<D, RT, RH>;
    constexpr int BN = GEO::BN, WN_WBUF = GEO::WBUF;
    constexpr int NTH = GEO::NTH, XW = GEO::XW, XP = GEO::XP, XBUF = GEO::XBUF, WP = GEO::WP, xw4 = XW / 4;
    // ONE LDS array (a second object beside LDS-DMA staging makes hipcc wait vmcnt(0) before every ds_read)
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                                   // [4][64] per-row parameters (device_common.h)
    float *Sp = smem_all + 4 * BN;                       // [Cin][2] SnakeBeta parameters of the input channels (SN)
    float *Xs = Sp + (SN ? 2 * a.Cin : 0);                  // [2][XBUF]: raw activation window, rows of XW floats
    float *Ws = Xs + 2 * XBUF;                              // [2][8][16][64]

    int nwg = gridDim.x, orig = blockIdx.x;                 // XCD-aware remap, as conv_gemm_kernel
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt = id / n_tiles;
    const int unit = pt / wg_per_unit;
    const int T0 = (pt - unit * wg_per_unit) * WN_NT;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int CinP = ((a.Cin + 31) / 32) * 32;
    const int ncb = a.Cin / WN_BK;

    // window of raw activations this workgroup needs: positions [pwin0, pwin0 + XW)
    const int p0 = 2 * D * (T0 / D) + T0 % D;
    const int pwin0 = (p0 - 6 * D) & ~3;                    // (floor to a multiple of 4, also when negative)
    const bool wg_zero = pwin0 < 0;                          // (workgroup-uniform) some pieces of the window lie left of the unit
    // this lane's tile: outputs p and p + D from the inputs p - 6D + i D, i = 0..7
    const int tw = wave & 3, rh = wave >> 2;                    // tile group of this wave, its row group (RT row tiles)
    const int T = T0 + tw * 32 + l31;
    const int tq = T / D, tph = T - tq * D;
    const int p = 2 * D * tq + tph;
    const int xoff = p - 6 * D - pwin0;                     // >= 0; xoff + 7 D < XW
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int pi = p + (i - 6) * D;
        dmask |= (pi >= 0 && pi < a.Lin) ? (1u << i) : 0u;
    }
    // Inputs that do not exist.  LEFT of the unit (causal zero padding): position 0 is a multiple of 4 in the window's
    // 16-byte pieces, so such inputs are whole pieces, and those are fetched from a page of zeros instead of the tensor (zsrc
    // below).  RIGHT of the unit (positions >= Lin = M): the only input of a tile that can lie there while the tile's first
    // output p is stored is x[p + D], which enters V7 alone, and M7 enters y1 = y[p + D] alone -- an output that is not stored.
    // So whatever the (clamped, finite) fetch delivers there never reaches memory, and the loop needs no per-lane masks.

    f32x16 acc[WN_PTS][RT];
#pragma unroll
    for (int k = 0; k < WN_PTS; k++)
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][i][r] = 0.0f;

    // ---- staging: global -> LDS directly (global_load_lds_dwordx4: no registers, no commit pass).  The LDS image of one
    // wave-instruction is lane-linear (64 x 16 B), so piece j of the window is float4 index 256 j + tid of the [16][XW]
    // image and its per-lane SOURCE address does the row / clamp arithmetic; weight tile k is a plain copy.
    const int xlim = ((a.Lin + 3) & ~3) - 4;
    unsigned xo[XP], zsrc = 0;
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int i = tid + NTH * j;
        int row = i / xw4;
        const int c4 = i - row * xw4;
        row = row < WN_BK ? row : 0;                         // (pieces past the window are not issued)
        int pos = pwin0 + 4 * c4;
        zsrc |= pos < 0 ? (1u << j) : 0u;
        pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
        xo[j] = (unsigned)row * (unsigned)a.x_cs + (unsigned)pos;
    }
    // weight piece k of this thread: float4 index tid + NTH k of the [8][16][64] image = (point, row, float4 column)
    unsigned wo[WP];
#pragma unroll
    for (int k = 0; k < WP; k++) {
        const int i = tid + NTH * k, q = i / (4 * BN), r = (i / (BN / 4)) & 15, c4 = i % (BN / 4);
        wo[k] = ((unsigned)q * (unsigned)CinP + (unsigned)r) * (unsigned)a.NP + (unsigned)c4 * 4u;
    }
    auto stage = [&](int cb, int buf) {
        const float *xb = xu + (long)cb * WN_BK * a.x_cs;
        float *xd = Xs + buf * XBUF + wave * 256;
#pragma unroll
        for (int j = 0; j < XP; j++) {
            if (NTH * (j + 1) <= WN_BK * xw4 || tid + NTH * j < WN_BK * xw4) {
                if (wg_zero) wn_dma16((WN_ZP && ((zsrc >> j) & 1u)) ? a.zeros : xb + xo[j], xd + j * (NTH * 4));
                else wn_dma16_s(xb, xo[j] * 4u, xd + j * (NTH * 4));
            }
        }
        const float *wb = a.ww + ((long)cb * WN_BK) * a.NP + n0;
        float *wd = Ws + buf * WN_WBUF + wave * 256;
#pragma unroll
        for (int k = 0; k < WP; k++)
            wn_dma16_s(wb, wo[k] * 4u, wd + k * (NTH * 4));
    };

    // the first block's DMA goes out first; the parameter tables (ordinary loads, then LDS writes) are filled while it flies
    // (the DMA is inline asm, so the compiler does not drain it at the first use of an ordinary load's result)
    stage(0, 0);
    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    if constexpr (SN) {
        for (int c = tid; c < a.Cin; c += NTH) {
            Sp[2 * c] = a.sn_a[c];
            Sp[2 * c + 1] = a.sn_ib[c];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int cb = 0; cb < ncb; cb++) {
        const int buf = cb & 1;
        if (cb + 1 < ncb) stage(cb + 1, buf ^ 1);            // lands during this block's MFMAs; the other buffer has no readers
        {
            const float *xs = Xs + buf * XBUF + xoff;
            const float *ws = Ws + buf * WN_WBUF + l31 + rh * (32 * RT);
            const float *sp = Sp + 2 * (cb * WN_BK + h);
            float dv[WN_DD][8], vv[2][WN_PTS], av[2][WN_PTS][RT];
            float2 sv[WN_DD];
            auto load_d = [&](int kk, int sl) {
                const float *xk = xs + (2 * kk + h) * XW;
#pragma unroll
                for (int i = 0; i < 8; i++) dv[sl][i] = xk[i * D];
                if constexpr (SN) sv[sl] = *reinterpret_cast<const float2 *>(sp + 4 * kk);
            };
            auto load_a = [&](int kk, int sl) {
                const float *wk = ws + (2 * kk + h) * BN;
#pragma unroll
                for (int q = 0; q < WN_PTS; q++)
#pragma unroll
                    for (int i = 0; i < RT; i++) av[sl][q][i] = wk[q * (WN_BK * BN) + 32 * i];
            };
            // SnakeBeta (snake(0) == 0 keeps the zero padding), then V = B^T d for the points 0, 1, -1, 2, -2, 1/2, -1/2, inf
            auto transform = [&](int ds, int vs) {
                float d[8];
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = WN_MASK ? (((dmask >> i) & 1u) ? dv[ds][i] : 0.0f) : dv[ds][i];
                if constexpr (SN) {
                    const f32x2 ea = (f32x2)(sv[ds].x), ib = (f32x2)(sv[ds].y);
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const f32x2 r = snake_beta2(mk2(d[i], d[i + 1]), ea, ib);
                        d[i] = r.x;
                        d[i + 1] = r.y;
                    }
                }
                const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5], d6 = d[6], d7 = d[7];
                // 24 VALU instructions (on gfx950 every fp32 VALU instruction takes its ~4 cycles from the fp32 matrix pipe --
                // tools/ubench/mfma_valu2.hip -- so this count, not latency, is what the transform costs).  The odd parts of
                // the +-2 and +-1/2 points are kept at the scale whose first coefficient is 1 and rescaled by the final fma.
                const float e12 = fmaf(-4.25f, d4, d2 + d6), o12 = fmaf(-4.25f, d3, d1 + d5);
                const float e34 = fmaf(0.25f, d2, fmaf(-1.25f, d4, d6)), o34 = fmaf(4.0f, d5, fmaf(-5.0f, d3, d1));      // 2 x odd part
                const float e56 = fmaf(4.0f, d2, fmaf(-5.0f, d4, d6)), o56 = fmaf(0.25f, d5, fmaf(-1.25f, d3, d1));     // odd part / 2
                vv[vs][0] = fmaf(5.25f, d4 - d2, d0 - d6);
                vv[vs][1] = e12 + o12;
                vv[vs][2] = e12 - o12;
                vv[vs][3] = fmaf(0.5f, o34, e34);
                vv[vs][4] = fmaf(-0.5f, o34, e34);
                vv[vs][5] = fmaf(2.0f, o56, e56);
                vv[vs][6] = fmaf(-2.0f, o56, e56);
                vv[vs][7] = fmaf(5.25f, d3 - d5, d7 - d1);
            };
            // One wave per SIMD: only this wave's own instruction stream hides latency.  Three-deep pipeline over the eight
            // k-pairs of the block: step kk issues the input reads of step kk+2 and the weight reads of step kk+1,
            // transforms the inputs of step kk+1, and feeds the matrix cores with step kk -- three independent streams
            // that sched_group_barrier interleaves around the MFMAs.
            load_d(0, 0);
            load_a(0, 0);
#pragma unroll
            for (int j = 1; j < WN_DD - 1; j++) load_d(j, j);
            transform(0, 0);
#pragma unroll
            for (int kk = 0; kk < WN_BK / 2; kk++) {
                __builtin_amdgcn_sched_barrier(0);
                // weights of the next step first, inputs of step kk + WN_DD - 1 after them: LDS returns in order, so the
                // next step's first MFMA waits for the weights only (lgkmcnt(n > 0)), and the inputs have a whole step more
                if (kk + 1 < WN_BK / 2) load_a(kk + 1, (kk + 1) & 1);
                if (kk + WN_DD - 1 < WN_BK / 2) load_d(kk + WN_DD - 1, (kk + WN_DD - 1) % WN_DD);
                if (wino_order_pin) __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < WN_BK / 2) transform((kk + 1) % WN_DD, (kk + 1) & 1);
#pragma unroll
                for (int q = 0; q < WN_PTS; q++)
#pragma unroll
                    for (int i = 0; i < RT; i++)
                        acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][q][i], vv[kk & 1][q], acc[q][i], 0, 0, 0);
                if constexpr (wino_order_pin) {
#pragma unroll
                    for (int g = 0; g < 8 * RT; g++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    }
                } else if constexpr (RT == 2) {
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);      // VALU in its shadow
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, WN_DG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, WN_VG, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next block's DMA has had this whole block to land; every wave is done reading this block's buffers
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- output transform  y = A^T M  and epilogue: bias, optional SnakeBeta of the consumer, store
    //   y0 = M0 + M1 + M2 + M3 + M4 + M5 + M6
    //   y1 = M1 - M2 + 2 (M3 - M4) + (M5 - M6) / 2 + M7
    float *yu = a.y + (long)unit * a.y_us;
    const bool ok0 = p < a.M, ok1 = p + D < a.M;
#pragma unroll
    for (int i = 0; i < RT; i++) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float y0[2], y1[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float m0 = acc[0][i][r + e], m1 = acc[1][i][r + e], m2 = acc[2][i][r + e], m3 = acc[3][i][r + e],
                            m4 = acc[4][i][r + e], m5 = acc[5][i][r + e], m6 = acc[6][i][r + e], m7 = acc[7][i][r + e];
                y0[e] = ((m1 + m2) + (m3 + m4)) + ((m5 + m6) + m0);
                y1[e] = fmaf(2.0f, m3 - m4, m1 - m2) + fmaf(0.5f, m5 - m6, m7);
            }
            const int row = (rh * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // rows row, row + 1
            f32x2 b = mk2(Pm[row], Pm[row + 1]);
            f32x2 o0 = mk2(y0[0], y0[1]), o1 = mk2(y1[0], y1[1]);
            if (a.bias) { o0 = o0 + b; o1 = o1 + b; }
            if (a.psn_a) {
                const f32x2 pa = mk2(Pm[2 * BN + row], Pm[2 * BN + row + 1]), pb = mk2(Pm[3 * BN + row], Pm[3 * BN + row + 1]);
                o0 = snake_beta2(o0, pa, pb);
                o1 = snake_beta2(o1, pa, pb);
            }
            float *yr = yu + (long)(n0 + row) * a.y_cs + p;
            if (ok0) { yr[0] = o0.x; yr[a.y_cs] = o0.y; }
            if (ok1) { yr[D] = o1.x; yr[a.y_cs + D] = o1.y; }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// POINT-SPLIT form of the F(2,7) convolution: wino_ps_kernel<D, RT, FUSE>.
//
// On gfx950 the fp32 MFMAs run on the SIMD's fp32 vector ALUs: every fp32 VALU instruction of any wave on the SIMD takes its
// ~4 cycles out of the matrix pipe's time, however many waves there are to "hide" it (tools/ubench/mfma_valu2.hip: 64.7
// cycles per MFMA alone, 78.6 with 4 VALU instructions beside it, 90.5 with 8 -- at 1, 2 or 4 waves per SIMD alike).  What
// the input transform costs is therefore its instruction count per MFMA: 24 per 8 MFMAs in wino_conv7_kernel's default shape.
// Here a wave owns only FOUR of the eight interpolation points, for RT row tiles of 32 output channels: waves 0-3 (one per
// tile group of 32 tiles) the points 0, +1, -1, inf, waves 4-7 the points +2, -2, +1/2, -1/2.  Each transforms its inputs for
// its own points only (12 instructions) and feeds 4 RT MFMAs with them: 1.5 VALU instructions per MFMA at RT = 2, 1.0 at
// RT = 3, with two waves per SIMD (4 RT x 16 accumulator registers each).  The output transform is linear in the M_k, so
// every wave reduces its four products to partial sums of y[p] and y[p + D]; the two waves of a tile group swap one of the
// two through LDS (the staging buffers, idle by then) and finish one output position each:
//     waves 0-3:  y[p]     = (M0 + M1 + M2)           + [ M3 + M4 + M5 + M6 ]            <- from wave + 4
//     waves 4-7:  y[p + D] = 2 (M3 - M4) + (M5 - M6)/2 + [ M1 - M2 + M7 ]                <- from wave - 4
// The weight tiles arrive in LDS already in that point order (the DMA's source addresses do the permutation).
//
// FUSE (RT = 3, N = Cin = 96: the residual units of the last decoder block): the workgroup holds ALL output channels of its
// positions, so the unit's second half runs in the same kernel as in conv_gemm_kernel<3, true>: H = snake2(y + b1) in
// registers is the B operand of the 1x1 conv (weights W2 staged whole, 36 KB), + b2 + residual, store; optionally
// snake_next(out) beside it (y2), because this kernel's own input must come activated (it is fetched by DMA).
template <int D, int RT, int NTG, int BK>
struct PsGeom {
    static constexpr int NTH = 2 * NTG * 64;                              // NTG tile groups x 2 point groups of waves
    static constexpr int NT = 32 * NTG;                                   // tiles per workgroup
    static constexpr int BN = 32 * RT;
    static constexpr int XW = (2 * NT + 1 + 10 * D + 3) & ~3;             // window bound as in wino_launch_one
    static constexpr int XP = (BK * (XW / 4) + NTH - 1) / NTH;
    static constexpr int XBUF = BK * XW;
    static constexpr int WF4 = WN_PTS * BK * BN / 4;                      // float4 of one block's weight tiles: a multiple of NTH
    static constexpr int WP = WF4 / NTH;
    static constexpr int WBUF = WN_PTS * BK * BN;
    static constexpr int EX = 2 * NTG * RT * 16 * 64;                     // floats of the partial-sum exchange: [wave][RT * 16][lane]
    static constexpr int EXH = NTG * RT * 16 * 64;                        // ... of one direction of it (two-phase exchange)
    static constexpr int W2F = BN * BN;                                   // FUSE: the 1x1 conv's weights [k][BN]
    static constexpr int W2P = (W2F / 4 + NTH - 1) / NTH;
    static constexpr int STAGE = 2 * XBUF + 2 * WBUF;
    // FUSE in 4-wave workgroups (NTG == 2): the two directions of the exchange go one after the other through the same LDS,
    // so that exchange + W2 stay under 80 KB and two workgroups share a CU
    static constexpr bool two_phase(bool fuse) { return fuse && NTG == 2; }
    static constexpr int WORK(bool fuse) {
        const int need = (two_phase(fuse) ? EXH : EX) + (fuse ? W2F : 0);
        return STAGE > need ? STAGE : need;
    }
    static_assert(WF4 % NTH == 0 && (NTG == 2 || NTG == 4) && BK % 2 == 0, "geometry");
};

template <int D, int RT, bool FUSE, int NTG, int BK>
__global__ __launch_bounds__((PsGeom<D, RT, NTG, BK>::NTH), (NTG == 2 ? (FUSE ? 2 : WN_PS_OCC) : 1)) void wino_ps_kernel(ConvGemmArgs a, int wg_per_unit, int n_tiles) {
    using GEO = PsGeom<D, RT, NTG, BK>;
    constexpr int NTH = GEO::NTH, BN = GEO::BN, XW = GEO::XW, XP = GEO::XP, XBUF = GEO::XBUF, WP = GEO::WP, WBUF = GEO::WBUF, xw4 = XW / 4;
    constexpr int PMF = FUSE ? 6 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                                   // per-row parameters (device_common.h: fill_param_table)
    float *Xs = smem_all + PMF * BN;                        // [2][XBUF]: raw (already activated) input window, rows of XW floats
    float *Ws = Xs + 2 * XBUF;                              // [2][8 point slots][16][BN]

    int nwg = gridDim.x, orig = blockIdx.x;                 // XCD-aware remap, as conv_gemm_kernel
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt = id / n_tiles;
    const int unit = pt / wg_per_unit;
    const int T0 = (pt - unit * wg_per_unit) * GEO::NT;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int CinP = ((a.Cin + 31) / 32) * 32;
    const int ncb = a.Cin / BK;
    const int tw = wave & (NTG - 1), pg = wave / NTG;       // tile group, point group

    const int p0 = 2 * D * (T0 / D) + T0 % D;
    const int pwin0 = (p0 - 6 * D) & ~3;
    const bool wg_zero = pwin0 < 0;                          // (workgroup-uniform) some pieces of the window lie left of the unit
    const int T = T0 + tw * 32 + l31;
    const int tq = T / D, tph = T - tq * D;
    const int p = 2 * D * tq + tph;                         // this lane's tile: outputs p and p + D
    const int xoff = p - 6 * D - pwin0;


    f32x16 acc[4][RT];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][i][r] = 0.0f;

    // staging by LDS-DMA as in wino_conv7_kernel (zero page for the causal left padding; nothing to mask on the right)
    const int xlim = ((a.Lin + 3) & ~3) - 4;
    unsigned xo[XP], zsrc = 0, wo[WP];
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int i = tid + NTH * j;
        int row = i / xw4;
        const int c4 = i - row * xw4;
        row = row < BK ? row : 0;
        int pos = pwin0 + 4 * c4;
        zsrc |= pos < 0 ? (1u << j) : 0u;
        pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
        xo[j] = (unsigned)row * (unsigned)a.x_cs + (unsigned)pos;
    }
#pragma unroll
    for (int k = 0; k < WP; k++) {                           // float4 index tid + NTH k of the [8 slots][16][BN] image
        const int i = tid + NTH * k, slot = i / (BK * BN / 4), rem = i - slot * (BK * BN / 4), r = rem / (BN / 4), c4 = rem - r * (BN / 4);
        const int point = slot < 3 ? slot : (slot == 3 ? 7 : slot - 1);           // slots: 0 +1 -1 inf | +2 -2 +1/2 -1/2
        wo[k] = ((unsigned)point * (unsigned)CinP + (unsigned)r) * (unsigned)a.NP + (unsigned)c4 * 4u;
    }
    auto stage = [&](int cb, int buf) {
        const float *xb = xu + (long)cb * BK * a.x_cs;
        float *xd = Xs + buf * XBUF + wave * 256;
#pragma unroll
        for (int j = 0; j < XP; j++) {
            if (NTH * (j + 1) <= BK * xw4 || tid + NTH * j < BK * xw4) {
                if (wg_zero) wn_dma16((zsrc >> j) & 1u ? a.zeros : xb + xo[j], xd + j * (NTH * 4));
                else wn_dma16_s(xb, xo[j] * 4u, xd + j * (NTH * 4));
            }
        }
        const float *wb = a.ww + ((long)cb * BK) * a.NP + n0;
        float *wd = Ws + buf * WBUF + wave * 256;
#pragma unroll
        for (int k = 0; k < WP; k++) wn_dma16_s(wb, wo[k] * 4u, wd + k * (NTH * 4));
    };

    // The K loop, specialised per point group (a wave-uniform choice made once, outside the loop).
    auto main_loop = [&](auto pgc) {
        constexpr int PG = decltype(pgc)::value;
        for (int cb = 0; cb < ncb; cb++) {
            const int buf = cb & 1;
            if (!(PS_ABL & 8) && cb + 1 < ncb) stage(cb + 1, buf ^ 1);
            {
                const float *xs = Xs + buf * XBUF + xoff;
                const float *ws = Ws + buf * WBUF + PG * (4 * BK * BN) + l31;
                float dv[3][8], vv[2][4], av[2][4][RT];
                auto load_d = [&](int kk, int sl) {
                    const float *xk = xs + (2 * kk + h) * XW;
#pragma unroll
                    for (int i = (PG ? 1 : 0); i < (PG ? 7 : 8); i++) dv[sl][i] = (PS_ABL & 2) ? (float)(kk + i) : WN_LDS(xk + i * D);     // (the +-2, +-1/2 points do not see d0, d7)
                };
                auto load_a = [&](int kk, int sl) {
                    const float *wk = ws + (2 * kk + h) * BN;
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int i = 0; i < RT; i++) av[sl][q][i] = (PS_ABL & 4) ? (float)(q + i + kk) : WN_LDS(wk + q * (BK * BN) + 32 * i);
                };
                auto transform = [&](int ds, int vs) {      // 12 VALU instructions either way
                    const float *d = dv[ds];
                    if constexpr ((PS_ABL & 1) != 0) {
#pragma unroll
                        for (int q = 0; q < 4; q++) vv[vs][q] = d[q + 1];
                    } else if constexpr (PG == 0) {
                        const float e12 = fmaf(-4.25f, d[4], d[2] + d[6]), o12 = fmaf(-4.25f, d[3], d[1] + d[5]);
                        vv[vs][0] = fmaf(5.25f, d[4] - d[2], d[0] - d[6]);
                        vv[vs][1] = e12 + o12;
                        vv[vs][2] = e12 - o12;
                        vv[vs][3] = fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
                    } else {
                        const float e34 = fmaf(0.25f, d[2], fmaf(-1.25f, d[4], d[6])), o34 = fmaf(4.0f, d[5], fmaf(-5.0f, d[3], d[1]));
                        const float e56 = fmaf(4.0f, d[2], fmaf(-5.0f, d[4], d[6])), o56 = fmaf(0.25f, d[5], fmaf(-1.25f, d[3], d[1]));
                        vv[vs][0] = fmaf(0.5f, o34, e34);
                        vv[vs][1] = fmaf(-0.5f, o34, e34);
                        vv[vs][2] = fmaf(2.0f, o56, e56);
                        vv[vs][3] = fmaf(-2.0f, o56, e56);
                    }
                };
                load_d(0, 0);
                load_a(0, 0);
                load_d(1, 1);
                transform(0, 0);
#pragma unroll
                for (int kk = 0; kk < BK / 2; kk++) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk + 1 < BK / 2) load_a(kk + 1, (kk + 1) & 1);
                    if (kk + 2 < BK / 2) load_d(kk + 2, (kk + 2) % 3);
                    if (kk + 1 < BK / 2) transform((kk + 1) % 3, (kk + 1) & 1);
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int i = 0; i < RT; i++)
                            acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][q][i], vv[kk & 1][q], acc[q][i], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4 * RT; g++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, RT == 2 ? 2 : 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(PS_ABL & 16)) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };

    stage(0, 0);                                             // (first: the parameter tables are filled while it flies)
    fill_param_table<BN, FUSE>(a, Pm, n0, tid, NTH);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (pg == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});

    // ---- the staging buffers are idle (the loop ended with a barrier): partial output transforms, exchange, finish
    float *Ex = Xs;                                          // [8 waves][RT * 16][64 lanes]
    constexpr bool TWO = GEO::two_phase(FUSE);
    float *W2s = Xs + (TWO ? GEO::EXH : GEO::EX);            // FUSE: [BN k rows][BN]
    if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < GEO::W2P; j++) {
            const int i = tid + NTH * j;
            if (NTH * (j + 1) <= GEO::W2F / 4 || i < GEO::W2F / 4) wn_dma16(a.w2 + 4 * (long)i, W2s + wave * 256 + j * (NTH * 4));
        }
    }
    float yv[RT][16], snd[TWO ? RT : 1][16];
    {
        float *exw = Ex + (wave * RT * 16) * 64 + lane;
        if (pg == 0) {                                       // M0 M1 M2 M7 here: keep M0 + M1 + M2, send M1 - M2 + M7
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float m0 = acc[0][i][r], m1 = acc[1][i][r], m2 = acc[2][i][r], m3 = acc[3][i][r];
                    yv[i][r] = (m1 + m2) + m0;
                    if constexpr (TWO) snd[i][r] = (m1 - m2) + m3;
                    else exw[(i * 16 + r) * 64] = (m1 - m2) + m3;
                }
        } else {                                             // M3 M4 M5 M6 here: keep 2 (M3 - M4) + (M5 - M6) / 2, send their sum
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float m0 = acc[0][i][r], m1 = acc[1][i][r], m2 = acc[2][i][r], m3 = acc[3][i][r];
                    yv[i][r] = fmaf(2.0f, m0 - m1, 0.5f * (m2 - m3));
                    if constexpr (TWO) snd[i][r] = (m0 + m1) + (m2 + m3);
                    else exw[(i * 16 + r) * 64] = (m0 + m1) + (m2 + m3);
                }
        }
    }
    const int pos = p + pg * D;                              // this wave finishes y[p] (point group 0) or y[p + D] (group 1)
    const bool ok = pos < a.M;
    // FUSE: the residual x[row][pos] of all BN rows is requested BEFORE the exchange barrier (the accumulators are dead, the
    // registers free) and used after the second product.  vmcnt retires in order and the W2 pieces were issued first, so the
    // barrier waits for those only: vmcnt(RT * 16) leaves the residual loads in flight.
    float resv[FUSE ? RT : 1][16];
    if constexpr (FUSE) {
        const float *rup = a.res + (long)unit * a.y_us + (ok ? pos : 0);
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) resv[i][r] = (PS_ABL & 128) ? 0.0f : rup[(long)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * a.y_cs];
        static_assert(RT * 16 <= 63, "vmcnt range");
    }
    if constexpr (TWO) {
        // one direction at a time through [tile group][RT * 16][lane]: group 1 -> group 0, then group 0 -> group 1
        float *exh = Ex + (tw * RT * 16) * 64 + lane;
        if (pg == 1) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) exh[(i * 16 + r) * 64] = snd[i][r];
        }
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((PS_ABL & 128) ? 0 : RT * 16) : "memory");     // (also: W2 has landed)
        if (pg == 0) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) yv[i][r] += exh[(i * 16 + r) * 64];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                              // group 0 has read
        if (pg == 0) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) exh[(i * 16 + r) * 64] = snd[i][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pg == 1) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) yv[i][r] += exh[(i * 16 + r) * 64];
        }
    } else {
        if constexpr (FUSE) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((PS_ABL & 128) ? 0 : RT * 16) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float *exr = Ex + ((wave ^ NTG) * RT * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) yv[i][r] += exr[(i * 16 + r) * 64];
    }
    float *yu = a.y + (long)unit * a.y_us;

    if constexpr (!FUSE) {
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;          // rows row, row + 1
                f32x2 o = mk2(yv[i][r], yv[i][r + 1]);
                if (a.bias) o = o + mk2(Pm[row], Pm[row + 1]);
                if (a.psn_a) o = snake_beta2(o, mk2(Pm[2 * BN + row], Pm[2 * BN + row + 1]), mk2(Pm[3 * BN + row], Pm[3 * BN + row + 1]));
                float *yr = yu + (long)(n0 + row) * a.y_cs + pos;
                if (ok) { yr[0] = o.x; yr[a.y_cs] = o.y; }
            }
    } else {
        // H = snake2(y + b1): the second product's B operand; register r of tile i is channel i*32 + (r&3) + 8*(r>>2) + 4h
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x2 hv = (PS_ABL & 64) ? mk2(yv[i][r] + Pm[n], yv[i][r + 1] + Pm[n + 1])
                                               : snake_beta2(mk2(yv[i][r] + Pm[n], yv[i][r + 1] + Pm[n + 1]), mk2(Pm[BN + n], Pm[BN + n + 1]),
                                                             mk2(Pm[2 * BN + n], Pm[2 * BN + n + 1]));
                yv[i][r] = hv.x;
                yv[i][r + 1] = hv.y;
            }
        float *y2u = a.y2 ? a.y2 + (long)unit * a.y_us : nullptr;
        // the 1x1 conv: all RT output tiles at once -- RT independent accumulator chains, each B value feeds RT MFMAs
        f32x16 acc2[RT];
#pragma unroll
        for (int o = 0; o < RT; o++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[o][r] = 0.0f;
        const float *ws2 = W2s + l31 + 4 * h * BN;
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = i * 32 + (r & 3) + 8 * (r >> 2);                    // this lane half's k is k + 4h
#pragma unroll
                for (int o = 0; o < RT; o++) acc2[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws2[k * BN + o * 32], yv[i][r], acc2[o], 0, 0, 0);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);              // (bounds how many LDS reads are hoisted)
            }
#pragma unroll
        for (int o = 0; o < RT; o++) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int n = o * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                f32x2 v = mk2(acc2[o][r] + Pm[3 * BN + n] + resv[o][r], acc2[o][r + 1] + Pm[3 * BN + n + 1] + resv[o][r + 1]);
                float *yr = yu + (long)n * a.y_cs + pos;
                const bool st = (PS_ABL & 32) ? (ok && v.x == 1234.5f) : ok;      // (timing-only ablation: no stores)
                if (a.psn_a) {
                    const f32x2 sv = (PS_ABL & 64) ? v : snake_beta2(v, mk2(Pm[4 * BN + n], Pm[4 * BN + n + 1]), mk2(Pm[5 * BN + n], Pm[5 * BN + n + 1]));
                    if (y2u) {
                        float *y2r = y2u + (long)n * a.y_cs + pos;
                        if (st) { y2r[0] = sv.x; y2r[a.y_cs] = sv.y; }
                    } else
                        v = sv;
                }
                if (st) { yr[0] = v.x; yr[a.y_cs] = v.y; }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// The decoder blocks' transposed convs (kernel 2s, stride s) by F(3,2).  Per output phase r such a conv is a 2-tap filter
// along the INPUT positions q:  y[co][q s + r] = g0 x[q + sh] + g1 x[q + sh + 1]  (g0 = W[..][r + s], g1 = W[..][r], rows
// n = (co, r); sh = 0, or -1 with the right-only trim).  F(3,2) makes three consecutive q from four products,
//     V = (d0 - d2, d1 + d2, d2 - d1, d3 - d1),   U = (g0, (g0 + g1)/2, (g0 - g1)/2, g1),
//     y0 = M0 + M1 + M2,  y1 = M1 - M2,  y2 = M1 + M2 + M3,
// so the matrix cores do 4/6 of the direct kernel's work; the constants are 1 and 1/2, the rounding is that of the direct
// form to within a few ulp.  Same structure as wino_conv7_kernel: lane = tile (three q), LDS-DMA staging, one barrier per
// block of 16 input channels, three-deep software pipeline.  Workgroup = RW wave-rows x 4 tile groups; a wave owns RT row
// tiles of 32 rows x 32 tiles and 4 x RT accumulator tiles: (RT, RW) = (2, 2) -> 128 rows, 8 waves, two per SIMD; (1, 3) ->
// 96 rows, 12 waves, three per SIMD (N = 288 of the last block, whose time goes into its stores).
static constexpr int WT_PTS = 4;

// NTG = tile groups of 32 tiles per workgroup: 4, or 3 (96 tiles: a unit of 85 tiles -- block 0's 256 input steps -- fills 89 % of
// such a workgroup instead of 66 % of a 128-tile one)
template <int RT, int RW, int NTG, int BK = WN_BK>
struct TconvGeom {
    static constexpr int NTH = NTG * RW * 64;
    static constexpr int NT = 32 * NTG;                                      // tiles per workgroup
    static constexpr int BN = 32 * RT * RW;                                  // rows per workgroup
    static constexpr int XW = 3 * NT + 8;                                    // window: positions 3 T0 + sh (floored to 4) .. + 3 NT + 3
    static constexpr int XP = (BK * (XW / 4) + NTH - 1) / NTH;
    static constexpr int XBUF = XP * NTH * 4;
    static constexpr int WF4 = WT_PTS * BK * BN / 4;                      // float4 of the weight tiles of one block
    static constexpr int WP = (WF4 + NTH - 1) / NTH;
    static constexpr int WBUF = WP * NTH * 4;
};

template <bool SN, int RT, int RW, int NTG, int BK = WN_BK>
__global__ __launch_bounds__((TconvGeom<RT, RW, NTG, BK>::NTH), (BK == 8 ? 3 : 1)) void wino_tconv_kernel(ConvGemmArgs a, int wg_per_unit, int n_tiles) {
    using GEO = TconvGeom<RT, RW, NTG, BK>;
    constexpr int NTH = GEO::NTH, BN = GEO::BN, XW = GEO::XW, XP = GEO::XP, XBUF = GEO::XBUF, WP = GEO::WP, WBUF = GEO::WBUF, xw4 = XW / 4;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                                   // [4][BN] per-row parameters: bias | - | psn_a | psn_ib
    float *Sp = smem_all + 4 * BN;                          // [Cin][2] (SN)
    float *Xs = Sp + (SN ? 2 * a.Cin : 0);                  // [2][XBUF]: rows of XW floats
    float *Ws = Xs + 2 * XBUF;                              // [2][4][16][BN]

    int nwg = gridDim.x, orig = blockIdx.x;                 // XCD-aware remap, as conv_gemm_kernel
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt = id / n_tiles;
    const int unit = pt / wg_per_unit;
    const int T0 = (pt - unit * wg_per_unit) * GEO::NT;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int tw = wave % NTG, rw = wave / NTG;             // tile group, wave-row
    const float *xu = a.x + (long)unit * a.x_us;
    const int CinP = ((a.Cin + 31) / 32) * 32;
    const int ncb = a.Cin / BK;
    const int sh = a.shift0;

    const int pwin0 = (3 * T0 + sh) & ~3;                   // first position of the window (floor to 4, also when negative)
    const int q0 = 3 * (T0 + tw * 32 + l31);                // this lane's tile: input steps q0, q0 + 1, q0 + 2
    const int xoff = q0 + sh - pwin0;                       // inputs at xoff .. xoff + 3
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pi = q0 + sh + i;
        dmask |= (pi >= 0 && pi < a.Lin) ? (1u << i) : 0u;
    }


    f32x16 acc[WT_PTS][RT];
#pragma unroll
    for (int k = 0; k < WT_PTS; k++)
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][i][r] = 0.0f;

    const int xlim = ((a.Lin + 3) & ~3) - 4;
    unsigned xo[XP], wo[WP];
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int i = tid + NTH * j;
        int row = i / xw4;
        const int c4 = i - row * xw4;
        row = row < BK ? row : 0;                         // pieces past the window land in the buffer's padding
        int pos = pwin0 + 4 * c4;
        pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
        xo[j] = (unsigned)row * (unsigned)a.x_cs + (unsigned)pos;
    }
#pragma unroll
    for (int k = 0; k < WP; k++) {                           // float4 index tid + NTH k of the [4][16][BN] image
        int i = tid + NTH * k;
        i = i < GEO::WF4 ? i : 0;                            // (pieces past the image re-read its start into the padding)
        const int q = i / (BK * BN / 4), rem = i - q * (BK * BN / 4), r = rem / (BN / 4), c4 = rem - r * (BN / 4);
        wo[k] = ((unsigned)q * (unsigned)CinP + (unsigned)r) * (unsigned)a.NP + (unsigned)c4 * 4u;
    }
    auto stage = [&](int cb, int buf) {
        const float *xb = xu + (long)cb * BK * a.x_cs;
        float *xd = Xs + buf * XBUF + wave * 256;
#pragma unroll
        for (int j = 0; j < XP; j++)
            wn_dma16(xb + xo[j], xd + j * (NTH * 4));
        const float *wb = a.ww + ((long)cb * BK) * a.NP + n0;
        float *wd = Ws + buf * WBUF + wave * 256;
#pragma unroll
        for (int k = 0; k < WP; k++)
            wn_dma16(wb + wo[k], wd + k * (NTH * 4));
    };

    stage(0, 0);                                             // (first: the parameter tables are filled while it flies)
    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    if constexpr (SN) {
        for (int c = tid; c < a.Cin; c += NTH) {
            Sp[2 * c] = a.sn_a[c];
            Sp[2 * c + 1] = a.sn_ib[c];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int cb = 0; cb < ncb; cb++) {
        const int buf = cb & 1;
        if (cb + 1 < ncb) stage(cb + 1, buf ^ 1);
        {
            const float *xs = Xs + buf * XBUF + xoff;
            const float *ws = Ws + buf * WBUF + l31 + rw * (32 * RT);
            const float *sp = Sp + 2 * (cb * BK + h);
            float dv[3][4], vv[2][WT_PTS], av[2][WT_PTS][RT];
            float2 sv[3];
            auto load_d = [&](int kk, int sl) {
                const float *xk = xs + (2 * kk + h) * XW;
#pragma unroll
                for (int i = 0; i < 4; i++) dv[sl][i] = WN_LDS(xk + i);
                if constexpr (SN) sv[sl] = *reinterpret_cast<const float2 *>(sp + 4 * kk);
            };
            auto load_a = [&](int kk, int sl) {
                const float *wk = ws + (2 * kk + h) * BN;
#pragma unroll
                for (int q = 0; q < WT_PTS; q++)
#pragma unroll
                    for (int i = 0; i < RT; i++) av[sl][q][i] = WN_LDS(wk + q * (BK * BN) + 32 * i);
            };
            auto transform = [&](int ds, int vs) {
                float d[4];
#pragma unroll
                for (int i = 0; i < 4; i++) d[i] = (dmask >> i) & 1u ? dv[ds][i] : 0.0f;
                if constexpr (SN) {
                    const f32x2 ea = (f32x2)(sv[ds].x), ib = (f32x2)(sv[ds].y);
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        const f32x2 r = snake_beta2(mk2(d[i], d[i + 1]), ea, ib);
                        d[i] = r.x;
                        d[i + 1] = r.y;
                    }
                }
                vv[vs][0] = d[0] - d[2];
                vv[vs][1] = d[1] + d[2];
                vv[vs][2] = d[2] - d[1];
                vv[vs][3] = d[3] - d[1];
            };
            load_d(0, 0);
            load_a(0, 0);
            load_d(1, 1);
            transform(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) {
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < BK / 2) load_d(kk + 2, (kk + 2) % 3);
                if (kk + 1 < BK / 2) {
                    load_a(kk + 1, (kk + 1) & 1);
                    transform((kk + 1) % 3, (kk + 1) & 1);
                }
#pragma unroll
                for (int q = 0; q < WT_PTS; q++)
#pragma unroll
                    for (int i = 0; i < RT; i++)
                        acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][q][i], vv[kk & 1][q], acc[q][i], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < WT_PTS * RT; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // LDS reads
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU in its shadow
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- output transform and epilogue.  A lane holds three input steps of 16 RT rows; in memory a row (co, phase) is
    // every S-th float of channel co, so storing from the accumulator layout would touch 3 S cache lines per 128 bytes
    // written (measured: the stores, not the MFMAs, set the pace).  Instead every wave transposes its 32-row tile through
    // its own slice of the (now idle) LDS and writes each channel's samples as one contiguous run, 256 bytes per
    // instruction.  Rows are channel-major (n = co S + phase; the loader packs them so for this kernel).
    float *yu = a.y + (long)unit * a.y_us;
    float *y2u = a.y2 ? a.y2 + (long)unit * a.y_us : nullptr;
    const int S = a.S;
    constexpr int LTW = 97;                                   // padded row of the transposition tile (96 input steps)
    float *Lt = Xs + wave * (32 * LTW);                       // this wave's slice (the loop ended with a barrier)
    const int q0w = 3 * (T0 + tw * 32);                       // first input step of the wave's tiles
    const int nq = a.M - q0w < 96 ? a.M - q0w : 96;           // steps of it that exist (<= 0: nothing to store)
    const unsigned magic = 0xffffffffu / (unsigned)S + 1u;    // m / S for m < 2^16 as a multiply-high
#pragma unroll
    for (int i = 0; i < RT; i++) {
        const int tr = (rw * RT + i) * 32;                    // first row of this tile inside the workgroup's rows
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float m0 = acc[0][i][r], m1 = acc[1][i][r], m2 = acc[2][i][r], m3 = acc[3][i][r];
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float b = a.bias ? Pm[tr + row] : 0.0f;
            float *lp = Lt + row * LTW + 3 * l31;
            lp[0] = ((m1 + m2) + m0) + b;
            lp[1] = (m1 - m2) + b;
            lp[2] = ((m1 + m2) + m3) + b;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (one wave: its LDS operations complete in order)
        const int nw = n0 + tr;                               // first global row of the tile
        const int co_lo = nw / S, co_hi = (nw + 31) / S;
        for (int co = co_lo; co <= co_hi; co++) {
            const int rbase = co * S - nw;                    // tile row of phase 0 of this channel (may be < 0)
            const int ph_lo = rbase < 0 ? -rbase : 0, ph_hi = rbase + S - 1 > 31 ? 31 - rbase : S - 1;
            const long base = (long)co * a.y_cs + (long)q0w * S;
            for (int m = lane; m < ((PS_ABL & 256) ? (nq * S == 12345 ? 1 : 0) : nq * S); m += 64) {
                const int ql = (int)__umulhi((unsigned)m, magic), ph = m - ql * S;
                if (ph < ph_lo || ph > ph_hi) continue;
                const int row = rbase + ph;
                float v = Lt[row * LTW + ql];
                if (a.psn_a) {
                    const float sv2 = snake_beta(v, Pm[2 * BN + tr + row], Pm[3 * BN + tr + row]);
                    if (y2u) y2u[base + m] = sv2;
                    else v = sv2;
                }
                yu[base + m] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next tile overwrites the slice
    }
}

// transposed conv of a decoder block in its Winograd form?
static bool wino_tconv_selected(const ConvGemmArgs &a) {
    return a.ntaps == 2 && a.S > 1 && a.shift_step == 1 && (a.shift0 == 0 || a.shift0 == -1) && a.units >= 1 && a.N == a.NP &&
           (a.N % 128 == 0 || a.N % 96 == 0) && a.N % a.S == 0 && a.Cin % WN_BK == 0 && !a.res && !a.rscale && !a.xb && !a.yb &&
           a.act == ACT_NONE && !a.in_elu && a.s2d <= 1 && a.y && (!a.y2 || a.psn_a) && a.M >= 1 && a.Lin >= 1 && !a.pm && a.S <= 32;
}

// a whole residual unit of the last decoder block (N = Cin = 96) on wino_ps_kernel<D, 3, true>?
static bool wino_unit_selected(const ConvGemmArgs &a) {
    const int d = a.shift_step;
    return a.ww && a.w2 && !a.wb && a.ntaps == 7 && a.S == 1 && (d == 1 || d == 3 || d == 9) && a.shift0 == -6 * d && a.units >= 1 &&
           a.N == 96 && a.NP == 96 && a.Cin == 96 && !a.sn_a && a.res && !a.rscale && !a.xb && !a.yb && a.act == ACT_NONE && !a.in_elu &&
           a.s2d <= 1 && a.M == a.Lin && a.y && a.bias && a.bias2 && a.sn2_a && a.sn2_ib && (!a.y2 || a.psn_a) && (!a.psn_a || a.psn_ib);
}

// 128-row workgroups (8 waves) or 96-row ones (12 waves)?  96 when N allows nothing else, and on small launches (a single
// utterance) where the 128-row grid would leave CUs idle: a third more workgroups, each a quarter shorter.  Per output element
// both run the same chain of MFMAs and the same transforms: bit-identical (the batch == single tests cover it).
static bool wino_tconv_rows128(const ConvGemmArgs &a) {
    if (a.N % 128) return false;
    if (a.N % 96) return true;
    const long n_tl = (a.M + 2) / 3, wgs = ((n_tl + WN_NT - 1) / WN_NT) * (long)a.units * (a.N / 128);
    return wgs > 192;
}

// ... and 96-tile workgroups (12 waves x one row tile, 128 rows) where they fill visibly better than 128-tile ones
static bool wino_tconv_tiles96(const ConvGemmArgs &a) {
    if (!wino_tconv_rows128(a)) return false;
    const char *e = getenv("HIPVOC_WINO_TCONV_NT96");
    if (e && atoi(e) == 0) return false;
    if (e && atoi(e) == 2) return true;                      // (experiment: always)
    const long n_tl = (a.M + 2) / 3;
    const double f128 = (double)n_tl / (double)(((n_tl + 127) / 128) * 128), f96 = (double)n_tl / (double)(((n_tl + 95) / 96) * 96);
    return f96 > f128 + 0.1;
}

// Workgroup shape of a transposed conv: {RT, RW, NTG}.  Large launches: the rules above (measured at 64 units).  Small launches
// (a single utterance: the default shape's grid is at most HIPVOC_WINO_TCONV_SMALL workgroups, default 400; 0: never) pick the
// cheapest of the instantiated shapes by rounds x work per SIMD: a grid of 128 12-wave workgroups leaves half the chip idle, one
// of 324 runs two rounds for 1.3.  The 64-row shapes (RT 1, RW 2: 6 or 8 waves) exist for this.  Per output element every shape
// runs the same chain of MFMAs and the same transforms: bit-identical (the batch == single tests cover it).
struct TconvShape { int rt, rw, ntg, bk = WN_BK; };
static TconvShape wino_tconv_shape(const ConvGemmArgs &a) {
    TconvShape dflt = wino_tconv_tiles96(a) ? TconvShape{1, 4, 3} : wino_tconv_rows128(a) ? TconvShape{2, 2, 4} : TconvShape{1, 3, 4};
    // HIPVOC_WINO_TCONV_W4=1 (experiment): 128-row x 128-tile layers in 4-wave workgroups of 64 rows, 8-channel blocks, three per CU
    // 128-row x 128-tile layers (blocks 1 and 2) run in 4-wave workgroups of 64 rows with 8-channel blocks, three per CU (50 KB of
    // LDS, 166 registers), so that one workgroup's barriers and epilogue overlap the others' MFMAs: 5.1 -> 4.75 ms per layer
    // (HIPVOC_WINO_TCONV_W4=0: the 8-wave shape).  Measured out for the other two layers: block 0 in 3 waves x 64 rows x 96 tiles
    // (4.04 -> 4.73 ms: K = 1536 is 192 barriers, and the weight tiles are fetched twice as often), block 3 in 6 waves x 96 rows x
    // 64 tiles (4.69 -> 5.36 ms).
    {
        const char *e4 = getenv("HIPVOC_WINO_TCONV_W4");
        if (dflt.rt == 2 && !(e4 && atoi(e4) == 0)) dflt = TconvShape{2, 1, 4, 8};
    }
    const char *e = getenv("HIPVOC_WINO_TCONV_SMALL");
    const long max_wgs = e ? atol(e) : 400;
    const long n_tl = (a.M + 2) / 3;
    auto wgs = [&](const TconvShape &t) { return ((n_tl + 32 * t.ntg - 1) / (32 * t.ntg)) * (long)a.units * (a.N / (32 * t.rt * t.rw)); };
    if (max_wgs <= 0 || wgs(dflt) > max_wgs) return dflt;
    auto cost = [&](const TconvShape &t) {                 // rounds over 256 CUs (one workgroup each) x MFMAs per SIMD and channel block
        return ((wgs(t) + 255) / 256) * t.rt * ((t.ntg * t.rw + 3) / 4);
    };
    static const TconvShape cand[] = {{1, 2, 3}, {1, 2, 4}, {1, 3, 4}, {1, 4, 3}, {2, 2, 4}};
    TconvShape best = dflt;
    for (const TconvShape &t : cand)
        if (a.N % (32 * t.rt * t.rw) == 0 && cost(t) < cost(best)) best = t;
    return best;
}

bool wino_selected(const ConvGemmArgs &a) {
    if (wino_unit_selected(a)) return true;
    if (!a.ww || a.wb || a.w2) return false;
    if (wino_tconv_selected(a)) return true;
    const int d = a.shift_step;
    return a.ntaps == 7 && a.S == 1 && (d == 1 || d == 3 || d == 9) && a.shift0 == -6 * d && a.units >= 1 && a.N == a.NP &&
           a.N % 64 == 0 && a.Cin % WN_BK == 0 && !a.res && !a.rscale && !a.y2 && !a.xb && !a.yb && a.act == ACT_NONE && !a.in_elu &&
           a.s2d <= 1 && a.M == a.Lin && a.y;
}

static int wino_rt();
// Small launches of the default conv7 form (a single utterance: decoder[0] is 96 workgroups, block 0's layers 384, for 768 slots):
// one row tile per wave instead of two -- twice the workgroups, each wave's MFMA chain half as long.  Used when the two-tile grid
// is at most HIPVOC_WINO_PS_SMALL workgroups (default 448; 0: never).  Same chains per output element: bit-identical.
static bool wino_ps_rt1(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_WINO_PS_SMALL");
    const long max_wgs = e ? atol(e) : 448;
    const int d = a.shift_step;
    if (max_wgs <= 0 || a.sn_a || d == 9) return false;      // (dilation 9: the window's 54-position halo outweighs it -- 0.217 -> 0.239 ms on block 0)
    const long n_pairs = (a.M + 2 * d - 1) / (2 * d) * d, wgs = ((n_pairs + 63) / 64) * (long)a.units * (a.N / 64);
    return wgs <= max_wgs;
}
// HIPVOC_WINO_UNIT_WG: shape of the fused Winograd unit: 8 = 8 waves, 128 tiles, 16-channel blocks, one workgroup per CU;
// 4 = 4 waves, 64 tiles, 8-channel blocks, two-phase exchange, two workgroups per CU
static bool wino_unit_small() {
    const char *e = getenv("HIPVOC_WINO_UNIT_WG");
    return (e ? atoi(e) : WN_UNIT_WG_DEFAULT) == 4;
}
const char *wino_kernel_name(const ConvGemmArgs &a) {      // as rocprofv3 prints it
    if (wino_unit_selected(a)) {
        if (wino_unit_small()) return a.shift_step == 1 ? "wino_ps_kernel<1,3,true,2,8>" : a.shift_step == 3 ? "wino_ps_kernel<3,3,true,2,8>" : "wino_ps_kernel<9,3,true,2,8>";
        return a.shift_step == 1 ? "wino_ps_kernel<1,3,true,4,16>" : a.shift_step == 3 ? "wino_ps_kernel<3,3,true,4,16>" : "wino_ps_kernel<9,3,true,4,16>";
    }
    if (wino_tconv_selected(a)) {
        static char tn[2][3][5][5][2][40];                  // (all five template arguments, as rocprofv3 prints the symbol)
        const TconvShape t = wino_tconv_shape(a);
        char *n = tn[a.sn_a ? 1 : 0][t.rt][t.rw][t.ntg][t.bk == 8];
        if (!n[0]) snprintf(n, 40, "wino_tconv_kernel<%s,%d,%d,%d,%d>", a.sn_a ? "true" : "false", t.rt, t.rw, t.ntg, t.bk);
        return n;
    }
    static char names[2][3][5][44];
    const int di = a.shift_step == 1 ? 0 : a.shift_step == 3 ? 1 : 2, sn = a.sn_a ? 1 : 0, mode = sn && (wino_rt() == 3 || wino_rt() == 5) ? 1 : wino_rt();
    if (mode == 5 && wino_ps_rt1(a)) return di == 0 ? "wino_ps_kernel<1,1,false,2,8>" : di == 1 ? "wino_ps_kernel<3,1,false,2,8>" : "wino_ps_kernel<9,1,false,2,8>";
    const int mi = mode == 1 ? 0 : mode == 2 ? 1 : mode == 4 ? 2 : mode == 3 ? 3 : 4;
    char *n = names[sn][di][mi];
    if (!n[0]) {
        if (mode == 3) snprintf(n, 44, "wino_ps_kernel<%d,2,false,4,16>", a.shift_step);
        else if (mode == 5) snprintf(n, 44, "wino_ps_kernel<%d,2,false,2,8>", a.shift_step);
        else snprintf(n, 44, "wino_conv7_kernel<%d,%s,%d,%d>", a.shift_step, sn ? "true" : "false", mode == 2 ? 2 : 1, mode == 1 ? 2 : 1);
    }
    return n;
}

// HIPVOC_WINO_RT: shape of a workgroup of wino_conv7_kernel<D, SN, RT, RH> (RT row tiles per wave, RH row groups of waves).
//   1:           RT 1, RH 2 -- 64 rows, 8 waves, two per SIMD
//   2:           RT 2, RH 1 -- 64 rows, 4 waves, one per SIMD (256 accumulator registers)
//   4:           RT 1, RH 1 -- 32 rows, 4 waves; two INDEPENDENT workgroups per CU, so one's barrier and block prologue
//                              overlap the other's MFMAs
//   3:           wino_ps_kernel<D, 2, false, 4, 16>: 64 rows, 8 waves, the eight points split over two waves (no SN form:
//                              such layers run in shape 1)
//   5 (default): wino_ps_kernel<D, 2, false, 2, 8>: the same in 4-wave workgroups of 64 tiles and 8-channel blocks, two
//                              independent workgroups per CU
static int wino_rt() {                                   // (read at every launch: a handful of getenv calls per pass; lets one process compare shapes)
    const char *e = getenv("HIPVOC_WINO_RT");
    const int x = e ? atoi(e) : WN_MODE_DEFAULT;
    return x == 2 ? 2 : x == 4 ? 4 : x == 3 ? 3 : x == 5 ? 5 : x == 1 ? 1 : WN_MODE_DEFAULT;
}

template <int D, bool SN, int RT, int RH>
static hipError_t wino_launch_one(const ConvGemmArgs &a, hipStream_t s) {
    using GEO = WinoGeom<D, RT, RH>;
    // tiles of a unit: the pairs (p, p + D) with floor(p / D) even and p < M
    const int n_pairs = (a.M + 2 * D - 1) / (2 * D) * D;
    const int wg_per_unit = (n_pairs + WN_NT - 1) / WN_NT;
    const int n_tiles = a.N / GEO::BN;
    const long grid = (long)wg_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    // Window bound: a workgroup's tiles T0 .. T0+127 have base positions within 2D (127/D + 1) + D - 1 <= 253 + 3D of the
    // first one, the window starts at most 6D + 3 before it and the last read is D past the base: XW >= 257 + 10 D.
    const size_t lds = sizeof(float) * (4 * GEO::BN + (SN ? 2 * (size_t)a.Cin : 0) + 2 * (size_t)GEO::XBUF + 2 * (size_t)GEO::WBUF);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv7_kernel<D, SN, RT, RH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((wino_conv7_kernel<D, SN, RT, RH>), dim3((unsigned)grid), dim3(GEO::NTH), lds, s, a, wg_per_unit, n_tiles);
    return hipGetLastError();
}

template <int D, int RT, bool FUSE, int NTG, int BK>
static hipError_t wino_ps_launch_one(const ConvGemmArgs &a, hipStream_t s) {
    using GEO = PsGeom<D, RT, NTG, BK>;
    if (a.Cin % BK) return hipErrorInvalidValue;
    const int n_pairs = (a.M + 2 * D - 1) / (2 * D) * D;     // tiles of a unit, as wino_launch_one
    const int wg_per_unit = (n_pairs + GEO::NT - 1) / GEO::NT;
    const int n_tiles = a.N / GEO::BN;
    const long grid = (long)wg_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * ((FUSE ? 6 : 4) * GEO::BN + (size_t)GEO::WORK(FUSE));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_ps_kernel<D, RT, FUSE, NTG, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((wino_ps_kernel<D, RT, FUSE, NTG, BK>), dim3((unsigned)grid), dim3(GEO::NTH), lds, s, a, wg_per_unit, n_tiles);
    return hipGetLastError();
}

template <bool SN, int RT, int RW, int NTG, int BK = WN_BK>
static hipError_t wino_tconv_launch_one(const ConvGemmArgs &a, hipStream_t s) {
    using GEO = TconvGeom<RT, RW, NTG, BK>;
    const int n_tl = (a.M + 2) / 3;                          // tiles of three input steps
    const int wg_per_unit = (n_tl + GEO::NT - 1) / GEO::NT;
    const int n_tiles = a.N / GEO::BN;
    const long grid = (long)wg_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    // window: a workgroup's 128 tiles start at input position 3 T0 + sh, the window at most 3 before it, and the last
    // read is 3*127 + 3 past the first tile's: XW = 3*128 + 8 covers it
    size_t work = 2 * (size_t)GEO::XBUF + 2 * (size_t)GEO::WBUF;                 // staging buffers ...
    const size_t lt = (size_t)(GEO::NTH / 64) * 32 * 97;                           // ... reused by the epilogue's transposition tiles
    if (lt > work) work = lt;
    const size_t lds = sizeof(float) * (4 * GEO::BN + (SN ? 2 * (size_t)a.Cin : 0) + work);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_tconv_kernel<SN, RT, RW, NTG, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((wino_tconv_kernel<SN, RT, RW, NTG, BK>), dim3((unsigned)grid), dim3(GEO::NTH), lds, s, a, wg_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_wino_conv7(const ConvGemmArgs &a, hipStream_t s) {
    if (!wino_selected(a)) return hipErrorInvalidValue;
    if (a.M <= 0) return hipSuccess;
    if (!a.zeros) return hipErrorInvalidValue;
    if (wino_unit_selected(a)) {
        if ((a.x_cs & 3) || (a.x_us & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.ww & 15) || ((uintptr_t)a.w2 & 15) || a.Lin < 1) return hipErrorInvalidValue;
        if (wino_unit_small())
            return a.shift_step == 1 ? wino_ps_launch_one<1, 3, true, 2, 8>(a, s) : a.shift_step == 3 ? wino_ps_launch_one<3, 3, true, 2, 8>(a, s) : wino_ps_launch_one<9, 3, true, 2, 8>(a, s);
        return a.shift_step == 1 ? wino_ps_launch_one<1, 3, true, 4, 16>(a, s) : a.shift_step == 3 ? wino_ps_launch_one<3, 3, true, 4, 16>(a, s) : wino_ps_launch_one<9, 3, true, 4, 16>(a, s);
    }
    if (wino_tconv_selected(a)) {
        if ((a.x_cs & 3) || (a.x_us & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.ww & 15) || (a.NP & 3) || (a.sn_a && !a.sn_ib) || a.Cin > 4096)
            return hipErrorInvalidValue;
        const bool sn = a.sn_a != nullptr;
        const TconvShape t = wino_tconv_shape(a);
#define WT_GO(RT_, RW_, NTG_) if (t.rt == RT_ && t.rw == RW_ && t.ntg == NTG_) return sn ? wino_tconv_launch_one<true, RT_, RW_, NTG_>(a, s) : wino_tconv_launch_one<false, RT_, RW_, NTG_>(a, s)
        if (t.bk == 8) return sn ? wino_tconv_launch_one<true, 2, 1, 4, 8>(a, s) : wino_tconv_launch_one<false, 2, 1, 4, 8>(a, s);
        WT_GO(2, 2, 4);
        WT_GO(1, 3, 4);
        WT_GO(1, 4, 3);
        WT_GO(1, 2, 3);
        WT_GO(1, 2, 4);
#undef WT_GO
        return hipErrorInvalidValue;
    }
    // host-side shape checks: what the kernel's 16-byte accesses assume
    if ((a.x_cs & 3) || (a.x_us & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.ww & 15) || (a.NP & 3) || a.Lin < 1 ||
        (a.sn_a && !a.sn_ib) || a.Cin > 4096)
        return hipErrorInvalidValue;
    const bool sn = a.sn_a != nullptr;
    const int mode = sn && (wino_rt() == 3 || wino_rt() == 5) ? 1 : wino_rt();
    if (mode == 3) return a.shift_step == 1 ? wino_ps_launch_one<1, 2, false, 4, 16>(a, s) : a.shift_step == 3 ? wino_ps_launch_one<3, 2, false, 4, 16>(a, s) : wino_ps_launch_one<9, 2, false, 4, 16>(a, s);
    if (mode == 5 && wino_ps_rt1(a)) return a.shift_step == 1 ? wino_ps_launch_one<1, 1, false, 2, 8>(a, s) : a.shift_step == 3 ? wino_ps_launch_one<3, 1, false, 2, 8>(a, s) : wino_ps_launch_one<9, 1, false, 2, 8>(a, s);
    if (mode == 5) return a.shift_step == 1 ? wino_ps_launch_one<1, 2, false, 2, 8>(a, s) : a.shift_step == 3 ? wino_ps_launch_one<3, 2, false, 2, 8>(a, s) : wino_ps_launch_one<9, 2, false, 2, 8>(a, s);
#define WN_GO2(D_, SN_) (mode == 1 ? wino_launch_one<D_, SN_, 1, 2>(a, s) : mode == 2 ? wino_launch_one<D_, SN_, 2, 1>(a, s) : wino_launch_one<D_, SN_, 1, 1>(a, s))
#define WN_GO(D_) (sn ? WN_GO2(D_, true) : WN_GO2(D_, false))
    switch (a.shift_step) {
        case 1: return WN_GO(1);
        case 3: return WN_GO(3);
        default: return WN_GO(9);
    }
#undef WN_GO
#undef WN_GO2
}

// -------------------------------------------------------------------------------------------------------------------------
// gemm_dma_kernel<TN>: the plain 1-tap GEMMs of large launches (1x1 convs of the residual units, pointwise convs) with BOTH
// operands staged by LDS-DMA.  Same tile shape (32 TN rows x 32 NW positions, wave = all rows x 32 positions), the same MFMA
// chain per output element (16-channel blocks, k pairs ascending, k = 2 kk + h) and the same epilogue as conv_gemm_kernel, so the
// results are bit-identical to it and to gemm1_kernel; what goes away is the register-staged prefetch and its commit pass --
// VALU instructions that take matrix-pipe cycles on this chip (see wino_ps_kernel's header).  Needs an input that is already
// activated (no sn_*), Cin % 16 == 0 and M == Lin; positions past the row's end are fetched from a clamped address and only
// ever reach outputs that are not stored.
template <int TN, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : 4)) void gemm_dma_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    constexpr int NTH = 64 * NW, BM = 32 * NW, BN = 32 * TN, GK = 16;
    constexpr int XBUF = GK * BM, WBUF = GK * BN, WF4 = GK * BN / 4, WPT = (WF4 + NTH - 1) / NTH;
    static_assert(GK * BM / 4 == 2 * NTH && WPT <= 2, "pieces per thread");
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                                   // [4][BN] per-row parameters
    float *Xs = smem_all + 4 * BN;                          // [2][16][BM]
    float *Ws = Xs + 2 * XBUF;                              // [2][16][BN]

    int nwg = gridDim.x, orig = blockIdx.x;                 // XCD-aware remap, as conv_gemm_kernel
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt = id / n_tiles;
    const int unit = pt / tiles_per_unit;
    const int m0 = (pt - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int ncb = a.Cin / GK;

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // pieces (16 B per lane): X image [16][BM] -> two per thread; W image [16][BN] -> at most two
    const int xlim = ((a.Lin + 3) & ~3) - 4;
    unsigned xvo[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int i = tid + NTH * j, row = i / (BM / 4), c4 = i - row * (BM / 4);
        int pos = m0 + 4 * c4;
        pos = pos > xlim ? xlim : pos;
        xvo[j] = ((unsigned)row * (unsigned)a.x_cs + (unsigned)pos) * 4u;
    }
    unsigned wvo[WPT];
#pragma unroll
    for (int j = 0; j < WPT; j++) {
        const int i = tid + NTH * j, wrow = i / (BN / 4), wc4 = i - wrow * (BN / 4);
        wvo[j] = ((unsigned)(wrow < GK ? wrow : 0) * (unsigned)a.NP + (unsigned)wc4 * 4u) * 4u;
    }
    auto stage = [&](int cb, int buf) {
        const float *xb = xu + (long)cb * GK * a.x_cs;
        float *xd = Xs + buf * XBUF + wave * 256;
        wn_dma16_s(xb, xvo[0], xd);
        wn_dma16_s(xb, xvo[1], xd + NTH * 4);
        const float *wb = a.w + ((long)cb * GK) * a.NP + n0;
#pragma unroll
        for (int j = 0; j < WPT; j++)
            if (WF4 >= NTH * (j + 1) || tid + NTH * j < WF4) wn_dma16_s(wb, wvo[j], Ws + buf * WBUF + wave * 256 + j * (NTH * 4));
    };

    stage(0, 0);                                             // (first: the parameter table is filled while it flies)
    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int cb = 0; cb < ncb; cb++) {
        const int buf = cb & 1;
        if (cb + 1 < ncb) stage(cb + 1, buf ^ 1);
        {
            const float *ws = Ws + buf * WBUF + l31;
            const float *xs = Xs + buf * XBUF + wave * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < GK / 2; kk++) {
                const int k = 2 * kk + h;
                float av[TN];
#pragma unroll
                for (int i = 0; i < TN; i++) av[i] = ws[k * BN + i * 32];        // (merged ds_read2 forms are faster here than WN_LDS: measured)
                const float bv = xs[k * BM];
#pragma unroll
                for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    float *yu = a.y + (long)unit * a.y_us;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
    // The 1x1 conv that closes a residual unit (bias + residual, every row exists; then the raw value, SnakeBeta of it, or both side
    // by side): conv_epilogue_tile's arithmetic in its order without its per-element offsets, predicates and layout bookkeeping -- one
    // base per tile, the sixteen row offsets wave-uniform -- and with the NEXT tile's residual requested before this tile's
    // arithmetic (the residual is added in place: in the generic form every tile's loads wait behind the previous tile's stores).
    // On this chip fp32 VALU instructions take their cycles out of the matrix pipe's time (wino_ps_kernel's header), so an epilogue
    // is priced in MFMAs.  dbg bit 1 (HIPVOC_GEMMD_EPI=0, set by the launcher): the generic epilogue (same bits).
    if (ru && a.bias && a.act == ACT_NONE && !a.rscale && a.s2d <= 1 && a.N == a.NP && !(a.dbg & 1)) {       // (workgroup-uniform; the launcher sets dbg bit 1 for tensors of 4 GB and more)
        const long cs = a.y_cs;
        const bool okm = m < a.M, psn = a.psn_a != nullptr;
        const int mc = okm ? m : a.M - 1;
        // addresses: a wave-uniform base per (tile, register) -- the tensor's base + (32 i + (r&3) + 8 (r>>2)) rows -- plus this lane's
        // 32-bit BYTE offset (its lane half's four rows and its position; host: the tensor is under 4 GB), the scalar-base form of
        // the global instructions: no 64-bit address arithmetic in vector registers
        const unsigned lob = (unsigned)(((long)(n0 + 4 * h) * cs + m) * 4), lobc = (unsigned)(((long)(n0 + 4 * h) * cs + mc) * 4);
        auto at = [](const float *base, long row_off, unsigned byte_off) -> float * {
            return reinterpret_cast<float *>(reinterpret_cast<char *>(const_cast<float *>(base + row_off)) + byte_off);
        };
        float t[2][16];
        auto fetch = [&](int i, float *d) {
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = *at(ru, (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * cs, lobc);
        };
        // phase A: v = residual + (acc + bias), in the residual's registers (the accumulator tile is dead after it: the next tile's
        // residual, requested between the phases, takes its place in the register budget); phase B: SnakeBeta and the stores
        auto tile_a = [&](const f32x16 &ac, int i, float *tt) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                const float v = ac[r] + Pm[row];
                tt[r] = tt[r] + v;
            }
        };
        auto tile_b = [&](int i, const float *tt) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row = i * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                const long o0 = (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * cs, o1 = o0 + cs;
                const float v0 = tt[r], v1 = tt[r + 1];
                if (psn) {
                    const f32x2 sv = snake_beta2(mk2(v0, v1), mk2(Pm[2 * BN + row], Pm[2 * BN + row + 1]), mk2(Pm[3 * BN + row], Pm[3 * BN + row + 1]));
                    if (okm) {
                        if (y2u) { *at(yu, o0, lob) = v0; *at(y2u, o0, lob) = sv.x; *at(yu, o1, lob) = v1; *at(y2u, o1, lob) = sv.y; }
                        else { *at(yu, o0, lob) = sv.x; *at(yu, o1, lob) = sv.y; }
                    }
                } else if (okm) {
                    *at(yu, o0, lob) = v0;
                    *at(yu, o1, lob) = v1;
                }
            }
        };
        static_assert(TN <= 4, "written out for up to four tiles");
        fetch(0, t[0]);
#define GD_TILE(i_)                                                   \
    if constexpr ((i_) < TN) {                                       \
        tile_a(acc[(i_) < TN ? (i_) : 0], (i_), t[(i_) & 1]);         \
        if constexpr ((i_) + 1 < TN) fetch((i_) + 1, t[((i_) + 1) & 1]); \
        asm volatile("" ::: "memory");                               \
        tile_b((i_), t[(i_) & 1]);                                    \
    }
        GD_TILE(0) GD_TILE(1) GD_TILE(2) GD_TILE(3)
#undef GD_TILE
        return;
    }
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, false>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u);
}

// HIPVOC_GEMMD=0 turns it off (conv_gemm_kernel runs instead: same bits)
bool gemmd_selected(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_GEMMD");
    const bool off = e && atoi(e) == 0;
    const bool sel = !a.wb && !a.w2 && a.w && a.ntaps == 1 && a.S == 1 && a.shift0 == 0 && !a.sn_a && !a.xb && !a.yb && !a.in_elu && a.s2d <= 1 &&
           (a.act == ACT_NONE || a.act == ACT_GELU || a.act == ACT_CLAMP1) && a.Cin >= 16 && a.Cin % 16 == 0 && a.M == a.Lin && a.units >= 1 &&
           a.TN >= 1 && a.TN <= 4 && a.NP % (32 * a.TN) == 0 && a.NP >= a.N && a.y && (!a.y2 || (a.psn_a && a.psn_ib));
    if (sel && getenv("HIPVOC_GEMMD_TRACE"))
        fprintf(stderr, "GEMMD cand: Cin %d N %d NP %d TN %d tn %d M %d Lin %d units %d x_cs %ld x_us %ld y_cs %ld y_us %ld act %d res %d rscale %d y2 %d psn %d bias %d S %d pm %d\n",
                a.Cin, a.N, a.NP, a.TN, detail::conv_gemm_launch_tn(a), a.M, a.Lin, a.units, (long)a.x_cs, (long)a.x_us, (long)a.y_cs, (long)a.y_us, a.act,
                a.res != nullptr, a.rscale != nullptr, a.y2 != nullptr, a.psn_a != nullptr, a.bias != nullptr, a.S, a.pm);
    return sel && !off;
}

// HIPVOC_GEMMD_NW: waves per workgroup: 4 (default: 128 positions, four workgroups per CU; 2-4 % faster on the 1x1 convs) or
// 8 (256 positions, two per CU)
static int gemmd_nw() {
    const char *e = getenv("HIPVOC_GEMMD_NW");
    return (e ? atoi(e) : GEMMD_NW_DEFAULT) == 4 ? 4 : 8;
}

// Tile height of a launch: conv_gemm_launch_tn's, except on small launches (a few rounds of the chip's workgroup slots), where the
// height with the fewest rounds x rows wins -- block 2's 1x1 conv on one utterance is 1 280 workgroups of 96 rows for 1 024 slots
// (two rounds for 1.25); 1 920 of 64 rows run two rounds of two thirds the length.  Any height gives the same bits.
static int gemmd_tn(const ConvGemmArgs &a) {
    const int tn0 = detail::conv_gemm_launch_tn(a);
    const int bm = 32 * gemmd_nw();
    const long slots = gemmd_nw() == 4 ? 1024 : 512, tiles_m = (long)((a.M + bm - 1) / bm) * a.units;
    auto wgs = [&](int t) { return tiles_m * (a.NP / (32 * t)); };
    if (tn0 <= 1 || wgs(tn0) <= slots || wgs(tn0) > 4 * slots) return tn0;      // (one round, or many: as it is)
    int best = tn0;
    long best_cost = ((wgs(tn0) + slots - 1) / slots) * tn0;
    for (int t = tn0 - 1; t >= 1; t--) {
        if (a.NP % (32 * t)) continue;
        const long c = ((wgs(t) + slots - 1) / slots) * t;
        if (c < best_cost) { best = t; best_cost = c; }
    }
    return best;
}

const char *gemmd_kernel_name(const ConvGemmArgs &a) {
    static const char *n[2][5] = {{"", "gemm_dma_kernel<1,8>", "gemm_dma_kernel<2,8>", "gemm_dma_kernel<3,8>", "gemm_dma_kernel<4,8>"},
                                  {"", "gemm_dma_kernel<1,4>", "gemm_dma_kernel<2,4>", "gemm_dma_kernel<3,4>", "gemm_dma_kernel<4,4>"}};
    const int tn = gemmd_tn(a);
    return n[gemmd_nw() == 4][tn >= 1 && tn <= 4 ? tn : 4];
}

template <int TN, int NW>
static hipError_t gemmd_launch_one(const ConvGemmArgs &a, hipStream_t s) {
    constexpr int BM = 32 * NW;
    const int tiles_per_unit = (a.M + BM - 1) / BM, n_tiles = a.NP / (32 * TN);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (4 * 32 * TN + 2 * 16 * BM + 2 * 16 * 32 * TN);
    ConvGemmArgs b = a;
    {
        const char *le = getenv("HIPVOC_GEMMD_EPI");                // (read per launch) 0: the generic epilogue
        b.dbg = ((le && atoi(le) == 0) || (long)a.NP * a.y_cs * 4 >= 0xffffffffL || a.units * a.y_us > a.y_cs) ? 1 : 0;      // (32-bit byte offsets in the lean form)
    }
    hipLaunchKernelGGL((gemm_dma_kernel<TN, NW>), dim3((unsigned)grid), dim3(64 * NW), lds, s, b, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_gemmd(const ConvGemmArgs &a, hipStream_t s) {
    if (!gemmd_selected(a)) return hipErrorInvalidValue;
    if (a.M <= 0) return hipSuccess;
    // what the kernel's 16-byte DMA pieces and 32-bit byte offsets (16 rows of a channel block) assume
    if ((a.x_cs & 3) || (a.x_us & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || (a.NP & 3) || a.Lin < 1 ||
        16.0 * (double)a.x_cs * 4.0 + 4.0 * (double)a.Lin >= 4294967296.0 || 16.0 * (double)a.NP * 4.0 >= 4294967296.0)
        return hipErrorInvalidValue;
    const int tn = gemmd_tn(a);
    if (gemmd_nw() == 4) {
        switch (tn) {
            case 1: return gemmd_launch_one<1, 4>(a, s);
            case 2: return gemmd_launch_one<2, 4>(a, s);
            case 3: return gemmd_launch_one<3, 4>(a, s);
            default: return gemmd_launch_one<4, 4>(a, s);
        }
    }
    switch (tn) {
        case 1: return gemmd_launch_one<1, 8>(a, s);
        case 2: return gemmd_launch_one<2, 8>(a, s);
        case 3: return gemmd_launch_one<3, 8>(a, s);
        default: return gemmd_launch_one<4, 8>(a, s);
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// gemm_dmas_kernel<TN, NW, DP>: the same GEMM for SMALL launches (one utterance: BASELINE configs[1]).  There a layer is a few dozen
// workgroups, one wave per SIMD, and nothing on the CU hides a memory round trip: gemm1_kernel's two-stage pipeline of 64-channel
// stages spends 2.2 us per stage where its 32 MFMAs take 0.85 (o_proj, K = 1024: 36 us).  Here the 16-channel stages of
// gemm_dma_kernel are DP deep in flight (LDS-DMA needs no registers to hold them), so after the first round trip the wave's MFMA chain
// -- 64 cycles per k pair, the floor of any kernel that sums in this order -- runs without waiting, and a workgroup is NW <= 2 waves
// of 32 positions (more, shorter workgroups on the idle chip).  Same MFMA chain and epilogue: bit-identical to gemm_dma_kernel,
// gemm1_kernel and conv_gemm_kernel (test_gemm_dma_bitwise), so the choice may depend on the launch size.
template <int TN, int NW, int DP>
__global__ __launch_bounds__(64 * NW, 1) void gemm_dmas_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    constexpr int NTH = 64 * NW, BM = 32 * NW, BN = 32 * TN, GK = 16;
    constexpr int XBUF = GK * BM, WBUF = GK * BN, XPT = GK * BM / 4 / NTH, WPT = GK * BN / 4 / NTH;
    constexpr int PPS = XPT + WPT;                          // DMA pieces per thread and stage: the same for every thread
    static_assert(XPT == 2 && WPT * NTH * 4 == GK * BN && (DP & (DP - 1)) == 0 && DP >= 4 && PPS * (DP - 2) <= 63, "uniform piece count, vmcnt range");
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    float *Pm = smem_all;                                   // [4][BN] per-row parameters
    float *Xs = smem_all + 4 * BN;                          // [DP][16][BM]
    float *Ws = Xs + DP * XBUF;                             // [DP][16][BN]

    const int id = blockIdx.x;                              // (a small launch: no XCD remap -- fewer workgroups than CUs)
    const int nt = id % n_tiles;
    const int pt = id / n_tiles;
    const int unit = pt / tiles_per_unit;
    const int m0 = (pt - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const float *xu = a.x + (long)unit * a.x_us;
    const int ncb = a.Cin / GK;

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    const int xlim = ((a.Lin + 3) & ~3) - 4;
    unsigned xvo[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int i = tid + NTH * j, row = i / (BM / 4), c4 = i - row * (BM / 4);
        int pos = m0 + 4 * c4;
        pos = pos > xlim ? xlim : pos;
        xvo[j] = ((unsigned)row * (unsigned)a.x_cs + (unsigned)pos) * 4u;
    }
    unsigned wvo[WPT];
#pragma unroll
    for (int j = 0; j < WPT; j++) {
        const int i = tid + NTH * j, wrow = i / (BN / 4), wc4 = i - wrow * (BN / 4);
        wvo[j] = ((unsigned)wrow * (unsigned)a.NP + (unsigned)wc4 * 4u) * 4u;
    }
    auto stage = [&](int cb) {
        const int buf = cb & (DP - 1);
        const float *xb = xu + (long)cb * GK * a.x_cs;
        float *xd = Xs + buf * XBUF + wave * 256;
        wn_dma16_s(xb, xvo[0], xd);
        wn_dma16_s(xb, xvo[1], xd + NTH * 4);
        const float *wb = a.w + ((long)cb * GK) * a.NP + n0;
#pragma unroll
        for (int j = 0; j < WPT; j++) wn_dma16_s(wb, wvo[j], Ws + buf * WBUF + wave * 256 + j * (NTH * 4));
    };

#pragma unroll
    for (int s = 0; s < DP - 1; s++)
        if (s < ncb) stage(s);
    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    // The operands of block cb + 1 are read from LDS into the other register set BEFORE the MFMAs of block cb (one wave per SIMD:
    // nothing else hides the ds_read latency); so the wait + barrier of iteration cb are for block cb + 1.
    float av[2][GK / 2][TN], bv[2][GK / 2];
    const float *ws0 = Ws + l31 + h * BN, *xs0 = Xs + wave * 32 + l31 + h * BM;      // lane bases: pair kk is rows 2 kk + h
    auto lds_read = [&](const float *ws, const float *xs, int sl, int kk) {            // (kk-dependent part: an immediate offset)
#pragma unroll
        for (int i = 0; i < TN; i++) av[sl][kk][i] = ws[2 * kk * BN + i * 32];
        bv[sl][kk] = xs[2 * kk * BM];
    };
    auto step = [&](int cb, int sl) {
        const bool more = cb + 1 < ncb;
        if (more) {
            // issued so far: stages .. cb + DP - 2; vmcnt retires in order: all but the younger ones = stage cb + 1 has landed
            if (cb + DP - 2 < ncb) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PPS * (DP - 3)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (cb + DP - 1 < ncb) stage(cb + DP - 1);      // into the buffer of block cb - 1: every wave has read it (the barrier)
        }
        // (pinned order: MFMA kk, then the reads of the next block's pair kk -- they return under the following MFMAs, and the
        // compiler's own lgkmcnt for MFMA kk then never waits for a read issued in this iteration)
        const int nb = (more ? cb + 1 : cb) & (DP - 1);     // (last block: a harmless re-read instead of a branch per pair)
        const float *ws = ws0 + nb * WBUF, *xs = xs0 + nb * XBUF;
#pragma unroll
        for (int kk = 0; kk < GK / 2; kk++) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl][kk][i], bv[sl][kk], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            lds_read(ws, xs, sl ^ 1, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (DP - 2 < ncb) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PPS * (DP - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < GK / 2; kk++) lds_read(ws0, xs0, 0, kk);
    for (int cb = 0; cb < ncb; cb += 2) {                   // (Cin % 32 == 0: an even number of blocks)
        step(cb, 0);
        step(cb + 1, 1);
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

// Shape of the small-launch form: HIPVOC_GEMMDS_NW waves (1 or 2; default 2) x TN = 1; used when that grid is at most
// HIPVOC_GEMMDS_WGS workgroups (default 640; 0 turns the kernel off: gemm1_kernel runs those layers, same bits).
static int gemmds_nw() {
    const char *e = getenv("HIPVOC_GEMMDS_NW");
    return e && atoi(e) == 1 ? 1 : 2;
}

bool gemmds_selected(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_GEMMDS_WGS");
    const long max_wgs = e ? atol(e) : 640;
    if (max_wgs <= 0 || !gemmd_selected(a) || a.Cin > 65536 || a.Cin % 32) return false;
    const int bm = 32 * gemmds_nw();
    return (long)((a.M + bm - 1) / bm) * a.units * (a.NP / 32) <= max_wgs;
}

const char *gemmds_kernel_name(const ConvGemmArgs &) { return gemmds_nw() == 1 ? "gemm_dmas_kernel<1,1,8>" : "gemm_dmas_kernel<1,2,8>"; }

template <int TN, int NW, int DP>
static hipError_t gemmds_launch_one(const ConvGemmArgs &a, hipStream_t s) {
    constexpr int BM = 32 * NW;
    const int tiles_per_unit = (a.M + BM - 1) / BM, n_tiles = a.NP / (32 * TN);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * (4 * 32 * TN + DP * 16 * (BM + 32 * TN));
    static_assert(sizeof(float) * (4 * 32 * TN + DP * 16 * (BM + 32 * TN)) <= 64 * 1024, "static LDS limit: no attribute call needed");
    hipLaunchKernelGGL((gemm_dmas_kernel<TN, NW, DP>), dim3((unsigned)grid), dim3(64 * NW), lds, s, a, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_gemmds(const ConvGemmArgs &a, hipStream_t s) {
    if (!gemmds_selected(a)) return hipErrorInvalidValue;
    if (a.M <= 0) return hipSuccess;
    // what the kernel's 16-byte DMA pieces and 32-bit byte offsets assume
    if ((a.x_cs & 3) || (a.x_us & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || (a.NP & 31) || a.Lin < 1 ||
        16.0 * (double)a.x_cs * 4.0 + 4.0 * (double)a.Lin >= 4294967296.0 || 16.0 * (double)a.NP * 4.0 >= 4294967296.0)
        return hipErrorInvalidValue;
    return gemmds_nw() == 1 ? gemmds_launch_one<1, 1, 8>(a, s) : gemmds_launch_one<1, 2, 8>(a, s);
}

}  // namespace hipvoc
