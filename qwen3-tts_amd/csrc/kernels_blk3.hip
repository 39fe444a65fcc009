// gfx950 (CDNA4 / MI355X), bf16 conv-GEMM mode: the THREE residual units of a 96-channel decoder block as ONE kernel
// that walks along time (resblock3_bf16_kernel).
//
// A residual unit is  y = x + W2 * snake2(conv7_dil(snake1(x)) + b1) + b2  (dilations 1, 3, 9).  One unit per
// launch (conv_gemm_bf16_kernel<3,true,true>) moves 12 bytes per element through HBM -- packed operand in, fp32 residual in,
// fp32 + packed out -- for 1 536 MACs: block 3 ran at 3.6 TB/s and a quarter of the bf16 MFMA rate, bound by neither.  Every
// conv of the chain is CAUSAL, so a workgroup that processes position tiles left to right can run all three units on a tile
// before it moves on, if it keeps the last 6 / 18 / 54 activated positions of each unit's input: nothing is recomputed and
// nothing but the block's input and output touches HBM (4 + 2 bytes in, 4 bytes out per element).
//
//   * one workgroup (8 waves, 256 positions per step, wave w = positions 32w..32w+31) per (unit, span of tiles); a span that
//     does not start at position 0 first runs the tile to its left without storing anything, which leaves exactly the
//     history a walk from position 0 would have left (the last 54 positions of that tile only depend on the tile itself);
//   * LDS: ONE activated bf16 tile [12 k-groups][54 + 256 positions][8 channels] -- unit k+1's operand overwrites unit k's
//     in place, because the fp32 residual tile lives in REGISTERS (48 per lane) and the 1x1 conv's operand is the conv7
//     accumulator itself; the three history stores (78 positions); a ring of three 24 KB groups of weight slots;
//   * the weights of the whole block are one cyclic stream of 72 slots of 6 KB (per unit: the 21 (channel block, tap) tiles
//     of the conv7 in the order the per-unit kernel accumulates them, then the three output tiles of the 1x1 conv), fetched
//     by LDS-DMA (global_load_lds_dwordx4) two groups ahead: no staging registers, no commit pass, one barrier per 4 slots;
//   * per output element the MFMA chain, the SnakeBeta evaluations, the roundings and the residual add are EXACTLY those of
//     three launches of conv_gemm_bf16_kernel<3,true,true>: the result is bit-identical to that path (tested), so the plan
//     may choose either by launch size without breaking batch == single.
#include "device_common.h"
#include "kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *b3_lptr_t;
// LDS operand reads by 32-bit LDS address (generic-pointer arithmetic is 64-bit VALU work per read)
typedef const __attribute__((address_space(3))) bf16x8 *b3_lds16_t;
#ifndef B3_ABL
#define B3_ABL 0
#endif
#ifndef B3_PF
#define B3_PF 1          // k-steps the LDS operand reads run ahead of their MFMAs
#endif
#ifndef B3_SKEW
#define B3_SKEW 0        // s_sleep argument (x 64 cycles) of waves 4-7 after every group barrier
#endif
__device__ __forceinline__ bf16x8 b3_lds16(unsigned addr) {
#if B3_ABL & 32
    bf16x8 v;
    for (int j = 0; j < 8; j++) v[j] = (__bf16)(float)(addr & 7);
    return v;
#else
    return *(b3_lds16_t)(size_t)addr;
#endif
}

namespace blk3 {
constexpr int C = 96, TN = 3, KG = 12;      // channels, 32-row tiles, k-groups of 8 channels
constexpr int BM = 256, NTH = 512;          // positions per step, threads (8 waves)
constexpr int HMAX = 54;                    // the largest history (dilation 9)
constexpr int HTOT = 78;                    // 6 + 18 + 54
constexpr int XWB = HMAX + BM + 2;          // tile positions per k-group in LDS (16-byte elements)
constexpr int SLOT = 6144;                  // bytes per weight slot: [4 k-groups][96 rows][8 bf16] = [6 k-steps][64 lanes][8 bf16]
constexpr int GSL = 4;                      // slots per DMA group
constexpr int GRP = GSL * SLOT;             // 24 KB: 24 wave-pieces of 1 KB, three per wave
constexpr int UGR = 6;                      // groups per unit (21 conv7 slots + 3 slots of the 1x1 conv)
constexpr int PM_BYTES = 3 * 6 * C * 4;
constexpr int A_OFF = PM_BYTES;
constexpr int A_BYTES = KG * XWB * 16;
constexpr int H_OFF = A_OFF + A_BYTES;
constexpr int H_BYTES = KG * HTOT * 16;
constexpr int R_OFF = H_OFF + H_BYTES;
constexpr int LDS_BYTES = R_OFF + 3 * GRP;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert((UGR * 3) % 3 == 0, "ring position of a unit's group g is g % 3");
}  // namespace blk3

// 16 bytes per lane from global memory straight into LDS: wave-uniform source base (SGPR pair) + per-lane byte offset,
// wave-uniform LDS base in M0 (+ 16 * lane).  Inline asm for the reason given at wn_dma16 (kernels_wino.hip): the builtin is
// booked as a flat access and turns every later LDS wait into lgkmcnt(0).
__device__ __forceinline__ void b3_dma16(const void *sbase, unsigned voff_bytes, unsigned lds_addr) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)), b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned long long bs = ((unsigned long long)b_hi << 32) | (unsigned long long)b_lo;
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff_bytes), "s"(bs) : "memory");
}

// SnakeBeta as this kernel evaluates it.  B3_ABL (timing-only experiments, tools/build_variant.sh; results are wrong with any
// bit set): 1 = identity instead of SnakeBeta, 2 = the v_sin_f32 form, 4 = no residual loads / output stores, 8 = no weight
// DMA after the first two groups, 16 = no MFMAs, 32 = no LDS operand reads, 64 = no group barriers.
#ifndef B3_ABL
#define B3_ABL 0
#endif
__device__ __forceinline__ f32x2 b3_snake2(f32x2 x, f32x2 ea, f32x2 ib) {
#if B3_ABL & 1
    return x + ib * ea * 1e-9f;
#elif B3_ABL & 2
    const f32x2 rev = (x * ea) * 0.15915494309189535f;
    const float s0 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.x)), s1 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.y));
    return fma2(ib, mk2(s0 * s0, s1 * s1), x);
#else
    return snake_beta2_t<true>(x, ea, ib);      // (the bf16 mode's form: device_common.h)
#endif
}

// Top of a group: wait until at most N of this wave's vector-memory operations are outstanding (they complete in issue
// order), then for its LDS writes, then for every wave.  N is the number of operations issued AFTER the DMA pieces of the
// group about to be read (see the table in the kernel): those stay in flight across the barrier.
template <int N>
__device__ __forceinline__ void b3_wait_bar() {
#if B3_ABL & 64
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ f32x16 b3_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
#if B3_ABL & 16
    c[0] += (float)a[0] + (float)b[0];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
// XRB: the block's input stream a.x is the packed bf16 tensor its trans-conv wrote (ResBlock3Args::xrb; layout of xb): twelve 8-byte
// loads per step into the LOWER HALF of each tile's residual registers, unpacked in place where unit 0's epilogue first needs them
template <bool XRB>
__global__ __launch_bounds__(blk3::NTH, 2) void resblock3_bf16_kernel(ResBlock3Args a) {
    using namespace blk3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);                        // [3 units][6][96]: b1 | sn2_a | sn2_ib | b2 | psn_a | psn_ib
    uint4 *Ab = reinterpret_cast<uint4 *>(smem + A_OFF);                // [12][XWB] activated tile, index 54 = the step's first position
    uint4 *Hs = reinterpret_cast<uint4 *>(smem + H_OFF);                // history stores, unit u at [12][6 d_u] from element hoff[u]
    const unsigned ring_lds = (unsigned)(size_t)(b3_lptr_t)(smem + R_OFF);
    const unsigned ab_lds = (unsigned)(size_t)(b3_lptr_t)(smem + A_OFF);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int unit = blockIdx.x / a.spans_per_unit, span = blockIdx.x - unit * a.spans_per_unit;
    const int total_tiles = (a.L + BM - 1) / BM;
    const int tile0 = span * a.span_tiles;
    int ntiles = total_tiles - tile0;
    ntiles = ntiles < a.span_tiles ? ntiles : a.span_tiles;
    if (ntiles <= 0) return;                                            // (workgroup-uniform)
    const int first = tile0 > 0 ? -1 : 0;                               // a warm-up step on the tile to the left
    const int nsteps = ntiles - first;
    const int ngroups = nsteps * 3 * UGR;

    const int d0 = a.dil[0], d1 = a.dil[1], d2 = a.dil[2];
    const int hoff1 = KG * 6 * d0, hoff2 = hoff1 + KG * 6 * d1;

    // ---- prologue: zero the histories and the tile's history region, parameter table, the first two weight groups, first tile
    for (int e = tid; e < KG * HTOT; e += NTH) Hs[e] = make_uint4(0, 0, 0, 0);
    for (int e = tid; e < KG * HMAX; e += NTH) Ab[(e / HMAX) * XWB + (e % HMAX)] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 3 * C; i += NTH) {
        const int u = i / C, c = i - u * C;
        float *p = Pm + u * 6 * C + c;
        p[0 * C] = a.bias1[u][c];
        p[1 * C] = a.sn2_a[u][c];
        p[2 * C] = a.sn2_ib[u][c];
        p[3 * C] = a.bias2[u][c];
        p[4 * C] = a.psn_a[u] ? a.psn_a[u][c] : 0.0f;
        p[5 * C] = a.psn_ib[u] ? a.psn_ib[u][c] : 0.0f;
    }
    const unsigned dma_voff = (unsigned)lane * 16u;
    auto dma_group = [&](int gi, int rbuf) {                            // group gi of the walk -> ring buffer rbuf
        if ((B3_ABL & 8) && gi > 1) return;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.ws) + (size_t)(gi % (3 * UGR)) * GRP;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = wave * 3 + j;
            b3_dma16(src + k * 1024, dma_voff, ring_lds + (unsigned)(rbuf * GRP + k * 1024));
        }
    };
    dma_group(0, 0);
    if (ngroups > 1) dma_group(1, 1);

    // the packed operand tile of a step: element (k-group 2j + tid/256, position tid % 256) per thread and j
    const uint4 *xbu = reinterpret_cast<const uint4 *>(a.xb) + (long)unit * a.us;
    const int xkg = tid >> 8, xpo = tid & 255;
    uint4 xn[6];
    auto x_fetch = [&](int t0) {
        int pos = t0 + xpo;
        pos = pos < a.L ? pos : a.L - 1;
#pragma unroll
        for (int j = 0; j < 6; j++) xn[j] = xbu[(long)(2 * j + xkg) * a.cs + pos];
    };
    auto x_commit = [&](int t0) {
        const bool ok = t0 + xpo < a.L;
#pragma unroll
        for (int j = 0; j < 6; j++) Ab[(2 * j + xkg) * XWB + HMAX + xpo] = ok ? xn[j] : make_uint4(0, 0, 0, 0);
    };
    x_fetch((tile0 + first) * BM);
    x_commit((tile0 + first) * BM);

    const float *xu = a.x + (long)unit * a.us;
    float *yu = a.y ? a.y + (long)unit * a.us : nullptr;
    // descriptor of the 32 rows of row tile i of a [C][G][Lp] fp32 tensor (base wave-uniform; 32 rows span < 2 GB: host check)
    const unsigned row_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)a.cs * 4u));
    auto row_tile_rsrc = [&](const float *base, int i) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long ba = reinterpret_cast<unsigned long long>(base + (long)i * 32 * a.cs);
        const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba), bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)bhi << 32) | blo), 0, 0x7fffffff, 0x00020000);
    };
    unsigned short *ybu = a.yb ? a.yb + (((long)unit * a.us) << 3) : nullptr;
    // XRB: descriptor of the four channel groups of row tile i of the packed stream (16-byte elements (c/8, unit, pos))
    const unsigned grp_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)a.cs * 16u));
    auto packed_tile_rsrc = [&](int i) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long ba = reinterpret_cast<unsigned long long>(a.x) + 16ull * (unsigned long long)((long)unit * a.us + (long)(4 * i) * a.cs);
        const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba), bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)bhi << 32) | blo), 0, 0x7fffffff, 0x00020000);
    };

    f32x16 y[TN];                                                       // the fp32 residual tile, rows (r&3) + 8(r>>2) + 4h of tile i
    int gi = 0;
    int prev_stores = 0;                                                // stores this wave issued at the end of the previous step
    for (int step = first; step < ntiles; step++) {
        const int t0 = (tile0 + step) * BM;
        const int m = t0 + wave * 32 + l31;
        const bool stored = step >= 0;
        const bool has_next = step + 1 < ntiles;
        // (wave-uniform) this wave stores its part of the step: then all its 48 store instructions are issued, under a lane mask
        const bool wave_stores = stored && t0 + wave * 32 < a.L;
        for (int u = 0; u < 3; u++) {
            const int d = u == 0 ? d0 : (u == 1 ? d1 : d2);
            const float *P1 = Pm + u * 6 * C;
            f32x16 acc[TN];
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
            bf16x8 hb[TN][2];
            bf16x8 w2q[2 * TN];                                         // the 1x1 conv's A operands of one output tile
            // LDS byte addresses of this lane's operands: tap t of its position is + 16 t d; k-group rows and slots are immediates
            const unsigned xs_addr = ab_lds + (unsigned)((h * XWB + HMAX + wave * 32 + l31 - 6 * d) * 16);
            const unsigned d16 = (unsigned)d * 16u;
            static_assert(UGR == 6, "six groups per unit");
#define B3_G 0
#include "kernels_blk3_group.inc"
#undef B3_G
#define B3_G 1
#include "kernels_blk3_group.inc"
#undef B3_G
#define B3_G 2
#include "kernels_blk3_group.inc"
#undef B3_G
#define B3_G 3
#include "kernels_blk3_group.inc"
#undef B3_G
#define B3_G 4
#include "kernels_blk3_group.inc"
#undef B3_G
#define B3_G 5
#include "kernels_blk3_group.inc"
#undef B3_G
        }
    }
}

// ---------------------------------------------------------------------------------------------------
bool resblock3_bf16_supported(int C, const int *dil) { return C == blk3::C && dil[0] == 1 && dil[1] == 3 && dil[2] == 9; }
size_t resblock3_bf16_stream_bytes() { return (size_t)3 * blk3::UGR * blk3::GRP; }
size_t resblock3_bf16_slot_bytes() { return blk3::SLOT; }

// spans per unit: fewest rounds of the chip's workgroup slots (one workgroup per CU: 152 KB of LDS) x steps per workgroup
static void blk3_spans(int units, int L, int n_cu, int &span_tiles, int &spans_per_unit) {
    const int tiles = (L + blk3::BM - 1) / blk3::BM;
    long best_cost = -1;
    int best_s = 1;
    for (int s = 1; s <= tiles && s <= 512; s++) {
        const int st = (tiles + s - 1) / s, sp = (tiles + st - 1) / st;
        if (sp != s) continue;
        const long rounds = ((long)units * s + n_cu - 1) / n_cu;
        const long cost = rounds * (st + (s > 1 ? 1 : 0));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_s = s; }
    }
    spans_per_unit = best_s;
    span_tiles = (tiles + best_s - 1) / best_s;
}

hipError_t launch_resblock3_bf16(const ResBlock3Args &a_in, hipStream_t s) {
    ResBlock3Args a = a_in;
    if (a.G <= 0 || a.L <= 0) return hipSuccess;
    // host-side shape checks: what the kernel's indexing assumes
    if (!a.xb || !a.x || !a.ws || (!a.y && !a.yb) || ((uintptr_t)a.xb & 15) || ((uintptr_t)a.ws & 15) || (a.yb && ((uintptr_t)a.yb & 15)) ||
        !resblock3_bf16_supported(blk3::C, a.dil) || a.us < a.L || a.cs < (long)a.G * a.us || (a.cs * (long)blk3::C) >= 0x7fffffffL * 4L ||
        (a.xrb && (((uintptr_t)a.x & 15) || a.cs * 16L * 4L >= 0x7fffffffL)))
        return hipErrorInvalidValue;
    for (int u = 0; u < 3; u++)
        if (!a.bias1[u] || !a.sn2_a[u] || !a.sn2_ib[u] || !a.bias2[u] || (u < 2 && (!a.psn_a[u] || !a.psn_ib[u])) || (!a.psn_a[u] != !a.psn_ib[u]))
            return hipErrorInvalidValue;
    if (a.yb && !a.psn_a[2]) return hipErrorInvalidValue;             // (a packed output carries the consumer's activation)
    static std::atomic<int> n_cu[64];
    static std::atomic<bool> done[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(resblock3_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(resblock3_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int cu = 0;
        e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        n_cu[dev] = cu > 0 ? cu : 256;
        done[dev] = true;
    }
    if (const char *f = getenv("HIPVOC_BLK3_SPANS")) {                  // (experiments: force the spans per unit)
        const int tiles = (a.L + blk3::BM - 1) / blk3::BM;
        int sp = atoi(f);
        sp = sp < 1 ? 1 : (sp > tiles ? tiles : sp);
        a.span_tiles = (tiles + sp - 1) / sp;
        a.spans_per_unit = (tiles + a.span_tiles - 1) / a.span_tiles;
    } else {
        blk3_spans(a.G, a.L, n_cu[dev], a.span_tiles, a.spans_per_unit);
    }
    const long grid = (long)a.G * a.spans_per_unit;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    if (a.xrb) hipLaunchKernelGGL(resblock3_bf16_kernel<true>, dim3((unsigned)grid), dim3(blk3::NTH), blk3::LDS_BYTES, s, a);
    else hipLaunchKernelGGL(resblock3_bf16_kernel<false>, dim3((unsigned)grid), dim3(blk3::NTH), blk3::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace hipvoc
