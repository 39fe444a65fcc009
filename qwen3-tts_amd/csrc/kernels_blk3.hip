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

__global__ __launch_bounds__(blk3::NTH, 2) void resblock3_bf16_kernel(ResBlock3Args a) {
    using namespace blk3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);                        // [3 units][6][96]: b1 | sn2_a | sn2_ib | b2 | psn_a | psn_ib
    uint4 *Ab = reinterpret_cast<uint4 *>(smem + A_OFF);                // [12][XWB] activated tile, index 54 = the step's first position
    uint4 *Hs = reinterpret_cast<uint4 *>(smem + H_OFF);                // history stores, unit u at [12][6 d_u] from element hoff[u]
    const unsigned ring_lds = (unsigned)(size_t)(b3_lptr_t)(smem + R_OFF);
    const unsigned char *Ring = smem + R_OFF;

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

    f32x16 y[TN];                                                       // the fp32 residual tile, rows (r&3) + 8(r>>2) + 4h of tile i
    int gi = 0;
    for (int step = first; step < ntiles; step++) {
        const int t0 = (tile0 + step) * BM;
        const int m = t0 + wave * 32 + l31;
        const bool stored = step >= 0;
        const bool has_next = step + 1 < ntiles;
        for (int u = 0; u < 3; u++) {
            const int d = u == 0 ? d0 : (u == 1 ? d1 : d2);
            const float *P1 = Pm + u * 6 * C;
            f32x16 acc[TN];
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
            bf16x8 hb[TN][2];
            const uint4 *xsb = Ab + HMAX + wave * 32 + l31 - 6 * d;    // tap t of this lane's position: + t d
#pragma unroll
            for (int g = 0; g < UGR; g++, gi++) {
                // every wave's pieces of the groups in flight have landed, its LDS writes are done; then: all of them
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (gi + 2 < ngroups) dma_group(gi + 2, (g + 2) % 3);
                if (g == 0 && u == 0) {                                 // the raw residual tile: 48 rows of this lane's position
                    // buffer loads: one lane offset + the row's offset in an SGPR (48 plain loads keep 96 address registers live)
                    const int mc = m < a.L ? m : a.L - 1;
                    const unsigned voff = (unsigned)(4 * h * a.cs + mc) * 4u;
#pragma unroll
                    for (int i = 0; i < TN; i++) {
                        __amdgpu_buffer_rsrc_t rs = row_tile_rsrc(xu, i);
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            y[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes, 0));
                    }
                }
                if (g == 0 && u == 2 && has_next) x_fetch(t0 + BM);
                const unsigned char *rb = Ring + (g % 3) * GRP;
#pragma unroll
                for (int s = 0; s < GSL; s++) {
                    const int st = g * GSL + s;
                    if (st < 21) {                                      // conv7 slot: channel block cb, tap
                        const int cb = st / 7, tap = st - cb * 7;
                        const bf16x8 *wsb = reinterpret_cast<const bf16x8 *>(rb + s * SLOT) + l31;
                        const bf16x8 *xt = reinterpret_cast<const bf16x8 *>(xsb + tap * d);
#pragma unroll
                        for (int ks = 0; ks < 2; ks++) {
                            const int kq = ks * 2 + h;
                            const bf16x8 bv = xt[(cb * 4 + kq) * XWB];
                            bf16x8 av[TN];
#pragma unroll
                            for (int i = 0; i < TN; i++) av[i] = wsb[kq * C + i * 32];
#pragma unroll
                            for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv, acc[i], 0, 0, 0);
                        }
                        if (st == 20) {
                            // H = snake2(acc + b1) -> the 1x1 conv's bf16 B operands (registers 8g..8g+7 of tile i = its eight k)
#pragma unroll
                            for (int i = 0; i < TN; i++) {
#pragma unroll
                                for (int r = 0; r < 16; r++) {
                                    const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                    if (r & 1) {
                                        f32x2 hv = snake_beta2(mk2(acc[i][r - 1] + P1[n - 1], acc[i][r] + P1[n]), mk2(P1[C + n - 1], P1[C + n]),
                                                               mk2(P1[2 * C + n - 1], P1[2 * C + n]));
                                        acc[i][r - 1] = hv.x;
                                        acc[i][r] = hv.y;
                                    }
                                    if ((r & 3) == 3) {
                                        asm volatile("" : "+v"(acc[i][r - 3]), "+v"(acc[i][r - 2]), "+v"(acc[i][r - 1]), "+v"(acc[i][r]));
                                        __builtin_amdgcn_sched_barrier(0);
                                    }
                                }
#pragma unroll
                                for (int gg = 0; gg < 2; gg++)
#pragma unroll
                                    for (int j = 0; j < 8; j++) hb[i][gg][j] = (__bf16)acc[i][8 * gg + j];
                            }
                            // every wave has read this unit's operand tile for the last time
                            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            // history: the last 6d positions of this unit's operand, kept by the waves that own them ...
                            const int hu = 6 * d, hbase = u == 0 ? 0 : (u == 1 ? hoff1 : hoff2);
                            if (wave * 32 + 32 > BM - hu) {             // (wave-uniform)
                                const int pj = wave * 32 + l31 - (BM - hu);
                                if (pj >= 0) {
#pragma unroll
                                    for (int k2 = 0; k2 < KG / 2; k2++) {
                                        const int kg = 2 * k2 + h;
                                        Hs[hbase + kg * hu + pj] = Ab[kg * XWB + HMAX + wave * 32 + l31];
                                    }
                                }
                            }
                            // ... and the next unit's history (kept during the previous step) in front of the tile
                            {
                                const int un = u == 2 ? 0 : u + 1;
                                const int dn = un == 0 ? d0 : (un == 1 ? d1 : d2), hn = 6 * dn, hnb = un == 0 ? 0 : (un == 1 ? hoff1 : hoff2);
                                for (int e = tid; e < KG * hn; e += NTH) {
                                    const int kg = e / hn, j = e - kg * hn;
                                    Ab[kg * XWB + HMAX - hn + j] = Hs[hnb + e];
                                }
                            }
                        }
                    } else {                                            // 1x1 conv, output tile o: y += W2 h + b2; then what follows the unit
                        const int o = st - 21;
                        f32x16 acc2;
#pragma unroll
                        for (int r = 0; r < 16; r++) acc2[r] = 0.0f;
                        const bf16x8 *ws2 = reinterpret_cast<const bf16x8 *>(rb + s * SLOT) + h * 32 + l31;
#pragma unroll
                        for (int i = 0; i < TN; i++)
#pragma unroll
                            for (int gg = 0; gg < 2; gg++) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws2[(i * 2 + gg) * 64], hb[i][gg], acc2, 0, 0, 0);
#pragma unroll
                        for (int half = 0; half < 2; half++) {
                            int n[8];
                            float v[8], w[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const int r = half * 8 + q;
                                n[q] = o * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                            }
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                v[q] = y[o][half * 8 + q] + (acc2[half * 8 + q] + P1[3 * C + n[q]]);
                                y[o][half * 8 + q] = v[q];
                                w[q] = v[q];
                            }
                            const bool act = u < 2 || a.psn_a[2] != nullptr;
                            if (act) {
#pragma unroll
                                for (int q = 0; q < 8; q += 2) {
                                    f32x2 sv = snake_beta2(mk2(v[q], v[q + 1]), mk2(P1[4 * C + n[q]], P1[4 * C + n[q + 1]]),
                                                           mk2(P1[5 * C + n[q]], P1[5 * C + n[q + 1]]));
                                    w[q] = sv.x; w[q + 1] = sv.y;
                                }
                            }
                            if (u < 2) {                                // the next unit's operand, in place
#pragma unroll
                                for (int g2 = 0; g2 < 2; g2++) {
                                    const int kg = n[4 * g2] >> 3;
                                    uint2 *dst = reinterpret_cast<uint2 *>(Ab + kg * XWB + HMAX + wave * 32 + l31) + h;
                                    *dst = pack4_bf16(w[4 * g2], w[4 * g2 + 1], w[4 * g2 + 2], w[4 * g2 + 3]);
                                }
                            } else if (stored && m < a.L) {             // the block's output
                                const unsigned voff = (unsigned)(4 * h * a.cs + m) * 4u;
                                if (yu) {
                                    __amdgpu_buffer_rsrc_t rs = row_tile_rsrc(yu, o);
#pragma unroll
                                    for (int q = 0; q < 8; q++) {
                                        const int r = half * 8 + q;
                                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ybu ? v[q] : w[q]), rs, voff,
                                                                              (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes, 0);
                                    }
                                }
                                if (ybu) {
#pragma unroll
                                    for (int g2 = 0; g2 < 2; g2++)
                                        *reinterpret_cast<uint2 *>(ybu + ((((long)(n[4 * g2] >> 3) * a.cs + m) << 3) + (n[4 * g2] & 4))) =
                                            pack4_bf16(w[4 * g2], w[4 * g2 + 1], w[4 * g2 + 2], w[4 * g2 + 3]);
                                }
                            }
                        }
                        if (st == 23 && u == 2 && has_next) {           // the next step's operand tile, over everybody's positions:
                            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... once the owners have kept their history
                            x_commit(t0 + BM);
                        }
                    }
                }
            }
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
        !resblock3_bf16_supported(blk3::C, a.dil) || a.us < a.L || a.cs < (long)a.G * a.us || (a.cs * (long)blk3::C) >= 0x7fffffffL * 4L)
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
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(resblock3_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    hipLaunchKernelGGL(resblock3_bf16_kernel, dim3((unsigned)grid), dim3(blk3::NTH), blk3::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace hipvoc
