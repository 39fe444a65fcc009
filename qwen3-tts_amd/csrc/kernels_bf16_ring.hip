// gfx950 (CDNA4 / MI355X), bf16 conv-GEMM mode: the causal k = 7 convs of the decoder blocks whose operand is a packed bf16
// tensor (ConvGemmArgs::xb), with BOTH operands staged by LDS-DMA through a ring (conv7_bf16_ring_kernel).
//
// conv_gemm_bf16_kernel stages every (channel block, tap) weight tile through registers ONE stage ahead, with a barrier per
// stage of 8 MFMAs per wave: the tile's L2 round trip (~1 us) does not fit into a stage (~0.4 us with four waves per SIMD), so
// every stage waits, and a wave reads five 16-byte LDS operands for four MFMAs (profiles/r03_pmc_sq_bf16_before.txt: matrix
// pipe busy 55-60 % of the cycles on these layers).  Here:
//   * the packed activation element [c/8][unit][t][8 bf16] IS the LDS tile element, and the weights lie in HBM in LDS tile
//     order, so both arrive by global_load_lds_dwordx4: no staging registers, no commit pass, and a weight tile is in flight
//     THREE stages ahead (ring of four 8 KB slots), the activation tile of the next channel block seven stages ahead
//     (double buffer);
//   * a wave owns 128 rows x 64 positions (4 x 2 accumulator tiles): six LDS operand reads feed eight MFMAs;
//   * 4-wave workgroups of 256 positions, 75 KB of LDS: two INDEPENDENT workgroups per CU, so one's barrier and epilogue
//     overlap the other's MFMAs (the move that paid off for the fp32 kernels);
//   * s_waitcnt vmcnt(N) at a stage's barrier names how many YOUNGER operations may stay in flight (they complete in issue
//     order): the weight tiles of the next two stages, plus the next activation tile during the three stages after its issue.
// Per output element the chain of MFMAs is conv_gemm_bf16_kernel's (channel blocks of 32 ascending, taps ascending, k-groups
// ascending with lane half h supplying group 2 ks + h) and the epilogue is the same function: results are BIT-IDENTICAL
// (tests/test_gpu_bf16.py), so the choice may depend on the launch size without breaking batch == single.
#include "device_common.h"
#include "kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

typedef __bf16 rg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *rg_lptr_t;
typedef const __attribute__((address_space(3))) rg_bf16x8 *rg_lds16_t;

#ifndef RG_ABL
#define RG_ABL 0          // timing-only experiments (results are wrong with any bit set): 1 = no weight DMA after the prologue,
#endif                    // 2 = no activation DMA after the prologue, 4 = no barriers, 8 = one store per lane instead of the epilogue, 16 = no LDS operand reads

namespace ring {
constexpr int NW = 4, NTH = 64 * NW;        // waves per workgroup, threads
constexpr int PT = 2;                       // 32-position tiles per wave
constexpr int BM = 32 * PT * NW;            // positions per workgroup
constexpr int XW = 320;                     // positions per k-group row of the activation tile: BM + the left halo (<= 64), whole pieces
constexpr int XBUF = 4 * XW * 16;           // bytes per activation tile [4 k-groups][XW][8 bf16]
constexpr int XP = 4 * XW / 64 / NW;        // 1 KB pieces per wave and activation tile
constexpr int RW = 4;                       // weight slots in the ring
constexpr int NTAPS = 7;
static_assert(4 * XW % (64 * NW) == 0, "every wave issues the same number of activation pieces");
}  // namespace ring

__device__ __forceinline__ rg_bf16x8 rg_lds16(unsigned addr) {
#if RG_ABL & 16
    rg_bf16x8 v;
    for (int j = 0; j < 8; j++) v[j] = (__bf16)(float)(addr & 7);
    return v;
#else
    return *(rg_lds16_t)(size_t)addr;
#endif
}

// 16 bytes per lane from global memory straight into LDS (see b3_dma16 / wn_dma16: inline asm, because the builtin is booked as
// a flat access and turns every later LDS wait into lgkmcnt(0)).  sbase and lds_addr are wave-uniform.
__device__ __forceinline__ void rg_dma16(unsigned long long sbase, unsigned voff_bytes, unsigned lds_addr) {
    const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sbase >> 32)), b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sbase);
    const unsigned long long bs = ((unsigned long long)b_hi << 32) | (unsigned long long)b_lo;
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff_bytes), "s"(bs) : "memory");
}

template <int N>
__device__ __forceinline__ void rg_wait_bar() {
#if RG_ABL & 4
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <int TN>
__global__ __launch_bounds__(ring::NTH, 2) void conv7_bf16_ring_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    using namespace ring;
    constexpr int BN = 32 * TN;
    constexpr int SLOTB = 4 * BN * 16;              // bytes per weight slot [4 k-groups][BN rows][8 bf16]
    constexpr int WP = SLOTB / 1024 / NW;           // 1 KB pieces per wave and slot
    constexpr int PPG = BN / 64;                    // pieces per k-group row
    static_assert(BN % 64 == 0 && SLOTB % (1024 * NW) == 0, "whole pieces per wave");
    constexpr int PMB = 4 * BN * 4;                 // per-row parameter table (device_common.h), first in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);
    uint4 *Xs = reinterpret_cast<uint4 *>(smem + PMB);
    const unsigned x_lds = (unsigned)(size_t)(rg_lptr_t)(smem + PMB);
    const unsigned w_lds = x_lds + 2 * XBUF;

    // XCD-aware bijective remap (as conv_gemm_bf16_kernel): the row tiles that share an activation tile run on one L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt_ = id / n_tiles;
    const int unit = pt_ / tiles_per_unit;
    const int m0 = (pt_ - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int d = a.shift_step, HL = -a.shift0;     // (host: HL == 6 d, BM + HL <= XW)
    const int ncb = a.Cin / 32;                     // (host: Cin % 32 == 0)

    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    // The causal left padding: the first tile of a unit keeps zeros at positions < 0 of both buffers -- the DMA's lanes for
    // those positions are masked off, so nothing overwrites them.  (Visible to every wave after the first stage's barrier.)
    if (m0 == 0) {
        for (int e = tid; e < 2 * 4 * HL; e += NTH) {
            const int row = e / HL, t = e - row * HL;
            Xs[row * XW + t] = make_uint4(0, 0, 0, 0);
        }
    }

    // ---- activation pieces of this wave: piece wave + NW j covers the tile's elements 64 (wave + NW j) + lane
    unsigned xvoff[XP];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = (wave + NW * j) * 64 + lane;
        const int kg = e / XW, t = e - kg * XW;
        int pos = m0 - HL + t;
        if (pos >= 0) xmask |= 1u << j;
        pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
        xvoff[j] = (unsigned)(((long)kg * a.x_cs + pos) * 16);
    }
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us) << 4);
    const unsigned long long xcb_bytes = (unsigned long long)a.x_cs * 64ull;          // 4 k-groups
    auto dma_x = [&](int cb, int buf) {
        if ((RG_ABL & 2) && cb > 0) return;
        const unsigned long long base = xbase + (unsigned long long)cb * xcb_bytes;
#pragma unroll
        for (int j = 0; j < XP; j++)
            if ((xmask >> j) & 1) rg_dma16(base, xvoff[j], x_lds + (unsigned)(buf * XBUF + (wave + NW * j) * 1024));
    };
    // ---- weight pieces: slot (cb, tap) is [4 k-groups][BN rows] of the packed tensor [tap][cb][4][NP]; piece p = k-group p / PPG,
    // rows 64 (p % PPG) ..
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wb) + ((unsigned long long)n0 << 4);
    const unsigned long long wkg_bytes = (unsigned long long)a.NP * 16ull;
    const unsigned wlane = (unsigned)lane * 16u;
    auto dma_w = [&](int cb, int tap, int slot) {
        if ((RG_ABL & 1) && (cb > 0 || tap > 2)) return;
        const unsigned long long base = wbase + (unsigned long long)((tap * ncb + cb) * 4) * wkg_bytes;
#pragma unroll
        for (int q = 0; q < WP; q++) {
            const int p = wave * WP + q, kg = p / PPG, part = p - kg * PPG;
            rg_dma16(base + (unsigned long long)kg * wkg_bytes + (unsigned long long)(part * 1024), wlane, w_lds + (unsigned)(slot * SLOTB + p * 1024));
        }
    };

    f32x16 acc[TN][PT];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int p = 0; p < PT; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][p][r] = 0.0f;

    dma_x(0, 0);
    dma_w(0, 0, 0);
    dma_w(0, 1, 1);
    dma_w(0, 2, 2);

    // this lane's operand addresses: B = element (k-group 2 ks + h, position 64 wave + 32 p + l31 + tap d) of the activation tile
    // (tile position 0 = m0 - HL, tap 0 reads position m - 6 d = tile position m - m0), A = (k-group 2 ks + h, row 32 i + l31)
    const unsigned xaddr0 = x_lds + (unsigned)((h * XW + wave * 64 + l31) * 16);
    const unsigned waddr0 = w_lds + (unsigned)((h * BN + l31) * 16);
    const unsigned d16 = (unsigned)d * 16u;

    rg_bf16x8 bq[2][PT];
    for (int cb = 0; cb < ncb; cb++) {
        const bool lastcb = cb + 1 == ncb;
#pragma unroll
        for (int tap = 0; tap < NTAPS; tap++) {
            const int st = cb * NTAPS + tap, slot = st & (RW - 1);
            // this wave's pieces of slot st (and, at tap 0, of this block's activation tile: issued earlier still) have landed;
            // younger: the slots of the next two stages, and during taps 1-3 the next block's activation tile (issued at tap 0)
            if (lastcb && tap == NTAPS - 1) rg_wait_bar<0>();
            else if (lastcb && tap == NTAPS - 2) rg_wait_bar<WP>();
            else if (!lastcb && tap >= 1 && tap <= 3) rg_wait_bar<2 * WP + XP>();
            else rg_wait_bar<2 * WP>();
            // every wave is past stage st - 1: its slot, and at tap 0 the other activation buffer, are free
            {
                const int tap3 = (tap + 3) % NTAPS, cb3 = cb + ((tap + 3) >= NTAPS ? 1 : 0);
                if (cb3 < ncb) dma_w(cb3, tap3, (st + 3) & (RW - 1));
            }
            if (tap == 0 && !lastcb) dma_x(cb + 1, (cb + 1) & 1);
            const unsigned xa = xaddr0 + (unsigned)((cb & 1) * XBUF) + (unsigned)tap * d16;
            const unsigned wa = waddr0 + (unsigned)(slot * SLOTB);
            // Operand reads, pinned ahead of the MFMAs (left alone, hipcc emits ds_read -> s_waitcnt -> two v_mfma in series through
            // sixteen operand registers).  The A operands of this stage first; then the B operands of the NEXT tap, whose activation
            // tile is this block's and already complete -- only at tap 0 does a stage read its own B operands behind the barrier.
            rg_bf16x8 aq[2][TN];
            if (tap == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bq[ks][p] = rg_lds16(xa + (unsigned)(ks * 2 * XW * 16 + p * 512));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TN; i++) aq[ks][i] = rg_lds16(wa + (unsigned)(ks * 2 * BN * 16 + i * 512));
            rg_bf16x8 bn[2][PT];
            if (tap + 1 < NTAPS) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bn[ks][p] = rg_lds16(xa + d16 + (unsigned)(ks * 2 * XW * 16 + p * 512));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TN; i++)
#pragma unroll
                    for (int p = 0; p < PT; p++) acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][i], bq[ks][p], acc[i][p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < NTAPS) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bq[ks][p] = bn[ks][p];
            }
        }
    }

    // epilogue (conv_gemm_bf16_kernel's): bias, residual, optional SnakeBeta, fp32 and / or packed stores
    if ((RG_ABL & 8) && a.yb) {                     // timing only: one 8-byte store per lane instead
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int p = 0; p < PT; p++)
#pragma unroll
                for (int r = 0; r < 16; r++) t += acc[i][p][r];
        const int m = m0 + wave * 64 + l31;
        if (m < a.M) *reinterpret_cast<uint2 *>(a.yb + ((((long)((n0 + 4 * h) >> 3) * a.y_cs + (long)unit * a.y_us + m) << 3) + ((n0 + 4 * h) & 4))) = pack4_bf16(t, t, t, t);
        return;
    }
    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[TN - 1][PT - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][PT - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    unsigned short *ybu0 = launder_after(a.yb, acc[TN - 1][PT - 1][15]);
    unsigned short *ybu = ybu0 ? ybu0 + (((long)unit * a.y_us) << 3) : nullptr;
    static_assert(PT == 2 && TN <= 6, "the epilogue below is written out for these");
    // The case every conv7 of the packed plans is (a residual unit's first conv: bias, SnakeBeta, packed bf16 out, nothing else; every
    // row exists): conv_epilogue_tile's arithmetic in its order, without its per-element offsets and predicates -- one base per
    // channel group.  HIPVOC_BF16_RING_EPI=0 (dbg bit 1, set by the launcher): the generic epilogue (same bits).
    if (ybu && !yu && !ru && !y2u && a.bias && a.act == ACT_NONE && !a.rscale && a.N == a.NP && !(a.dbg & 1)) {      // (workgroup-uniform)
        const bool psn = a.psn_a != nullptr;
        auto lean = [&](const f32x16 &ac, int i, int m) {
            if (m >= a.M) return;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int row = i * 32 + 8 * g + 4 * h, c = n0 + row;
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = ac[4 * g + j] + Pm[row + j];
                if (psn) {
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 sv = snake_beta2_t<true>(mk2(w[j], w[j + 1]), mk2(Pm[2 * BN + row + j], Pm[2 * BN + row + j + 1]),
                                                             mk2(Pm[3 * BN + row + j], Pm[3 * BN + row + j + 1]));
                        w[j] = sv.x; w[j + 1] = sv.y;
                    }
                }
                *reinterpret_cast<uint2 *>(ybu + ((((long)(c >> 3) * a.y_cs + m) << 3) + (c & 4))) = pack4_bf16(w[0], w[1], w[2], w[3]);
            }
        };
#define RG_LEAN(i_)                                                                      \
    if constexpr ((i_) < TN) {                                                          \
        lean(acc[(i_) < TN ? (i_) : 0][0], (i_), m0 + wave * 64 + l31);                  \
        lean(acc[(i_) < TN ? (i_) : 0][1], (i_), m0 + wave * 64 + 32 + l31);             \
    }
        RG_LEAN(0) RG_LEAN(1) RG_LEAN(2) RG_LEAN(3) RG_LEAN(4) RG_LEAN(5)
#undef RG_LEAN
        return;
    }
    // (written out: left as loops, hipcc keeps them rolled -- the body is large -- and the accumulators move to scratch)
#define RG_EPI(i_)                                                                                                                    \
    if constexpr ((i_) < TN) {                                                                                                       \
        conv_epilogue_tile<BN, false, false, true>(a, acc[(i_) < TN ? (i_) : 0][0], Pm, n0, (i_)*32 + 4 * h, m0 + wave * 64 + l31, yu, ru, y2u, ybu);      \
        conv_epilogue_tile<BN, false, false, true>(a, acc[(i_) < TN ? (i_) : 0][1], Pm, n0, (i_)*32 + 4 * h, m0 + wave * 64 + 32 + l31, yu, ru, y2u, ybu); \
    }
    RG_EPI(0) RG_EPI(1) RG_EPI(2) RG_EPI(3) RG_EPI(4) RG_EPI(5)
#undef RG_EPI
}

// ---------------------------------------------------------------------------------------------------
// ONE tap: the 1x1 conv that closes an un-fused residual unit (blocks 0 / 1: y = res + W2 h + b2, h packed from the unit's conv7;
// conv1_bf16_ring_kernel<TN>).  With one tap nothing of an activation tile is reused across stages, so the tile has to run ahead
// like the weights: a ring of THREE stages, each the 32-channel activation tile (16 KB: [4 k-groups][256 positions]) and weight
// slot (8 KB at TN = 4) of one channel block, 74 KB in all -- two workgroups per CU.  The wait at a stage's barrier is a constant:
// younger than this stage's pieces are the next stage's XP + WP.  conv_gemm_bf16_kernel<4,false,true>, which this replaces on
// large launches, fetches each channel block through registers behind a barrier (a global-load latency exposed per 32 channels,
// twelve to twenty-four per workgroup, for 0.08 ms of MFMA time in a 0.7 ms launch) and then a residual round trip per output
// tile; here the residual of tile k + 1 is requested before tile k's epilogue.  Per output element the same chain of MFMAs
// (channel blocks in order, k-steps 0 / 1, lane half h = k-group 2 ks + h) and the same epilogue arithmetic: bit-identical.
namespace ring1 {
constexpr int NW = 4, NTH = 64 * NW, PT = 2, BM = 32 * PT * NW;
constexpr int XBUF = 4 * BM * 16;           // bytes per activation tile
constexpr int XP = 4 * BM / 64 / NW;        // 1 KB pieces per wave and tile
constexpr int R = 3;                        // stages in the ring
}  // namespace ring1

template <int TN>
__global__ __launch_bounds__(ring1::NTH, 2) void conv1_bf16_ring_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    using namespace ring1;
    constexpr int BN = 32 * TN;
    constexpr int SLOTB = 4 * BN * 16;
    constexpr int WP = SLOTB / 1024 / NW;
    constexpr int PPG = BN / 64;
    static_assert(BN % 64 == 0 && SLOTB % (1024 * NW) == 0, "whole pieces per wave");
    constexpr int PMB = 4 * BN * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);
    const unsigned x_lds = (unsigned)(size_t)(rg_lptr_t)(smem + PMB);
    const unsigned w_lds = x_lds + R * XBUF;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt_ = id / n_tiles;
    const int unit = pt_ / tiles_per_unit;
    const int m0 = (pt_ - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int ncb = a.Cin / 32;                     // (host: Cin % 32 == 0, >= 1)

    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);

    // activation pieces of this wave: piece wave + NW j = elements 64 (wave + NW j) + lane of [4 k-groups][BM]; positions past the
    // input repeat the last one and only reach outputs that are not stored
    unsigned xvoff[XP];
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = (wave + NW * j) * 64 + lane;
        const int kg = e / BM, t = e - kg * BM;
        int pos = m0 + t;
        pos = pos >= a.Lin ? a.Lin - 1 : pos;
        xvoff[j] = (unsigned)(((long)kg * a.x_cs + pos) * 16);
    }
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us) << 4);
    const unsigned long long xcb_bytes = (unsigned long long)a.x_cs * 64ull;
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wb) + ((unsigned long long)n0 << 4);
    const unsigned long long wkg_bytes = (unsigned long long)a.NP * 16ull;
    const unsigned wlane = (unsigned)lane * 16u;
    auto dma_stage = [&](int cb, int buf) {
        const unsigned long long xb_ = xbase + (unsigned long long)cb * xcb_bytes;
#pragma unroll
        for (int j = 0; j < XP; j++) rg_dma16(xb_, xvoff[j], x_lds + (unsigned)(buf * XBUF + (wave + NW * j) * 1024));
        const unsigned long long wb_ = wbase + (unsigned long long)(cb * 4) * wkg_bytes;
#pragma unroll
        for (int q = 0; q < WP; q++) {
            const int p = wave * WP + q, kg = p / PPG, part = p - kg * PPG;
            rg_dma16(wb_ + (unsigned long long)kg * wkg_bytes + (unsigned long long)(part * 1024), wlane, w_lds + (unsigned)(buf * SLOTB + p * 1024));
        }
    };

    f32x16 acc[TN][PT];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int p = 0; p < PT; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][p][r] = 0.0f;

    dma_stage(0, 0);
    if (ncb > 1) dma_stage(1, 1);

    const unsigned xaddr0 = x_lds + (unsigned)((h * BM + wave * 64 + l31) * 16);
    const unsigned waddr0 = w_lds + (unsigned)((h * BN + l31) * 16);
    int buf = 0;
    for (int cb = 0; cb < ncb; cb++) {
        // this wave's pieces of stage cb have landed (younger: the next stage's); then every wave's, and every wave is past stage
        // cb - 1, whose buffers take stage cb + 2
        if (cb + 1 < ncb) rg_wait_bar<XP + WP>();
        else rg_wait_bar<0>();
        if (cb + 2 < ncb) dma_stage(cb + 2, buf == 0 ? 2 : buf - 1);
        const unsigned xa = xaddr0 + (unsigned)(buf * XBUF);
        const unsigned wa = waddr0 + (unsigned)(buf * SLOTB);
        rg_bf16x8 aq[2][TN], bq[2][PT];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int p = 0; p < PT; p++) bq[ks][p] = rg_lds16(xa + (unsigned)(ks * 2 * BM * 16 + p * 512));
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TN; i++) aq[ks][i] = rg_lds16(wa + (unsigned)(ks * 2 * BN * 16 + i * 512));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int p = 0; p < PT; p++) acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][i], bq[ks][p], acc[i][p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == R - 1 ? 0 : buf + 1;
    }

    // epilogue, lean (S == 1, every row exists, bias + residual + optional SnakeBeta; the arithmetic of conv_epilogue_tile in its order:
    // (acc + bias), residual + that, SnakeBeta on row pairs): per tile the residual of the NEXT tile is requested first
    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[TN - 1][PT - 1][15]);
    const float *ru = ru0 + (long)unit * a.y_us;                           // (host: res != null)
    unsigned short *ybu0 = launder_after(a.yb, acc[TN - 1][PT - 1][15]);
    unsigned short *ybu = ybu0 + (((long)unit * a.y_us) << 3);             // (host: yb != null)
    unsigned short *yrb = (yu && a.rbf == 2) ? rbf_packed(a.y, yu) : nullptr;
    const unsigned short *rrb = a.rbi == 2 ? rbf_packed(a.res, ru) : nullptr;
    const int mw = m0 + wave * 64 + l31;
    const bool psn = a.psn_a != nullptr;
    const int rnd = a.rbf;
    static_assert(PT == 2 && TN == 4, "the epilogue below is written out for eight tiles per wave");
    // tile k = (row tile k >> 1, position tile k & 1); channel group g of it: channels n0 + 32 (k >> 1) + 8 g + 4 h .. + 3 at position m
    auto tile_m = [&](int k) { return mw + 32 * (k & 1); };
    auto epi = [&](const f32x16 &ac, int k, const float *t) {
        const int m = tile_m(k);
        if (m >= a.M) return;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = (k >> 1) * 32 + 8 * g + 4 * h, c = n0 + row;
            float v[4], w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = ac[4 * g + j] + Pm[row + j];
                v[j] = t[4 * g + j] + v[j];
                w[j] = v[j];
            }
            if (psn) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 sv = snake_beta2_t<true>(mk2(v[j], v[j + 1]), mk2(Pm[2 * BN + row + j], Pm[2 * BN + row + j + 1]),
                                                         mk2(Pm[3 * BN + row + j], Pm[3 * BN + row + j + 1]));
                    w[j] = sv.x; w[j + 1] = sv.y;
                }
            }
            const long pk = ((((long)(c >> 3) * a.y_cs + m) << 3) + (c & 4));
            if (yrb) {
                *reinterpret_cast<uint2 *>(yrb + pk) = pack4_bf16(v[0], v[1], v[2], v[3]);
            } else if (yu) {
                float *yp = yu + (long)c * a.y_cs + m;
#pragma unroll
                for (int j = 0; j < 4; j++) yp[(long)j * a.y_cs] = rnd ? bf16_round(v[j]) : v[j];
            }
            *reinterpret_cast<uint2 *>(ybu + pk) = pack4_bf16(w[0], w[1], w[2], w[3]);
        }
    };
    if (rrb) {
        ResPacked rq[2];
        auto fetch = [&](int k) {
            const int m = tile_m(k) < a.M ? tile_m(k) : a.M - 1;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int c = n0 + (k >> 1) * 32 + 8 * g + 4 * h;
                rq[k & 1].q[g] = *reinterpret_cast<const uint2 *>(rrb + ((((long)(c >> 3) * a.y_cs + m) << 3) + (c & 4)));
            }
        };
#define R1_EPI(k_)                                         \
    {                                                     \
        if ((k_) + 1 < 8) fetch((k_) + 1);                 \
        float t[16];                                      \
        fused_unpack_res(rq[(k_) & 1], t);                 \
        epi(acc[(k_) >> 1][(k_) & 1], (k_), t);            \
    }
        fetch(0);
        R1_EPI(0) R1_EPI(1) R1_EPI(2) R1_EPI(3) R1_EPI(4) R1_EPI(5) R1_EPI(6) R1_EPI(7)
#undef R1_EPI
    } else {
        float rf[2][16];
        auto fetch = [&](int k) {
            const int m = tile_m(k) < a.M ? tile_m(k) : a.M - 1;
            const float *rp = ru + (long)(n0 + (k >> 1) * 32 + 4 * h) * a.y_cs + m;
#pragma unroll
            for (int r = 0; r < 16; r++) rf[k & 1][r] = rp[(long)((r & 3) + 8 * (r >> 2)) * a.y_cs];
        };
#define R1_EPI(k_)                                         \
    {                                                     \
        if ((k_) + 1 < 8) fetch((k_) + 1);                 \
        epi(acc[(k_) >> 1][(k_) & 1], (k_), rf[(k_) & 1]); \
    }
        fetch(0);
        R1_EPI(0) R1_EPI(1) R1_EPI(2) R1_EPI(3) R1_EPI(4) R1_EPI(5) R1_EPI(6) R1_EPI(7)
#undef R1_EPI
    }
}

// ---------------------------------------------------------------------------------------------------
// The 4-wave ring kernel for any tap count >= 2 and any tile height (conv_bf16_ring_kernel<TN, NTAPS, EPI, RW>): also the
// decoder blocks' TRANSPOSED convs (two taps; rows = (channel, phase) in the octet-interleaved or phase-tile order of
// device_common.h, EPI = 2 / 3 = ConvGemmArgs::pm).  What differs from the kernel above:
//   * the wait at a stage's barrier is COUNTED at run time: the wave keeps the number of vector-memory operations it has issued
//     and, per ring slot and activation buffer, the count right after that tile's last piece; "operations younger than what
//     this stage needs" is a subtraction of wave-uniform scalars, dispatched to the s_waitcnt immediates through a switch.  No
//     per-schedule table of constants: with two taps the next activation tile is issued only two stages ahead and is YOUNGER
//     than the weight slot of the stage that first reads it, which the fixed counts above do not cover;
//   * a weight slot is [4 k-groups][BN rows] for any BN: 1 KB pieces of 64 consecutive elements with per-lane source offsets
//     (a piece may straddle two k-group rows), dealt to the waves round-robin (waves may differ by one piece per slot);
//   * RW = 3 slots where 32 TN rows x 4 slots would not leave room for two workgroups per CU (TN = 5).
template <int N>
__device__ __forceinline__ void rg_wait_only() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void rg_wait_dyn(int n) {          // n wave-uniform; a smaller immediate than n only waits longer
#define RG_W(k_) case k_: rg_wait_only<k_>(); break;
    switch (n) {
        RG_W(0) RG_W(1) RG_W(2) RG_W(3) RG_W(4) RG_W(5) RG_W(6) RG_W(7) RG_W(8) RG_W(9) RG_W(10) RG_W(11) RG_W(12)
        RG_W(13) RG_W(14) RG_W(15) RG_W(16) RG_W(17) RG_W(18) RG_W(19) RG_W(20) RG_W(21) RG_W(22) RG_W(23)
        default: rg_wait_only<24>(); break;
    }
#undef RG_W
}

template <int TN, int NTAPS, int EPI, int RW>
__global__ __launch_bounds__(ring::NTH, 2) void conv_bf16_ring_kernel(ConvGemmArgs a, int HL, int tiles_per_unit, int n_tiles) {
    using namespace ring;
    static_assert(NTAPS >= 2 && RW >= 3 && RW <= 4, "see the header comment");
    constexpr int BN = 32 * TN;
    constexpr int SLOTB = 4 * BN * 16;              // bytes per weight slot [4 k-groups][BN rows][8 bf16]
    constexpr int NPIECES = SLOTB / 1024;           // 2 TN
    constexpr int WPMAX = (NPIECES + NW - 1) / NW;
    constexpr int PMB = 4 * BN * 4;
    constexpr bool BNEXT = NTAPS > 2 && TN <= 4;    // the next tap's B operands are read during this stage's MFMAs (16 registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);
    uint4 *Xs = reinterpret_cast<uint4 *>(smem + PMB);
    const unsigned x_lds = (unsigned)(size_t)(rg_lptr_t)(smem + PMB);
    const unsigned w_lds = x_lds + 2 * XBUF;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt_ = id / n_tiles;
    const int unit = pt_ / tiles_per_unit;
    const int m0 = (pt_ - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int ncb = a.Cin / 32;                     // (host: Cin % 32 == 0)
    const int nstage = ncb * NTAPS;

    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    // tile position t holds input position m0 - HL + t (HL = the left halo, host); positions < 0 stay zero in both buffers (the
    // DMA's lanes for them are masked off), positions >= Lin repeat the last one and only reach outputs that are not stored
    if (m0 - HL < 0) {
        const int nz = HL - m0;
        for (int e = tid; e < 2 * 4 * nz; e += NTH) {
            const int row = e / nz, t = e - row * nz;
            Xs[row * XW + t] = make_uint4(0, 0, 0, 0);
        }
    }
    unsigned xvoff[XP];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = (wave + NW * j) * 64 + lane;
        const int kg = e / XW, t = e - kg * XW;
        int pos = m0 - HL + t;
        if (pos >= 0) xmask |= 1u << j;
        pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
        xvoff[j] = (unsigned)(((long)kg * a.x_cs + pos) * 16);
    }
    int issued = 0;                                 // vector-memory operations (DMA pieces) this wave has issued: wave-uniform
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us) << 4);
    const unsigned long long xcb_bytes = (unsigned long long)a.x_cs * 64ull;
    auto dma_x = [&](int cb, int buf) {
        const unsigned long long base = xbase + (unsigned long long)cb * xcb_bytes;
#pragma unroll
        for (int j = 0; j < XP; j++)
            if ((xmask >> j) & 1) rg_dma16(base, xvoff[j], x_lds + (unsigned)(buf * XBUF + (wave + NW * j) * 1024));
        issued += XP;                               // (every piece has at least one lane inside the unit: HL < 64)
    };
    // weight pieces of this wave: piece wave + NW q covers the slot's elements 64 (wave + NW q) + lane = (k-group, row)
    unsigned wvoff[WPMAX];
#pragma unroll
    for (int q = 0; q < WPMAX; q++) {
        const int e = (wave + NW * q) * 64 + lane;
        const int kg = e / BN, r = e - kg * BN;
        wvoff[q] = (unsigned)(((long)(kg < 4 ? kg : 0) * a.NP + r) * 16);
    }
    const int wcount = (NPIECES - wave + NW - 1) / NW;            // pieces per slot of this wave (wave-uniform)
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wb) + ((unsigned long long)n0 << 4);
    const unsigned long long wkg_bytes = (unsigned long long)a.NP * 16ull;
    auto dma_w = [&](int cb, int tap, int slot) {
        const unsigned long long base = wbase + (unsigned long long)((tap * ncb + cb) * 4) * wkg_bytes;
#pragma unroll
        for (int q = 0; q < WPMAX; q++)
            if (q < wcount) rg_dma16(base, wvoff[q], w_lds + (unsigned)(slot * SLOTB + (wave + NW * q) * 1024));
        issued += wcount;
    };

    f32x16 acc[PT][TN];
#pragma unroll
    for (int p = 0; p < PT; p++)
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[p][i][r] = 0.0f;

    // marks: `issued` right after the last piece of the tile in ring slot j / activation buffer j
    int wmark[RW], xmark[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < RW; j++) wmark[j] = 0;
    dma_x(0, 0);
    xmark[0] = issued;
#pragma unroll
    for (int j = 0; j < RW - 1; j++) {              // slots of stages 0 .. RW - 2 (stage j = (cb, tap) = (j / NTAPS, j % NTAPS))
        if (j < nstage) { dma_w(j / NTAPS, j % NTAPS, j); wmark[j] = issued; }
    }

    const unsigned xaddr0 = x_lds + (unsigned)((h * XW + wave * 64 + l31 + HL + a.shift0) * 16);
    const unsigned waddr0 = w_lds + (unsigned)((h * BN + l31) * 16);
    const unsigned d16 = (unsigned)a.shift_step * 16u;

    rg_bf16x8 bq[2][PT];
    int slot = 0;                                   // ring slot of the current stage = stage mod RW
    int cb3 = (RW - 1) / NTAPS, tap3 = (RW - 1) % NTAPS;          // (cb, tap) of the stage RW - 1 ahead
    for (int cb = 0; cb < ncb; cb++) {
#pragma unroll
        for (int tap = 0; tap < NTAPS; tap++) {
            // this wave's pieces of the stage's slot -- and, at tap 0, of the block's activation tile -- have landed: everything it
            // issued after them may stay in flight; then every wave's
            {
                int need = 0;
#pragma unroll
                for (int j = 0; j < RW; j++) need = slot == j ? wmark[j] : need;
                if (tap == 0) { const int xm = (cb & 1) ? xmark[1] : xmark[0]; need = xm > need ? xm : need; }
                rg_wait_dyn(issued - need);
                rg_wait_bar<63>();                  // (lgkmcnt(0) + barrier; the vmcnt part is the counted wait above)
            }
            // every wave is past the previous stage: its slot, and at tap 0 the other activation buffer, are free
            if (cb3 < ncb) {
                int fs = slot == 0 ? RW - 1 : slot - 1;
                dma_w(cb3, tap3, fs);
#pragma unroll
                for (int j = 0; j < RW; j++) wmark[j] = fs == j ? issued : wmark[j];
            }
            if (++tap3 == NTAPS) { tap3 = 0; cb3++; }
            if (tap == 0 && cb + 1 < ncb) {
                dma_x(cb + 1, (cb + 1) & 1);
                if ((cb + 1) & 1) xmark[1] = issued; else xmark[0] = issued;
            }
            const unsigned xa = xaddr0 + (unsigned)((cb & 1) * XBUF) + (unsigned)tap * d16;
            const unsigned wa = waddr0 + (unsigned)(slot * SLOTB);
            rg_bf16x8 aq[2][TN];
            if (tap == 0 || !BNEXT) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bq[ks][p] = rg_lds16(xa + (unsigned)(ks * 2 * XW * 16 + p * 512));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TN; i++) aq[ks][i] = rg_lds16(wa + (unsigned)(ks * 2 * BN * 16 + i * 512));
            rg_bf16x8 bn[BNEXT ? 2 : 1][PT];
            if (BNEXT && tap + 1 < NTAPS) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bn[ks][p] = rg_lds16(xa + d16 + (unsigned)(ks * 2 * XW * 16 + p * 512));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TN; i++)
#pragma unroll
                    for (int p = 0; p < PT; p++) acc[p][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][i], bq[ks][p], acc[p][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (BNEXT && tap + 1 < NTAPS) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int p = 0; p < PT; p++) bq[ks][p] = bn[ks][p];
            }
            slot = slot + 1 == RW ? 0 : slot + 1;
        }
    }

    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[PT - 1][TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[PT - 1][TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    unsigned short *ybu0 = launder_after(a.yb, acc[PT - 1][TN - 1][15]);
    unsigned short *ybu = ybu0 ? ybu0 + (((long)unit * a.y_us) << 3) : nullptr;
    static_assert(PT == 2 && TN <= 6, "the epilogue below is written out for these");
    if constexpr (EPI == 3) {
        tconv_pt_epilogue<BN, TN, true>(a, acc[0], Pm, n0, h, m0 + wave * 64 + l31, yu, y2u, ybu);
        tconv_pt_epilogue<BN, TN, true>(a, acc[1], Pm, n0, h, m0 + wave * 64 + 32 + l31, yu, y2u, ybu);
    } else {
        // (written out: left as loops, hipcc keeps them rolled -- the body is large -- and the accumulators move to scratch)
#define RG_EPI(i_)                                                                                                                    \
    if constexpr ((i_) < TN) {                                                                                                       \
        conv_epilogue_tile<BN, false, EPI == 2, true>(a, acc[0][(i_) < TN ? (i_) : 0], Pm, n0, (i_)*32 + 4 * h, m0 + wave * 64 + l31, yu, ru, y2u, ybu);      \
        conv_epilogue_tile<BN, false, EPI == 2, true>(a, acc[1][(i_) < TN ? (i_) : 0], Pm, n0, (i_)*32 + 4 * h, m0 + wave * 64 + 32 + l31, yu, ru, y2u, ybu); \
    }
        RG_EPI(0) RG_EPI(1) RG_EPI(2) RG_EPI(3) RG_EPI(4) RG_EPI(5)
#undef RG_EPI
    }
}

// ---------------------------------------------------------------------------------------------------
// A whole residual unit of a 192-channel block (block 2) on the ring (resunit_bf16_ring_kernel<6>):
//     y = x + W2 snake2(conv7_dil(xb) + b1) + b2      (xb = snake1(x), packed, written by the producer)
// conv_gemm_bf16_kernel<6,true,true> runs it in ONE 8-wave workgroup of 256 positions per CU: barrier -> operand reads -> twelve
// MFMAs -> register-staged weight commit in lock-step, then an epilogue (two SnakeBeta evaluations per element, the 1x1 conv with
// a barrier per output tile, residual loads, stores) that nothing overlaps -- matrix pipe busy 32 % of the cycles, 46 % of all
// wave cycles parked (profiles/r03_pmc_sq_bf16.txt).  Here: 4-wave workgroups of 128 positions (a wave = 192 rows x 32 positions,
// six accumulator tiles), 77 KB of LDS, so TWO independent workgroups share a CU and one's epilogue runs under the other's main
// loop; weights by LDS-DMA three stages ahead through four 12 KB slots -- the six output tiles of the 1x1 conv's weights simply
// follow the conv7's 42 tiles in the same stream of slots -- activation tile double-buffered.  (The 4-wave shape was tried with
// register-staged weights in round 3 and lost 5 %: twice the staging instructions per position.  The DMA issues no such
// instructions; what doubles is L2 -> LDS traffic.)  Per output element the MFMA chains, SnakeBeta evaluations, roundings and
// the residual add are conv_gemm_bf16_kernel<6,true,true>'s: bit-identical (tests/test_gpu_bf16.py).
namespace ringu {
constexpr int NW = 4, NTH = 64 * NW, BM = 32 * NW;                  // one 32-position tile per wave
constexpr int XW = 192;                     // BM + the left halo (<= 64)
constexpr int XBUF = 4 * XW * 16;
constexpr int XP = 4 * XW / 64 / NW;        // 3 pieces per wave and activation tile
constexpr int RW = 4;
constexpr int NTAPS = 7;
static_assert(4 * XW % (64 * NW) == 0, "every wave issues the same number of activation pieces");
}  // namespace ringu

// RB: the residual stream is the packed bf16 tensor (ConvGemmArgs::rbi == 2): four 8-byte loads per tile instead of sixteen 4-byte ones
template <int TN, bool RB>
__global__ __launch_bounds__(ringu::NTH, 2) void resunit_bf16_ring_kernel(ConvGemmArgs a, int tiles_per_unit) {
    using namespace ringu;
    constexpr int BN = 32 * TN;
    constexpr int SLOTB = 4 * BN * 16;              // conv7 slot [4 k-groups][BN rows]; also one output tile of the 1x1 conv: [2 TN k-steps][64 lanes]
    constexpr int WP = SLOTB / 1024 / NW;
    static_assert(SLOTB % (1024 * NW) == 0 && 2 * TN * 64 * 16 == SLOTB, "whole pieces per wave; a 1x1 tile fills a slot");
    constexpr int PMB = 6 * BN * 4;                 // fused parameter table (device_common.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);
    uint4 *Xs = reinterpret_cast<uint4 *>(smem + PMB);
    const unsigned x_lds = (unsigned)(size_t)(rg_lptr_t)(smem + PMB);
    const unsigned w_lds = x_lds + 2 * XBUF;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int unit = id / tiles_per_unit;
    const int m0 = (id - unit * tiles_per_unit) * BM;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int d = a.shift_step, HL = -a.shift0;     // (host: HL == 6 d <= 64)
    const int ncb = a.Cin / 32;
    const int nstage = ncb * NTAPS;

    fill_param_table<BN, true>(a, Pm, 0, tid, NTH);
    if (m0 == 0) {
        for (int e = tid; e < 2 * 4 * HL; e += NTH) {
            const int row = e / HL, t = e - row * HL;
            Xs[row * XW + t] = make_uint4(0, 0, 0, 0);
        }
    }
    unsigned xvoff[XP];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = (wave + NW * j) * 64 + lane;
        const int kg = e / XW, t = e - kg * XW;
        int pos = m0 - HL + t;
        if (pos >= 0) xmask |= 1u << j;
        pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
        xvoff[j] = (unsigned)(((long)kg * a.x_cs + pos) * 16);
    }
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us) << 4);
    const unsigned long long xcb_bytes = (unsigned long long)a.x_cs * 64ull;
    auto dma_x = [&](int cb, int buf) {
        if ((RG_ABL & 2) && cb > 0) return;
        const unsigned long long base = xbase + (unsigned long long)cb * xcb_bytes;
#pragma unroll
        for (int j = 0; j < XP; j++)
            if ((xmask >> j) & 1) rg_dma16(base, xvoff[j], x_lds + (unsigned)(buf * XBUF + (wave + NW * j) * 1024));
    };
    // the weight stream: stages 0 .. nstage - 1 are the conv7's (cb, tap) slots [4][NP] (NP == BN: contiguous), stages
    // nstage .. nstage + TN - 1 the output tiles of the 1x1 conv (ConvGemmArgs::wb2: [TN tiles][2 TN k-steps][64 lanes], SLOTB each)
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wb);
    const unsigned long long w2base = reinterpret_cast<unsigned long long>(a.wb2);
    const unsigned wlane = (unsigned)lane * 16u;
    auto dma_w = [&](int st, int slot) {            // (st < nstage + TN: caller)
        if ((RG_ABL & 1) && st > 2) return;
        unsigned long long base;
        if (st < nstage) {
            const int cb = st / NTAPS, tap = st - cb * NTAPS;
            base = wbase + (unsigned long long)(tap * ncb + cb) * SLOTB;
        } else {
            base = w2base + (unsigned long long)(st - nstage) * SLOTB;
        }
#pragma unroll
        for (int q = 0; q < WP; q++) rg_dma16(base + (unsigned long long)((wave * WP + q) * 1024), wlane, w_lds + (unsigned)(slot * SLOTB + (wave * WP + q) * 1024));
    };

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    dma_x(0, 0);
    dma_w(0, 0);
    dma_w(1, 1);
    dma_w(2, 2);

    const unsigned xaddr0 = x_lds + (unsigned)((h * XW + wave * 32 + l31) * 16);
    const unsigned waddr0 = w_lds + (unsigned)((h * BN + l31) * 16);
    const unsigned d16 = (unsigned)d * 16u;

    // The barrier at the top of stage st certifies the slot of stage st + 1 (each wave waits for its own pieces of THAT slot
    // first), so the operands of k-step (st + 1, 0) can be read during the MFMAs of k-step (st, 1), across the barrier: no MFMA
    // waits for an LDS read issued behind the barrier it follows.  Two operand sets alternate per k-step.
    rg_bf16x8 aq[2][TN], bq[2];
#define RU_READ(set_, xa_, wa_, ks_)                                                                                \
    {                                                                                                               \
        bq[set_] = rg_lds16((xa_) + (unsigned)((ks_)*2 * XW * 16));                                                 \
        _Pragma("unroll") for (int i = 0; i < TN; i++) aq[set_][i] = rg_lds16((wa_) + (unsigned)((ks_)*2 * BN * 16 + i * 512)); \
    }
#define RU_MFMA(set_)                                                                                               \
    { _Pragma("unroll") for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[set_][i], bq[set_], acc[i], 0, 0, 0); }
    rg_wait_bar<2 * WP>();                          // slot 0 and the first activation tile (younger: slots 1, 2)
    RU_READ(0, xaddr0, waddr0, 0)
    for (int cb = 0; cb < ncb; cb++) {
        const bool lastcb = cb + 1 == ncb;
#pragma unroll
        for (int tap = 0; tap < NTAPS; tap++) {
            const int st = cb * NTAPS + tap, slot = st & (RW - 1);
            // slot st + 1 (issued two stages ago; the 1x1 conv's tiles continue the stream, so it always exists): younger are the
            // slot of stage st + 2 and, during taps 1-2, the next block's activation tile (issued at tap 0, behind that stage's slot)
            if (!lastcb && tap >= 1 && tap <= 2) rg_wait_bar<WP + XP>();
            else rg_wait_bar<WP>();
            // every wave is past stage st - 1 (whose operands it read before this barrier): that slot, and at tap 0 the other
            // activation buffer, are free
            dma_w(st + 3, (st + 3) & (RW - 1));     // (st + 3 <= nstage + 2 < nstage + TN)
            if (tap == 0 && !lastcb) dma_x(cb + 1, (cb + 1) & 1);
            const unsigned xa = xaddr0 + (unsigned)((cb & 1) * XBUF) + (unsigned)tap * d16;
            const unsigned wa = waddr0 + (unsigned)(slot * SLOTB);
            RU_READ(1, xa, wa, 1)
            __builtin_amdgcn_sched_barrier(0);
            RU_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            if (!(lastcb && tap == NTAPS - 1)) {    // k-step (st + 1, 0): the next tap of this tile, or tap 0 of the next block's
                const unsigned xa1 = tap + 1 < NTAPS ? xa + d16 : xaddr0 + (unsigned)(((cb + 1) & 1) * XBUF);
                const unsigned wa1 = waddr0 + (unsigned)(((st + 1) & (RW - 1)) * SLOTB);
                RU_READ(0, xa1, wa1, 0)
            }
            __builtin_amdgcn_sched_barrier(0);
            RU_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef RU_READ
#undef RU_MFMA

    const int m = m0 + wave * 32 + l31;
    // the residual of output tile o: sixteen unconditional loads from a clamped position (N == BN: every row exists), requested one
    // tile AHEAD; tile 0's before the SnakeBeta pass below
    const float *rup = a.res + (long)unit * a.y_us + (m < a.M ? m : a.M - 1) + (long)(4 * h) * a.y_cs;
    const float *rupk = a.res + (long)unit * a.y_us;
    float resv[RB ? 1 : 2][16];
    ResPacked resq[RB ? 2 : 1];
    auto fetch_res = [&](int o) {
        if constexpr (RB) {
            fused_fetch_res_packed(a, rupk, o * 32 + 4 * h, m < a.M ? m : a.M - 1, true, resq[o & 1]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) resv[o & 1][r] = rup[(long)(o * 32 + (r & 3) + 8 * (r >> 2)) * a.y_cs];
        }
    };
    fetch_res(0);
    // the first output tile's weights (slot nstage: certified by the last stage's barrier) are read under the SnakeBeta pass too
    const unsigned w2addr0 = w_lds + (unsigned)lane * 16u;
    rg_bf16x8 w2q[2][2 * TN];
#pragma unroll
    for (int k = 0; k < 2 * TN; k++) w2q[0][k] = rg_lds16(w2addr0 + (unsigned)((nstage & (RW - 1)) * SLOTB + k * 1024));
    // ---- H = snake2(acc + b1) -> the 1x1 conv's bf16 B operands: a lane's registers 8g .. 8g+7 of tile i are its eight k of k-step 2i + g
    rg_bf16x8 hb[TN][2];
#pragma unroll
    for (int i = 0; i < TN; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (r & 1) {
                f32x2 hv = snake_beta2_t<true>(mk2(acc[i][r - 1] + Pm[n - 1], acc[i][r] + Pm[n]), mk2(Pm[BN + n - 1], Pm[BN + n]),
                                       mk2(Pm[2 * BN + n - 1], Pm[2 * BN + n]));
                acc[i][r - 1] = hv.x;
                acc[i][r] = hv.y;
            }
            if ((r & 3) == 3) {
                asm volatile("" : "+v"(acc[i][r - 3]), "+v"(acc[i][r - 2]), "+v"(acc[i][r - 1]), "+v"(acc[i][r]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int j = 0; j < 8; j++) hb[i][g][j] = (__bf16)acc[i][8 * g + j];
    }
    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    unsigned short *ybu = a.yb ? a.yb + (((long)unit * a.y_us) << 3) : nullptr;
#pragma unroll
    for (int o = 0; o < TN; o++) {
        const int st = nstage + o;
        // certify the next tile's slot (younger: the tile after it -- and, unknown to this count, residual loads: the smaller
        // immediate only waits for them too); then that slot of the ring is everybody's to read, and tile o - 1's is free
        asm volatile("" ::: "memory");
        if (o + 2 < TN) rg_wait_bar<WP>();
        else rg_wait_bar<0>();
        if (o + 3 < TN) dma_w(st + 3, (st + 3) & (RW - 1));
        if (o + 1 < TN) fetch_res(o + 1);
        asm volatile("" ::: "memory");
        if (o + 1 < TN) {
#pragma unroll
            for (int k = 0; k < 2 * TN; k++) w2q[(o + 1) & 1][k] = rg_lds16(w2addr0 + (unsigned)(((st + 1) & (RW - 1)) * SLOTB + k * 1024));
        }
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; r++) acc2[r] = 0.0f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int g = 0; g < 2; g++) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2q[o & 1][i * 2 + g], hb[i][g], acc2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RB) {
            // the production case (packed bf16 stream in, packed stream and / or packed activated tensor out; N == BN: every row
            // exists): fused_epilogue_tile's arithmetic in its order, one base per channel group, no per-element predicates
            if (ybu && (!yu || a.rbf == 2) && !(a.dbg & 1)) {                   // (workgroup-uniform)
                fused_unpack_res(resq[o & 1], resv[0]);
                if (m < a.M) {
                    unsigned short *yrb = yu ? rbf_packed(a.y, yu) : nullptr;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int row = o * 32 + 8 * g + 4 * h;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = resv[0][4 * g + j] + (acc2[4 * g + j] + Pm[3 * BN + row + j]);
                        const long pk = ((((long)(row >> 3) * a.y_cs + m) << 3) + (row & 4));
                        if (yrb) *reinterpret_cast<uint2 *>(yrb + pk) = pack4_bf16(v[0], v[1], v[2], v[3]);
                        if (a.psn_a) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const f32x2 sv = snake_beta2_t<true>(mk2(v[j], v[j + 1]), mk2(Pm[4 * BN + row + j], Pm[4 * BN + row + j + 1]),
                                                                     mk2(Pm[5 * BN + row + j], Pm[5 * BN + row + j + 1]));
                                v[j] = sv.x; v[j + 1] = sv.y;
                            }
                        }
                        *reinterpret_cast<uint2 *>(ybu + pk) = pack4_bf16(v[0], v[1], v[2], v[3]);
                    }
                }
            } else {
                fused_unpack_res(resq[o & 1], resv[0]);
                fused_epilogue_tile<BN, true>(a, acc2, resv[0], Pm, o * 32 + 4 * h, m, yu, ybu);
            }
        } else {
            fused_epilogue_tile<BN, true>(a, acc2, resv[o & 1], Pm, o * 32 + 4 * h, m, yu, ybu);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Measured out for the residual unit (all bit-identical; block 2's three units per 64-unit pass on MI355X): TWO position tiles per
// wave (192 rows x 64 positions, 192 accumulator registers, 256 positions per workgroup, three weight slots and an activation tile
// of exactly 312 positions so that two workgroups still fit a CU): half the operand reads per MFMA and half the L2 -> LDS weight
// traffic, but operands single-buffered per k-step and 77 registers spilled around the SnakeBeta pass -- 6.57 ms against 6.26
// (its floor without DMA, barriers and operand reads: 5.05 ms against 4.55).  Timing ablations of the kernel above: without the
// weight DMA 5.00 ms, without the LDS operand reads 4.97, without barriers 6.24, without all three 4.55; reading the operands of
// k-step (st + 1, 0) across the barrier (the certify-ahead form it has now) moved 6.26 to 6.18: what the DMA and the operand reads
// cost is LDS and L2 bandwidth next to each other (13 GB of slots per launch: 6.3 TB/s), not exposed latency.

// ---------------------------------------------------------------------------------------------------
// The same conv in ONE 8-wave workgroup of 512 positions per CU, with a ring of EIGHT weight slots (conv7_bf16_ring8_kernel).
// Timing ablations of the 4-wave form on block 1 (4.02 ms per pass for its three layers): without the weight DMA 3.24, without
// any DMA 3.05, and without barriers 2.96 -- a weight tile three stages ahead is issued ~1.5 us before its use, about the
// DMA's issue-to-landing time under load, and every position tile of 256 fetches its own copy of the weights.  Here a slot
// feeds 512 positions (half the L2 -> LDS weight traffic), is issued SEVEN stages ahead, and the barrier at the top of stage s
// certifies the slot of stage s + 1 (each wave waits for its own piece of that slot before the barrier): the operands of
// k-step (s + 1, 0) are read from LDS during the MFMAs of k-step (s, 1), i.e. across the barrier, so no MFMA waits for an LDS
// read issued behind the barrier it follows.  Two operand register sets alternate per k-step.
namespace ring8 {
constexpr int BM = 512;                     // positions per workgroup: NW waves x PT 32-position tiles
constexpr int XW = 640;                     // BM + the left halo (<= 64) in whole pieces per wave
constexpr int XBUF = 4 * XW * 16;
constexpr int RW = 8;
constexpr int NTAPS = 7;
static_assert(RW - 1 == NTAPS, "the slot issued at (cb, tap) is (cb + 1, tap)");
}  // namespace ring8

// NW x PT = 16: eight waves of 64 positions (two per SIMD; the instance that is built), or four waves of 128 positions -- one
// wave per SIMD with 4 x 4 accumulator tiles in 256 AGPRs, eight LDS operand reads per sixteen MFMAs, the shape library GEMMs
// use.  That instance was built, is bit-identical too, and measured MUCH slower here (block 1's layers 3.90 -> 5.77 ms per
// pass, block 0's 2.77 -> 3.53): with one wave per SIMD nothing covers the stage barrier, the DMA's scalar address arithmetic
// or -- at K = 2 688 -- the epilogue of sixteen tiles, which is a third of the main loop's length.
template <int TN, int NW, int PT>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 1 : 2)) void conv7_bf16_ring8_kernel(ConvGemmArgs a, int tiles_per_unit, int n_tiles) {
    using namespace ring8;
    constexpr int NTH = 64 * NW;
    constexpr int XP = 4 * XW / 64 / NW;            // activation pieces per wave: 5, or 10
    static_assert(32 * PT * NW == BM && 4 * XW % (64 * NW) == 0, "geometry");
    constexpr int BN = 32 * TN;
    constexpr int SLOTB = 4 * BN * 16;
    constexpr int WP = SLOTB / 1024 / NW;           // 1 for 128 rows
    constexpr int PPG = BN / 64;
    static_assert(BN % 64 == 0 && SLOTB % (1024 * NW) == 0 && WP >= 1, "whole pieces per wave");
    constexpr int PMB = 4 * BN * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Pm = reinterpret_cast<float *>(smem);
    uint4 *Xs = reinterpret_cast<uint4 *>(smem + PMB);
    const unsigned x_lds = (unsigned)(size_t)(rg_lptr_t)(smem + PMB);
    const unsigned w_lds = x_lds + 2 * XBUF;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int nt = id % n_tiles;
    const int pt_ = id / n_tiles;
    const int unit = pt_ / tiles_per_unit;
    const int m0 = (pt_ - unit * tiles_per_unit) * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int d = a.shift_step, HL = -a.shift0;
    const int ncb = a.Cin / 32;

    fill_param_table<BN, false>(a, Pm, n0, tid, NTH);
    if (m0 == 0) {
        for (int e = tid; e < 2 * 4 * HL; e += NTH) {
            const int row = e / HL, t = e - row * HL;
            Xs[row * XW + t] = make_uint4(0, 0, 0, 0);
        }
    }
    unsigned xvoff[XP];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int e = (wave + NW * j) * 64 + lane;
        const int kg = e / XW, t = e - kg * XW;
        int pos = m0 - HL + t;
        if (pos >= 0) xmask |= 1u << j;
        pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
        xvoff[j] = (unsigned)(((long)kg * a.x_cs + pos) * 16);
    }
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us) << 4);
    const unsigned long long xcb_bytes = (unsigned long long)a.x_cs * 64ull;
    auto dma_x = [&](int cb, int buf) {
        if ((RG_ABL & 2) && cb > 0) return;
        const unsigned long long base = xbase + (unsigned long long)cb * xcb_bytes;
#pragma unroll
        for (int j = 0; j < XP; j++)
            if ((xmask >> j) & 1) rg_dma16(base, xvoff[j], x_lds + (unsigned)(buf * XBUF + (wave + NW * j) * 1024));
    };
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wb) + ((unsigned long long)n0 << 4);
    const unsigned long long wkg_bytes = (unsigned long long)a.NP * 16ull;
    const unsigned wlane = (unsigned)lane * 16u;
    auto dma_w = [&](int cb, int tap, int slot) {
        if ((RG_ABL & 1) && cb > 0) return;
        const unsigned long long base = wbase + (unsigned long long)((tap * ncb + cb) * 4) * wkg_bytes;
#pragma unroll
        for (int q = 0; q < WP; q++) {
            const int p = wave * WP + q, kg = p / PPG, part = p - kg * PPG;
            rg_dma16(base + (unsigned long long)kg * wkg_bytes + (unsigned long long)(part * 1024), wlane, w_lds + (unsigned)(slot * SLOTB + p * 1024));
        }
    };

    f32x16 acc[TN][PT];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int p = 0; p < PT; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][p][r] = 0.0f;

    // prologue: the first activation tile, then the slots of stages 0 .. 6 (slot j of the ring = stage j mod 8)
    dma_x(0, 0);
#pragma unroll
    for (int t = 0; t < NTAPS; t++) dma_w(0, t, t);

    const unsigned xaddr0 = x_lds + (unsigned)((h * XW + wave * (32 * PT) + l31) * 16);
    const unsigned waddr0 = w_lds + (unsigned)((h * BN + l31) * 16);
    const unsigned d16 = (unsigned)d * 16u;

    // two operand sets, alternating per k-step: set ks holds the operands of k-step (stage, ks)
    rg_bf16x8 aq[2][TN], bq[2][PT];
#define RG8_READ(set_, xa_, wa_, ks_)                                                                               \
    {                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < PT; p++) bq[set_][p] = rg_lds16((xa_) + (unsigned)((ks_)*2 * XW * 16 + p * 512)); \
        _Pragma("unroll") for (int i = 0; i < TN; i++) aq[set_][i] = rg_lds16((wa_) + (unsigned)((ks_)*2 * BN * 16 + i * 512)); \
    }
#define RG8_MFMA(set_)                                                                                              \
    {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < TN; i++)                                                              \
            _Pragma("unroll") for (int p = 0; p < PT; p++)                                                          \
                acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[set_][i], bq[set_][p], acc[i][p], 0, 0, 0);  \
    }
    // slot 0 and the first activation tile: this wave's pieces (younger: the slots of stages 1 .. 6), then everybody's
    rg_wait_bar<6 * WP>();
    RG8_READ(0, xaddr0, waddr0, 0)

    for (int cb = 0; cb < ncb; cb++) {
        const bool lastcb = cb + 1 == ncb;
#pragma unroll
        for (int tap = 0; tap < NTAPS; tap++) {
            const int st = cb * NTAPS + tap;
            // Certify the slot of stage st + 1 (issued seven stages ago; at tap 6 also the next activation tile, issued before it):
            // younger operations of this wave are the slots of stages st + 2 .. st + 6, plus, during taps 1 .. 5, the activation
            // tile of the next block issued at tap 0; in the last block the ring drains (5 - tap slots left behind it).
            if (lastcb) {
                if (tap == 0) rg_wait_bar<5 * WP>();
                else if (tap == 1) rg_wait_bar<4 * WP>();
                else if (tap == 2) rg_wait_bar<3 * WP>();
                else if (tap == 3) rg_wait_bar<2 * WP>();
                else if (tap == 4) rg_wait_bar<1 * WP>();
                else rg_wait_bar<0>();
            } else if (tap >= 1 && tap <= 5) rg_wait_bar<5 * WP + XP>();
            else rg_wait_bar<5 * WP>();
            // every wave is past stage st - 1 (whose operands it read before this barrier): that slot, and at tap 0 the other
            // activation buffer, are free.  The activation tile goes FIRST: the wait for slot (cb + 1, 0) at tap 6 then covers it.
            if (!lastcb) {
                if (tap == 0) dma_x(cb + 1, (cb + 1) & 1);
                dma_w(cb + 1, tap, (st + RW - 1) & (RW - 1));
            }
            const unsigned xa = xaddr0 + (unsigned)((cb & 1) * XBUF) + (unsigned)tap * d16;
            const unsigned wa = waddr0 + (unsigned)((st & (RW - 1)) * SLOTB);
            RG8_READ(1, xa, wa, 1)
            __builtin_amdgcn_sched_barrier(0);
            RG8_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            if (!(lastcb && tap == NTAPS - 1)) {      // k-step (st + 1, 0): the next tap of this tile, or tap 0 of the next block's
                const unsigned xa1 = tap + 1 < NTAPS ? xa + d16 : xaddr0 + (unsigned)(((cb + 1) & 1) * XBUF);
                const unsigned wa1 = waddr0 + (unsigned)(((st + 1) & (RW - 1)) * SLOTB);
                RG8_READ(0, xa1, wa1, 0)
            }
            __builtin_amdgcn_sched_barrier(0);
            RG8_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef RG8_READ
#undef RG8_MFMA

    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[TN - 1][PT - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][PT - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    unsigned short *ybu0 = launder_after(a.yb, acc[TN - 1][PT - 1][15]);
    unsigned short *ybu = ybu0 ? ybu0 + (((long)unit * a.y_us) << 3) : nullptr;
    static_assert(PT <= 4 && TN <= 6, "the epilogue below is written out for these");
    // (written out: left as loops, hipcc keeps them rolled -- the body is large -- and the accumulators move to scratch)
#define RG_EPI1(i_, p_)                                                                                                              \
    if constexpr ((i_) < TN && (p_) < PT)                                                                                            \
        conv_epilogue_tile<BN, false, false, true>(a, acc[(i_) < TN ? (i_) : 0][(p_) < PT ? (p_) : 0], Pm, n0, (i_)*32 + 4 * h,            \
                                             m0 + wave * (32 * PT) + (p_)*32 + l31, yu, ru, y2u, ybu);
#define RG_EPI(i_) RG_EPI1(i_, 0) RG_EPI1(i_, 1) RG_EPI1(i_, 2) RG_EPI1(i_, 3)
    RG_EPI(0) RG_EPI(1) RG_EPI(2) RG_EPI(3) RG_EPI(4) RG_EPI(5)
#undef RG_EPI
#undef RG_EPI1
}

// ---------------------------------------------------------------------------------------------------
static int ring_mode() {
    // (read per launch: the tests switch it within one process) 0 = off, 1 = the 4-wave form (default), 2 = the 8-wave form, which
    // measured SLOWER on MI355X (block 1's layers 4.06 -> 4.33 ms per pass, block 0's 2.87 -> 2.93): one workgroup per CU moves in
    // lock-step again, and neither the deeper ring nor the halved weight traffic makes up for it; 3 = the 4-wave form through the
    // general kernel (conv_bf16_ring_kernel<4,7,0,4>: counted waits through a switch and per-lane weight offsets cost it 9 % on these
    // layers: 4.04 -> 4.41 and 2.90 -> 3.19 ms)
    const char *e = getenv("HIPVOC_BF16_RING");
    return e ? atoi(e) : 1;
}
static bool ring_tconv_enabled() {
    const char *e = getenv("HIPVOC_BF16_RING_TCONV");      // the transposed convs on the ring kernel (0: conv_gemm_bf16_oi / _pt_kernel)
    return e ? atoi(e) != 0 : true;
}

bool conv7_bf16_ring_selected(const ConvGemmArgs &a) {
    if (ring_mode() <= 0) return false;
    if (!a.wb || !a.xb || a.wb2 || a.wbm || a.sn_a || a.S != 1 || a.ntaps != ring::NTAPS || a.in_elu || a.s2d > 1) return false;
    if (a.shift_step < 1 || a.shift0 != -6 * a.shift_step || -a.shift0 > 64) return false;
    if ((a.Cin % 32) || a.NP % 128 || a.TN != 4 || a.units < 1 || a.M < 1 || a.Lin < a.M) return false;
    if ((long)a.x_cs * 16 * 5 >= 0x7fffffffL) return false;          // 32-bit byte offsets of a lane inside a channel block
    return true;
}

// the decoder blocks' transposed convs (kernel = 2 stride: two taps) with octet-interleaved (pm 2, tile height 4) or phase-tile
// rows (pm 3, tile height = stride 3 / 5): what conv_gemm_bf16_oi_kernel / _pt_kernel run, and their host checks
static bool tconv_ring_selected(const ConvGemmArgs &a) {
    if (ring_mode() <= 0 || !ring_tconv_enabled()) return false;
    if (!a.wb || !a.xb || a.wb2 || a.wbm || a.sn_a || a.S < 2 || a.ntaps != 2 || a.shift_step != 1 || a.in_elu || a.s2d > 1) return false;
    if (a.shift0 < -1 || a.shift0 > 0 || a.res || a.rscale || a.act != ACT_NONE || (a.Cin % 32) || a.units < 1 || a.M < 1 || a.Lin < a.M) return false;
    if ((long)a.x_cs * 16 * 5 >= 0x7fffffffL || a.N % a.S || a.NP < a.N) return false;
    const int tn = detail::conv_gemm_launch_tn(a);
    if (a.pm == 2) return tn == 4 && (a.S == 4 || a.S == 8) && a.N % (8 * a.S) == 0 && a.NP % 128 == 0 && !(a.y_cs & 1) && !(a.y_us & 1);
    // (measured on MI355X, per pass: stride 8 0.80 -> 0.73 ms, 5 1.34 -> 0.98, 4 1.37 -> 1.29)
    if (a.pm == 3) {
        if (tn != a.S || (a.N / a.S) % 32 != 0 || a.NP != a.N) return false;
        if (a.S == 5) return true;
        // stride 3 (block 3's: Cin = 192 is twelve stages in all, the launch is its epilogue): 1.80 ms on either kernel while it wrote
        // an fp32 stream; with the packed stream of DESIGN 7d 1.57 -> 1.45 ms here (HIPVOC_BF16_RING_TCONV3=0: the old kernel)
        const char *e3 = getenv("HIPVOC_BF16_RING_TCONV3");       // (read per launch)
        return a.S == 3 && !(e3 && atoi(e3) == 0);
    }
    return false;
}
static int ring_unit_mode() {
    const char *e = getenv("HIPVOC_BF16_RING_UNIT");       // (read per launch) 0: conv_gemm_bf16_kernel<6,true,true>
    return e ? atoi(e) : 1;
}
// the fused residual unit of a 192-channel block (what conv_gemm_bf16_kernel<6,true,true> runs, and its host checks)
static bool resunit_ring_selected(const ConvGemmArgs &a) {
    if (ring_mode() <= 0 || ring_unit_mode() == 0) return false;
    if (!a.wb || !a.wb2 || !a.xb || a.wbm || a.sn_a || a.S != 1 || a.ntaps != ringu::NTAPS || a.in_elu || a.s2d > 1 || a.act != ACT_NONE || a.rscale || a.y2) return false;
    if (a.shift_step < 1 || a.shift0 != -6 * a.shift_step || -a.shift0 > 64) return false;
    if (a.N != 192 || a.NP != 192 || a.Cin != 192 || !a.res || !a.bias || !a.bias2 || !a.sn2_a || !a.sn2_ib || a.units < 1 || a.M < 1 || a.Lin < a.M) return false;
    if ((long)a.x_cs * 16 * 5 >= 0x7fffffffL) return false;
    return true;
}
// the 1x1 conv that closes an un-fused residual unit (what conv_gemm_bf16_kernel<4,false,true> runs), on large launches
static bool conv1_ring_selected(const ConvGemmArgs &a) {
    if (ring_mode() <= 0) return false;
    const char *e = getenv("HIPVOC_BF16_RING_C1");              // (read per launch) 0: conv_gemm_bf16_kernel<4,false,true>
    if (e && atoi(e) == 0) return false;
    if (!a.wb || a.wb2 || a.wbm || !a.xb || a.sn_a || a.S != 1 || a.ntaps != 1 || a.shift0 != 0 || a.in_elu || a.s2d > 1 || a.act != ACT_NONE || a.rscale || a.y2) return false;
    if (!a.res || !a.yb || !a.bias || a.N != a.NP || (a.N % 128) || (a.Cin % 32) || a.Cin < 32 || a.units < 1 || a.M < 1 || a.Lin < 1) return false;
    if (a.psn_a && !a.psn_ib) return false;
    if ((long)a.x_cs * 16 * 5 >= 0x7fffffffL) return false;
    // large launches only: below two workgroups per CU the small-launch / 8-wave forms keep the chip fuller
    const long wgs = (long)((a.M + ring1::BM - 1) / ring1::BM) * a.units * (a.N / 128);
    const char *t = getenv("HIPVOC_BF16_RING_C1_WGS");
    return wgs >= (t ? atol(t) : 1024);
}
bool conv_bf16_ring_selected(const ConvGemmArgs &a) {
    return conv7_bf16_ring_selected(a) || tconv_ring_selected(a) || resunit_ring_selected(a) || conv1_ring_selected(a);
}

const char *conv_bf16_ring_kernel_name(const ConvGemmArgs &a) {
    if (resunit_ring_selected(a)) return a.rbi == 2 ? "resunit_bf16_ring_kernel<6,true>" : "resunit_bf16_ring_kernel<6,false>";
    if (tconv_ring_selected(a)) return a.pm == 2 ? "conv_bf16_ring_kernel<4,2,2,4>" : (a.S == 3 ? "conv_bf16_ring_kernel<3,2,3,4>" : "conv_bf16_ring_kernel<5,2,3,3>");
    if (conv1_ring_selected(a)) return "conv1_bf16_ring_kernel<4>";
    const int m = ring_mode();
    return m == 1 ? "conv7_bf16_ring_kernel<4>" : (m == 2 ? "conv7_bf16_ring8_kernel<4,8,2>" : "conv_bf16_ring_kernel<4,7,0,4>");
}

template <int TN, int NTAPS, int EPI, int RW>
static hipError_t ring_launch_general(const ConvGemmArgs &a, int HL, hipStream_t s) {
    using namespace ring;
    const int tiles_per_unit = (a.M + BM - 1) / BM;
    const int n_tiles = a.NP / (32 * TN);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    constexpr size_t lds = 4 * 32 * TN * 4 + 2 * XBUF + (size_t)RW * 4 * 32 * TN * 16;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static std::atomic<bool> done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_bf16_ring_kernel<TN, NTAPS, EPI, RW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((conv_bf16_ring_kernel<TN, NTAPS, EPI, RW>), dim3((unsigned)grid), dim3(NTH), lds, s, a, HL, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_conv_bf16_ring(const ConvGemmArgs &a, hipStream_t s) {
    constexpr int TN = 4;
    // host-side shape checks: what the kernels' indexing and vector accesses assume
    if (((uintptr_t)a.wb & 15) || ((uintptr_t)a.xb & 15) || (a.yb && ((uintptr_t)a.yb & 15)) || (!a.y && !a.yb) || (a.yb && ((a.N / a.S) & 7)) ||
        (a.y2 && (!a.psn_a || !a.psn_ib || a.yb)) || a.NP < a.N)
        return hipErrorInvalidValue;
    if ((a.rbf == 2 || a.rbi == 2) && (((a.N / a.S) & 7) || (a.N % a.S) || (a.S > 1 && !a.pm) || a.y2)) return hipErrorInvalidValue;
    if (a.rbf == 2 && ((a.y && ((uintptr_t)a.y & 15)) || (a.y && !a.yb))) return hipErrorInvalidValue;
    if (a.rbi == 2 && a.res && ((uintptr_t)a.res & 15)) return hipErrorInvalidValue;
    if (resunit_ring_selected(a)) {
        using namespace ringu;
        if (((uintptr_t)a.wb2 & 15) || (a.yb && !a.psn_a) || (a.psn_a && !a.psn_ib)) return hipErrorInvalidValue;
        const int tiles_per_unit = (a.M + BM - 1) / BM;
        const long grid = (long)tiles_per_unit * a.units;
        if (grid > 0x7fffffffL) return hipErrorInvalidValue;
        constexpr size_t lds = 6 * 192 * 4 + 2 * XBUF + (size_t)RW * 4 * 192 * 16;
        static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
        static std::atomic<bool> doneu[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !doneu[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(resunit_bf16_ring_kernel<6, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(resunit_bf16_ring_kernel<6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            doneu[dev] = true;
        }
        ConvGemmArgs b = a;
        {
            const char *le = getenv("HIPVOC_BF16_RING_EPI");        // (read per launch) 0: the generic fused epilogue
            b.dbg = (le && atoi(le) == 0) ? 1 : 0;
        }
        if (a.rbi == 2) hipLaunchKernelGGL((resunit_bf16_ring_kernel<6, true>), dim3((unsigned)grid), dim3(NTH), lds, s, b, tiles_per_unit);
        else hipLaunchKernelGGL((resunit_bf16_ring_kernel<6, false>), dim3((unsigned)grid), dim3(NTH), lds, s, a, tiles_per_unit);
        return hipGetLastError();
    }
    if (tconv_ring_selected(a)) {
        const int HL = a.shift0 < 0 ? -a.shift0 : 0;                   // taps at q + shift0, q + shift0 + 1: one position of halo, left or right
        if (a.pm == 2) {
            if ((a.y && ((uintptr_t)a.y & 7)) || (a.y2 && ((uintptr_t)a.y2 & 7))) return hipErrorInvalidValue;
            return ring_launch_general<4, 2, 2, 4>(a, HL, s);
        }
        if (a.S == 3) return ring_launch_general<3, 2, 3, 4>(a, HL, s);
        return ring_launch_general<5, 2, 3, 3>(a, HL, s);
    }
    if (conv1_ring_selected(a)) {
        using namespace ring1;
        if (a.Lin < a.M || ((uintptr_t)a.res & 15) || (a.y && ((uintptr_t)a.y & 15))) return hipErrorInvalidValue;
        const int tiles_per_unit = (a.M + BM - 1) / BM;
        const int n_tiles = a.NP / (32 * TN);
        const long grid = (long)tiles_per_unit * a.units * n_tiles;
        if (grid > 0x7fffffffL) return hipErrorInvalidValue;
        constexpr size_t lds = 4 * 32 * TN * 4 + R * XBUF + (size_t)R * 4 * 32 * TN * 16;
        static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
        static std::atomic<bool> done1[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !done1[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_bf16_ring_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            done1[dev] = true;
        }
        hipLaunchKernelGGL(conv1_bf16_ring_kernel<TN>, dim3((unsigned)grid), dim3(NTH), lds, s, a, tiles_per_unit, n_tiles);
        return hipGetLastError();
    }
    if (!conv7_bf16_ring_selected(a)) return hipErrorInvalidValue;
    const int mode = ring_mode();
    if (mode >= 3) return ring_launch_general<4, 7, 0, 4>(a, -a.shift0, s);
    const bool w8 = mode == 2;
    const int bm = w8 ? ring8::BM : ring::BM;
    const int tiles_per_unit = (a.M + bm - 1) / bm;
    const int n_tiles = a.NP / (32 * TN);
    const long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    constexpr size_t lds4 = 4 * 32 * TN * 4 + 2 * ring::XBUF + ring::RW * 4 * 32 * TN * 16;
    constexpr size_t lds8 = 4 * 32 * TN * 4 + 2 * ring8::XBUF + ring8::RW * 4 * 32 * TN * 16;
    static_assert(2 * lds4 <= 160 * 1024 && lds8 <= 160 * 1024, "two 4-wave workgroups, or one 8-wave workgroup, per CU");
    static std::atomic<bool> done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_bf16_ring_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_bf16_ring8_kernel<TN, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    ConvGemmArgs b = a;
    {
        const char *le = getenv("HIPVOC_BF16_RING_EPI");            // (read per launch) 0: the generic epilogue in conv7_bf16_ring_kernel
        b.dbg = (le && atoi(le) == 0) ? 1 : 0;
    }
    if (w8) hipLaunchKernelGGL((conv7_bf16_ring8_kernel<TN, 8, 2>), dim3((unsigned)grid), dim3(512), lds8, s, a, tiles_per_unit, n_tiles);
    else hipLaunchKernelGGL(conv7_bf16_ring_kernel<TN>, dim3((unsigned)grid), dim3(ring::NTH), lds4, s, b, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

}  // namespace hipvoc
