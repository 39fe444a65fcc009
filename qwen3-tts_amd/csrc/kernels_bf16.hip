// gfx950 (CDNA4 / MI355X) bf16 conv-GEMM mode of the vocoder path (HIPVOC_DTYPE_BF16, BASELINE configs[4]).
//
// Same orientation and tiling as the fp32 kernel (kernels.hip): weights are the A operand, activations the
// B operand (time on the lane), 8 waves x 32 positions per workgroup, accumulators in fp32.  What changes:
//   * v_mfma_f32_32x32x16_bf16: each lane supplies EIGHT consecutive k of its row / column, so both LDS
//     tiles are laid out as [k/8][row or position][8 bf16]: one ds_read_b128 per operand, and a conv tap is
//     still a shift along the position axis of the same activation tile;
//   * the activation tile is built from the channel-first fp32 tensors in HBM: a thread gathers the eight
//     channels of one (k-group, position), applies SnakeBeta in fp32, rounds to bf16 (RNE) and writes one
//     16-byte LDS element; everything in HBM stays fp32 (residual stream, biases, epilogues);
//   * weights are rounded to bf16 once at load and stored in exactly the LDS tile order;
//   * fused residual unit: the fp32 accumulator becomes the second product's B operand by a pure
//     register-local pack (a lane's registers 8g..8g+7 are its eight k of MFMA g); the k order this implies
//     is baked into the packing of the 1x1 conv's weights on the host (hipvoc.cpp: pack_w2_bf16).
#include "device_common.h"
#include "kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#ifndef HIPVOC_ABLATE
#define HIPVOC_ABLATE 0
#endif
#define ABL(bit_) (HIPVOC_ABLATE && (a.dbg & (bit_)))   // timing-only ablations, see kernels.hip

static constexpr int BMB = 256;   // positions per workgroup
static constexpr int BKB = 32;    // input channels per K block (two MFMA k-steps)
static constexpr int NTHB = 512;  // 8 waves

__device__ __forceinline__ bf16x8 pack8(const f32x8 &v) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (__bf16)v[j];
    return o;
}

// XB: the input is a packed bf16 tensor (ConvGemmArgs::xb, activation already applied by its producer): staging a
// (k-group, position) pair is ONE 16-byte load and one 16-byte LDS store -- no gather, no SnakeBeta, no conversion.
// X3 ("bf16x3", HIPVOC_DTYPE_BF16X3): every operand is split into two bf16 planes, v = hi + mid with hi = bf16(v) and
// mid = bf16(v - hi) (16 mantissa bits together), and a product is three MFMAs: hi*hi + hi*mid + mid*hi (the dropped
// mid*mid term is 2^-16 of the product).  fp32-grade results at a third of the bf16 MFMA rate -- still several
// times the fp32 MFMA rate.  Weights are split once at load (ConvGemmArgs::wbm / wb2m), activations at commit.
// NW: waves per workgroup (32 positions each).  8 by default; 4 (HIPVOC_BF16_W4=1) for the packed fused residual units: two
// INDEPENDENT workgroups per CU instead of one, the move that paid off for the fp32 kernels -- here it does not
// (see bf16_w4 below).  Same MFMA chain per output element either way.
template <int TN, bool FUSE, bool XB, bool X3, int OI = 0, int NW = 8>      // OI: 0, or ConvGemmArgs::pm of a transposed conv with its own epilogue (2, 3)
__device__ __forceinline__ void conv_gemm_bf16_body(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    constexpr int BMB = 32 * NW, NTHB = 64 * NW;      // (shadow the 8-wave constants of this file)
    constexpr int BN = TN * 32;
    constexpr bool FS = !X3;                           // the bf16 mode's hardware SnakeBeta (device_common.h: snake_beta2_hw); bf16x3 keeps the polynomial
    constexpr int WF = 4 * BN;                        // 16-byte elements per W tile [4 k-groups][BN rows]
    constexpr int WPT = (WF + NTHB - 1) / NTHB;       // per thread: 1 (TN <= 4) or 2 (TN = 6); 3 for TN = 6 in 4 waves
    static_assert(WPT <= 3, "W staging registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    constexpr int PMF = (FUSE ? 6 : 4) * BN;                     // per-row parameter table (device_common.h), first in LDS
    float *Pm = reinterpret_cast<float *>(smem_all);
    unsigned char *smem_raw = smem_all + PMF * sizeof(float);
    bf16x8 *Xs = reinterpret_cast<bf16x8 *>(smem_raw);          // [2][4][XW]
    bf16x8 *Xm = Xs + 2 * 4 * XW;                                // X3: the mid plane, same shape
    bf16x8 *Ws = Xs + (X3 ? 4 : 2) * 4 * XW;                     // [2][4][BN]
    bf16x8 *Wm = Ws + 2 * WF;                                    // X3: the mid plane
    float *Sn = reinterpret_cast<float *>(Ws + (X3 ? 4 : 2) * WF);   // [2][CinP]: exp(alpha) | 1/(exp(beta)+1e-9)

    int nwg = gridDim.x, orig = blockIdx.x;
    int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int nt = id % n_tiles;
    int pt = id / n_tiles;
    int unit = pt / tiles_per_unit;
    int m0 = (pt - unit * tiles_per_unit) * BMB;
    int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;

    const float *xu = a.x + (long)unit * a.x_us;
    const int ncb = (a.Cin + BKB - 1) / BKB;
    const int CinP = ncb * BKB;
    const int nstage = ncb * a.ntaps;

    fill_param_table<BN, FUSE>(a, Pm, n0, tid, NTHB);           // read after the main loop (barriers in between)
    if (!XB && a.sn_a) {
        for (int i = tid; i < CinP; i += NTHB) {
            Sn[i] = i < a.Cin ? a.sn_a[i] : 0.0f;
            Sn[CinP + i] = i < a.Cin ? a.sn_ib[i] : 0.0f;
        }
        __syncthreads();
    }

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // ---- W staging: the packed tensor is [tap][cb][4 k-groups][NP rows] of 16-byte elements
    const uint4 *wg = reinterpret_cast<const uint4 *>(a.wb);
    const uint4 *wgm = reinterpret_cast<const uint4 *>(a.wbm);
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2r = w0, wm0 = w0, wm1 = w0;
    const int wq0 = tid / BN, wr0 = tid - wq0 * BN;
    const int wq1 = (tid + NTHB) / BN, wr1 = tid + NTHB - wq1 * BN;
    const int wq2 = (tid + 2 * NTHB) / BN, wr2 = tid + 2 * NTHB - wq2 * BN;
    static_assert(!(X3 && WPT > 2), "the split-operand mode runs in 8-wave workgroups");
#define W_LOAD(stage_)                                                                          \
    {                                                                                           \
        int cb_ = (stage_) / a.ntaps, tap_ = (stage_)-cb_ * a.ntaps;                            \
        const uint4 *wp_ = wg + ((long)(tap_ * ncb + cb_) * 4) * a.NP + n0;                     \
        if (WF >= NTHB || tid < WF) w0 = wp_[(long)wq0 * a.NP + wr0];                           \
        if constexpr (WPT > 1) if (WF >= 2 * NTHB || tid + NTHB < WF) w1 = wp_[(long)wq1 * a.NP + wr1];           \
        if constexpr (WPT > 2) if (tid + 2 * NTHB < WF) w2r = wp_[(long)wq2 * a.NP + wr2];      \
        if constexpr (X3) {                                                                     \
            const uint4 *wpm_ = wgm + ((long)(tap_ * ncb + cb_) * 4) * a.NP + n0;               \
            if (WF >= NTHB || tid < WF) wm0 = wpm_[(long)wq0 * a.NP + wr0];                     \
            if constexpr (WPT > 1) if (tid + NTHB < WF) wm1 = wpm_[(long)wq1 * a.NP + wr1];     \
        }                                                                                       \
    }
#define W_STORE(buf_)                                                                           \
    {                                                                                           \
        uint4 *ws_ = reinterpret_cast<uint4 *>(Ws) + (buf_)*WF + tid;                           \
        if (WF >= NTHB || tid < WF) *ws_ = w0;                                                  \
        if constexpr (WPT > 1) if (WF >= 2 * NTHB || tid + NTHB < WF) ws_[NTHB] = w1;           \
        if constexpr (WPT > 2) if (tid + 2 * NTHB < WF) ws_[2 * NTHB] = w2r;                    \
        if constexpr (X3) {                                                                     \
            uint4 *wsm_ = reinterpret_cast<uint4 *>(Wm) + (buf_)*WF + tid;                      \
            if (WF >= NTHB || tid < WF) *wsm_ = wm0;                                            \
            if constexpr (WPT > 1) if (tid + NTHB < WF) wsm_[NTHB] = wm1;                       \
        }                                                                                       \
    }

    // ---- X staging: (k-group, position) pairs p = tid + 512*round, k-group = p / XW, position = p % XW; the LDS
    // element index of a pair is p itself.  Loop-invariant per-thread state is kept to four registers: one 32-bit
    // global offset per round (k-group row + clamped position) and one word of packed flags
    // (per round: k-group [2 bits], pair exists [1], position inside [0, Lin) [1]).
    f32x8 xa, xb, xc;
    unsigned offa, offb, offc, xmeta = 0;
    {
        auto prep = [&](int round, unsigned &off) {
            int p = tid + round * NTHB;
            int kq = p / XW, t = p - kq * XW;
            int pos = m0 - HL + t;
            bool ok = pos >= 0 && pos < a.Lin;
            pos = pos < 0 ? 0 : (pos >= a.Lin ? a.Lin - 1 : pos);
            bool ex = kq < 4;
            off = (unsigned)((ex ? kq : 0) * (XB ? 1 : 8)) * (unsigned)a.x_cs + (unsigned)pos;
            xmeta |= (unsigned)((ex ? kq : 0) | (ex ? 4 : 0) | (ok ? 8 : 0)) << (4 * round);
        };
        prep(0, offa); prep(1, offb); prep(2, offc);
    }
    // eight unconditional loads from clamped (always valid) addresses; zeroing happens at commit (see kernels.hip).
    // The channel block and the channel within the group are wave-uniform, so a load is uniform base + lane
    // offset: no per-element 64-bit address arithmetic in the loop.  (Register pressure matters doubly here: a
    // spill reload inside the loop counts on vmcnt, which retires in order, so it drains the whole prefetch.)
    auto x_fetch1 = [&](int cb, int round, unsigned off) -> f32x8 {
        f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned m = xmeta >> (4 * round);
        if (m & 4) {      // rounds past the tile (whole waves, mostly) fetch nothing
            // buffer loads: uniform descriptor (this channel block's base) + lane offset in ONE register + the
            // channel's row offset in an SGPR: no 64-bit per-row address pairs (16 VGPRs with plain loads).  The
            // descriptor ends after the last real channel, so rows >= Cin of a partial last block read as 0.
            const long rows_left = a.Cin - cb * BKB;
            const long bytes = rows_left * a.x_cs * 4;
            // (every input is wave-uniform, but the 64-bit address arithmetic runs on the VALU: without the explicit
            // readfirstlane hipcc wraps each load in a waterfall loop)
            const unsigned long long ba = reinterpret_cast<unsigned long long>(xu + (long)cb * BKB * a.x_cs);
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)ba), bhi = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32));
            const int nrec = __builtin_amdgcn_readfirstlane(bytes < 0x7fffffffL ? (int)bytes : 0x7fffffff);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<float *>(((unsigned long long)bhi << 32) | blo), 0, nrec, 0x00020000);
            const unsigned row = __builtin_amdgcn_readfirstlane((unsigned)a.x_cs * 4u);
#pragma unroll
            for (int j = 0; j < 8; j++)
                v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off * 4u, (unsigned)j * row, 0));
        }
        return v;
    };
    auto x_commit1 = [&](int cb, int buf, int round, f32x8 v) {
        const unsigned m = xmeta >> (4 * round);
        if (!(m & 4)) return;
        const int kq = m & 3, c0 = cb * BKB + kq * 8;
        const bool ok = (m & 8) != 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = ok ? v[j] : 0.0f;     // (channels >= Cin already read as 0)
        if (a.sn_a) {
            const float4 *sa = reinterpret_cast<const float4 *>(Sn + c0);
            const float4 *sb = reinterpret_cast<const float4 *>(Sn + CinP + c0);
            float4 a0 = sa[0], a1 = sa[1], b0 = sb[0], b1 = sb[1];
            f32x2 p0 = snake_beta2_t<FS>(mk2(v[0], v[1]), mk2(a0.x, a0.y), mk2(b0.x, b0.y));
            f32x2 p1 = snake_beta2_t<FS>(mk2(v[2], v[3]), mk2(a0.z, a0.w), mk2(b0.z, b0.w));
            f32x2 p2 = snake_beta2_t<FS>(mk2(v[4], v[5]), mk2(a1.x, a1.y), mk2(b1.x, b1.y));
            f32x2 p3 = snake_beta2_t<FS>(mk2(v[6], v[7]), mk2(a1.z, a1.w), mk2(b1.z, b1.w));
            v[0] = p0.x; v[1] = p0.y; v[2] = p1.x; v[3] = p1.y; v[4] = p2.x; v[5] = p2.y; v[6] = p3.x; v[7] = p3.y;
        }
        const bf16x8 hi = pack8(v);
        Xs[buf * 4 * XW + tid + round * NTHB] = hi;
        if constexpr (X3) {
            f32x8 r;
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = v[j] - (float)hi[j];
            Xm[buf * 4 * XW + tid + round * NTHB] = pack8(r);
        }
    };
    // packed input: element (k-group cb*4+kq, unit, pos) of a.xb; the descriptor ends after the last real group
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa, qc = qa;
    const int ngroups = (a.Cin + 7) >> 3;
    auto xq_fetch1 = [&](int cb, int round, unsigned off) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        const unsigned m = xmeta >> (4 * round);
        if (m & 4) {
            const long groups_left = ngroups - cb * 4;
            const long bytes = groups_left * a.x_cs * 16;
            const unsigned long long ba = reinterpret_cast<unsigned long long>(a.xb) + ((unsigned long long)((long)unit * a.x_us + (long)cb * 4 * a.x_cs) << 4);
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)ba), bhi = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32));
            const int nrec = __builtin_amdgcn_readfirstlane(bytes < 0x7fffffffL ? (int)bytes : 0x7fffffff);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(((unsigned long long)bhi << 32) | blo), 0, nrec, 0x00020000);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, off * 16u, 0, 0);
            v = make_uint4(r[0], r[1], r[2], r[3]);
        }
        return v;
    };
    auto xq_commit1 = [&](int buf, int round, uint4 v) {
        const unsigned m = xmeta >> (4 * round);
        if (!(m & 4)) return;
        if (!(m & 8)) v = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(Xs)[buf * 4 * XW + tid + round * NTHB] = v;
    };
#define X_FETCH(cb_)                                                                                              \
    {                                                                                                             \
        if constexpr (XB) { qa = xq_fetch1(cb_, 0, offa); qb = xq_fetch1(cb_, 1, offb); qc = xq_fetch1(cb_, 2, offc); } \
        else { xa = x_fetch1(cb_, 0, offa); xb = x_fetch1(cb_, 1, offb); xc = x_fetch1(cb_, 2, offc); }             \
    }
#define X_COMMIT(cb_, buf_)                                                                                       \
    {                                                                                                             \
        if constexpr (XB) { xq_commit1(buf_, 0, qa); xq_commit1(buf_, 1, qb); xq_commit1(buf_, 2, qc); }            \
        else { x_commit1(cb_, buf_, 0, xa); x_commit1(cb_, buf_, 1, xb); x_commit1(cb_, buf_, 2, xc); }             \
    }

    X_FETCH(0)
    W_LOAD(0);
    X_COMMIT(0, 0)
    W_STORE(0);
    for (int cb = 0; cb < ncb; cb++) {
        const bool xnext = cb + 1 < ncb && !ABL(1);
        if (xnext) X_FETCH(cb + 1)
        for (int tap = 0; tap < a.ntaps; tap++) {
            const int st = cb * a.ntaps + tap;
            // unconditional (the last stage re-loads itself into the idle buffer): with the load and the store under
            // a condition hipcc cannot pair them and waits vmcnt(0) BEFORE every load, draining the activation prefetch
            if (!ABL(2)) W_LOAD(st + 1 < nstage ? st + 1 : st);
            if (!ABL(4)) __syncthreads();
            {
                const bf16x8 *wsb = Ws + (st & 1) * WF + l31;
                const bf16x8 *xsb = Xs + (cb & 1) * 4 * XW + wave * 32 + l31 + HL + a.shift0 + tap * a.shift_step;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int kq = ks * 2 + h;
                    bf16x8 bv = xsb[kq * XW];
                    bf16x8 av[TN];
#pragma unroll
                    for (int i = 0; i < TN; i++) av[i] = wsb[kq * BN + i * 32];
                    if constexpr (X3) {
                        const bf16x8 bm = (xsb + (Xm - Xs))[kq * XW];
                        bf16x8 am[TN];
#pragma unroll
                        for (int i = 0; i < TN; i++) am[i] = (wsb + (Wm - Ws))[kq * BN + i * 32];
                        // small terms first; one term over all TN tiles at a time, so that consecutive MFMAs never
                        // wait on each other's accumulator
#pragma unroll
                        for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bv, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bm, acc[i], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv, acc[i], 0, 0, 0);
                }
            }
            if (!ABL(2)) W_STORE((st + 1) & 1);
        }
        if (xnext) X_COMMIT(cb + 1, (cb + 1) & 1)
    }
#undef X_FETCH
#undef X_COMMIT
#undef W_LOAD
#undef W_STORE

    if (ABL(8)) {   // timing-only: main loop alone
        if (ABL(64)) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) t += acc[i][r];
            a.y[tid] = t;
        }
        return;
    }
    if constexpr (FUSE) {
        // H = snake2(acc + b1) -> bf16 B operands hb[i][g] (lane's registers 8g..8g+7 of tile i)
        constexpr int W2N = 128 * TN;     // 16-byte elements per W2 tile: [2*TN k-steps][2 lane halves][32 rows]
        const uint4 *w2g = reinterpret_cast<const uint4 *>(launder_after(a.wb2, acc[0][15]));
        if (tid < W2N) w0 = w2g[tid];
        if constexpr (W2N > NTHB) if (tid + NTHB < W2N) w1 = w2g[tid + NTHB];
        if constexpr (W2N > 2 * NTHB) if (tid + 2 * NTHB < W2N) w2r = w2g[tid + 2 * NTHB];
        static_assert(W2N <= 3 * NTHB, "W2 staging registers");
        if constexpr (X3) {
            const uint4 *w2gm = reinterpret_cast<const uint4 *>(launder_after(a.wb2m, acc[0][15]));
            if (tid < W2N) wm0 = w2gm[tid];
            if constexpr (W2N > NTHB) if (tid + NTHB < W2N) wm1 = w2gm[tid + NTHB];
        }
        bf16x8 hb[TN][2];
        bf16x8 hm[X3 ? TN : 1][2];
#pragma unroll
        for (int i = 0; i < TN; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (r & 1) {   // registers r-1, r are channels n-1, n: one packed evaluation for the pair
                    f32x2 hv = snake_beta2_t<FS>(mk2(acc[i][r - 1] + Pm[n - 1], acc[i][r] + Pm[n]), mk2(Pm[BN + n - 1], Pm[BN + n]),
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
                for (int j = 0; j < 8; j++) {
                    hb[i][g][j] = (__bf16)acc[i][8 * g + j];
                    if constexpr (X3) hm[i][g][j] = (__bf16)(acc[i][8 * g + j] - (float)hb[i][g][j]);
                }
        }
        float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
        unsigned short *ybu = a.yb ? a.yb + (((long)unit * a.y_us) << 3) : nullptr;
        const int m = m0 + wave * 32 + l31;
        uint4 *w2s = reinterpret_cast<uint4 *>(smem_raw);
        uint4 *w2sm = w2s + 2 * W2N;              // X3: the mid planes of the two tile buffers
        __syncthreads();                          // every wave is done with the main loop's tiles
        if (tid < W2N) w2s[tid] = w0;
        if constexpr (W2N > NTHB) if (tid + NTHB < W2N) w2s[tid + NTHB] = w1;
        if constexpr (W2N > 2 * NTHB) if (tid + 2 * NTHB < W2N) w2s[tid + 2 * NTHB] = w2r;
        if constexpr (X3) {
            if (tid < W2N) w2sm[tid] = wm0;
            if constexpr (W2N > NTHB) if (tid + NTHB < W2N) w2sm[tid + NTHB] = wm1;
        }
        float dep2 = acc[TN - 1][15];
#pragma unroll
        for (int o = 0; o < TN; o++) {
            if (o + 1 < TN) {
                const uint4 *wq_ = reinterpret_cast<const uint4 *>(launder_after(a.wb2, dep2)) + (o + 1) * W2N;
                if (tid < W2N) w0 = wq_[tid];
                if constexpr (W2N > NTHB) if (tid + NTHB < W2N) w1 = wq_[tid + NTHB];
                if constexpr (W2N > 2 * NTHB) if (tid + 2 * NTHB < W2N) w2r = wq_[tid + 2 * NTHB];
                if constexpr (X3) {
                    const uint4 *wqm_ = reinterpret_cast<const uint4 *>(launder_after(a.wb2m, dep2)) + (o + 1) * W2N;
                    if (tid < W2N) wm0 = wqm_[tid];
                    if constexpr (W2N > NTHB) if (tid + NTHB < W2N) wm1 = wqm_[tid + NTHB];
                }
            }
            __syncthreads();
            float resv[16];   // this tile's residual, in flight during its product
            ResPacked resq;   // ... from the packed bf16 stream (ConvGemmArgs::rbi == 2)
            {
                const float *rup = launder_after(a.res, dep2) + (long)unit * a.y_us;
                if (a.rbi == 2) {
                    fused_fetch_res_packed(a, rup, o * 32 + 4 * h, m, m < a.M, resq);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        int n = o * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        resv[r] = (n < a.N && m < a.M) ? rup[(long)n * a.y_cs + m] : 0.0f;
                    }
                }
            }
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[r] = 0.0f;
            const bf16x8 *ws = reinterpret_cast<const bf16x8 *>(smem_raw) + (o & 1) * W2N + h * 32 + l31;
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    if constexpr (X3) {
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ws + 2 * W2N)[(i * 2 + g) * 64], hb[i][g], acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[(i * 2 + g) * 64], hm[i][g], acc2, 0, 0, 0);
                    }
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[(i * 2 + g) * 64], hb[i][g], acc2, 0, 0, 0);
                }
            if (o + 1 < TN) {
                uint4 *wd_ = w2s + ((o + 1) & 1) * W2N;
                if (tid < W2N) wd_[tid] = w0;
                if constexpr (W2N > NTHB) if (tid + NTHB < W2N) wd_[tid + NTHB] = w1;
                if constexpr (W2N > 2 * NTHB) if (tid + 2 * NTHB < W2N) wd_[tid + 2 * NTHB] = w2r;
                if constexpr (X3) {
                    uint4 *wdm_ = w2sm + ((o + 1) & 1) * W2N;
                    if (tid < W2N) wdm_[tid] = wm0;
                    if constexpr (W2N > NTHB) if (tid + NTHB < W2N) wdm_[tid + NTHB] = wm1;
                }
            }
            dep2 = acc2[15];
            if (a.rbi == 2) fused_unpack_res(resq, resv);
            fused_epilogue_tile<BN, FS>(a, acc2, resv, Pm, o * 32 + 4 * h, m, yu, ybu);
        }
        return;
    }

    // epilogue (fp32): bias, activation, per-channel scale, residual, optional SnakeBeta, store
    float *yu = a.y ? a.y + (long)unit * a.y_us : nullptr;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * a.y_us : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
    unsigned short *ybu0 = launder_after(a.yb, acc[TN - 1][15]);
    unsigned short *ybu = ybu0 ? ybu0 + (((long)unit * a.y_us) << 3) : nullptr;
    if constexpr (OI == 3) {
        tconv_pt_epilogue<BN, TN, FS>(a, acc, Pm, n0, h, m, yu, y2u, ybu);
        return;
    }
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, false, OI == 2, FS>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u, ybu);
}

template <int TN, bool FUSE, bool XB>
__global__ __launch_bounds__(NTHB, (TN > 4) ? 2 : 4) void conv_gemm_bf16_kernel(ConvGemmArgs a, int HL, int XW,
                                                                                 int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, FUSE, XB, false>(a, HL, XW, tiles_per_unit, n_tiles);
}
// the packed fused residual unit in 4-wave workgroups of 128 positions (two independent workgroups per CU)
template <int TN>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16_w4_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, true, true, false, 0, 4>(a, HL, XW, tiles_per_unit, n_tiles);
}
// the split-operand mode (fp32 exchange between kernels; one 8-wave workgroup per CU)
template <int TN, bool FUSE>
__global__ __launch_bounds__(NTHB, 2) void conv_gemm_bf16x3_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit,
                                                                   int n_tiles) {
    conv_gemm_bf16_body<TN, FUSE, false, true>(a, HL, XW, tiles_per_unit, n_tiles);
}

// the transposed convs with octet-interleaved rows (ConvGemmArgs::pm == 2; device_common.h: tconv_oi_epilogue_tile)
template <int TN, bool XB>
__global__ __launch_bounds__(NTHB, 4) void conv_gemm_bf16_oi_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, false, XB, false, 2>(a, HL, XW, tiles_per_unit, n_tiles);
}
template <int TN>
__global__ __launch_bounds__(NTHB, 2) void conv_gemm_bf16x3_oi_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, false, false, true, 2>(a, HL, XW, tiles_per_unit, n_tiles);
}

// ... and with phase-tile rows (pm == 3: strides 3 and 5; TN = stride; device_common.h: tconv_pt_epilogue)
template <int TN, bool XB>
__global__ __launch_bounds__(NTHB, (TN > 4) ? 2 : 4) void conv_gemm_bf16_pt_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, false, XB, false, 3>(a, HL, XW, tiles_per_unit, n_tiles);
}
template <int TN>
__global__ __launch_bounds__(NTHB, 2) void conv_gemm_bf16x3_pt_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit, int n_tiles) {
    conv_gemm_bf16_body<TN, false, false, true, 3>(a, HL, XW, tiles_per_unit, n_tiles);
}

// ---------------------------------------------------------------------------------------------------
// HIPVOC_BF16_W4=1: the packed fused residual units in 4-wave workgroups (conv_gemm_bf16_w4_kernel).  Built, bit-identical
// (tests/test_gpu_bf16.py) and measured SLOWER on MI355X -- block 2's units 6.94 -> 7.32 ms, block 3's 7.08 -> 7.45 ms per pass
// (profiles/r03_steps_bf16_w4_vs_w8.txt): twice the weight staging per position and a 43 % halo per MFMA outweigh the two
// independent workgroups per CU -- so the default stays 8 waves.
static bool bf16_w4(const ConvGemmArgs &a) {
    const char *e = getenv("HIPVOC_BF16_W4");      // (read per launch: the tests switch it within one process)
    const bool on = e ? (atoi(e) != 0) : false;
    return on && a.wb2 && a.xb && !a.wbm && (a.NP == 96 || a.NP == 192);
}

size_t conv_gemm_bf16_lds_bytes(const ConvGemmArgs &a) {
    int HL, XW;
    detail::conv_gemm_geometry(a, HL, XW, bf16_w4(a) ? 128 : 256);
    const int tn = detail::conv_gemm_launch_tn(a);
    const int CinP = (a.Cin + BKB - 1) / BKB * BKB;
    const size_t planes = a.wbm ? 2 : 1;
    size_t b = planes * 16 * (2 * 4 * (size_t)XW + 2 * 4 * (size_t)32 * tn) + (a.sn_a ? 2 * sizeof(float) * CinP : 0);
    if (a.wb2) b = b > planes * 2 * 16 * (size_t)128 * tn ? b : planes * 2 * 16 * (size_t)128 * tn;
    b += sizeof(float) * (a.wb2 ? 6 : 4) * 32 * tn;     // per-row parameter table
    return b;
}

const char *conv_gemm_bf16_kernel_name(const ConvGemmArgs &a) {
    if (gemm1_bf16_selected(a)) return gemm1_bf16_kernel_name(a);
    if (convs_bf16_selected(a)) return convs_bf16_kernel_name(a);
    if (conv_bf16_ring_selected(a)) return conv_bf16_ring_kernel_name(a);
    static char names[2][2][8][56];
    int tn = detail::conv_gemm_launch_tn(a);
    int fuse = a.wb2 ? 1 : 0, xbf = a.xb ? 1 : 0;
    if (a.S > 1 && a.pm == 3) {
        static char np[3][8][48];
        char *m = np[a.wbm ? 2 : xbf][tn & 7];
        if (!m[0]) {
            if (a.wbm) snprintf(m, 48, "conv_gemm_bf16x3_pt_kernel<%d>", tn);
            else snprintf(m, 48, "conv_gemm_bf16_pt_kernel<%d,%s>", tn, xbf ? "true" : "false");
        }
        return m;
    }
    if (a.S > 1 && a.pm == 2) {
        static char no[3][8][48];
        char *m = no[a.wbm ? 2 : xbf][tn & 7];
        if (!m[0]) {
            if (a.wbm) snprintf(m, 48, "conv_gemm_bf16x3_oi_kernel<%d>", tn);
            else snprintf(m, 48, "conv_gemm_bf16_oi_kernel<%d,%s>", tn, xbf ? "true" : "false");
        }
        return m;
    }
    if (bf16_w4(a)) return tn == 3 ? "conv_gemm_bf16_w4_kernel<3>" : "conv_gemm_bf16_w4_kernel<6>";
    char *n = names[fuse][xbf][tn & 7];
    if (a.wbm) {
        static char n3[2][8][48];
        char *m = n3[fuse][tn & 7];
        if (!m[0]) snprintf(m, 48, "conv_gemm_bf16x3_kernel<%d,%s>", tn, fuse ? "true" : "false");
        return m;
    }
    if (!n[0]) snprintf(n, 56, "conv_gemm_bf16_kernel<%d,%s,%s>", tn, fuse ? "true" : "false", xbf ? "true" : "false");
    return n;
}

template <int TN, bool FUSE, bool XB>
static hipError_t launch_one_bf16(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid,
                                  size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_bf16_kernel<TN, FUSE, XB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((conv_gemm_bf16_kernel<TN, FUSE, XB>), dim3((unsigned)grid), dim3(NTHB), lds, s, a, HL, XW,
                       tiles_per_unit, n_tiles);
    return hipGetLastError();
}

// (any kernel of this file's signature: the one-off raise of the dynamic LDS limit per device, then the launch)
template <auto KERN>
static hipError_t launch_fn_bf16(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL(KERN, dim3((unsigned)grid), dim3(NTHB), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
    return hipGetLastError();
}

template <int TN, bool FUSE>
static hipError_t launch_one_bf16x3(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles, long grid,
                                    size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_bf16x3_kernel<TN, FUSE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((conv_gemm_bf16x3_kernel<TN, FUSE>), dim3((unsigned)grid), dim3(NTHB), lds, s, a, HL, XW,
                       tiles_per_unit, n_tiles);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_bf16(const ConvGemmArgs &a_in, hipStream_t s) {
    if (gemm1_bf16_selected(a_in)) return launch_gemm1_bf16(a_in, s);
    if (convs_bf16_selected(a_in)) return launch_convs_bf16(a_in, s);
    if (conv_bf16_ring_selected(a_in)) return launch_conv_bf16_ring(a_in, s);
    ConvGemmArgs a = a_in;
    {
        static int dbg = -1;
        if (dbg < 0) { const char *e = getenv("HIPVOC_DBG"); dbg = e ? atoi(e) : 0; }
        a.dbg = dbg;
    }
    int HL, XW;
    const bool w4 = bf16_w4(a);
    const int nth = w4 ? 256 : NTHB, bm = w4 ? 128 : BMB;
    detail::conv_gemm_geometry(a, HL, XW, bm);
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    // host-side shape checks: what the kernel's staging loops and vector accesses assume
    if (!a.wb || (a.NP % (32 * a.TN)) || a.NP < a.N || a.TN < 1 || (a.TN > 4 && !(a.pm == 3 && a.TN == 5)) || ((uintptr_t)a.wb & 15) || a.ntaps < 1 ||
        a.S < 1 || 4 * XW > 3 * nth ||
        (long)a.x_cs * 4 * (BKB + 1) >= 0x7fffffffL)   // 32-bit byte offsets of the buffer loads (a channel block + a row)
        return hipErrorInvalidValue;
    const int tn = detail::conv_gemm_launch_tn(a);
    if (a.y2 && (!a.psn_a || !a.psn_ib || a.wb2 || a.yb)) return hipErrorInvalidValue;
    if (a.xb && ((a.Cin & 7) || a.sn_a || ((uintptr_t)a.xb & 15) || (long)a.x_cs * 16 * 5 >= 0x7fffffffL)) return hipErrorInvalidValue;
    if (a.yb && (((a.N / a.S) & 7) || (a.N % a.S) || ((uintptr_t)a.yb & 15))) return hipErrorInvalidValue;
    if (!a.y && !a.yb) return hipErrorInvalidValue;
    // packed bf16 residual stream behind y / res: grouped rows, whole channel groups, 16-byte aligned, packed mode only
    if ((a.rbf == 2 || a.rbi == 2) && (a.wbm || ((a.N / a.S) & 7) || (a.N % a.S) || (a.S > 1 && !a.pm) || a.y2 || a.s2d > 1)) return hipErrorInvalidValue;
    if (a.rbf == 2 && ((a.y && ((uintptr_t)a.y & 15)) || (a.y && !a.yb))) return hipErrorInvalidValue;
    if (a.rbi == 2 && a.res && ((uintptr_t)a.res & 15)) return hipErrorInvalidValue;
    // octet-interleaved transposed-conv rows (device_common.h: tconv_oi_epilogue_tile): what that epilogue assumes
    if (a.S > 1 && a.pm == 2 && ((a.S != 4 && a.S != 8) || a.N % (8 * a.S) || a.res || a.rscale || a.act != ACT_NONE || (a.y_cs & 1) || (a.y_us & 1) ||
                                 (a.y && ((uintptr_t)a.y & 7)) || (a.y2 && ((uintptr_t)a.y2 & 7)) || (!a.y && !a.yb)))
        return hipErrorInvalidValue;
    if (a.wb2 && (a.S != 1 || a.NP != 32 * tn || (a.N + 31) / 32 != tn || tn == 5 || tn > 6 || !a.res || !a.sn2_a ||
                  !a.sn2_ib || !a.bias || !a.bias2 || ((uintptr_t)a.wb2 & 15)))
        return hipErrorInvalidValue;
    int tiles_per_unit = (a.M + bm - 1) / bm;
    int n_tiles = a.NP / (32 * tn);
    long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    size_t lds = conv_gemm_bf16_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (w4) {
        static std::atomic<bool> done4[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !done4[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_bf16_w4_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_bf16_w4_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            done4[dev] = true;
        }
        if (tn == 3) hipLaunchKernelGGL(conv_gemm_bf16_w4_kernel<3>, dim3((unsigned)grid), dim3(256), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
        else hipLaunchKernelGGL(conv_gemm_bf16_w4_kernel<6>, dim3((unsigned)grid), dim3(256), lds, s, a, HL, XW, tiles_per_unit, n_tiles);
        return hipGetLastError();
    }
    if (a.S > 1 && a.pm == 3) {     // phase-tile rows: tile height = stride
        if (a.wb2 || tn != a.S || (tn != 3 && tn != 5) || (a.N / a.S) % 32 || a.N % a.S || a.NP != a.N || a.res || a.rscale || a.act != ACT_NONE)
            return hipErrorInvalidValue;
#define GPT(K_) return launch_fn_bf16<K_>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
        if (a.wbm) {
            if (a.xb || a.yb || ((uintptr_t)a.wbm & 15)) return hipErrorInvalidValue;
            if (tn == 3) GPT(conv_gemm_bf16x3_pt_kernel<3>);
            GPT(conv_gemm_bf16x3_pt_kernel<5>);
        }
        if (a.xb) {
            if (tn == 3) GPT((conv_gemm_bf16_pt_kernel<3, true>));
            GPT((conv_gemm_bf16_pt_kernel<5, true>));
        }
        if (tn == 3) GPT((conv_gemm_bf16_pt_kernel<3, false>));
        GPT((conv_gemm_bf16_pt_kernel<5, false>));
#undef GPT
    }
    if (a.S > 1 && a.pm == 2) {     // octet-interleaved transposed conv: its own kernel instances (never fused)
        if (a.wb2 || (tn != 1 && tn != 2 && tn != 4)) return hipErrorInvalidValue;
#define GOI(K_) return launch_fn_bf16<K_>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
        if (a.wbm) {
            if (a.xb || a.yb || ((uintptr_t)a.wbm & 15)) return hipErrorInvalidValue;
            if (tn == 1) GOI(conv_gemm_bf16x3_oi_kernel<1>);
            if (tn == 2) GOI(conv_gemm_bf16x3_oi_kernel<2>);
            GOI(conv_gemm_bf16x3_oi_kernel<4>);
        }
        if (a.xb) {
            if (tn == 1) GOI((conv_gemm_bf16_oi_kernel<1, true>));
            if (tn == 2) GOI((conv_gemm_bf16_oi_kernel<2, true>));
            GOI((conv_gemm_bf16_oi_kernel<4, true>));
        }
        if (tn == 1) GOI((conv_gemm_bf16_oi_kernel<1, false>));
        if (tn == 2) GOI((conv_gemm_bf16_oi_kernel<2, false>));
        GOI((conv_gemm_bf16_oi_kernel<4, false>));
#undef GOI
    }
#define GOX(TN_, F_, X_) return launch_one_bf16<TN_, F_, X_>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
#define SW(F_, X_)                  \
    switch (tn) {                   \
        case 1: GOX(1, F_, X_);     \
        case 2: GOX(2, F_, X_);     \
        case 3: GOX(3, F_, X_);     \
        case 4: GOX(4, F_, X_);     \
        default: break;             \
    }
    if (a.wbm) {     // split-operand mode: fp32 exchange only
        if (a.xb || a.yb || ((uintptr_t)a.wbm & 15) || (a.wb2 && (!a.wb2m || ((uintptr_t)a.wb2m & 15)))) return hipErrorInvalidValue;
#define GO3(TN_, F_) return launch_one_bf16x3<TN_, F_>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
        if (a.wb2) {
            switch (tn) { case 1: GO3(1, true); case 2: GO3(2, true); case 3: GO3(3, true); case 4: GO3(4, true); default: GO3(6, true); }
        }
        switch (tn) { case 1: GO3(1, false); case 2: GO3(2, false); case 3: GO3(3, false); default: GO3(4, false); }
#undef GO3
    }
    if (a.wb2) {
        if (a.xb) { SW(true, true) GOX(6, true, true); }
        SW(true, false) GOX(6, true, false);
    }
    if (a.xb) { SW(false, true) }
    SW(false, false)
    return hipErrorInvalidValue;
#undef GOX
#undef SW
}

}  // namespace hipvoc
