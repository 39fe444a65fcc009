// gfx950 (CDNA4 / MI355X) kernels of the vocoder path.  Hand-written HIP, wave64, fp32 MFMA.
//
// Orientation of every GEMM-shaped op: the A operand of v_mfma_f32_32x32x2_f32 holds WEIGHTS
// (row i = output channel), the B operand holds ACTIVATIONS (column j = time position), so the
// accumulator tile has time on the lane and output channels in the 16 registers.  With the
// channel-first / time-contiguous activation layout this makes every global load and store of an
// activation a run of consecutive floats across a wavefront, lets the 7 taps of a causal conv
// re-read ONE LDS halo tile at shifted offsets, and evaluates SnakeBeta once per staged element.
#include "kernels.h"
#include "device_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace hipvoc {

// Timing-only ablation switches (HIPVOC_DBG bits; results are WRONG with any of them set) exist only in a
// build with -DHIPVOC_ABLATE=1 (make ABLATE=1): in the production build the conditions fold away (the runtime
// tests alone cost ~2 % of conv7).
#ifndef HIPVOC_ABLATE
#define HIPVOC_ABLATE 0
#endif
#define ABL(bit_) (HIPVOC_ABLATE && (a.dbg & (bit_)))

static constexpr int BM = 256;   // time positions per workgroup (8 waves x 32)
static constexpr int BK = 16;    // input channels per K stage

// -------------------------------------------------------------------------------------------------
// conv_gemm: LDS-tiled implicit GEMM on v_mfma_f32_32x32x2_f32.
//   workgroup: 8 waves; tile = BN (= 32*TN rows: output channels x phases) x 256 time positions;
//   wave w: all BN rows x positions [32w, 32w+32) -> TN accumulator tiles of 32x32.
//   K loop, 16 input channels per block "cb":
//     * the activation halo tile Xs[16][XW] is double-buffered: the tile of cb+1 is fetched into
//       registers before the MFMAs of cb start and committed to LDS (SnakeBeta applied once per
//       element) after its last tap, so its HBM/L2 latency and its VALU work hide under the MFMAs;
//     * per tap a weight tile Ws[16][BN] (double-buffered, next one prefetched into a register
//       during the MFMAs); the taps of a causal conv re-read the SAME Xs at shifted offsets;
//     * one barrier per (cb, tap).
//   FUSE (residual units whose channels fit one tile): the accumulator H = conv7(snake(x)) becomes
//   snake2(H + b1) in registers and is fed straight back as the B operand of the 1x1 conv (time
//   stays on the lane; H's channel index is the k of the second product), then + b2 + x.
//   (Measured on MI355X, G=32: staging X without the prefetch cost 11-13% of conv7 time and 26% of the
//    2-tap transposed convs; weights straight from L2 instead of LDS were 15% slower.)
// -------------------------------------------------------------------------------------------------

// ENC: the encoder path's variant (ELU on the input, ELU / tanh-GELU epilogues); compiled out of the vocoder's kernels
template <int TN, bool FUSE, int NW, bool ENC = false>
__global__ __launch_bounds__(NW * 64, (FUSE && TN > 4) ? 3 : 4) void conv_gemm_kernel(ConvGemmArgs a, int HL, int XW, int tiles_per_unit,
                                                                           int n_tiles) {
    constexpr int NTH = NW * 64;                     // NW waves; the workgroup covers BM = 32*NW positions
    constexpr int BM = NW * 32;
    constexpr int BN = TN * 32;
    constexpr int WF4 = 4 * BN;                      // float4 per W tile [BK][BN]
    constexpr int WPT = (WF4 + NTH - 1) / NTH;       // float4 per thread
    static_assert(WPT <= 3, "W staging registers");
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    constexpr int PMF = (FUSE ? 6 : 4) * BN;         // per-row parameter table (device_common.h) at the start of LDS
    float *Pm = smem_all;
    float *smem = smem_all + PMF;                    // Xs[2][BK][XW], then Ws[2][BK][BN]
    const int xtile = BK * XW;
    float *Ws = smem + 2 * xtile;

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
    const int CinP = ((a.Cin + 31) / 32) * 32;       // row count per tap of the packed weight tensor
    const int nstage = ncb * a.ntaps;
    const int xw4 = XW >> 2;

    fill_param_table<BN, FUSE>(a, Pm, n0, tid, NTH);     // read after the main loop (barriers in between)

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // ---- W staging through named registers
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, wq2 = w0;
    const int wk0 = tid / (BN / 4), wc0 = (tid - wk0 * (BN / 4)) * 4;
    const int wk1 = (tid + NTH) / (BN / 4), wc1 = (tid + NTH - wk1 * (BN / 4)) * 4;
    const int wk2 = (tid + 2 * NTH) / (BN / 4), wc2 = (tid + 2 * NTH - wk2 * (BN / 4)) * 4;
    // tile = BK rows starting at wp_ (row stride ld_), columns n0..n0+BN
#define W_LOAD_FROM(wp_, ld_)                                                                              \
    {                                                                                                      \
        if (WF4 >= NTH || tid < WF4) w0 = *reinterpret_cast<const float4 *>((wp_) + (long)wk0 * (ld_) + wc0); \
        if constexpr (WPT > 1) if (WF4 >= 2 * NTH || tid + NTH < WF4) w1 = *reinterpret_cast<const float4 *>((wp_) + (long)wk1 * (ld_) + wc1); \
        if constexpr (WPT > 2) if (WF4 >= 3 * NTH || tid + 2 * NTH < WF4) wq2 = *reinterpret_cast<const float4 *>((wp_) + (long)wk2 * (ld_) + wc2); \
    }
#define W_LOAD(stage_)                                                                             \
    {                                                                                              \
        int cb_ = (stage_) / a.ntaps, tap_ = (stage_)-cb_ * a.ntaps;                               \
        const float *wp_ = a.w + ((long)tap_ * CinP + (long)cb_ * BK) * a.NP + n0;                 \
        W_LOAD_FROM(wp_, a.NP)                                                                     \
    }
#define W_STORE(buf_)                                                                              \
    {                                                                                              \
        float *ws_ = Ws + (buf_) * (BK * BN) + tid * 4;                                            \
        if (WF4 >= NTH || tid < WF4) *reinterpret_cast<float4 *>(ws_) = w0;                        \
        if constexpr (WPT > 1) if (WF4 >= 2 * NTH || tid + NTH < WF4) *reinterpret_cast<float4 *>(ws_ + 4 * NTH) = w1; \
        if constexpr (WPT > 2) if (WF4 >= 3 * NTH || tid + 2 * NTH < WF4) *reinterpret_cast<float4 *>(ws_ + 8 * NTH) = wq2; \
    }

    // ---- X tile: fetch (global -> 3 float4 registers per thread) and commit (SnakeBeta, -> LDS)
    float4 xr0, xr1, xr2;
    int xk0, xk1, xk2, xc0, xc1, xc2;                // (row, float4 column) of this thread's pieces
    {
        int i0 = tid, i1 = tid + NTH, i2 = tid + 2 * NTH;
        xk0 = i0 / xw4; xc0 = i0 - xk0 * xw4;
        xk1 = i1 / xw4; xc1 = i1 - xk1 * xw4;
        xk2 = i2 / xw4; xc2 = i2 - xk2 * xw4;
    }
    // Fetch: ONE unconditional float4 load per piece from a clamped (always valid) address; what must read as
    // zero (halo before the unit, positions >= Lin, channels >= Cin) is zeroed at commit time.  (With the loads
    // under per-lane conditions hipcc put an s_waitcnt vmcnt(0) between the pieces: three dependent memory round
    // trips in front of every channel block's MFMAs instead of a prefetch that overlaps them.)
    // The lane offset (row inside the block + clamped position) is loop-invariant and 32-bit, the channel block is
    // wave-uniform: a load is uniform base + lane offset, with no 64-bit address arithmetic in the loop.
    const int xlim = ((a.Lin + 3) & ~3) - 4;         // last whole float4 of a row (rows are padded to 4)
    auto lane_off = [&](int k, int c4) -> unsigned {
        int pos = m0 - HL + c4 * 4;
        pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
        return (unsigned)(k < BK ? k : 0) * (unsigned)a.x_cs + (unsigned)pos;
    };
    const unsigned xo0 = lane_off(xk0, xc0), xo1 = lane_off(xk1, xc1), xo2 = lane_off(xk2, xc2);
    auto x_fetch1 = [&](int cb, int k, int c4, unsigned off) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < BK) {     // pieces past the tile (whole waves, mostly) fetch nothing
            if ((cb + 1) * BK <= a.Cin) {
                v = *reinterpret_cast<const float4 *>(xu + (long)cb * BK * a.x_cs + off);
            } else {      // last, partial channel block: rows past Cin read row 0 (discarded at commit)
                int ci = cb * BK + k;
                ci = ci < a.Cin ? ci : 0;
                int pos = m0 - HL + c4 * 4;
                pos = pos < 0 ? 0 : (pos > xlim ? xlim : pos);
                v = *reinterpret_cast<const float4 *>(xu + (long)ci * a.x_cs + pos);
            }
        }
        return v;
    };
    // SnakeBeta parameters of each piece's channel travel with the prefetch (loading them at commit
    // time would expose an L2 round trip per channel block)
    float2 xp0 = make_float2(0.f, 0.f), xp1 = xp0, xp2 = xp0;
    auto p_fetch1 = [&](int cb, int k) -> float2 {
        int ci = cb * BK + k;
        ci = (k < BK && ci < a.Cin) ? ci : 0;        // vectors are zero-padded; a clamped read is discarded
        if (a.sn_a) return make_float2(a.sn_a[ci], a.sn_ib[ci]);
        return make_float2(0.f, 0.f);
    };
    auto x_commit1 = [&](int cb, int buf, int k, int c4, float4 v, float2 sp) {
        if (k >= BK) return;
        const int pos = m0 - HL + c4 * 4;
        const bool ok = cb * BK + k < a.Cin && pos >= 0;     // pos is a multiple of 4: the halo test is per float4
        v.x = ok && pos + 0 < a.Lin ? v.x : 0.0f;
        v.y = ok && pos + 1 < a.Lin ? v.y : 0.0f;
        v.z = ok && pos + 2 < a.Lin ? v.z : 0.0f;
        v.w = ok && pos + 3 < a.Lin ? v.w : 0.0f;
        if (a.sn_a) {   // SnakeBeta: x + 1/(e^beta+1e-9) * sin(x e^alpha)^2 ; snake(0) == 0 keeps the zero padding
            f32x2 lo = snake_beta2(mk2(v.x, v.y), (f32x2)(sp.x), (f32x2)(sp.y));
            f32x2 hi = snake_beta2(mk2(v.z, v.w), (f32x2)(sp.x), (f32x2)(sp.y));
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        if constexpr (ENC) {
            if (a.in_elu) v = make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));   // elu(0) == 0 keeps the padding
        }
        *reinterpret_cast<float4 *>(smem + buf * xtile + k * XW + c4 * 4) = v;
    };
#define X_FETCH(cb_) { xr0 = x_fetch1(cb_, xk0, xc0, xo0); xr1 = x_fetch1(cb_, xk1, xc1, xo1); xr2 = x_fetch1(cb_, xk2, xc2, xo2); \
                       xp0 = p_fetch1(cb_, xk0); xp1 = p_fetch1(cb_, xk1); xp2 = p_fetch1(cb_, xk2); }
#define X_COMMIT(cb_, buf_) { x_commit1(cb_, buf_, xk0, xc0, xr0, xp0); x_commit1(cb_, buf_, xk1, xc1, xr1, xp1); x_commit1(cb_, buf_, xk2, xc2, xr2, xp2); }

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
            if (!ABL(4)) __syncthreads();   // Ws[st&1] (and, at tap 0, the committed Xs[cb&1]) are visible
            {
                const float *ws = Ws + (st & 1) * (BK * BN) + l31;
                const float *xs = smem + (cb & 1) * xtile + wave * 32 + l31 + HL + a.shift0 + tap * a.shift_step;
#pragma unroll
                for (int kk = 0; kk < BK / 2; kk++) {
                    int k = 2 * kk + h;
                    float av[TN];
#pragma unroll
                    for (int i = 0; i < TN; i++) av[i] = ws[k * BN + i * 32];
                    float bv = xs[k * XW];
#pragma unroll
                    for (int i = 0; i < TN; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
                }
            }
            if (!ABL(2)) W_STORE((st + 1) & 1);
        }
        // Xs[(cb+1)&1] was last read during cb-1; every wave has passed a later barrier since
        if (xnext) X_COMMIT(cb + 1, (cb + 1) & 1)
    }
#undef X_FETCH
#undef X_COMMIT

    if (ABL(8)) {   // timing-only: main loop alone (one never-taken store keeps the MFMAs alive)
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
        // H = snake2(acc + b1), in place.  Register r of tile i is channel i*32 + (r&3) + 8*(r>>2) + 4*h.
        // bias / snake vectors are zero-padded by the loader, so padded rows stay exactly 0.
        // staging of the second product's weight tiles: 256*TN float4 per tile through up to six registers
        constexpr int W2F4 = 256 * TN;
        static_assert(W2F4 <= 6 * NTH, "W2 staging registers");
#define W2_PIECE(j_, reg_, base_)                                                                            \
    if constexpr (W2F4 > (j_) * NTH)                                                                         \
        if (W2F4 >= ((j_) + 1) * NTH || tid + (j_) * NTH < W2F4)                                             \
            reg_ = *reinterpret_cast<const float4 *>((base_) + (long)((tid + (j_) * NTH) >> 3) * a.NP + (tid & 7) * 4);
#define W2_FETCH_FROM(base_) { W2_PIECE(0, xr0, base_) W2_PIECE(1, xr1, base_) W2_PIECE(2, xr2, base_) \
                               W2_PIECE(3, w0, base_) W2_PIECE(4, w1, base_) W2_PIECE(5, wq2, base_) }
#define W2_PUT(j_, reg_, dst_)                                                                               \
    if constexpr (W2F4 > (j_) * NTH)                                                                         \
        if (W2F4 >= ((j_) + 1) * NTH || tid + (j_) * NTH < W2F4) *reinterpret_cast<float4 *>((dst_) + 4 * (j_) * NTH) = reg_;
        {
            // tile 0: in flight during the activation below
            const float *w2p = launder_after(a.w2, acc[0][15]);
            W2_FETCH_FROM(w2p)
        }
#pragma unroll
        for (int i = 0; i < TN; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (r & 1) {   // registers r-1, r are channels n-1, n: one packed evaluation for the pair
                    f32x2 hv = snake_beta2(mk2(acc[i][r - 1] + Pm[n - 1], acc[i][r] + Pm[n]), mk2(Pm[BN + n - 1], Pm[BN + n]),
                                           mk2(Pm[2 * BN + n - 1], Pm[2 * BN + n]));
                    acc[i][r - 1] = hv.x;
                    acc[i][r] = hv.y;
                }
                if ((r & 3) == 3) {
                    // ... and the results are pinned here: without it LLVM sinks the polynomial tails to their first
                    // use in the second product, where all 16*TN chains become live at once
                    asm volatile("" : "+v"(acc[i][r - 3]), "+v"(acc[i][r - 2]), "+v"(acc[i][r - 1]), "+v"(acc[i][r]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // second product, ONE 32-row output tile at a time (16 accumulator registers instead of 16*TN, so
        // the C=96 unit still runs four waves per SIMD).  Its weights W2[:, 32o..32o+32) are staged as WHOLE
        // [32*TN k rows][32 output rows] tiles, double-buffered over the (now free) Xs/Ws area through the
        // activation prefetch registers: one barrier per output tile, 16*TN MFMAs between barriers.
        float *yu = a.y + (long)unit * a.y_us;
        const int m = m0 + wave * 32 + l31;
        constexpr int W2T = 32 * TN * 32;         // floats per tile
#define W2_FETCH(o_) { const float *wq_ = launder_after(a.w2, dep2) + (o_) * 32; W2_FETCH_FROM(wq_) }
#define W2_COMMIT(buf_)                                                                                      \
    {                                                                                                        \
        float *wd_ = smem + (buf_) * W2T + tid * 4;                                                          \
        W2_PUT(0, xr0, wd_) W2_PUT(1, xr1, wd_) W2_PUT(2, xr2, wd_) W2_PUT(3, w0, wd_) W2_PUT(4, w1, wd_) W2_PUT(5, wq2, wd_) \
    }
        __syncthreads();                          // every wave is done with the main loop's Xs / Ws
        W2_COMMIT(0)
        float dep2 = acc[TN - 1][15];             // pins each tile's weight fetch behind the previous tile's product
#pragma unroll
        for (int o = 0; o < TN; o++) {
            if (o + 1 < TN) W2_FETCH(o + 1)
            __syncthreads();                      // tile o visible; tile o-1's readers are past
            // this tile's residual is fetched before its product, so HBM latency hides under the MFMAs
            float resv[16];
            {
                const float *rup = launder_after(a.res, dep2) + (long)unit * a.y_us;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int n = o * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    resv[r] = (n < a.N && m < a.M) ? rup[(long)n * a.y_cs + m] : 0.0f;
                }
            }
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[r] = 0.0f;
            const float *ws = smem + (o & 1) * W2T + l31 + 4 * h * 32;
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    // register r of tile i holds channel i*32 + (r&3) + 8*(r>>2) + 4h: that is this lane half's k
                    const int k = i * 32 + (r & 3) + 8 * (r >> 2);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[k * 32], acc[i][r], acc2, 0, 0, 0);
                    // keep the LDS reads of at most 8 steps in flight (unbounded, all 16*TN are hoisted and spill)
                    if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            if (o + 1 < TN) W2_COMMIT((o + 1) & 1)
            dep2 = acc2[15];
            // + b2 + residual x, store this 32-row tile
            fused_epilogue_tile<BN>(a, acc2, resv, Pm, o * 32 + 4 * h, m, yu);
        }
#undef W2_FETCH
#undef W2_COMMIT
#undef W2_PIECE
#undef W2_FETCH_FROM
#undef W2_PUT
        return;
    }

    // epilogue: bias, activation, per-channel scale, residual, store (time on the lane: coalesced)
    float *yu = a.y + (long)unit * a.y_us;
    const float *ru0 = launder_after(a.res, acc[TN - 1][15]);
    const float *ru = ru0 ? ru0 + (long)unit * ((ENC && a.s2d > 1) ? a.x_us : a.y_us) : nullptr;
    float *y2u0 = launder_after(a.y2, acc[TN - 1][15]);
    float *y2u = y2u0 ? y2u0 + (long)unit * a.y_us : nullptr;
    const int m = m0 + wave * 32 + l31;
#pragma unroll
    for (int i = 0; i < TN; i++) conv_epilogue_tile<BN, ENC>(a, acc[i], Pm, n0, i * 32 + 4 * h, m, yu, ru, y2u);
}
#undef W_LOAD_FROM
#undef W_LOAD
#undef W_STORE

// waves per workgroup of the fp32 kernel for these args: 8 (256 positions), or 4 (128 positions) for the fused
// residual unit (HIPVOC_FUSE_WAVES, default 4): its long epilogue (activation, second product, residual, store)
// then overlaps the main loops of the OTHER workgroups of the CU instead of stalling all eight waves at once
static int conv_gemm_waves(const ConvGemmArgs &a) {
    static int fw = -1;
    if (fw < 0) { const char *e = getenv("HIPVOC_FUSE_WAVES"); fw = e && atoi(e) == 8 ? 8 : 4; }
    return a.w2 ? fw : 8;
}

void detail::conv_gemm_geometry(const ConvGemmArgs &a, int &HL, int &XW, int bm) {
    int smin = a.shift0, smax = a.shift0;
    int last = a.shift0 + (a.ntaps - 1) * a.shift_step;
    if (last < smin) smin = last;
    if (last > smax) smax = last;
    int hl = smin < 0 ? -smin : 0, hr = smax > 0 ? smax : 0;
    HL = (hl + 3) & ~3;
    XW = bm + HL + ((hr + 3) & ~3);
}

// Tile height actually launched: the packed TN, lowered for small grids (more workgroups), or
// NP/32 for the fused residual unit (one tile must hold every channel).
int detail::conv_gemm_launch_tn(const ConvGemmArgs &a) {
    if (a.w2 || a.wb2) return a.NP / 32;
    if (a.S > 1 && a.pm == 3) return a.S;               // phase-tile rows of a transposed conv: one row tile per phase
    int tiles_m = ((a.M + BM - 1) / BM) * a.units;
    int tn = a.TN;
    static int min_wgs = -1;
    if (min_wgs < 0) { const char *e = getenv("HIPVOC_TN_WGS"); min_wgs = e ? atoi(e) : 384; }
    while (tn > 1 && (long)tiles_m * (a.NP / (32 * tn)) < min_wgs) {
        int t2 = tn == 4 ? 2 : 1;
        if (a.NP % (32 * t2)) break;
        tn = t2;
    }
    return tn;
}

using detail::conv_gemm_geometry;
using detail::conv_gemm_launch_tn;

const char *conv_gemm_kernel_name(const ConvGemmArgs &a) {
    if (a.wb) return conv_gemm_bf16_kernel_name(a);
    if (wino_selected(a)) return wino_kernel_name(a);
    if (gemmds_selected(a)) return gemmds_kernel_name(a);
    if (gemm1_selected(a)) return gemm1_kernel_name(a);
    if (convs_selected(a)) return convs_kernel_name(a);
    if (gemmd_selected(a)) return gemmd_kernel_name(a);
    static char names[3][2][8][48];
    int tn = conv_gemm_launch_tn(a);
    int fuse = a.w2 ? 1 : 0, nw = conv_gemm_waves(a);
    const bool enc = a.in_elu || a.act == ACT_ELU || a.act == ACT_GELU_TANH || a.s2d > 1;
    char *n = names[enc ? 2 : fuse][nw == 4][tn & 7];
    if (!n[0]) snprintf(n, 48, "conv_gemm_kernel<%d,%s,%d,%s>", tn, fuse ? "true" : "false", nw, enc ? "true" : "false");   // as rocprofv3 prints it
    return n;
}

size_t conv_gemm_lds_bytes(const ConvGemmArgs &a) {
    int HL, XW;
    conv_gemm_geometry(a, HL, XW, 32 * conv_gemm_waves(a));
    const int tn = conv_gemm_launch_tn(a);
    size_t fl = 2 * (size_t)BK * XW + 2 * (size_t)BK * 32 * tn;
    if (a.w2) fl = fl > 2 * (size_t)32 * tn * 32 ? fl : 2 * (size_t)32 * tn * 32;   // two W2 tiles of the fused second product
    fl += (size_t)(a.w2 ? 6 : 4) * 32 * tn;                                          // per-row parameter table
    return sizeof(float) * fl;
}

template <int TN, bool FUSE, int NW, bool ENC = false>
static hipError_t conv_gemm_launch_one(const ConvGemmArgs &a, int HL, int XW, int tiles_per_unit, int n_tiles,
                                       long grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> done[64] = {};      // (the worker calls from several threads per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !done[dev]) {   // dynamic LDS above 64 KiB must be enabled per function and device
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<TN, FUSE, NW, ENC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<TN, FUSE, NW, ENC>), dim3((unsigned)grid), dim3(NW * 64), lds, s, a, HL, XW, tiles_per_unit,
                       n_tiles);
    return hipGetLastError();
}

hipError_t launch_conv_gemm(const ConvGemmArgs &a_in, hipStream_t s) {
    if (a_in.wb) return launch_conv_gemm_bf16(a_in, s);
    if (wino_selected(a_in)) return launch_wino_conv7(a_in, s);
    if (gemmds_selected(a_in)) return launch_gemmds(a_in, s);
    if (gemm1_selected(a_in)) return launch_gemm1(a_in, s);
    if (convs_selected(a_in)) return launch_convs(a_in, s);
    if (gemmd_selected(a_in)) return launch_gemmd(a_in, s);
    ConvGemmArgs a = a_in;
    {   // timing-only ablation switch (HIPVOC_DBG bit0: stage the activation tile only once; bit1: the weight tile only once; bit2: no barriers; bit3: main loop only).  Results are
        // WRONG with it set; never set in production.
        static int dbg = -1;
        if (dbg < 0) { const char *e = getenv("HIPVOC_DBG"); dbg = e ? atoi(e) : 0; }
        a.dbg = dbg;
    }
    int HL, XW;
    const int nw = conv_gemm_waves(a), bm = 32 * nw;
    conv_gemm_geometry(a, HL, XW, bm);
    if (a.M <= 0 || a.units <= 0) return hipSuccess;
    // host-side shape checks: what the kernel's vector accesses assume
    if ((a.x_cs & 3) || (a.x_us & 3) || (a.NP % (32 * a.TN)) || a.NP < a.N || a.TN < 1 || a.TN > 4 ||
        ((uintptr_t)a.x & 15) || ((uintptr_t)a.w & 15) || a.ntaps < 1 || a.S < 1 || BK * (XW / 4) > 3 * 64 * nw)
        return hipErrorInvalidValue;
    const int tn = conv_gemm_launch_tn(a);
    if (a.y2 && (!a.psn_a || !a.psn_ib || a.w2)) return hipErrorInvalidValue;
    if (a.pm >= 2) return hipErrorInvalidValue;      // (octet-interleaved / phase-tile trans-conv rows: bf16-mode kernels only)
    if ((a.in_elu || a.act == ACT_ELU || a.act == ACT_GELU_TANH || a.s2d > 1) && (a.w2 || a.sn_a)) return hipErrorInvalidValue;
    if (a.s2d > 1 && (a.S != 1 || a.M % a.s2d || a.y2 || a.psn_a || a.units < 1)) return hipErrorInvalidValue;
    if (a.w2 && (a.S != 1 || a.NP != 32 * tn || (a.N + 31) / 32 != tn || tn == 5 || tn > 6 || !a.res || !a.sn2_a ||
                 !a.sn2_ib || !a.bias || !a.bias2 || ((uintptr_t)a.w2 & 15)))
        return hipErrorInvalidValue;
    int tiles_per_unit = (a.M + bm - 1) / bm;
    int n_tiles = a.NP / (32 * tn);
    long grid = (long)tiles_per_unit * a.units * n_tiles;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    size_t lds = conv_gemm_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define GO(TN_, F_) return conv_gemm_launch_one<TN_, F_, 8>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
#define GO4(TN_) return conv_gemm_launch_one<TN_, true, 4>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
    if (a.w2 && nw == 4) {
        switch (tn) {
            case 1: GO4(1);
            case 2: GO4(2);
            case 3: GO4(3);
            case 4: GO4(4);
            default: GO4(6);
        }
    }
    if (a.w2) {
        switch (tn) {
            case 1: GO(1, true);
            case 2: GO(2, true);
            case 3: GO(3, true);
            case 4: GO(4, true);
            default: GO(6, true);
        }
    }
    if (a.in_elu || a.act == ACT_ELU || a.act == ACT_GELU_TANH || a.s2d > 1) {
#define GOE(TN_) return conv_gemm_launch_one<TN_, false, 8, true>(a, HL, XW, tiles_per_unit, n_tiles, grid, lds, s)
        switch (tn) {
            case 1: GOE(1);
            case 2: GOE(2);
            case 3: GOE(3);
            default: GOE(4);
        }
#undef GOE
    }
    switch (tn) {
        case 1: GO(1, false);
        case 2: GO(2, false);
        case 3: GO(3, false);
        default: GO(4, false);
    }
#undef GO
#undef GO4
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
        // (four rows' loads in flight at a time, summed in codebook order: the kernel is one dependent round trip per
        // loop iteration otherwise -- 0.12 ms for a single utterance)
        float v0 = 0.0f, v1 = 0.0f;
        for (int q0 = 0; q0 < a.nq; q0 += 4) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = q0 + j < a.nq ? q0 + j : a.nq - 1;      // clamped: the copy is not summed
                long long c = cp[q];
                if (c < 0 || c >= a.cb_size) {
                    if (d == 0) *a.err = 1;
                    c = 0;
                }
                e[j] = a.embed[((long)q * a.cb_size + c) * a.D + d];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = q0 + j;
                if (q == 0) v0 = 0.0f + e[j];
                else if (q < a.nq) v1 = v1 + e[j];
            }
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
        // the two chains keep their k-ascending order; eight weights of each are loaded before they are used
        float p = 0.0f, r = 0.0f;
        int k = 0;
        for (; k + 8 <= a.D; k += 8) {
            float wf[8], wr[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                wf[j] = a.wf_t[(long)(k + j) * a.Q + n];
                wr[j] = a.wr_t[(long)(k + j) * a.Q + n];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) p = __builtin_fmaf(wf[j], s0[k + j], p);
#pragma unroll
            for (int j = 0; j < 8; j++) r = __builtin_fmaf(wr[j], s1[k + j], r);
        }
        for (int kk = k; kk < a.D; kk++) p = __builtin_fmaf(a.wf_t[(long)kk * a.Q + n], s0[kk], p);
        for (int kk = k; kk < a.D; kk++) r = __builtin_fmaf(a.wr_t[(long)kk * a.Q + n], s1[kk], r);
        a.out[(long)n * a.G * a.Lp + (long)u * a.Lp + t] = p + r;
    }
}

hipError_t launch_rvq(const RvqArgs &a, hipStream_t s) {
    if (a.G <= 0 || a.T <= 0) return hipSuccess;
    hipLaunchKernelGGL(rvq_kernel, dim3(a.G * a.T), dim3(256), sizeof(float) * 2 * a.D, s, a);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// RMSNorm over channels of x[C][P].  Workgroup = 32 positions x 8 channel slices: lanes 0-31 of a
// wave walk consecutive positions (128-B rows), slices are reduced through LDS in a fixed order.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float *x, float *y, const float *w, int C, long P,
                                                       float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    long p = (long)blockIdx.x * 32 + lp;
    bool ok = p < P;
    float ss = 0.0f;
    if (ok)
        for (int c = sl; c < C; c += 8) {
            float v = x[(long)c * P + p];
            ss += v * v;
        }
    red[sl][lp] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) tot += red[i][lp];
    float r = 1.0f / sqrtf(tot / (float)C + eps);
    if (ok)
        for (int c = sl; c < C; c += 8) y[(long)c * P + p] = w[c] * (x[(long)c * P + p] * r);
}

// Same workgroup shape, same slice assignment (channel c = slice + 8 j) and the same summation orders -- so the same
// bits -- with the thread's C/8 channels held in registers: all loads are issued before the first use and x is read
// once.  (The loop form above waits one memory latency per channel: 29 us per launch at C = 512, whatever the batch.)
template <int NPT>
__global__ __launch_bounds__(256) void rmsnorm_reg_kernel(const float *x, float *y, const float *w, int C, long P,
                                                           float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    long p = (long)blockIdx.x * 32 + lp;
    bool ok = p < P;
    long pc = ok ? p : P - 1;
    float v[NPT], wv[NPT];        // (the weights travel with the first batch of loads: no second round trip after the barrier)
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        int c = sl + 8 * j;
        v[j] = c < C ? x[(long)c * P + pc] : 0.0f;
        wv[j] = c < C ? w[c] : 0.0f;
    }
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < NPT; j++) ss += v[j] * v[j];
    red[sl][lp] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) tot += red[i][lp];
    float r = 1.0f / sqrtf(tot / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        int c = sl + 8 * j;
        if (ok && c < C) y[(long)c * P + p] = wv[j] * (v[j] * r);
    }
}

const char *rmsnorm_kernel_name(int C) {
    return C <= 64 ? "rmsnorm_reg_kernel<8>" : C <= 256 ? "rmsnorm_reg_kernel<32>" : C <= 512 ? "rmsnorm_reg_kernel<64>"
           : C <= 1024 ? "rmsnorm_reg_kernel<128>" : "rmsnorm_kernel";
}

hipError_t launch_rmsnorm(const float *x, float *y, const float *w, int C, long P, float eps, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    const dim3 grid((unsigned)((P + 31) / 32));
#define RN(N_) { hipLaunchKernelGGL(rmsnorm_reg_kernel<N_>, grid, dim3(256), 0, s, x, y, w, C, P, eps); return hipGetLastError(); }
    if (C <= 64) RN(8)
    if (C <= 256) RN(32)
    if (C <= 512) RN(64)
    if (C <= 1024) RN(128)
#undef RN
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, x, y, w, C, P, eps);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Attention: one workgroup per (unit, head, block of 32 queries).  RoPE (rotate-half) is applied
// while q and k are staged into LDS.  Mask: kv <= q && kv > q - window.  fp32 softmax.
// -------------------------------------------------------------------------------------------------
// QB queries per workgroup, 256 / QB threads per query row (QB = 32 for the vocoder's window of 72; the encoder's
// window of 250 needs QB = 16 to fit K, V and the score rows into LDS)
template <int QB>
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
    // scores + softmax: TPR threads per query row
    constexpr int TPR = 256 / QB;
    int tq = tid / TPR, l8 = tid % TPR;
    float scaling = 1.0f / sqrtf((float)hd);
    int t_abs = q0 + tq;
    float mx = -INFINITY;
    if (tq < nq) {
        for (int tk = l8; tk < nk; tk += TPR) {
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
#pragma unroll
    for (int sh = 1; sh < TPR; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    float den = 0.0f;
    if (tq < nq) {
        for (int tk = l8; tk < nk; tk += TPR) {
            float sc = Ps[tq * NKP + tk];
            float e = sc == -INFINITY ? 0.0f : expf(sc - mx);
            Ps[tq * NKP + tk] = e;
            den += e;
        }
    }
#pragma unroll
    for (int sh = 1; sh < TPR; sh <<= 1) den += __shfl_xor(den, sh);
    __syncthreads();
    if (tq < nq) {
        float inv = 1.0f / den;
        for (int d = l8; d < hd; d += TPR) {
            float acc = 0.0f;
            for (int tk = 0; tk < nk; tk++) acc += (Ps[tq * NKP + tk] * inv) * Vs[d * NKP + tk];
            out[(long)(hh * hd + d) * cs + base + t_abs] = acc;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Attention on the fp32 matrix cores (head_dim 32 or 64): one wave per (unit, head, 32 queries), flash-style over
// key tiles of 32, in the TRANSPOSED formulation so that every softmax statistic is lane-local:
//   S^T tile = K_tile (A: rows = keys) x Q^T (B: columns = queries)  -> lane l31 holds ONE query's scores for the
//   16 keys of its half; running max / sum are per lane (+ one exchange between the two lane halves);
//   O^T += V^T (A: rows = head dims) x P^T (B): the score accumulator is fed straight back as the B operand -- register
//   r of lane half h is key (r&3) + 8(r>>2) + 4h of the tile, and the A operand is read from LDS in that key order.
// RoPE (rotate-half) is applied to q in registers and to k while it is staged.  Mask: kv <= q && kv > q - window.
// -------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(64) void attention_mfma_kernel(const float *qkv, float *out, const float *cos_t,
                                                            const float *sin_t, int nh, int G, int T, int Lp, int window) {
    __shared__ float Ks[HD][32];        // [d][key]
    __shared__ float Vs[32][HD + 1];    // [key][d]
    constexpr int half = HD / 2;
    const int u = blockIdx.x, hh = blockIdx.y, q0 = blockIdx.z * 32;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const long cs = (long)G * Lp, base = (long)u * Lp;
    const float *qp = qkv + (long)(hh * HD) * cs + base;
    const float *kp = qkv + (long)((nh + hh) * HD) * cs + base;
    const float *vp = qkv + (long)((2 * nh + hh) * HD) * cs + base;
    const int tq = q0 + l31;
    const bool qok = tq < T;
    const int tqc = qok ? tq : T - 1;
    // this lane's query, head dims d = 2*kk + h (the k index the lane supplies to MFMA step kk), RoPE applied
    float qv[HD / 2];
#pragma unroll
    for (int kk = 0; kk < HD / 2; kk++) qv[kk] = qp[(long)(2 * kk + h) * cs + tqc];
#pragma unroll
    for (int kk = 0; kk < HD / 4; kk++) {
        const int d = 2 * kk + h;                         // d < half; its partner d + half is step kk + HD/4
        const float a = qv[kk], b = qv[kk + HD / 4];
        const float c = cos_t[tqc * half + d], s = sin_t[tqc * half + d];
        qv[kk] = a * c + (-b) * s;
        qv[kk + HD / 4] = b * c + a * s;
    }
    f32x16 o[HD / 32];
#pragma unroll
    for (int i = 0; i < HD / 32; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const float scaling = 1.0f / sqrtf((float)HD);
    const int k_lo = max(0, q0 - window + 1);
    const int qlast = min(q0 + 31, T - 1);
    // The next key tile (k, v, RoPE factors: 3 HD / 2 + HD / 2 values per lane) is fetched into registers while the
    // current one is being multiplied: a tile's loads are one memory round trip, about what its MFMAs take.
    float ka[HD / 4], kb[HD / 4], kcs[HD / 4], ksn[HD / 4], vv[HD / 2];
    auto tile_fetch = [&](int kt) {
        const int key = kt * 32 + l31;
        const int kc = key < T ? key : T - 1;              // clamped: keys >= T are masked below
#pragma unroll
        for (int j = 0; j < HD / 4; j++) {
            const int d = h * (HD / 4) + j;
            ka[j] = kp[(long)d * cs + kc];
            kb[j] = kp[(long)(d + half) * cs + kc];
            kcs[j] = cos_t[kc * half + d];
            ksn[j] = sin_t[kc * half + d];
        }
#pragma unroll
        for (int j = 0; j < HD / 2; j++) vv[j] = vp[(long)(h * (HD / 2) + j) * cs + kc];
    };
    const int kt_last = qlast / 32;
    tile_fetch(k_lo / 32);
    for (int kt = k_lo / 32; kt <= kt_last; kt++) {
        __syncthreads();                                   // the previous tile's LDS reads are done
#pragma unroll
        for (int j = 0; j < HD / 4; j++) {
            const int d = h * (HD / 4) + j;
            const float a = ka[j], b = kb[j], c = kcs[j], s = ksn[j];
            Ks[d][l31] = a * c + (-b) * s;
            Ks[d + half][l31] = b * c + a * s;
        }
#pragma unroll
        for (int j = 0; j < HD / 2; j++) Vs[l31][h * (HD / 2) + j] = vv[j];
        tile_fetch(kt < kt_last ? kt + 1 : kt);            // unconditional (the last tile re-fetches itself)
        __syncthreads();
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < HD / 2; kk++) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[2 * kk + h][l31], qv[kk], sc, 0, 0, 0);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = key <= tq && key > tq - window && qok;
            sc[r] = ok ? sc[r] * scaling : -INFINITY;
            tmax = fmaxf(tmax, sc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        if (mn != -INFINITY) {                             // (per lane; no MFMA inside the branch)
            const float alpha = m == -INFINITY ? 0.0f : expf(m - mn);
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sc[r] = sc[r] == -INFINITY ? 0.0f : expf(sc[r] - mn);
                ps += sc[r];
            }
            l = l * alpha + ps;
#pragma unroll
            for (int i = 0; i < HD / 32; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] *= alpha;
            m = mn;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) sc[r] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int i = 0; i < HD / 32; i++) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kl][i * 32 + l31], sc[r], o[i], 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32);
    if (qok) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int i = 0; i < HD / 32; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(long)(hh * HD + d) * cs + base + tq] = o[i][r] * inv;
            }
    }
}

static int attention_use_mfma() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("HIPVOC_ATTN_MFMA"); v = e ? atoi(e) : 1; }
    return v;
}

// kernel symbol launched for these shapes (profiling labels)
const char *attention_kernel_name(int hd, int T, int window) {
    if (attention_use_mfma() && hd == 64) return "attention_mfma_kernel<64>";
    if (attention_use_mfma() && hd == 32) return "attention_mfma_kernel<32>";
    for (int qb = 32; qb >= 8; qb >>= 1) {
        int nkmax = qb + (window > 1 ? window - 1 : 0);
        if (nkmax > T) nkmax = T;
        int NKP = nkmax | 1;
        size_t lds = sizeof(float) * ((size_t)hd * qb + 2 * (size_t)hd * NKP + (size_t)qb * NKP);
        if (lds <= 160 * 1024) return qb == 32 ? "attention_kernel<32>" : qb == 16 ? "attention_kernel<16>" : "attention_kernel<8>";
    }
    return "attention_kernel<8>";
}

hipError_t launch_attention(const float *qkv, float *out, const float *cos_t, const float *sin_t, int nh, int hd,
                            int G, int T, int Lp, int window, hipStream_t s) {
    if (G <= 0 || T <= 0) return hipSuccess;
    if (hd & 1) return hipErrorInvalidValue;
    if (attention_use_mfma() && (hd == 64 || hd == 32)) {
        dim3 g(G, nh, (T + 31) / 32);
        if (hd == 64) hipLaunchKernelGGL(attention_mfma_kernel<64>, g, dim3(64), 0, s, qkv, out, cos_t, sin_t, nh, G, T, Lp, window);
        else hipLaunchKernelGGL(attention_mfma_kernel<32>, g, dim3(64), 0, s, qkv, out, cos_t, sin_t, nh, G, T, Lp, window);
        return hipGetLastError();
    }
    for (int qb = 32; qb >= 8; qb >>= 1) {
        int nkmax = qb + (window > 1 ? window - 1 : 0);
        if (nkmax > T) nkmax = T;
        int NKP = nkmax | 1;    // odd stride: conflict-free column walks
        size_t lds = sizeof(float) * ((size_t)hd * qb + 2 * (size_t)hd * NKP + (size_t)qb * NKP);
        if (lds > 160 * 1024) continue;
        dim3 g(G, nh, (T + qb - 1) / qb);
#define ATT_GO(QB_)                                                                                                  \
        {                                                                                                            \
            if (lds > 64 * 1024) {                                                                                   \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel<QB_>),            \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);          \
                if (e != hipSuccess) return e;                                                                       \
            }                                                                                                        \
            hipLaunchKernelGGL(attention_kernel<QB_>, g, dim3(256), lds, s, qkv, out, cos_t, sin_t, nh, hd, G, T, Lp, window, NKP); \
            return hipGetLastError();                                                                                \
        }
        if (qb == 32) ATT_GO(32)
        if (qb == 16) ATT_GO(16)
        ATT_GO(8)
#undef ATT_GO
    }
    return hipErrorInvalidValue;
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
// ConvNeXt front half: depthwise causal conv (K taps, left pad K-1) + LayerNorm over channels
// (two-pass variance).  Workgroup = 32 positions x 8 channel slices; y holds the conv output between
// the passes (each thread re-reads only what it wrote itself).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float *x, float *y, const float *dw_w,
                                                         const float *dw_b, const float *ln_w, const float *ln_b,
                                                         int C, int K, int G, int L, int Lp, float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    int t = blockIdx.x * 32 + lp;
    int u = blockIdx.y;
    bool ok = t < L;
    long cs = (long)G * Lp, base = (long)u * Lp + t;
    float sum = 0.0f;
    if (ok)
        for (int c = sl; c < C; c += 8) {
            float acc = dw_b[c];
            const float *xr = x + (long)c * cs + (long)u * Lp;
            for (int j = 0; j < K; j++) {
                int sidx = t - (K - 1 - j);
                if (sidx >= 0) acc += dw_w[c * K + j] * xr[sidx];
            }
            y[(long)c * cs + base] = acc;
            sum += acc;
        }
    red[sl][lp] = sum;
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mean += red[i][lp];
    mean /= (float)C;
    __syncthreads();
    float var = 0.0f;
    if (ok)
        for (int c = sl; c < C; c += 8) {
            float d = y[(long)c * cs + base] - mean;
            var += d * d;
        }
    red[sl][lp] = var;
    __syncthreads();
    float vt = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) vt += red[i][lp];
    float r = 1.0f / sqrtf(vt / (float)C + eps);
    if (ok)
        for (int c = sl; c < C; c += 8) {
            float v = y[(long)c * cs + base];
            y[(long)c * cs + base] = (v - mean) * r * ln_w[c] + ln_b[c];
        }
}

// Register-resident form (same slices, same orders, same bits): the conv outputs of the thread's C/8 channels stay in
// registers between the three passes, y is written once and never re-read.
template <int NPT, int K>
__global__ __launch_bounds__(256) void dwconv_ln_reg_kernel(const float *x, float *y, const float *dw_w,
                                                             const float *dw_b, const float *ln_w, const float *ln_b,
                                                             int C, int G, int L, int Lp, float eps) {
    __shared__ float red[8][32];
    int lp = threadIdx.x & 31, sl = threadIdx.x >> 5;
    int t = blockIdx.x * 32 + lp;
    int u = blockIdx.y;
    bool ok = t < L;
    int tc = ok ? t : L - 1;
    long cs = (long)G * Lp, base = (long)u * Lp + t;
    float v[NPT];
    float sum = 0.0f;
    // Every load is unconditional from a clamped address (a load under a per-lane condition makes hipcc branch around it
    // and wait for it: 896 dependent round trips per thread, 145 us per launch), and the loads of EIGHT channels are
    // issued before the first of them is used; what must not count is selected away afterwards.
    constexpr int CG = 8;
    static_assert(NPT % CG == 0, "channel groups");
#pragma unroll
    for (int jb = 0; jb < NPT; jb += CG) {
        float xv[CG][K], wv[CG][K], bb[CG];
#pragma unroll
        for (int g = 0; g < CG; g++) {
            const int c = sl + 8 * (jb + g), cc = c < C ? c : C - 1;
            const float *xr = x + (long)cc * cs + (long)u * Lp;
#pragma unroll
            for (int q = 0; q < K; q++) {
                const int sidx = tc - (K - 1 - q);
                xv[g][q] = xr[sidx < 0 ? 0 : sidx];
                wv[g][q] = dw_w[cc * K + q];
            }
            bb[g] = dw_b[cc];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < CG; g++) {
            float acc = bb[g];
#pragma unroll
            for (int q = 0; q < K; q++) acc = (tc - (K - 1 - q) >= 0) ? acc + wv[g][q] * xv[g][q] : acc;
            v[jb + g] = (sl + 8 * (jb + g) < C) ? acc : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NPT; j++) sum += v[j];
    red[sl][lp] = sum;
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mean += red[i][lp];
    mean /= (float)C;
    __syncthreads();
    float var = 0.0f;
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        float d = v[j] - mean;
        var += (sl + 8 * j < C) ? d * d : 0.0f;
    }
    red[sl][lp] = var;
    __syncthreads();
    float vt = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) vt += red[i][lp];
    float r = 1.0f / sqrtf(vt / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        int c = sl + 8 * j;
        if (ok && c < C) y[(long)c * cs + base] = (v[j] - mean) * r * ln_w[c] + ln_b[c];
    }
}

const char *dwconv_ln_kernel_name(int C, int K) {
    if (K != 7 || C > 1024) return "dwconv_ln_kernel";
    return C <= 64 ? "dwconv_ln_reg_kernel<8,7>" : C <= 512 ? "dwconv_ln_reg_kernel<64,7>" : "dwconv_ln_reg_kernel<128,7>";
}

hipError_t launch_dwconv_ln(const float *x, float *y, const float *dw_w, const float *dw_b, const float *ln_w,
                            const float *ln_b, int C, int K, int G, int L, int Lp, float eps, hipStream_t s) {
    if (G <= 0 || L <= 0) return hipSuccess;
    if (K == 7 && C <= 1024) {
        const dim3 grid((L + 31) / 32, G);
#define DL(N_) { hipLaunchKernelGGL((dwconv_ln_reg_kernel<N_, 7>), grid, dim3(256), 0, s, x, y, dw_w, dw_b, ln_w, ln_b, C, G, L, Lp, eps); return hipGetLastError(); }
        if (C <= 64) DL(8)
        if (C <= 512) DL(64)
        DL(128)
#undef DL
    }
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3((L + 31) / 32, G), dim3(256), 0, s, x, y, dw_w, dw_b, ln_w, ln_b, C, K,
                       G, L, Lp, eps);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Final stage: SnakeBeta -> causal conv C -> 1 channel (K <= 9 taps) -> clamp[-1,1].
// HBM-bound (reads C floats per output sample): no LDS, every thread produces 4 consecutive samples
// and recomputes the K-1 halo activations it needs from aligned float4 loads (neighbours' loads hit L1/L2).
// sn_a == null: x is already activated (the last residual unit applied the final SnakeBeta in its epilogue, once per
// element instead of 2.5 times here) and the kernel is a plain 7-tap filter over C rows.
// -------------------------------------------------------------------------------------------------
template <int K, bool SN, bool FAST = false, bool RND = false>      // SN: x is the raw tensor and the kernel applies the SnakeBeta itself (a compile-time choice: as a run-time
                                // branch in the channel loop it kept hipcc from issuing the next channels' loads ahead of the arithmetic)
                                // FAST: ... in the bf16 mode's form (device_common.h: snake_beta2_hw), so that WHERE the final activation
                                // is applied (HIPVOC_FIN_POST) changes no bit in that mode either
                                // RND: ... and rounded to bf16, as the producer does when it hands the activated tensor over packed
                                // (final_conv_packed_kernel below): same values, same chain of fmas, same bits
__global__ __launch_bounds__(256) void final_conv_kernel(const float *x, float *out, const float *w /*[C][K]*/,
                                                          const float *bias, const float *sn_a, const float *sn_ib,
                                                          int C, int G, int L, int Lp, long out_stride,
                                                          int vec_store) {
    // 4 consecutive samples per thread: they need the activations of positions t-6 .. t+3, i.e. 5 aligned pairs for
    // the packed SnakeBeta.  (8 samples per thread -- 7 pairs for 8 outputs -- was measured 45 % SLOWER: the kernel is
    // bound by its access pattern, and a 32-byte lane stride halves the use of every load instruction.)
    constexpr int NO = 4;
    static_assert(K <= 7, "halo of 8 positions");
    int t = (blockIdx.x * 256 + threadIdx.x) * NO;
    int u = blockIdx.y;
    if (t >= L) return;
    long cs = (long)G * Lp;
    const float *xu = x + (long)u * Lp;
    float acc[NO];
    float b0 = bias ? bias[0] : 0.0f;
#pragma unroll
    for (int i = 0; i < NO; i++) acc[i] = b0;
    const int t6 = t >= 8 ? t - 6 : 0, t4 = t >= 4 ? t - 4 : 0;
    // Channels in batches of CB: all loads of a batch are issued before its arithmetic (program order keeps them so; as one loop
    // hipcc waited for each channel's three loads before issuing the next channel's -- three loads in flight per wave).
    // Unconditional loads from clamped addresses, zeroed by selects: written as `t >= 8 ? load : 0` hipcc split each halo load
    // into per-dword loads under exec-mask branches (seven load instructions and six branches per channel; found in the ISA).
    // Positions t-6, t-5 come as one 8-byte load (t is a multiple of 4).
    constexpr int CB = 4;
    for (int c0 = 0; c0 < C; c0 += CB) {
        float2 v0[CB];
        float4 v1[CB], v2[CB];
#pragma unroll
        for (int q = 0; q < CB; q++) {
            const int c = c0 + q < C ? c0 + q : C - 1;
            const float *xr = xu + (long)c * cs;
            v0[q] = *reinterpret_cast<const float2 *>(xr + t6);
            v1[q] = *reinterpret_cast<const float4 *>(xr + t4);
            v2[q] = *reinterpret_cast<const float4 *>(xr + t);
        }
#pragma unroll
        for (int q = 0; q < CB; q++) {
            const bool live = c0 + q < C;                 // (a batch past the last channel: clamped loads, zero weights -- no branch,
            const int c = live ? c0 + q : C - 1;          //  which would let hipcc sink each channel's loads to its arithmetic again)
            const float wl = live ? 1.0f : 0.0f;
            if (t < 8) v0[q] = make_float2(0.f, 0.f);
            if (t < 4) v1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            float sv[12] = {0.f, 0.f, v0[q].x, v0[q].y, v1[q].x, v1[q].y, v1[q].z, v1[q].w, v2[q].x, v2[q].y, v2[q].z, v2[q].w};
            if constexpr (SN) {
                const f32x2 sa = (f32x2)(sn_a[c]), sib = (f32x2)(sn_ib[c]);
#pragma unroll
                for (int i = 2; i < 12; i += 2) {         // (sv[0], sv[1] are never read: K - 1 <= 6)
                    f32x2 s2 = snake_beta2_t<FAST>(mk2(sv[i], sv[i + 1]), sa, sib);
                    if constexpr (RND) s2 = mk2(bf16_round(s2.x), bf16_round(s2.y));
                    int pos = t - 8 + i;
                    sv[i] = (pos >= 0 && pos < L) ? s2.x : 0.0f;        // zero padding applies to the activation output
                    sv[i + 1] = (pos + 1 >= 0 && pos + 1 < L) ? s2.y : 0.0f;
                }
            }                                             // else: positions < 0 are the zeros above; positions >= L only reach samples >= L
#pragma unroll
            for (int j = 0; j < K; j++) {
                const float wv = w[c * K + j] * wl;       // (unconditional: a uniform address stays a scalar load)
#pragma unroll
                for (int i = 0; i < NO; i++) acc[i] = __builtin_fmaf(wv, sv[8 + i - (K - 1) + j], acc[i]);   // sample t+i reads activation t+i-(K-1-j); an explicit fma: both instances round alike
            }
        }
    }
    float *o = out + (long)u * out_stride + t;
#pragma unroll
    for (int i = 0; i < NO; i++) acc[i] = fminf(fmaxf(acc[i], -1.0f), 1.0f);
    if (vec_store && t + NO <= L) {
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int i = 0; i < NO; i++)
            if (t + i < L) o[i] = acc[i];
    }
}

// bf16 mode: the same filter over the PACKED activated tensor (16-byte elements [C/8][G][Lp] of eight bf16 channels, written by the
// last residual unit): half the bytes of this HBM-bound stage, ten 16-byte loads per eight channels where the fp32 form issues
// twenty-four.  Channels and taps in the fp32 kernel's order: one chain of fmas per sample, the same bits as
// final_conv_kernel<7, true, true, true> on the raw tensor.
template <int K>
__global__ __launch_bounds__(256) void final_conv_packed_kernel(const uint4 *xb, float *out, const float *w /*[C][K]*/, const float *bias,
                                                                 int C, int G, int L, int Lp, long out_stride, int vec_store) {
    constexpr int NO = 4, NP = NO + K - 1;               // positions t - (K-1) .. t + 3
    int t = (blockIdx.x * 256 + threadIdx.x) * NO;
    int u = blockIdx.y;
    if (t >= L) return;
    const long cs = (long)G * Lp;
    const uint4 *xu = xb + (long)u * Lp;
    float acc[NO];
    const float b0 = bias ? bias[0] : 0.0f;
#pragma unroll
    for (int i = 0; i < NO; i++) acc[i] = b0;
    constexpr int GB = 2;                                // channel groups per batch: all loads of a batch before its arithmetic
    const int ng = C >> 3;                               // (host: C % 8 == 0)
    for (int g0 = 0; g0 < ng; g0 += GB) {
        uint4 e[GB][NP];
#pragma unroll
        for (int q = 0; q < GB; q++) {
            const int g = g0 + q < ng ? g0 + q : ng - 1;
            const uint4 *xr = xu + (long)g * cs;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int pos = t - (K - 1) + i;         // (t + 3 < Lp: t is a multiple of 4 below L, Lp a multiple of 4)
                e[q][i] = xr[pos < 0 ? 0 : pos];
            }
        }
#pragma unroll
        for (int q = 0; q < GB; q++) {
            const bool live = g0 + q < ng;               // (a batch past the last group: clamped loads, zero weights -- no branch)
            const int g = live ? g0 + q : ng - 1;
            const float wl = live ? 1.0f : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float sv[NP];
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const unsigned wd = j < 2 ? e[q][i].x : (j < 4 ? e[q][i].y : (j < 6 ? e[q][i].z : e[q][i].w));
                    const float v = __uint_as_float((j & 1) ? (wd & 0xffff0000u) : (wd << 16));
                    sv[i] = (t - (K - 1) + i) >= 0 ? v : 0.0f;              // causal zero padding
                }
                const int c = g * 8 + j;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const float wv = w[c * K + k] * wl;
#pragma unroll
                    for (int i = 0; i < NO; i++) acc[i] = __builtin_fmaf(wv, sv[i + k], acc[i]);
                }
            }
        }
    }
    float *o = out + (long)u * out_stride + t;
#pragma unroll
    for (int i = 0; i < NO; i++) acc[i] = fminf(fmaxf(acc[i], -1.0f), 1.0f);
    if (vec_store && t + NO <= L) {
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int i = 0; i < NO; i++)
            if (t + i < L) o[i] = acc[i];
    }
}

hipError_t launch_final_conv_packed(const unsigned short *xb, float *out, const float *w, const float *bias, int C, int K, int G, int L, int Lp,
                                    long out_stride, hipStream_t s) {
    if (G <= 0 || L <= 0) return hipSuccess;
    if (K != 7 || (Lp & 3) || (C & 7) || !xb || ((uintptr_t)xb & 15)) return hipErrorInvalidValue;
    const int vec = ((out_stride & 3) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const int nthr = (L + 3) / 4;
    hipLaunchKernelGGL((final_conv_packed_kernel<7>), dim3((nthr + 255) / 256, G), dim3(256), 0, s, reinterpret_cast<const uint4 *>(xb), out, w, bias, C,
                       G, L, Lp, out_stride, vec);
    return hipGetLastError();
}

hipError_t launch_final_conv(const float *x, float *out, const float *w, const float *bias, const float *sn_a,
                             const float *sn_ib, int C, int K, int G, int L, int Lp, long out_stride, hipStream_t s, bool fast_snake, bool round_act) {
    if (G <= 0 || L <= 0) return hipSuccess;
    if (K != 7 || (Lp & 3) || ((uintptr_t)x & 15)) return hipErrorInvalidValue;
    int vec = ((out_stride & 3) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    int nthr = (L + 3) / 4;
    if ((sn_a == nullptr) != (sn_ib == nullptr)) return hipErrorInvalidValue;
    if (round_act && !(sn_a && fast_snake)) return hipErrorInvalidValue;      // (only the bf16 mode's raw-input form rounds)
    if (round_act)
        hipLaunchKernelGGL((final_conv_kernel<7, true, true, true>), dim3((nthr + 255) / 256, G), dim3(256), 0, s, x, out, w, bias, sn_a, sn_ib, C,
                           G, L, Lp, out_stride, vec);
    else if (sn_a && fast_snake)
        hipLaunchKernelGGL((final_conv_kernel<7, true, true>), dim3((nthr + 255) / 256, G), dim3(256), 0, s, x, out, w, bias, sn_a, sn_ib, C,
                           G, L, Lp, out_stride, vec);
    else if (sn_a)
        hipLaunchKernelGGL((final_conv_kernel<7, true>), dim3((nthr + 255) / 256, G), dim3(256), 0, s, x, out, w, bias, sn_a, sn_ib, C,
                           G, L, Lp, out_stride, vec);
    else
        hipLaunchKernelGGL((final_conv_kernel<7, false>), dim3((nthr + 255) / 256, G), dim3(256), 0, s, x, out, w, bias, sn_a, sn_ib, C,
                           G, L, Lp, out_stride, vec);
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
