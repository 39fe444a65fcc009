// Launch interface of the gfx950 kernels (kernels.hip).  Host code (model.cpp) sees only this.
//
// Activation layout everywhere: channel-first, time-contiguous, batched over the G units of a pass:
//   x[c][u][t]  at  x + c*(G*Lp) + u*Lp + t     (Lp = L rounded up to 4 floats)
// so consecutive lanes of a wavefront touch consecutive time steps (coalesced), a causal tap is a
// shift along the contiguous axis, and the MFMA "position" index lives on the lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hipvoc {

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_CLAMP1 = 2,
           // encoder path only (fp32 kernel, ENC variant): applied AFTER the residual add
           ACT_ELU = 3, ACT_GELU_TANH = 4 };

// Implicit-GEMM convolution family:  y[n][m] = epi( sum_tap sum_ci W[tap][ci][n] * pro(x[ci][m + shift(tap)]) )
// rows n = output channel (x S output phases for transposed convs), columns m = time positions.
struct ConvGemmArgs {
    const float *x;
    float *y;
    const float *w;       // packed [ntaps][CinP][NP], CinP = roundup(Cin,32), NP = roundup(N, 32*TN)
    const float *bias;    // [N/S] or null
    const float *res;     // residual (layout of y) or null
    const float *rscale;  // [N/S] scale on (acc+bias) before the residual add, or null
    const float *sn_a;    // SnakeBeta prologue: exp(alpha)[Cin], or null
    const float *sn_ib;   //                     1/(exp(beta)+1e-9)[Cin]
    // fused residual unit (w2 != null): y = res + W2 * snake2(conv(snake(x)) + bias) + bias2
    const float *w2;      // packed 1x1 conv [1][roundup(N,32)][NP]
    const float *bias2;   // [N] or null
    const float *sn2_a;   // [N]
    const float *sn2_ib;  // [N]
    // optional SnakeBeta applied to the OUTPUT (after bias/residual): the activation that opens the next
    // decoder block, evaluated once here instead of once per row tile of the consumer
    const float *psn_a;   // [N] or null
    const float *psn_ib;
    float *y2;            // with psn_*: y keeps the raw value and y2 (layout of y) gets the activated one
    // bf16 conv-GEMM mode (kernels_bf16.hip): when wb != null the operands are bf16 (fp32 accumulate) and
    // w / w2 are unused.  wb: [ntaps][ceil(Cin/32)][4][NP][8] bf16; wb2: fused 1x1 conv, [NP/32 output tiles]
    // [2*NP/32 k-steps][2][32][8] bf16 in the k order of the accumulator registers (hipvoc.cpp)
    const unsigned short *wb;
    const unsigned short *wb2;
    // split-operand mode (HIPVOC_DTYPE_BF16X3): the "mid" planes bf16(w - hi) of wb / wb2, same layouts; non-null wbm
    // selects conv_gemm_bf16x3_kernel (three MFMAs per product: hi*hi + hi*mid + mid*hi), fp32 exchange only
    const unsigned short *wbm;
    const unsigned short *wb2m;
    // bf16 mode, packed activations (optional): tensors that are only ever consumed as GEMM operands are exchanged
    // as bf16 in channel-group-major form -- 16-byte element (c/8, u, t) at index (c/8)*cs + u*us + t, i.e. the
    // strides of the fp32 layout counted in 16-byte elements -- with the consumer's SnakeBeta already applied by
    // the producer.  xb: this conv's input in that form (x is then unused; needs Cin % 8 == 0, no sn_*).
    // yb: also write act(out) in that form, act = psn_* if given else identity (y then keeps the RAW fp32 value and
    // may be null when nobody needs it; y2 is not used together with yb; needs (N/S) % 8 == 0).
    const unsigned short *xb;
    unsigned short *yb;
    // fp32 mode, causal k = 7 convs: the Winograd F(2,7)-transformed weights [8 points][CinP][NP] (kernels_wino.hip); when
    // set and the layer qualifies (wino_selected) the conv runs on wino_conv7_kernel -- at EVERY launch size, so that
    // batch == single holds bit for bit -- else on the direct kernels from `w`
    const float *ww;
    const float *zeros;              // >= 16 bytes of zeros in device memory (wino_conv7_kernel: source of the causal left padding)
    int Cin, N, NP, TN;
    int ntaps, shift0, shift_step;   // shift(tap) = shift0 + tap*shift_step
    int S;                           // out channel = n / S, out position = m*S + n % S
    int pm;                          // S > 1, 1: phase-major rows instead: out channel = n % (N/S), phase = n / (N/S); needs (N/S) % 8 == 0
                                     // 2 (S = 4 or 8, (N/S) % 8 == 0): octet-interleaved rows -- see tconv_oi_row() in device_common.h
                                     // 3 (S = 3 or 5, (N/S) % 32 == 0, TN = S): phase-tile rows -- see tconv_pt_epilogue() there
    int units;                       // G (1 in flat mode)
    int M;                           // GEMM columns per unit
    int Lin;                         // valid input positions per unit
    long x_us, x_cs, y_us, y_cs;     // unit / channel strides in floats
    int act;
    int in_elu;                      // encoder path (fp32 kernel, ENC variant): ELU(alpha=1) on the input in the staging loop
    int s2d;                         // encoder path, > 1: store the output in space-to-depth form for a stride-s2d conv:
                                     // element (c, m) goes to channel c*s2d + m % s2d, position m / s2d (y_cs / y_us are the
                                     // strides of THAT tensor; M must be a multiple of s2d); the residual keeps x's strides
    int dbg;                         // timing-only ablation bits (set by the launcher from HIPVOC_DBG; 0 in production)
    // bf16 mode, residual stream in bf16 (HIPVOC_RES_BF16, hipvoc.cpp): the RAW output of this step is the residual stream of a
    // decoder block and is rounded to bf16 where it is written.  1: y / res are fp32 tensors that hold the rounded values (debug
    // plans, the fp32 exchange); 2: y and res POINT TO packed bf16 tensors in the layout and strides of yb (the float types of the
    // two fields are then nominal; needs yb, (N/S) % 8 == 0, S == 1 or pm != 0).  Same values either way, bit for bit.
    int rbf;
    // ... and of the residual READ by this step (res): 2 = packed bf16 tensor, else fp32.  (Apart from the units that open the
    // block that keeps its stream in registers, rbi == 2 goes with rbf == 2.)
    int rbi;
};
hipError_t launch_conv_gemm(const ConvGemmArgs &a, hipStream_t s);
hipError_t launch_conv_gemm_bf16(const ConvGemmArgs &a, hipStream_t s);
// kernels_gemm1.hip: the small-launch form of the fp32 1-tap GEMMs (bit-identical results); launch_conv_gemm
// dispatches to it when gemm1_selected()
bool gemm1_selected(const ConvGemmArgs &a);
hipError_t launch_gemm1(const ConvGemmArgs &a, hipStream_t s);
const char *gemm1_kernel_name(const ConvGemmArgs &a);
bool convs_selected(const ConvGemmArgs &a);           // multi-tap convs on a small launch (fp32)
hipError_t launch_convs(const ConvGemmArgs &a, hipStream_t s);
const char *convs_kernel_name(const ConvGemmArgs &a);
bool gemm1_bf16_selected(const ConvGemmArgs &a);      // the same for the bf16 mode (launch_conv_gemm_bf16 dispatches)
hipError_t launch_gemm1_bf16(const ConvGemmArgs &a, hipStream_t s);
const char *gemm1_bf16_kernel_name(const ConvGemmArgs &a);
bool convs_bf16_selected(const ConvGemmArgs &a);      // conv7 on a small launch, bf16 mode
hipError_t launch_convs_bf16(const ConvGemmArgs &a, hipStream_t s);
const char *convs_bf16_kernel_name(const ConvGemmArgs &a);
const char *conv_gemm_bf16_kernel_name(const ConvGemmArgs &a);
// kernels_bf16_ring.hip (bf16 mode): conv7 and the transposed convs over a packed operand with both operands staged by LDS-DMA through a ring;
// bit-identical to conv_gemm_bf16_kernel (launch_conv_gemm_bf16 dispatches; HIPVOC_BF16_RING=0 turns it off)
bool conv_bf16_ring_selected(const ConvGemmArgs &a);
hipError_t launch_conv_bf16_ring(const ConvGemmArgs &a, hipStream_t s);
const char *conv_bf16_ring_kernel_name(const ConvGemmArgs &a);
bool wino_selected(const ConvGemmArgs &a);            // kernels_wino.hip: F(2,7) Winograd form of the fp32 k = 7 convs
hipError_t launch_wino_conv7(const ConvGemmArgs &a, hipStream_t s);
const char *wino_kernel_name(const ConvGemmArgs &a);
// Octet-interleaved row order of a transposed conv (ConvGemmArgs::pm == 2; device_common.h: tconv_oi_row is the inverse):
// row of (channel c, phase ph) at stride S = 4 or 8.
inline int tconv_oi_n(int c, int ph, int S) {
    const int sup = c >> 3, j = c & 7, h = ph / (S >> 1), rem = ph - h * (S >> 1), ti = rem >> 1, r = 2 * j + (rem & 1);
    return sup * 8 * S + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
}
bool gemmd_selected(const ConvGemmArgs &a);           // kernels_wino.hip: large 1-tap GEMMs with both operands staged by LDS-DMA
hipError_t launch_gemmd(const ConvGemmArgs &a, hipStream_t s);
const char *gemmd_kernel_name(const ConvGemmArgs &a);
bool gemmds_selected(const ConvGemmArgs &a);          // kernels_wino.hip: the same GEMMs on a small launch (deep LDS-DMA pipeline, 1-2 waves per workgroup)
hipError_t launch_gemmds(const ConvGemmArgs &a, hipStream_t s);
const char *gemmds_kernel_name(const ConvGemmArgs &a);
size_t conv_gemm_bf16_lds_bytes(const ConvGemmArgs &a);
namespace detail {
void conv_gemm_geometry(const ConvGemmArgs &a, int &HL, int &XW, int bm = 256);   // halo (left, multiple of 4) and LDS tile width for bm positions
int conv_gemm_launch_tn(const ConvGemmArgs &a);                     // row-tile height actually launched
}
// kernel symbol family actually launched for these args (profiling labels)
const char *conv_gemm_kernel_name(const ConvGemmArgs &a);
size_t conv_gemm_lds_bytes(const ConvGemmArgs &a);

// kernels_blk3.hip (bf16 mode): the three residual units of a 96-channel decoder block in ONE launch that walks along time
// (dilations 1, 3, 9); bit-identical to three launches of the fused per-unit kernel.
struct ResBlock3Args {
    const unsigned short *xb;      // packed activated operand of unit 0, snake_s1(x): [C/8][G][Lp] 16-byte elements
    const float *x;                // the raw residual stream [C][G][Lp] -- or, with xrb, the packed bf16 tensor of it (layout of xb)
    int xrb;
    float *y;                      // fp32 output (the raw value when yb is given, else psn[2](out) if psn[2] is given), or null with yb
    unsigned short *yb;            // optional packed output psn[2](out)
    const unsigned short *ws;      // the block's weight stream: [3 units][24 slots][6 KB] (hipvoc.cpp: blk3_stream)
    const float *bias1[3], *sn2_a[3], *sn2_ib[3], *bias2[3];
    const float *psn_a[3], *psn_ib[3];   // activation applied to a unit's output: the next unit's, then what follows the block (may be null)
    int dil[3];
    int G, L;                      // units, positions per unit
    long us, cs;                   // unit / channel strides in floats (= in 16-byte elements of the packed tensors)
    int span_tiles, spans_per_unit;      // set by the launcher
};
bool resblock3_bf16_supported(int C, const int *dil);
size_t resblock3_bf16_stream_bytes();
size_t resblock3_bf16_slot_bytes();
hipError_t launch_resblock3_bf16(const ResBlock3Args &a, hipStream_t s);

struct RvqArgs {
    const long long *codes;   // [G][T][nq]
    const float *embed;       // [nq][cb_size][D]  (embed_sum / clamp(usage,1e-5), done on the host)
    const float *wf_t;        // [D][Q] transposed output_proj of the semantic group
    const float *wr_t;        // [D][Q] ... acoustic group
    float *out;               // [Q][G][Lp]
    float *sums;              // optional [2D][G][Lp] raw sums (debug) or null
    int *err;                 // device flag: set to 1 on an out-of-range code
    int nq, cb_size, D, Q, T, Lp, G;
};
hipError_t launch_rvq(const RvqArgs &a, hipStream_t s);

// RMSNorm over channels of x[C][P] (P flat positions, channel stride P): y = w * (x * rsqrt(mean x^2 + eps))
hipError_t launch_rmsnorm(const float *x, float *y, const float *w, int C, long P, float eps, hipStream_t s);
const char *rmsnorm_kernel_name(int C);            // the symbol launch_rmsnorm picks, as rocprof prints it

// Causal sliding-window attention with RoPE.  qkv[3*nh*hd][G][Lp] (rows: q | k | v), out[nh*hd][G][Lp].
hipError_t launch_attention(const float *qkv, float *out, const float *cos_t, const float *sin_t, int nh, int hd,
                            int G, int T, int Lp, int window, hipStream_t s);

const char *attention_kernel_name(int hd, int T, int window);   // kernel symbol launched for these shapes

// g[i] = silu(g[i]) * u[i]  over n elements
hipError_t launch_swiglu(float *g, const float *u, long n, hipStream_t s);

// ConvNeXt front half: depthwise causal conv k=K then LayerNorm over channels (eps) with affine.
hipError_t launch_dwconv_ln(const float *x, float *y, const float *dw_w, const float *dw_b, const float *ln_w,
                            const float *ln_b, int C, int K, int G, int L, int Lp, float eps, hipStream_t s);
const char *dwconv_ln_kernel_name(int C, int K);

// Final stage: SnakeBeta -> causal conv C->1 (w[C][K], K == 7; other K: use conv_gemm) -> clamp; out[u*out_stride + t]
hipError_t launch_final_conv(const float *x, float *out, const float *w, const float *bias, const float *sn_a,
                             const float *sn_ib, int C, int K, int G, int L, int Lp, long out_stride, hipStream_t s, bool fast_snake = false,
                             bool round_act = false);
// bf16 mode: the final conv over the packed bf16 tensor of the ACTIVATED input ([C/8][G][Lp] 16-byte elements; C % 8 == 0);
// round_act above is its twin on the raw fp32 tensor (SnakeBeta in the bf16 mode's form, rounded to bf16, same fma chains)
hipError_t launch_final_conv_packed(const unsigned short *xb, float *out, const float *w, const float *bias, int C, int K, int G, int L, int Lp,
                                    long out_stride, hipStream_t s);

// f32 -> i16 PCM: (s*32767.0).clamp(-32768.0, 32767.0) as i16   (src/worker/mod.rs:521-524)
hipError_t launch_pcm16(const float *x, int16_t *y, long n, hipStream_t s);

// ---- speech-tokenizer encoder (kernels_enc.hip; /root/reference/src/speech_tokenizer.rs) -------------------------
// y[(c*r + j)][u][q] = act(x[c][u][q*r + j]) for q < L / r: the space-to-depth form in which a conv of kernel 2r and
// stride r (CausalConv1d with pad k - s = r) is a 2-tap conv over c*r channels.  elu: apply ELU(alpha=1) first.
hipError_t launch_s2d(const float *x, float *y, int C, int G, int L, int Lp, int r, int Lq, int Lqp, int elu, hipStream_t s);
// y[j][u][t] = x[u][t + j - (K-1)] (0 before the start): the taps of a single-channel causal conv as K channels
hipError_t launch_tap_stack(const float *x, float *y, int K, int G, int L, int Lp, hipStream_t s);
// LayerNorm with bias over channels of x[C][P] (speech_tokenizer.rs:394-403): biased variance, eps inside the sqrt
hipError_t launch_layernorm(const float *x, float *y, const float *w, const float *b, int C, long P, float eps, hipStream_t s);
const char *layernorm_kernel_name(int C);
struct RvqEncArgs {
    const float *h;        // latent [dim][G][Lp]
    const float *wp_t;     // [2][dim][D]: transposed input_proj of the semantic, then the acoustic quantiser
    const float *emb_t;    // [nq][D][cb_size]: codebooks, transposed (layer 0.. = semantic layers, then acoustic)
    const float *e_sq;     // [nq][cb_size]
    long long *codes;      // [G][T][nq]
    float *proj;           // optional [2*D][G][Lp]: the projected latents (debug) or null
    int dim, D, cb_size, n_sem, n_ac, T, Lp, G;
};
// Nearest-codeword search, bit-exact contract of the encoder test oracle (k-ascending fmaf chains, first minimum)
hipError_t launch_rvq_encode(const RvqEncArgs &a, hipStream_t s);

}  // namespace hipvoc
