/*
 * voc_oracle.c -- CPU restatement (fp32, plain C + OpenMP) of the qwen3-tts vocoder path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under qwen3-tts_amd/ (the product) may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline.
 *
 * What it restates (reference = /root/reference, TF = transformers 5.15 sibling implementation,
 * see SURVEY.md section 8a/8c):
 *   - chunk loop, zero-code padding, truncate        src/worker/vocoder.rs:60-85
 *   - f32 -> i16 PCM conversion                      src/worker/mod.rs:521-524
 *   - decoder arithmetic (NOT in the reference tree: it is an opaque vocoder.onnx exported by
 *     scripts/export_vocoder.py:25-57 from Qwen3-TTS speech_tokenizer.decoder).  Restated from
 *     the published building blocks:
 *       RVQ decode                                   TF:models/mimi/modeling_mimi.py:979-983,1004-1006,1070-1081,1128-1137
 *       causal Conv1d                                TF:models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3180-3213
 *       causal ConvTranspose1d + trim                TF:...:3216-3228
 *       ConvNeXt block                               TF:...:3231-3263
 *       attention / MLP / RMSNorm / LayerScale       TF:...:3267-3450, eager attention TF:...:484-506
 *       RoPE                                         TF:...:860-864,1400-1422,2400-2435
 *       sliding-window causal mask                   TF:masking_utils.py:92-99,134-138
 *       SnakeBeta                                    TF:...:3566-3580
 *       residual unit / decoder block                TF:...:3592-3633
 *       decoder assembly, clamp                      TF:...:3660-3684 (+ quantizer / pre_conv /
 *                                                    input_proj / output_proj of the Qwen3-TTS
 *                                                    12 Hz tokenizer decoder)
 *
 * PARITY STATUS: the reference holds no golden vectors for this path and its arithmetic (ONNX
 * Runtime + vocoder.onnx) is absent offline, so parity against the shipped ONNX graph is
 * UNPINNED.  This restatement IS pinned against the transformers building blocks listed above
 * (tests/golden/make_golden.py generated the tests/golden .npz fixtures from them with seeded weights;
 * tests/test_oracle_golden.py checks this file against those fixtures).
 *
 * Layout: every activation is channel-first [C][L] row-major (PyTorch NCL with N=1).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_TENSORS 1024
#define MAX_NAME 160
#define VOC_NSTAGE_MAX 64

typedef struct {
    char name[MAX_NAME];
    const float *p;
    int64_t shape[4];
    int nd;
} Tensor;

typedef struct VocOracle {
    Tensor t[MAX_TENSORS];
    int nt;
    /* hyper-parameters that tensor shapes cannot tell (defaults = Qwen3-TTS 12Hz tokenizer) */
    int n_heads;
    int sliding_window;
    float rope_theta;
    float rms_eps;
    /* 1: emulate the bf16 conv-GEMM mode of the HIP library: the input and the weights of every conv / linear /
     * transposed conv with more than one output channel are rounded to bf16 (nearest even), products and sums
     * stay fp32.  0 (default): the reference's fp32 arithmetic. */
    int bf16_gemm;
    int trim_right_only; /* trans-conv trim policy, see transconv_causal (default 0 = both sides) */
    /* derived */
    int finalized;
    int n_q, cb_size, cb_dim, q_out, latent, hidden, n_layers, qkv_dim, head_dim, inter;
    int n_up, up_ratio[8];
    int dec_dim, n_blocks, rates[8];
    float *embed[64]; /* embed_sum / clamp(cluster_usage, 1e-5), per codebook */
    char err[256];
} VocOracle;

/* ------------------------------------------------------------------------------------------ */
VocOracle *voc_oracle_new(void) {
    VocOracle *o = (VocOracle *)calloc(1, sizeof(VocOracle));
    o->n_heads = 16;
    o->sliding_window = 72;
    o->rope_theta = 10000.0f;
    o->rms_eps = 1e-5f;
    return o;
}

void voc_oracle_free(VocOracle *o) {
    if (!o) return;
    for (int i = 0; i < 64; i++) free(o->embed[i]);
    free(o);
}

const char *voc_oracle_error(VocOracle *o) { return o->err; }

int voc_oracle_set_param(VocOracle *o, const char *name, double v) {
    if (!strcmp(name, "num_attention_heads")) o->n_heads = (int)v;
    else if (!strcmp(name, "sliding_window")) o->sliding_window = (int)v;
    else if (!strcmp(name, "rope_theta")) o->rope_theta = (float)v;
    else if (!strcmp(name, "rms_norm_eps")) o->rms_eps = (float)v;
    else if (!strcmp(name, "bf16_gemm")) o->bf16_gemm = v != 0;
    else if (!strcmp(name, "transconv_trim_right_only")) o->trim_right_only = v != 0;
    else return -1;
    return 0;
}

/* Borrows `data` (caller keeps it alive for the oracle's lifetime). */
int voc_oracle_set_tensor(VocOracle *o, const char *name, const float *data, const int64_t *shape, int nd) {
    if (o->nt >= MAX_TENSORS || nd > 4 || strlen(name) >= MAX_NAME) return -1;
    Tensor *t = &o->t[o->nt++];
    strcpy(t->name, name);
    t->p = data;
    t->nd = nd;
    for (int i = 0; i < 4; i++) t->shape[i] = i < nd ? shape[i] : 1;
    return 0;
}

static const Tensor *find(const VocOracle *o, const char *name) {
    for (int i = 0; i < o->nt; i++)
        if (!strcmp(o->t[i].name, name)) return &o->t[i];
    return NULL;
}

static const Tensor *need(VocOracle *o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static const Tensor *need(VocOracle *o, const char *fmt, ...) {
    char name[MAX_NAME];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    const Tensor *t = find(o, name);
    if (!t && !o->err[0]) snprintf(o->err, sizeof o->err, "missing tensor %s", name);
    return t;
}

static int has(const VocOracle *o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static int has(const VocOracle *o, const char *fmt, ...) {
    char name[MAX_NAME];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    return find(o, name) != NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* Derive the architecture from tensor shapes; precompute the codebook embeddings
 * (TF:mimi:979-983: embed = embed_sum / cluster_usage.clamp(min=1e-5)[:, None]). */
int voc_oracle_finalize(VocOracle *o) {
    o->err[0] = 0;
    const Tensor *e0 = need(o, "decoder.quantizer.rvq_first.vq.layers.0._codebook.embedding_sum");
    if (!e0) return -1;
    o->cb_size = (int)e0->shape[0];
    o->cb_dim = (int)e0->shape[1];
    int nrest = 0;
    while (has(o, "decoder.quantizer.rvq_rest.vq.layers.%d._codebook.embedding_sum", nrest)) nrest++;
    o->n_q = 1 + nrest;
    if (o->n_q > 64) return -1;
    for (int q = 0; q < o->n_q; q++) {
        const Tensor *es, *cu;
        if (q == 0) {
            es = need(o, "decoder.quantizer.rvq_first.vq.layers.0._codebook.embedding_sum");
            cu = need(o, "decoder.quantizer.rvq_first.vq.layers.0._codebook.cluster_usage");
        } else {
            es = need(o, "decoder.quantizer.rvq_rest.vq.layers.%d._codebook.embedding_sum", q - 1);
            cu = need(o, "decoder.quantizer.rvq_rest.vq.layers.%d._codebook.cluster_usage", q - 1);
        }
        if (!es || !cu) return -1;
        free(o->embed[q]);
        o->embed[q] = (float *)malloc(sizeof(float) * o->cb_size * o->cb_dim);
        for (int i = 0; i < o->cb_size; i++) {
            float u = cu->p[i];
            if (u < 1e-5f) u = 1e-5f;
            for (int d = 0; d < o->cb_dim; d++) o->embed[q][i * o->cb_dim + d] = es->p[i * o->cb_dim + d] / u;
        }
    }
    const Tensor *t;
    if (!(t = need(o, "decoder.quantizer.rvq_first.output_proj.weight"))) return -1;
    o->q_out = (int)t->shape[0];
    if (!(t = need(o, "decoder.pre_conv.conv.weight"))) return -1;
    o->latent = (int)t->shape[0];
    if (!(t = need(o, "decoder.pre_transformer.input_proj.weight"))) return -1;
    o->hidden = (int)t->shape[0];
    o->n_layers = 0;
    while (has(o, "decoder.pre_transformer.layers.%d.self_attn.q_proj.weight", o->n_layers)) o->n_layers++;
    if (o->n_layers) {
        t = need(o, "decoder.pre_transformer.layers.0.self_attn.q_proj.weight");
        o->qkv_dim = (int)t->shape[0];
        o->head_dim = o->qkv_dim / o->n_heads;
        t = need(o, "decoder.pre_transformer.layers.0.mlp.gate_proj.weight");
        if (!t) return -1;
        o->inter = (int)t->shape[0];
    }
    o->n_up = 0;
    while (has(o, "decoder.upsample.%d.0.conv.weight", o->n_up)) {
        t = need(o, "decoder.upsample.%d.0.conv.weight", o->n_up);
        o->up_ratio[o->n_up] = (int)t->shape[2];
        o->n_up++;
    }
    if (!(t = need(o, "decoder.decoder.0.conv.weight"))) return -1;
    o->dec_dim = (int)t->shape[0];
    o->n_blocks = 0;
    while (has(o, "decoder.decoder.%d.block.1.conv.weight", o->n_blocks + 1)) {
        t = need(o, "decoder.decoder.%d.block.1.conv.weight", o->n_blocks + 1);
        o->rates[o->n_blocks] = (int)t->shape[2] / 2;
        o->n_blocks++;
    }
    o->finalized = 1;
    return o->err[0] ? -1 : 0;
}

/* frames -> samples: every trans-conv with k=2r,s=r trims r on both sides (TF:3221-3228): (L-1)*r; with the
 * right-only policy (see transconv_causal) it keeps L*r */
int64_t voc_oracle_out_len(const VocOracle *o, int T) {
    int64_t L = T;
    for (int i = 0; i < o->n_up; i++) L *= o->up_ratio[i];
    for (int i = 0; i < o->n_blocks; i++) L = o->trim_right_only ? L * o->rates[i] : (L - 1) * o->rates[i];
    return L;
}

/* ------------------------------------------------------------------------------------------ */
/* Primitive ops, channel-first [C][L].                                                        */

static int g_bf16 = 0; /* set from o->bf16_gemm on entry of decoder_forward (the oracle is not re-entrant) */

static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
static float *bf16_copy(const float *x, int64_t n) {
    float *r = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) r[i] = bf16_round(x[i]);
    return r;
}

/* Causal Conv1d, stride 1, groups=1 (TF:3180-3213): left pad (K-1)*dil zeros, no right pad.
 * w[Cout][Cin][K].  y[co][t] = b[co] + sum_ci sum_j w[co][ci][j] * x[ci][t-(K-1-j)*dil]. */
static void conv1d_causal_f32(const float *x, int Cin, int64_t L, const float *w, const float *b, int Cout, int K,
                              int dil, float *y);
static void conv1d_causal(const float *x, int Cin, int64_t L, const float *w, const float *b, int Cout, int K,
                          int dil, float *y) {
    if (g_bf16 && Cout > 1) {
        float *xr = bf16_copy(x, (int64_t)Cin * L), *wr = bf16_copy(w, (int64_t)Cout * Cin * K);
        conv1d_causal_f32(xr, Cin, L, wr, b, Cout, K, dil, y);
        free(xr); free(wr);
        return;
    }
    conv1d_causal_f32(x, Cin, L, w, b, Cout, K, dil, y);
}
static void conv1d_causal_f32(const float *x, int Cin, int64_t L, const float *w, const float *b, int Cout, int K,
                              int dil, float *y) {
    enum { TB = 512, CB = 4 };
    int64_t ntb = (L + TB - 1) / TB;
    int ncb = (Cout + CB - 1) / CB;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int cb = 0; cb < ncb; cb++)
        for (int64_t tb = 0; tb < ntb; tb++) {
            float acc[CB][TB];
            int64_t t0 = tb * TB, t1 = t0 + TB < L ? t0 + TB : L;
            int co0 = cb * CB, nco = co0 + CB <= Cout ? CB : Cout - co0;
            for (int c = 0; c < nco; c++) {
                float bv = b ? b[co0 + c] : 0.0f;
                for (int64_t t = 0; t < t1 - t0; t++) acc[c][t] = bv;
            }
            for (int ci = 0; ci < Cin; ci++) {
                const float *xr = x + (int64_t)ci * L;
                for (int j = 0; j < K; j++) {
                    int64_t sh = (int64_t)(K - 1 - j) * dil;
                    int64_t lo = t0 > sh ? t0 : sh;
                    if (lo >= t1) continue;
                    if (nco == CB) {
                        float w0 = w[((int64_t)(co0 + 0) * Cin + ci) * K + j];
                        float w1 = w[((int64_t)(co0 + 1) * Cin + ci) * K + j];
                        float w2 = w[((int64_t)(co0 + 2) * Cin + ci) * K + j];
                        float w3 = w[((int64_t)(co0 + 3) * Cin + ci) * K + j];
                        float *a0 = acc[0] - t0, *a1 = acc[1] - t0, *a2 = acc[2] - t0, *a3 = acc[3] - t0;
#pragma omp simd
                        for (int64_t t = lo; t < t1; t++) {
                            float xv = xr[t - sh];
                            a0[t] += w0 * xv;
                            a1[t] += w1 * xv;
                            a2[t] += w2 * xv;
                            a3[t] += w3 * xv;
                        }
                    } else {
                        for (int c = 0; c < nco; c++) {
                            float wv = w[((int64_t)(co0 + c) * Cin + ci) * K + j];
                            float *a = acc[c] - t0;
#pragma omp simd
                            for (int64_t t = lo; t < t1; t++) a[t] += wv * xr[t - sh];
                        }
                    }
                }
            }
            for (int c = 0; c < nco; c++) memcpy(y + (int64_t)(co0 + c) * L + t0, acc[c], sizeof(float) * (t1 - t0));
        }
}

/* Depthwise causal conv (groups = C), w[C][1][K] (ConvNeXt dwconv, TF:3234-3240). */
static void dwconv_causal(const float *x, int C, int64_t L, const float *w, const float *b, int K, float *y) {
#pragma omp parallel for
    for (int c = 0; c < C; c++)
        for (int64_t t = 0; t < L; t++) {
            float a = b[c];
            for (int j = 0; j < K; j++) {
                int64_t s = t - (K - 1 - j);
                if (s >= 0) a += w[c * K + j] * x[(int64_t)c * L + s];
            }
            y[(int64_t)c * L + t] = a;
        }
}

/* ConvTranspose1d(k, stride s) + trim of pad = k - s samples.  w[Cin][Cout][K].
 * full[co][i*s+j] = b[co] + sum_ci x[ci][i]*w[ci][co][j]
 * Trim policy (which one the shipped vocoder.onnx has cannot be verified offline, so it is a switch):
 *   both sides (default; the sibling Qwen3OmniMoeCausalTransConvNet, TF:3216-3228): y = full[:, pad : Lfull - pad],
 *     length (L-1)*s -- hence 1920*T - 555 samples per unit;
 *   right only (`transconv_trim_right_only`; Mimi's causal MimiConvTranspose1d with trim_right_ratio = 1,
 *     TF:models/mimi/modeling_mimi.py): y = full[:, 0 : Lfull - pad], length L*s -- exactly 1920*T samples per unit. */
static int g_trim_right = 0; /* set from o->trim_right_only on entry of decoder_forward */
static int64_t transconv_len(int64_t L, int K, int s) { return g_trim_right ? (L - 1) * s + K - (K - s) : (L - 1) * s + K - 2 * (K - s); }

static void transconv_causal(const float *x_in, int Cin, int64_t L, const float *w_in, const float *b, int Cout, int K,
                             int s, float *y) {
    float *xr = g_bf16 ? bf16_copy(x_in, (int64_t)Cin * L) : NULL, *wr = g_bf16 ? bf16_copy(w_in, (int64_t)Cin * Cout * K) : NULL;
    const float *x = xr ? xr : x_in, *w = wr ? wr : w_in;
    int pad = K - s;
    int64_t Lf = (L - 1) * s + K, Lo = g_trim_right ? Lf - pad : Lf - 2 * pad;
#pragma omp parallel
    {
        float *full = (float *)malloc(sizeof(float) * Lf);
        float *tmp = (float *)malloc(sizeof(float) * L);
#pragma omp for schedule(dynamic, 1)
        for (int co = 0; co < Cout; co++) {
            for (int64_t t = 0; t < Lf; t++) full[t] = b ? b[co] : 0.0f;
            for (int j = 0; j < K; j++) {
                for (int64_t i = 0; i < L; i++) tmp[i] = 0.0f;
                for (int ci = 0; ci < Cin; ci++) {
                    float wv = w[((int64_t)ci * Cout + co) * K + j];
                    const float *xr = x + (int64_t)ci * L;
#pragma omp simd
                    for (int64_t i = 0; i < L; i++) tmp[i] += wv * xr[i];
                }
                for (int64_t i = 0; i < L; i++) full[i * s + j] += tmp[i];
            }
            memcpy(y + (int64_t)co * Lo, full + (g_trim_right ? 0 : pad), sizeof(float) * Lo);
        }
        free(full);
        free(tmp);
    }
    free(xr); free(wr);
}

/* SnakeBeta (TF:3566-3580): x + (1/(exp(beta)+1e-9)) * sin(x*exp(alpha))^2, per channel. */
static void snake_beta(float *x, int C, int64_t L, const float *alpha, const float *beta) {
#pragma omp parallel for
    for (int c = 0; c < C; c++) {
        float a = expf(alpha[c]);
        float ib = 1.0f / (expf(beta[c]) + 0.000000001f);
        float *r = x + (int64_t)c * L;
        for (int64_t t = 0; t < L; t++) {
            float s = sinf(r[t] * a);
            r[t] = r[t] + ib * (s * s);
        }
    }
}

/* RMSNorm over channels (TF:3356-3370): w * (x * rsqrt(mean(x^2)+eps)). */
static void rmsnorm(const float *x, int C, int64_t L, const float *w, float eps, float *y) {
#pragma omp parallel for
    for (int64_t t = 0; t < L; t++) {
        float ss = 0.0f;
        for (int c = 0; c < C; c++) ss += x[(int64_t)c * L + t] * x[(int64_t)c * L + t];
        float r = 1.0f / sqrtf(ss / (float)C + eps);
        for (int c = 0; c < C; c++) y[(int64_t)c * L + t] = w[c] * (x[(int64_t)c * L + t] * r);
    }
}

/* LayerNorm over channels (nn.LayerNorm(dim, eps=1e-6), TF:3241,3252), biased variance. */
static void layernorm(const float *x, int C, int64_t L, const float *w, const float *b, float eps, float *y) {
#pragma omp parallel for
    for (int64_t t = 0; t < L; t++) {
        float m = 0.0f;
        for (int c = 0; c < C; c++) m += x[(int64_t)c * L + t];
        m /= (float)C;
        float v = 0.0f;
        for (int c = 0; c < C; c++) {
            float d = x[(int64_t)c * L + t] - m;
            v += d * d;
        }
        float r = 1.0f / sqrtf(v / (float)C + eps);
        for (int c = 0; c < C; c++) y[(int64_t)c * L + t] = (x[(int64_t)c * L + t] - m) * r * w[c] + b[c];
    }
}

static inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static inline float silu(float x) { return x / (1.0f + expf(-x)); }

/* ------------------------------------------------------------------------------------------ */
/* RVQ dequantisation.  codes[T][n_q] (wire order, vocoder.rs:166 feeds [1,T,16]).
 * sums[2][cb_dim][T]: group 0 = semantic (codebook 0), group 1 = acoustic: ((0+q1)+q2)+... in
 * codebook order (TF:mimi:1072-1077).  out[q_out][T] = proj_first(sum0) + proj_rest(sum1), each 1x1
 * projection a k-ascending fmaf chain from 0 (bit-exact contract shared with the HIP kernel). */
static int rvq_decode(VocOracle *o, const int64_t *codes, int T, float *sums, float *out) {
    const Tensor *pf = need(o, "decoder.quantizer.rvq_first.output_proj.weight");
    const Tensor *pr = need(o, "decoder.quantizer.rvq_rest.output_proj.weight");
    if (!pf || !pr) return -1;
    int D = o->cb_dim;
    for (int t = 0; t < T; t++)
        for (int q = 0; q < o->n_q; q++) {
            int64_t c = codes[(int64_t)t * o->n_q + q];
            if (c < 0 || c >= o->cb_size) {
                snprintf(o->err, sizeof o->err, "code %lld out of range at frame %d book %d", (long long)c, t, q);
                return -2;
            }
        }
    for (int t = 0; t < T; t++)
        for (int d = 0; d < D; d++) {
            float s0 = 0.0f + o->embed[0][codes[(int64_t)t * o->n_q] * D + d];
            float s1 = 0.0f;
            for (int q = 1; q < o->n_q; q++) s1 = s1 + o->embed[q][codes[(int64_t)t * o->n_q + q] * D + d];
            sums[(int64_t)d * T + t] = s0;
            sums[(int64_t)(D + d) * T + t] = s1;
        }
    for (int n = 0; n < o->q_out; n++)
        for (int t = 0; t < T; t++) {
            float a = 0.0f, b = 0.0f;
            for (int k = 0; k < D; k++) a = fmaf(pf->p[n * D + k], sums[(int64_t)k * T + t], a);
            for (int k = 0; k < D; k++) b = fmaf(pr->p[n * D + k], sums[(int64_t)(D + k) * T + t], b);
            out[(int64_t)n * T + t] = a + b;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int want;        /* stage id to capture, or -1 */
    float *out;      /* capture buffer */
    size_t cap;
    int C;
    int64_t L;
    int got;
} Tap;

static void tap(Tap *tp, int id, const float *x, int C, int64_t L) {
    if (!tp || tp->want != id) return;
    tp->C = C;
    tp->L = L;
    tp->got = 1;
    size_t n = (size_t)C * (size_t)L;
    if (n > tp->cap) n = tp->cap;
    memcpy(tp->out, x, n * sizeof(float));
}

/* Transformer layer on h[hidden][T] in place (TF:3391-3450). */
static int transformer_layer(VocOracle *o, int li, float *h, int T, const float *cs, const float *sn) {
    int H = o->hidden, Q = o->qkv_dim, hd = o->head_dim, nh = o->n_heads, I = o->inter;
    const Tensor *ln1 = need(o, "decoder.pre_transformer.layers.%d.input_layernorm.weight", li);
    const Tensor *ln2 = need(o, "decoder.pre_transformer.layers.%d.post_attention_layernorm.weight", li);
    const Tensor *wq = need(o, "decoder.pre_transformer.layers.%d.self_attn.q_proj.weight", li);
    const Tensor *wk = need(o, "decoder.pre_transformer.layers.%d.self_attn.k_proj.weight", li);
    const Tensor *wv = need(o, "decoder.pre_transformer.layers.%d.self_attn.v_proj.weight", li);
    const Tensor *wo = need(o, "decoder.pre_transformer.layers.%d.self_attn.o_proj.weight", li);
    const Tensor *ls1 = need(o, "decoder.pre_transformer.layers.%d.self_attn_layer_scale.scale", li);
    const Tensor *ls2 = need(o, "decoder.pre_transformer.layers.%d.mlp_layer_scale.scale", li);
    const Tensor *wg = need(o, "decoder.pre_transformer.layers.%d.mlp.gate_proj.weight", li);
    const Tensor *wu = need(o, "decoder.pre_transformer.layers.%d.mlp.up_proj.weight", li);
    const Tensor *wd = need(o, "decoder.pre_transformer.layers.%d.mlp.down_proj.weight", li);
    if (!ln1 || !ln2 || !wq || !wk || !wv || !wo || !ls1 || !ls2 || !wg || !wu || !wd) return -1;

    float *xn = (float *)malloc(sizeof(float) * H * T);
    float *q = (float *)malloc(sizeof(float) * Q * T);
    float *k = (float *)malloc(sizeof(float) * Q * T);
    float *v = (float *)malloc(sizeof(float) * Q * T);
    float *ao = (float *)malloc(sizeof(float) * Q * T);
    float *proj = (float *)malloc(sizeof(float) * H * T);

    rmsnorm(h, H, T, ln1->p, o->rms_eps, xn);
    conv1d_causal(xn, H, T, wq->p, NULL, Q, 1, 1, q);
    conv1d_causal(xn, H, T, wk->p, NULL, Q, 1, 1, k);
    conv1d_causal(xn, H, T, wv->p, NULL, Q, 1, 1, v);
    /* RoPE, rotate_half form (TF:860-864,1418-1422): cos/sin index d % (hd/2) */
    int half = hd / 2;
    for (int hh = 0; hh < nh; hh++)
        for (int t = 0; t < T; t++)
            for (int d = 0; d < half; d++) {
                float c = cs[t * half + d], s = sn[t * half + d];
                float *ql = &q[(int64_t)(hh * hd + d) * T + t], *qh = &q[(int64_t)(hh * hd + d + half) * T + t];
                float *kl = &k[(int64_t)(hh * hd + d) * T + t], *kh = &k[(int64_t)(hh * hd + d + half) * T + t];
                float a = *ql, b = *qh;
                *ql = a * c + (-b) * s;
                *qh = b * c + a * s;
                a = *kl, b = *kh;
                *kl = a * c + (-b) * s;
                *kh = b * c + a * s;
            }
    /* causal sliding-window attention (mask: kv <= q && kv > q - window), fp32 softmax */
    float scaling = 1.0f / sqrtf((float)hd);
#pragma omp parallel for collapse(2)
    for (int hh = 0; hh < nh; hh++)
        for (int tq = 0; tq < T; tq++) {
            float sc[4096];
            int lo = tq - o->sliding_window + 1;
            if (lo < 0) lo = 0;
            float mx = -INFINITY;
            for (int tk = lo; tk <= tq; tk++) {
                float a = 0.0f;
                for (int d = 0; d < hd; d++) a += q[(int64_t)(hh * hd + d) * T + tq] * k[(int64_t)(hh * hd + d) * T + tk];
                a *= scaling;
                sc[tk - lo] = a;
                if (a > mx) mx = a;
            }
            float den = 0.0f;
            for (int tk = lo; tk <= tq; tk++) {
                sc[tk - lo] = expf(sc[tk - lo] - mx);
                den += sc[tk - lo];
            }
            for (int d = 0; d < hd; d++) {
                float a = 0.0f;
                for (int tk = lo; tk <= tq; tk++) a += (sc[tk - lo] / den) * v[(int64_t)(hh * hd + d) * T + tk];
                ao[(int64_t)(hh * hd + d) * T + tq] = a;
            }
        }
    conv1d_causal(ao, Q, T, wo->p, NULL, H, 1, 1, proj);
    for (int c = 0; c < H; c++)
        for (int t = 0; t < T; t++) h[(int64_t)c * T + t] = h[(int64_t)c * T + t] + ls1->p[c] * proj[(int64_t)c * T + t];

    /* MLP: down(silu(gate(x)) * up(x)) (TF:3350-3352) */
    float *g = (float *)malloc(sizeof(float) * I * T);
    float *u = (float *)malloc(sizeof(float) * I * T);
    rmsnorm(h, H, T, ln2->p, o->rms_eps, xn);
    conv1d_causal(xn, H, T, wg->p, NULL, I, 1, 1, g);
    conv1d_causal(xn, H, T, wu->p, NULL, I, 1, 1, u);
    for (int64_t i = 0; i < (int64_t)I * T; i++) g[i] = silu(g[i]) * u[i];
    conv1d_causal(g, I, T, wd->p, NULL, H, 1, 1, proj);
    for (int c = 0; c < H; c++)
        for (int t = 0; t < T; t++) h[(int64_t)c * T + t] = h[(int64_t)c * T + t] + ls2->p[c] * proj[(int64_t)c * T + t];
    free(xn); free(q); free(k); free(v); free(ao); free(proj); free(g); free(u);
    return 0;
}

/* One forward of the decoder on T frames: codes[T][n_q] -> wav[out_len(T)].
 * This is what OnnxVocoder::run (vocoder.rs:164-172) computes for T = max_tokens. */
static int decoder_forward(VocOracle *o, const int64_t *codes, int T, float *wav, Tap *tp) {
    if (!o->finalized) { snprintf(o->err, sizeof o->err, "not finalized"); return -1; }
    if (T < 1 || T > 4096) { snprintf(o->err, sizeof o->err, "bad T"); return -1; }
    o->err[0] = 0;
    g_bf16 = o->bf16_gemm;
    g_trim_right = o->trim_right_only;
    int st = 0;
    /* g1: RVQ */
    float *sums = (float *)malloc(sizeof(float) * 2 * o->cb_dim * T);
    float *x = (float *)malloc(sizeof(float) * o->q_out * T);
    int rc = rvq_decode(o, codes, T, sums, x);
    if (tp && tp->want == 1000) tap(tp, 1000, sums, 2 * o->cb_dim, T); /* raw dequant sums */
    free(sums);
    if (rc) { free(x); return rc; }
    tap(tp, st++, x, o->q_out, T);
    /* g2: pre_conv k=3 */
    const Tensor *w = need(o, "decoder.pre_conv.conv.weight"), *b = need(o, "decoder.pre_conv.conv.bias");
    if (!w || !b) { free(x); return -1; }
    float *y = (float *)malloc(sizeof(float) * o->latent * T);
    conv1d_causal(x, o->q_out, T, w->p, b->p, o->latent, (int)w->shape[2], 1, y);
    free(x);
    tap(tp, st++, y, o->latent, T);
    /* g3: pre-transformer */
    w = need(o, "decoder.pre_transformer.input_proj.weight"); b = need(o, "decoder.pre_transformer.input_proj.bias");
    if (!w || !b) { free(y); return -1; }
    float *h = (float *)malloc(sizeof(float) * o->hidden * T);
    conv1d_causal(y, o->latent, T, w->p, b->p, o->hidden, 1, 1, h);
    tap(tp, st++, h, o->hidden, T);
    int half = o->head_dim / 2;
    float *cs = (float *)malloc(sizeof(float) * T * half), *sn = (float *)malloc(sizeof(float) * T * half);
    for (int d = 0; d < half; d++) {
        /* inv_freq = 1/(theta^(2d/hd)) in fp32 (TF:2419); freqs = inv_freq*pos; cos/sin fp32 */
        float inv = 1.0f / powf(o->rope_theta, (float)(2 * d) / (float)o->head_dim);
        for (int t = 0; t < T; t++) {
            float f = inv * (float)t;
            cs[t * half + d] = cosf(f);
            sn[t * half + d] = sinf(f);
        }
    }
    for (int li = 0; li < o->n_layers; li++) {
        if (transformer_layer(o, li, h, T, cs, sn)) { free(y); free(h); free(cs); free(sn); return -1; }
        tap(tp, st++, h, o->hidden, T);
    }
    free(cs); free(sn);
    const Tensor *nw = need(o, "decoder.pre_transformer.norm.weight");
    w = need(o, "decoder.pre_transformer.output_proj.weight"); b = need(o, "decoder.pre_transformer.output_proj.bias");
    if (!nw || !w || !b) { free(y); free(h); return -1; }
    float *hn = (float *)malloc(sizeof(float) * o->hidden * T);
    rmsnorm(h, o->hidden, T, nw->p, o->rms_eps, hn);
    conv1d_causal(hn, o->hidden, T, w->p, b->p, o->latent, 1, 1, y);
    free(h); free(hn);
    tap(tp, st++, y, o->latent, T);
    /* g4: upsample stages */
    int C = o->latent;
    int64_t L = T;
    x = y;
    for (int i = 0; i < o->n_up; i++) {
        int r = o->up_ratio[i];
        w = need(o, "decoder.upsample.%d.0.conv.weight", i); b = need(o, "decoder.upsample.%d.0.conv.bias", i);
        if (!w || !b) { free(x); return -1; }
        int64_t L2 = transconv_len(L, r, r);
        y = (float *)malloc(sizeof(float) * C * L2);
        transconv_causal(x, C, L, w->p, b->p, C, r, r, y);
        free(x); x = y; L = L2;
        tap(tp, st++, x, C, L);
        /* ConvNeXt (TF:3247-3263) */
        const Tensor *dw = need(o, "decoder.upsample.%d.1.dwconv.conv.weight", i), *db = need(o, "decoder.upsample.%d.1.dwconv.conv.bias", i);
        const Tensor *lw = need(o, "decoder.upsample.%d.1.norm.weight", i), *lb = need(o, "decoder.upsample.%d.1.norm.bias", i);
        const Tensor *w1 = need(o, "decoder.upsample.%d.1.pwconv1.weight", i), *b1 = need(o, "decoder.upsample.%d.1.pwconv1.bias", i);
        const Tensor *w2 = need(o, "decoder.upsample.%d.1.pwconv2.weight", i), *b2 = need(o, "decoder.upsample.%d.1.pwconv2.bias", i);
        const Tensor *gm = need(o, "decoder.upsample.%d.1.gamma", i);
        if (!dw || !db || !lw || !lb || !w1 || !b1 || !w2 || !b2 || !gm) { free(x); return -1; }
        int C4 = (int)w1->shape[0];
        float *d1 = (float *)malloc(sizeof(float) * C * L), *d2 = (float *)malloc(sizeof(float) * C * L);
        float *m = (float *)malloc(sizeof(float) * C4 * L);
        dwconv_causal(x, C, L, dw->p, db->p, (int)dw->shape[2], d1);
        layernorm(d1, C, L, lw->p, lb->p, 1e-6f, d2);
        conv1d_causal(d2, C, L, w1->p, b1->p, C4, 1, 1, m);
        for (int64_t j = 0; j < (int64_t)C4 * L; j++) m[j] = gelu_erf(m[j]);
        conv1d_causal(m, C4, L, w2->p, b2->p, C, 1, 1, d1);
        for (int c = 0; c < C; c++)
            for (int64_t t = 0; t < L; t++) x[(int64_t)c * L + t] = x[(int64_t)c * L + t] + gm->p[c] * d1[(int64_t)c * L + t];
        free(d1); free(d2); free(m);
        tap(tp, st++, x, C, L);
    }
    /* g5: decoder[0] conv k=7 */
    w = need(o, "decoder.decoder.0.conv.weight"); b = need(o, "decoder.decoder.0.conv.bias");
    if (!w || !b) { free(x); return -1; }
    y = (float *)malloc(sizeof(float) * o->dec_dim * L);
    conv1d_causal(x, C, L, w->p, b->p, o->dec_dim, (int)w->shape[2], 1, y);
    free(x); x = y; C = o->dec_dim;
    tap(tp, st++, x, C, L);
    /* g6: decoder blocks */
    static const int dils[3] = {1, 3, 9};
    for (int bi = 0; bi < o->n_blocks; bi++) {
        int di = bi + 1, r = o->rates[bi];
        const Tensor *al = need(o, "decoder.decoder.%d.block.0.alpha", di), *be = need(o, "decoder.decoder.%d.block.0.beta", di);
        w = need(o, "decoder.decoder.%d.block.1.conv.weight", di); b = need(o, "decoder.decoder.%d.block.1.conv.bias", di);
        if (!al || !be || !w || !b) { free(x); return -1; }
        int Co = (int)w->shape[1];
        snake_beta(x, C, L, al->p, be->p);
        int64_t L2 = transconv_len(L, 2 * r, r);
        y = (float *)malloc(sizeof(float) * Co * L2);
        transconv_causal(x, C, L, w->p, b->p, Co, 2 * r, r, y);
        free(x); x = y; C = Co; L = L2;
        tap(tp, st++, x, C, L);
        for (int u = 0; u < 3; u++) {
            const Tensor *a1 = need(o, "decoder.decoder.%d.block.%d.act1.alpha", di, u + 2), *be1 = need(o, "decoder.decoder.%d.block.%d.act1.beta", di, u + 2);
            const Tensor *a2 = need(o, "decoder.decoder.%d.block.%d.act2.alpha", di, u + 2), *be2 = need(o, "decoder.decoder.%d.block.%d.act2.beta", di, u + 2);
            const Tensor *w1 = need(o, "decoder.decoder.%d.block.%d.conv1.conv.weight", di, u + 2), *b1 = need(o, "decoder.decoder.%d.block.%d.conv1.conv.bias", di, u + 2);
            const Tensor *w2 = need(o, "decoder.decoder.%d.block.%d.conv2.conv.weight", di, u + 2), *b2 = need(o, "decoder.decoder.%d.block.%d.conv2.conv.bias", di, u + 2);
            if (!a1 || !be1 || !a2 || !be2 || !w1 || !b1 || !w2 || !b2) { free(x); return -1; }
            float *s1 = (float *)malloc(sizeof(float) * C * L), *s2 = (float *)malloc(sizeof(float) * C * L);
            memcpy(s1, x, sizeof(float) * C * L);
            snake_beta(s1, C, L, a1->p, be1->p);
            conv1d_causal(s1, C, L, w1->p, b1->p, C, (int)w1->shape[2], dils[u], s2);
            snake_beta(s2, C, L, a2->p, be2->p);
            conv1d_causal(s2, C, L, w2->p, b2->p, C, 1, 1, s1);
            for (int64_t j = 0; j < (int64_t)C * L; j++) x[j] = s1[j] + x[j];
            free(s1); free(s2);
            if (u < 2) tap(tp, st++, x, C, L); else tap(tp, st++, x, C, L);
        }
    }
    /* g7: final SnakeBeta, conv 7 -> 1 channel, clamp */
    {
        int di = o->n_blocks + 1;
        const Tensor *al = need(o, "decoder.decoder.%d.alpha", di), *be = need(o, "decoder.decoder.%d.beta", di);
        w = need(o, "decoder.decoder.%d.conv.weight", di + 1); b = need(o, "decoder.decoder.%d.conv.bias", di + 1);
        if (!al || !be || !w || !b) { free(x); return -1; }
        snake_beta(x, C, L, al->p, be->p);
        y = (float *)malloc(sizeof(float) * L);
        conv1d_causal(x, C, L, w->p, b->p, 1, (int)w->shape[2], 1, y);
        for (int64_t t = 0; t < L; t++) {
            float vv = y[t];
            vv = vv < -1.0f ? -1.0f : vv;
            vv = vv > 1.0f ? 1.0f : vv;
            y[t] = vv;
        }
        free(x);
        tap(tp, st++, y, 1, L);
        if (wav) memcpy(wav, y, sizeof(float) * L);
        free(y);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Public API                                                                                  */

int voc_oracle_n_stages(const VocOracle *o) { return 3 + o->n_layers + 1 + 2 * o->n_up + 1 + 4 * o->n_blocks + 1; }


/* Shape of stage `stage` for a T-frame unit (same numbering as decoder_forward's taps). */
int voc_oracle_stage_shape(const VocOracle *o, int T, int stage, int *C, int64_t *L) {
    int st = 0;
    int64_t l = T;
    int c;
#define VOC_ST(cc, ll) do { if (st++ == stage) { *C = (cc); *L = (ll); return 0; } } while (0)
    if (stage == 1000) { *C = 2 * o->cb_dim; *L = T; return 0; }
    VOC_ST(o->q_out, l);
    VOC_ST(o->latent, l);
    VOC_ST(o->hidden, l);
    for (int i = 0; i < o->n_layers; i++) VOC_ST(o->hidden, l);
    VOC_ST(o->latent, l);
    c = o->latent;
    for (int i = 0; i < o->n_up; i++) {
        l *= o->up_ratio[i];
        VOC_ST(c, l);
        VOC_ST(c, l);
    }
    c = o->dec_dim;
    VOC_ST(c, l);
    for (int b = 0; b < o->n_blocks; b++) {
        c /= 2;
        l = o->trim_right_only ? l * o->rates[b] : (l - 1) * o->rates[b];
        for (int u = 0; u < 4; u++) VOC_ST(c, l);
    }
    VOC_ST(1, l);
#undef VOC_ST
    return -4;
}

#ifdef _OPENMP
#include <omp.h>
#endif
/* Known-answer hook: the transposed conv alone under either trim policy (tests/golden/transconv_trim.npz holds the
 * outputs of transformers' Qwen3OmniMoeCausalTransConvNet = both sides and MimiConvTranspose1d = right only).
 * Returns the output length; y may be NULL to query it. */
int64_t voc_oracle_transconv(const float *x, int Cin, int64_t L, const float *w, const float *b, int Cout, int K, int s,
                             int right_only, float *y) {
    int keep_b = g_bf16, keep_t = g_trim_right;
    g_bf16 = 0;
    g_trim_right = right_only != 0;
    int64_t Lo = transconv_len(L, K, s);
    if (y) transconv_causal(x, Cin, L, w, b, Cout, K, s, y);
    g_bf16 = keep_b;
    g_trim_right = keep_t;
    return Lo;
}

/* Number of OpenMP threads the oracle uses from now on (returns what is in effect). */
int voc_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* OnnxVocoder::run equivalent (vocoder.rs:164-172): one unit of exactly T frames, no chunking. */
int voc_oracle_run_unit(VocOracle *o, const int64_t *codes, int T, float *out, size_t cap, size_t *n_out) {
    int64_t L = voc_oracle_out_len(o, T);
    if (L <= 0 || (size_t)L > cap) { snprintf(o->err, sizeof o->err, "output capacity"); return -3; }
    int rc = decoder_forward(o, codes, T, out, NULL);
    if (!rc && n_out) *n_out = (size_t)L;
    return rc;
}

/* Intermediate stage `stage` (0..n_stages-1; 1000 = raw dequant sums [2*cb_dim][T]). */
int voc_oracle_stage(VocOracle *o, const int64_t *codes, int T, int stage, float *out, size_t cap, int *C, int64_t *L) {
    Tap tp = {stage, out, cap, 0, 0, 0};
    int rc = decoder_forward(o, codes, T, NULL, &tp);
    if (rc) return rc;
    if (!tp.got) { snprintf(o->err, sizeof o->err, "no such stage %d", stage); return -4; }
    *C = tp.C;
    *L = tp.L;
    return 0;
}

/* Vocoder::synthesize (vocoder.rs:60-85): independent max_tokens-frame units, zero-CODE padding,
 * keep min(chunk_len*1920, len) samples of each unit, concatenate. */
int voc_oracle_synthesize(VocOracle *o, const int64_t *codes, int n_tokens, int max_tokens, float *out, size_t cap,
                          size_t *n_out) {
    const size_t SAMPLES_PER_TOKEN = 1920; /* vocoder.rs:6, protocol.rs:122 */
    int nq = o->n_q;
    int64_t Lu = voc_oracle_out_len(o, max_tokens);
    int64_t *padded = (int64_t *)malloc(sizeof(int64_t) * max_tokens * nq);
    float *unit = (float *)malloc(sizeof(float) * (size_t)Lu);
    size_t n = 0;
    int rc = 0;
    for (int start = 0; start < n_tokens; start += max_tokens) {
        int end = start + max_tokens < n_tokens ? start + max_tokens : n_tokens;
        int len = end - start;
        memset(padded, 0, sizeof(int64_t) * max_tokens * nq);
        memcpy(padded, codes + (int64_t)start * nq, sizeof(int64_t) * len * nq);
        rc = decoder_forward(o, padded, max_tokens, unit, NULL);
        if (rc) break;
        size_t keep = (size_t)len * SAMPLES_PER_TOKEN;
        if (keep > (size_t)Lu) keep = (size_t)Lu;
        if (n + keep > cap) { snprintf(o->err, sizeof o->err, "output capacity"); rc = -3; break; }
        memcpy(out + n, unit, keep * sizeof(float));
        n += keep;
    }
    free(padded);
    free(unit);
    if (!rc && n_out) *n_out = n;
    return rc;
}

/* handle_vocode's PCM conversion (mod.rs:521-524): (s*32767.0).clamp(-32768.0, 32767.0) as i16.
 * Rust's `as i16` truncates toward zero and maps NaN to 0. */
void voc_oracle_pcm16(const float *x, size_t n, int16_t *out) {
    for (size_t i = 0; i < n; i++) {
        float v = x[i] * 32767.0f;
        if (v != v) { out[i] = 0; continue; }
        if (v < -32768.0f) v = -32768.0f;
        if (v > 32767.0f) v = 32767.0f;
        out[i] = (int16_t)v;
    }
}

/* i64 little-endian bytes -> codes (mod.rs:513-516). */
void voc_oracle_codes_from_le(const uint8_t *bytes, size_t n_codes, int64_t *out) {
    for (size_t i = 0; i < n_codes; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)bytes[i * 8 + b] << (8 * b);
        out[i] = (int64_t)v;
    }
}
