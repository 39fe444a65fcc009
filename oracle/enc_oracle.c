/*
 * CPU restatement of the reference's speech-tokenizer ENCODER: 24 kHz mono waveform ->
 * [n_frames, 16] codec tokens at 12.5 Hz.  TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker of the HIP path; the product never links or calls it.
 *
 * Follows /root/reference/src/speech_tokenizer.rs line by line (cited per function):
 *   SEANet encoder (171-244) -> 8-layer transformer (248-472) -> stride-2 downsample conv (590-604) ->
 *   split residual vector quantiser, encode direction (476-586); top level SpeechTokenizer::encode (646-675).
 * The reference runs this with candle on the CPU; candle's matmul / conv summation ORDER is not specified, so the
 * floating-point part is pinned only to round-off ("parity unpinned" against the Rust binary, which cannot be
 * built here: no cargo).  What this file fixes, and what the HIP path must reproduce bit for bit, is the integer
 * part: the nearest-codeword search given a latent -- distance = (|x|^2 + |e|^2) - 2 * <x, e> evaluated as
 * written at 499-504, with <x, e> a k-ascending fmaf chain, |.|^2 a k-ascending sum of rounded squares, and
 * argmin = first minimum (507).
 *
 * Architecture is derived from tensor shapes (ratios from the down-conv kernel sizes), so narrow test
 * configurations run through the same code; what shapes cannot tell comes in as parameters
 * (n_heads, context, n_use_codebooks, rope_max_period), defaults = the constants at speech_tokenizer.rs:15-33.
 * Layout: channel-first [C][L] row-major unless said otherwise.
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_BLOCKS 8
#define MAX_LAYERS 64
#define MAX_Q 64

typedef struct {
    char name[192];
    const float *data;
    int64_t shape[4];
    int nd;
} Tensor;

typedef struct {
    float *embed; /* [cb_size][cb_dim] = embed_sum / cluster_usage (482-491: plain division, no clamp) */
    float *e_sq;  /* [cb_size] */
} Codebook;

typedef struct EncOracle {
    Tensor *t;
    int nt, cap;
    char err[256];
    /* parameters (speech_tokenizer.rs:24-33) */
    int n_heads, context, n_use;
    double rope_max_period;
    float ln_eps;
    /* derived */
    int n_filters, n_blocks, ratio[MAX_BLOCKS], dim, n_layers, ff, ds_stride;
    int cb_size, cb_dim, n_sem, n_ac;
    Codebook sem[MAX_Q], ac[MAX_Q];
    int final_idx; /* layer index of the final conv inside encoder.encoder.layers */
    int ready;
} EncOracle;

EncOracle *enc_oracle_new(void) {
    EncOracle *o = (EncOracle *)calloc(1, sizeof *o);
    o->n_heads = 8;
    o->context = 250;
    o->n_use = 16;
    o->rope_max_period = 10000.0;
    o->ln_eps = 1e-5f; /* LayerNormBias::load, 390 */
    return o;
}

void enc_oracle_free(EncOracle *o) {
    if (!o) return;
    for (int i = 0; i < MAX_Q; i++) {
        free(o->sem[i].embed); free(o->sem[i].e_sq);
        free(o->ac[i].embed); free(o->ac[i].e_sq);
    }
    free(o->t);
    free(o);
}

const char *enc_oracle_error(EncOracle *o) { return o->err; }

int enc_oracle_set_param(EncOracle *o, const char *name, double v) {
    if (!strcmp(name, "n_heads")) o->n_heads = (int)v;
    else if (!strcmp(name, "context")) o->context = (int)v;
    else if (!strcmp(name, "n_use_codebooks")) o->n_use = (int)v;
    else if (!strcmp(name, "rope_max_period")) o->rope_max_period = v;
    else return -1;
    return 0;
}

/* Borrows `data` (the caller keeps it alive for the oracle's lifetime). */
int enc_oracle_set_tensor(EncOracle *o, const char *name, const float *data, const int64_t *shape, int nd) {
    if (nd > 4 || strlen(name) >= sizeof o->t[0].name) return -1;
    if (o->nt == o->cap) {
        o->cap = o->cap ? 2 * o->cap : 256;
        o->t = (Tensor *)realloc(o->t, sizeof(Tensor) * o->cap);
    }
    Tensor *t = &o->t[o->nt++];
    memset(t, 0, sizeof *t);
    strcpy(t->name, name);
    t->data = data;
    t->nd = nd;
    for (int i = 0; i < nd; i++) t->shape[i] = shape[i];
    return 0;
}

static const Tensor *find(const EncOracle *o, const char *name) {
    for (int i = 0; i < o->nt; i++)
        if (!strcmp(o->t[i].name, name)) return &o->t[i];
    return NULL;
}
static const Tensor *need(EncOracle *o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static const Tensor *need(EncOracle *o, const char *fmt, ...) {
    char name[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    const Tensor *t = find(o, name);
    if (!t && !o->err[0]) snprintf(o->err, sizeof o->err, "missing tensor %s", name);
    return t;
}
static const Tensor *opt(const EncOracle *o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static const Tensor *opt(const EncOracle *o, const char *fmt, ...) {
    char name[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    return find(o, name);
}

#define ENC "encoder.encoder."
#define TRF "encoder.encoder_transformer."
#define QNT "encoder.quantizer."

/* EuclideanCodebook::load (482-491) + the |e|^2 term of encode (501). */
static int load_codebook(EncOracle *o, Codebook *cb, const char *prefix, int i) {
    const Tensor *es = need(o, "%slayers.%d.codebook.embed_sum", prefix, i);
    const Tensor *us = need(o, "%slayers.%d.codebook.cluster_usage", prefix, i);
    if (!es || !us) return -1;
    if (es->nd != 2 || us->nd != 1 || us->shape[0] != es->shape[0]) {
        snprintf(o->err, sizeof o->err, "%slayers.%d.codebook: bad shapes", prefix, i);
        return -1;
    }
    if (!o->cb_size) { o->cb_size = (int)es->shape[0]; o->cb_dim = (int)es->shape[1]; }
    if (es->shape[0] != o->cb_size || es->shape[1] != o->cb_dim) {
        snprintf(o->err, sizeof o->err, "%slayers.%d.codebook: codebooks differ in shape", prefix, i);
        return -1;
    }
    cb->embed = (float *)malloc(sizeof(float) * o->cb_size * o->cb_dim);
    cb->e_sq = (float *)malloc(sizeof(float) * o->cb_size);
    for (int c = 0; c < o->cb_size; c++) {
        float s = 0.0f;
        for (int d = 0; d < o->cb_dim; d++) {
            float e = es->data[(size_t)c * o->cb_dim + d] / us->data[c];
            cb->embed[(size_t)c * o->cb_dim + d] = e;
            float sq = e * e; /* sqr() then sum(1): the squares are rounded before they are added */
            s = s + sq;
        }
        cb->e_sq[c] = s;
    }
    return 0;
}

/* Shapes -> architecture (SEANetEncoder::load 178-232, SpeechTokenizer::load 617-641). */
int enc_oracle_finalize(EncOracle *o) {
    const Tensor *w0 = need(o, ENC "layers.0.conv.weight");
    if (!w0) return -1;
    o->n_filters = (int)w0->shape[0];
    int idx = 1, nb = 0;
    while (nb < MAX_BLOCKS && opt(o, ENC "layers.%d.block.1.conv.weight", idx)) {
        const Tensor *dw = need(o, ENC "layers.%d.conv.weight", idx + 2);
        if (!dw) return -1;
        if (dw->nd != 3 || dw->shape[2] % 2) {
            snprintf(o->err, sizeof o->err, "down conv %d: kernel must be 2*ratio", idx + 2);
            return -1;
        }
        o->ratio[nb++] = (int)dw->shape[2] / 2;
        idx += 3;
    }
    o->n_blocks = nb;
    o->final_idx = idx + 1; /* skip the ELU layer (218) */
    const Tensor *fw = need(o, ENC "layers.%d.conv.weight", o->final_idx);
    if (!fw) return -1;
    o->dim = (int)fw->shape[0];
    int nl = 0;
    while (nl < MAX_LAYERS && opt(o, TRF "layers.%d.self_attn.q_proj.weight", nl)) nl++;
    o->n_layers = nl;
    if (nl) {
        const Tensor *f1 = need(o, TRF "layers.0.mlp.fc1.weight");
        if (!f1) return -1;
        o->ff = (int)f1->shape[0];
    }
    const Tensor *dsw = need(o, "encoder.downsample.conv.weight");
    if (!dsw) return -1;
    o->ds_stride = (int)dsw->shape[2] / 2; /* kernel = 2 * stride (596) */
    if (o->dim % o->n_heads || (o->dim / o->n_heads) % 2) {
        snprintf(o->err, sizeof o->err, "dim %d not divisible into %d even heads", o->dim, o->n_heads);
        return -1;
    }
    int ns = 0, na = 0;
    while (ns < MAX_Q && opt(o, QNT "semantic_residual_vector_quantizer.layers.%d.codebook.embed_sum", ns)) ns++;
    while (na < MAX_Q && opt(o, QNT "acoustic_residual_vector_quantizer.layers.%d.codebook.embed_sum", na)) na++;
    o->n_sem = ns; o->n_ac = na;
    if (!ns) { snprintf(o->err, sizeof o->err, "no semantic codebook"); return -1; }
    for (int i = 0; i < ns; i++)
        if (load_codebook(o, &o->sem[i], QNT "semantic_residual_vector_quantizer.", i)) return -1;
    for (int i = 0; i < na; i++)
        if (load_codebook(o, &o->ac[i], QNT "acoustic_residual_vector_quantizer.", i)) return -1;
    if (!need(o, QNT "semantic_residual_vector_quantizer.input_proj.weight")) return -1;
    if (na && !need(o, QNT "acoustic_residual_vector_quantizer.input_proj.weight")) return -1;
    o->ready = 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Primitives                                                                                 */

/* CausalConv1d::forward (65-79) and ResidualBlock::conv1_dilated_forward (152-166):
 * left-pad `pad` zeros, then a plain conv1d(stride, dilation), no right pad.  w[Cout][Cin][K]. */
static int64_t conv_out_len(int64_t L, int K, int stride, int dil, int pad) {
    int64_t span = (int64_t)(K - 1) * dil + 1;
    if (L + pad < span) return 0;
    return (L + pad - span) / stride + 1;
}
static void conv1d(const float *x, int Cin, int64_t L, const float *w, const float *b, int Cout, int K, int stride,
                   int dil, int pad, float *y, int64_t Lout) {
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; co++) {
        float *yr = y + (size_t)co * Lout;
        for (int64_t t = 0; t < Lout; t++) yr[t] = 0.0f;
        for (int ci = 0; ci < Cin; ci++) {
            const float *xr = x + (size_t)ci * L;
            for (int k = 0; k < K; k++) {
                float wv = w[((size_t)co * Cin + ci) * K + k];
                /* y[t] += w * xpad[t*stride + k*dil], xpad[i] = x[i - pad] */
                int64_t off = (int64_t)k * dil - pad;
                int64_t t0 = 0;
                if (off < 0) t0 = (-off + stride - 1) / stride;
                if (stride == 1) {
                    for (int64_t t = t0; t < Lout; t++) yr[t] = fmaf(wv, xr[t + off], yr[t]);
                } else {
                    for (int64_t t = t0; t < Lout; t++) yr[t] = fmaf(wv, xr[t * stride + off], yr[t]);
                }
            }
        }
        if (b)
            for (int64_t t = 0; t < Lout; t++) yr[t] = yr[t] + b[co]; /* broadcast_add after the conv (75-78) */
    }
}

/* elu (84-92): max(x,0) + alpha*(exp(min(x,0)) - 1), alpha = 1 */
static inline float elu1(float v) {
    float pos = v > 0.0f ? v : 0.0f, neg = v < 0.0f ? v : 0.0f;
    return pos + (expf(neg) - 1.0f) * 1.0f;
}
static void elu(const float *x, size_t n, float *y) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) y[i] = elu1(x[i]);
}

/* LayerNormBias::forward (394-403) over the channel axis of x[D][T]: biased variance, eps inside the sqrt. */
static void layernorm(const float *x, int D, int64_t T, const float *w, const float *b, float eps, float *y) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < T; t++) {
        float s = 0.0f;
        for (int d = 0; d < D; d++) s += x[(size_t)d * T + t];
        float mean = s / (float)D;
        float v = 0.0f;
        for (int d = 0; d < D; d++) {
            float c = x[(size_t)d * T + t] - mean;
            v += c * c;
        }
        float den = sqrtf(v / (float)D + eps);
        for (int d = 0; d < D; d++) y[(size_t)d * T + t] = (x[(size_t)d * T + t] - mean) / den * w[d] + b[d];
    }
}

/* y[N][T] = W[N][D] x[D][T] (broadcast_matmul with the transposed weight, e.g. 280-282) */
static void linear(const float *x, int D, int64_t T, const float *w, int N, float *y) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        float *yr = y + (size_t)n * T;
        for (int64_t t = 0; t < T; t++) yr[t] = 0.0f;
        for (int d = 0; d < D; d++) {
            float wv = w[(size_t)n * D + d];
            const float *xr = x + (size_t)d * T;
            for (int64_t t = 0; t < T; t++) yr[t] = fmaf(wv, xr[t], yr[t]);
        }
    }
}

/* candle's Tensor::gelu() is the tanh approximation (438):
 * 0.5 v (1 + tanh(sqrt(2/pi) v (1 + 0.044715 v^2))) */
static inline float gelu_tanh(float v) {
    const float k = 0.7978845608028654f;
    return 0.5f * v * (1.0f + tanhf(k * v * (1.0f + 0.044715f * v * v)));
}

/* RoPEAttention::forward (276-309) on h[D][T] -> out[D][T] (before o_proj). */
static void attention(const EncOracle *o, const float *q, const float *k, const float *v, int64_t T, float *out) {
    const int H = o->n_heads, hd = o->dim / H, half = hd / 2;
    float *cs = (float *)malloc(sizeof(float) * T * half), *sn = (float *)malloc(sizeof(float) * T * half);
    for (int64_t t = 0; t < T; t++)
        for (int i = 0; i < half; i++) {
            /* apply_rope (311-344): inv_freq computed in f64 and cast, angle = f32 product */
            float inv = (float)(1.0 / pow(o->rope_max_period, 2.0 * (double)i / (double)hd));
            float ang = (float)t * inv;
            cs[t * half + i] = cosf(ang);
            sn[t * half + i] = sinf(ang);
        }
    const float scale = (float)(1.0 / sqrt((double)hd)); /* `/ scale` on a tensor is affine(1/scale) (300-301) */
#pragma omp parallel for schedule(dynamic)
    for (int hh = 0; hh < H; hh++) {
        float *qr = (float *)malloc(sizeof(float) * T * hd), *kr = (float *)malloc(sizeof(float) * T * hd);
        float *p = (float *)malloc(sizeof(float) * T);
        for (int64_t t = 0; t < T; t++)
            for (int i = 0; i < half; i++) {
                float c = cs[t * half + i], s = sn[t * half + i];
                float q1 = q[(size_t)(hh * hd + i) * T + t], q2 = q[(size_t)(hh * hd + half + i) * T + t];
                float k1 = k[(size_t)(hh * hd + i) * T + t], k2 = k[(size_t)(hh * hd + half + i) * T + t];
                qr[t * hd + i] = q1 * c - q2 * s;
                qr[t * hd + half + i] = q1 * s + q2 * c;
                kr[t * hd + i] = k1 * c - k2 * s;
                kr[t * hd + half + i] = k1 * s + k2 * c;
            }
        for (int64_t i = 0; i < T; i++) {
            /* apply_causal_sliding_mask (346-362): keys j in [i - context + 1, i] */
            int64_t start = i >= o->context ? i - o->context + 1 : 0;
            float mx = -INFINITY;
            for (int64_t j = start; j <= i; j++) {
                float acc = 0.0f;
                for (int d = 0; d < hd; d++) acc = fmaf(qr[i * hd + d], kr[j * hd + d], acc);
                p[j] = acc * scale;
                if (p[j] > mx) mx = p[j];
            }
            float den = 0.0f;
            for (int64_t j = start; j <= i; j++) {
                p[j] = expf(p[j] - mx);
                den += p[j];
            }
            for (int d = 0; d < hd; d++) {
                float acc = 0.0f;
                for (int64_t j = start; j <= i; j++) acc = fmaf(p[j] / den, v[(size_t)(hh * hd + d) * T + j], acc);
                out[(size_t)(hh * hd + d) * T + i] = acc;
            }
        }
        free(qr); free(kr); free(p);
    }
    free(cs); free(sn);
}

/* ------------------------------------------------------------------------------------------ */
/* Quantiser, encode direction                                                                 */

/* EuclideanCodebook::encode (494-509) for one frame: x[cb_dim] -> index of the nearest codeword. */
static int nearest(const EncOracle *o, const Codebook *cb, const float *x) {
    const int D = o->cb_dim;
    float x_sq = 0.0f;
    for (int d = 0; d < D; d++) {
        float sq = x[d] * x[d];
        x_sq = x_sq + sq;
    }
    int best = 0;
    float bd = INFINITY;
    for (int c = 0; c < o->cb_size; c++) {
        const float *e = cb->embed + (size_t)c * D;
        float xe = 0.0f;
        for (int d = 0; d < D; d++) xe = fmaf(x[d], e[d], xe);
        float two = xe * 2.0f;
        float dist = (x_sq + cb->e_sq[c]) - two;
        if (dist < bd) { bd = dist; best = c; } /* argmin: first minimum */
    }
    return best;
}

/* ResidualVectorQuantizer::encode (536-556) on an already projected latent proj[cb_dim][T];
 * codes[q * T + t], n_layers rows.  `proj` is consumed (it becomes the residual). */
static void rvq_encode(const EncOracle *o, const Codebook *cbs, int n_layers, float *proj, int64_t T, int64_t *codes) {
    const int D = o->cb_dim;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < T; t++) {
        float *x = (float *)malloc(sizeof(float) * D);
        for (int d = 0; d < D; d++) x[d] = proj[(size_t)d * T + t];
        for (int q = 0; q < n_layers; q++) {
            int c = nearest(o, &cbs[q], x);
            codes[(size_t)q * T + t] = c;
            const float *e = cbs[q].embed + (size_t)c * D;
            for (int d = 0; d < D; d++) x[d] = x[d] - e[d]; /* residual - quantized (552) */
        }
        free(x);
    }
}

/* SplitResidualVectorQuantizer::encode (581-585) + the [..use_q] cut and transpose of SpeechTokenizer::encode
 * (664-668): latent h[dim][T] -> codes_out[T][use_q].  Only the first use_q layers are evaluated (later layers do
 * not influence earlier ones).  psem / pac (optional, [cb_dim][T]): the projected latents. */
static int quantize(EncOracle *o, const float *h, int64_t T, int64_t *codes_out, float *psem, float *pac) {
    const int D = o->cb_dim;
    int use_q = o->n_sem + o->n_ac < o->n_use ? o->n_sem + o->n_ac : o->n_use;
    int use_sem = o->n_sem < use_q ? o->n_sem : use_q, use_ac = use_q - use_sem;
    const Tensor *ps = need(o, QNT "semantic_residual_vector_quantizer.input_proj.weight");
    if (!ps) return -1;
    float *proj = (float *)malloc(sizeof(float) * D * (T ? T : 1));
    int64_t *codes = (int64_t *)malloc(sizeof(int64_t) * use_q * (T ? T : 1));
    linear(h, o->dim, T, ps->data, D, proj); /* 1x1 conv, no bias (538) */
    if (psem) memcpy(psem, proj, sizeof(float) * D * T);
    rvq_encode(o, o->sem, use_sem, proj, T, codes);
    if (use_ac) {
        const Tensor *pa = need(o, QNT "acoustic_residual_vector_quantizer.input_proj.weight");
        if (!pa) { free(proj); free(codes); return -1; }
        linear(h, o->dim, T, pa->data, D, proj);
        if (pac) memcpy(pac, proj, sizeof(float) * D * T);
        rvq_encode(o, o->ac, use_ac, proj, T, codes + (size_t)use_sem * T);
    }
    for (int64_t t = 0; t < T; t++)
        for (int q = 0; q < use_q; q++) codes_out[t * use_q + q] = codes[(size_t)q * T + t];
    free(proj); free(codes);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Forward with stage taps                                                                     */

typedef struct {
    int want;     /* stage to capture, -1 none */
    float *out;
    size_t cap;
    int C;
    int64_t L;
    int hit;
} Tap;

static void tap(Tap *tp, int id, const float *x, int C, int64_t L) {
    if (!tp || tp->want != id) return;
    tp->C = C; tp->L = L; tp->hit = 1;
    size_t n = (size_t)C * L;
    if (n <= tp->cap) memcpy(tp->out, x, n * sizeof(float));
}

/* stage numbering: 0 initial conv; 1+2i residual block i; 2+2i down conv i; 1+2nb final conv;
 * 2+2nb+l transformer layer l; 2+2nb+nl downsample; +1 semantic projection; +2 acoustic projection */
int enc_oracle_n_stages(const EncOracle *o) { return 2 + 2 * o->n_blocks + o->n_layers + 3; }

int enc_oracle_use_q(const EncOracle *o) { return o->n_sem + o->n_ac < o->n_use ? o->n_sem + o->n_ac : o->n_use; }
int enc_oracle_dim(const EncOracle *o) { return o->dim; }

/* frames produced for n samples: each strided conv floors (CausalConv1d::forward with pad = k - s) */
int64_t enc_oracle_n_frames(const EncOracle *o, int64_t n_samples) {
    int64_t L = n_samples;
    for (int i = 0; i < o->n_blocks; i++) L = conv_out_len(L, 2 * o->ratio[i], o->ratio[i], 1, o->ratio[i]);
    return conv_out_len(L, 2 * o->ds_stride, o->ds_stride, 1, o->ds_stride);
}

/* SEANetEncoder::forward (234-243) + EncoderTransformer::forward (463-471) + downsample (601-603):
 * audio[n] -> latent h[dim][T] (malloc'd, returned through *h_out). */
static int encoder_forward(EncOracle *o, const float *audio, int64_t n, float **h_out, int64_t *T_out, Tap *tp) {
    if (!o->ready) { snprintf(o->err, sizeof o->err, "finalize() not called"); return -1; }
    int st = 0;
    const Tensor *w = need(o, ENC "layers.0.conv.weight");
    const Tensor *b = opt(o, ENC "layers.0.conv.bias");
    int C = o->n_filters, K = (int)w->shape[2];
    int64_t L = conv_out_len(n, K, 1, 1, K - 1);
    if (L <= 0) { snprintf(o->err, sizeof o->err, "audio too short"); return -1; }
    float *h = (float *)malloc(sizeof(float) * C * L);
    conv1d(audio, 1, n, w->data, b ? b->data : NULL, C, K, 1, 1, K - 1, h, L);
    tap(tp, st++, h, C, L);
    int idx = 1;
    for (int i = 0; i < o->n_blocks; i++) {
        /* ResidualBlock::forward (139-149): x + conv2(elu(conv1(elu(x)))) */
        const Tensor *w1 = need(o, ENC "layers.%d.block.1.conv.weight", idx);
        const Tensor *b1 = opt(o, ENC "layers.%d.block.1.conv.bias", idx);
        const Tensor *w2 = need(o, ENC "layers.%d.block.3.conv.weight", idx);
        const Tensor *b2 = opt(o, ENC "layers.%d.block.3.conv.bias", idx);
        const Tensor *wd = need(o, ENC "layers.%d.conv.weight", idx + 2);
        const Tensor *bd = opt(o, ENC "layers.%d.conv.bias", idx + 2);
        if (!w1 || !w2 || !wd) { free(h); return -1; }
        int hid = (int)w1->shape[0], K1 = (int)w1->shape[2];
        float *a = (float *)malloc(sizeof(float) * C * L);
        float *m = (float *)malloc(sizeof(float) * hid * L);
        float *r = (float *)malloc(sizeof(float) * C * L);
        elu(h, (size_t)C * L, a);
        /* dilation 1 (193): effective kernel K1, padding K1 - 1 (153) */
        conv1d(a, C, L, w1->data, b1 ? b1->data : NULL, hid, K1, 1, 1, K1 - 1, m, L);
        elu(m, (size_t)hid * L, m);
        conv1d(m, hid, L, w2->data, b2 ? b2->data : NULL, C, 1, 1, 1, 0, r, L);
        for (size_t j = 0; j < (size_t)C * L; j++) h[j] = h[j] + r[j];
        free(a); free(m); free(r);
        tap(tp, st++, h, C, L);
        /* elu, then the strided conv: kernel 2r, stride r, left pad k - s = r (68-70, 204-211) */
        int ratio = o->ratio[i], Co = (int)wd->shape[0];
        int64_t L2 = conv_out_len(L, 2 * ratio, ratio, 1, ratio);
        if (L2 <= 0) { free(h); snprintf(o->err, sizeof o->err, "audio too short"); return -1; }
        float *e = (float *)malloc(sizeof(float) * C * L);
        elu(h, (size_t)C * L, e);
        float *d = (float *)malloc(sizeof(float) * Co * L2);
        conv1d(e, C, L, wd->data, bd ? bd->data : NULL, Co, 2 * ratio, ratio, 1, ratio, d, L2);
        free(e); free(h);
        h = d; C = Co; L = L2;
        tap(tp, st++, h, C, L);
        idx += 3;
    }
    {   /* final ELU + conv (241-242) */
        const Tensor *wf = need(o, ENC "layers.%d.conv.weight", o->final_idx);
        const Tensor *bf = opt(o, ENC "layers.%d.conv.bias", o->final_idx);
        int Kf = (int)wf->shape[2];
        float *e = (float *)malloc(sizeof(float) * C * L);
        elu(h, (size_t)C * L, e);
        float *d = (float *)malloc(sizeof(float) * o->dim * L);
        conv1d(e, C, L, wf->data, bf ? bf->data : NULL, o->dim, Kf, 1, 1, Kf - 1, d, L);
        free(e); free(h);
        h = d; C = o->dim;
        tap(tp, st++, h, C, L);
    }
    /* transformer (426-442 per layer); kept channel-first, which is the reference's [B,T,D] transposed */
    const int D = o->dim;
    for (int l = 0; l < o->n_layers; l++) {
        const Tensor *ln1w = need(o, TRF "layers.%d.input_layernorm.weight", l);
        const Tensor *ln1b = need(o, TRF "layers.%d.input_layernorm.bias", l);
        const Tensor *wq = need(o, TRF "layers.%d.self_attn.q_proj.weight", l);
        const Tensor *wk = need(o, TRF "layers.%d.self_attn.k_proj.weight", l);
        const Tensor *wv = need(o, TRF "layers.%d.self_attn.v_proj.weight", l);
        const Tensor *wo = need(o, TRF "layers.%d.self_attn.o_proj.weight", l);
        const Tensor *s1 = need(o, TRF "layers.%d.self_attn_layer_scale.scale", l);
        const Tensor *ln2w = need(o, TRF "layers.%d.post_attention_layernorm.weight", l);
        const Tensor *ln2b = need(o, TRF "layers.%d.post_attention_layernorm.bias", l);
        const Tensor *f1 = need(o, TRF "layers.%d.mlp.fc1.weight", l);
        const Tensor *f2 = need(o, TRF "layers.%d.mlp.fc2.weight", l);
        const Tensor *s2 = need(o, TRF "layers.%d.mlp_layer_scale.scale", l);
        if (!ln1w || !ln1b || !wq || !wk || !wv || !wo || !s1 || !ln2w || !ln2b || !f1 || !f2 || !s2) { free(h); return -1; }
        float *xn = (float *)malloc(sizeof(float) * D * L);
        float *q = (float *)malloc(sizeof(float) * D * L), *k = (float *)malloc(sizeof(float) * D * L);
        float *v = (float *)malloc(sizeof(float) * D * L), *ao = (float *)malloc(sizeof(float) * D * L);
        layernorm(h, D, L, ln1w->data, ln1b->data, o->ln_eps, xn);
        linear(xn, D, L, wq->data, D, q);
        linear(xn, D, L, wk->data, D, k);
        linear(xn, D, L, wv->data, D, v);
        attention(o, q, k, v, L, ao);
        linear(ao, D, L, wo->data, D, q);
        for (int d = 0; d < D; d++)
            for (int64_t t = 0; t < L; t++) h[(size_t)d * L + t] = h[(size_t)d * L + t] + q[(size_t)d * L + t] * s1->data[d];
        layernorm(h, D, L, ln2w->data, ln2b->data, o->ln_eps, xn);
        float *m = (float *)malloc(sizeof(float) * o->ff * L);
        linear(xn, D, L, f1->data, o->ff, m);
        for (size_t j = 0; j < (size_t)o->ff * L; j++) m[j] = gelu_tanh(m[j]);
        linear(m, o->ff, L, f2->data, D, q);
        for (int d = 0; d < D; d++)
            for (int64_t t = 0; t < L; t++) h[(size_t)d * L + t] = h[(size_t)d * L + t] + q[(size_t)d * L + t] * s2->data[d];
        free(xn); free(q); free(k); free(v); free(ao); free(m);
        tap(tp, st++, h, D, L);
    }
    {   /* ConvDownsample1d (590-604): kernel 2s, stride s, causal pad s */
        const Tensor *wd = need(o, "encoder.downsample.conv.weight");
        const Tensor *bd = opt(o, "encoder.downsample.conv.bias");
        int s = o->ds_stride;
        int64_t L2 = conv_out_len(L, 2 * s, s, 1, s);
        if (L2 <= 0) { free(h); snprintf(o->err, sizeof o->err, "audio too short"); return -1; }
        float *d = (float *)malloc(sizeof(float) * D * L2);
        conv1d(h, D, L, wd->data, bd ? bd->data : NULL, D, 2 * s, s, 1, s, d, L2);
        free(h);
        h = d; L = L2;
        tap(tp, st++, h, D, L);
    }
    *h_out = h;
    *T_out = L;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Public API                                                                                  */

int enc_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* SpeechTokenizer::encode (646-675): audio[n] (24 kHz mono f32) -> codes[T][use_q] (i64). */
int enc_oracle_encode(EncOracle *o, const float *audio, int64_t n, int64_t *codes, size_t cap, int64_t *n_frames) {
    o->err[0] = 0;
    float *h = NULL;
    int64_t T = 0;
    if (encoder_forward(o, audio, n, &h, &T, NULL)) return -1;
    int uq = enc_oracle_use_q(o);
    *n_frames = T;
    if ((size_t)T * uq > cap) { free(h); snprintf(o->err, sizeof o->err, "output buffer too small"); return -2; }
    int rc = quantize(o, h, T, codes, NULL, NULL);
    free(h);
    return rc;
}

/* The quantiser alone on a given latent h[dim][T] (what the HIP path must reproduce bit for bit). */
int enc_oracle_quantize(EncOracle *o, const float *h, int64_t T, int64_t *codes) {
    o->err[0] = 0;
    if (!o->ready) { snprintf(o->err, sizeof o->err, "finalize() not called"); return -1; }
    return quantize(o, h, T, codes, NULL, NULL);
}

/* Squared distances of one frame's residual to every codeword of layer (group 0 = semantic, 1 = acoustic), as the
 * search evaluates them: lets a test decide whether a differing code was a near-tie. */
int enc_oracle_distances(EncOracle *o, int group, int layer, const float *x, float *dist) {
    const Codebook *cb = group ? &o->ac[layer] : &o->sem[layer];
    if (layer < 0 || layer >= (group ? o->n_ac : o->n_sem)) return -1;
    const int D = o->cb_dim;
    float x_sq = 0.0f;
    for (int d = 0; d < D; d++) { float sq = x[d] * x[d]; x_sq = x_sq + sq; }
    for (int c = 0; c < o->cb_size; c++) {
        float xe = 0.0f;
        for (int d = 0; d < D; d++) xe = fmaf(x[d], cb->embed[(size_t)c * D + d], xe);
        float two = xe * 2.0f;
        dist[c] = (x_sq + cb->e_sq[c]) - two;
    }
    return 0;
}
int enc_oracle_cb_size(const EncOracle *o) { return o->cb_size; }
int enc_oracle_cb_dim(const EncOracle *o) { return o->cb_dim; }
/* embed[cb_size][cb_dim] of a layer (borrowed) */
const float *enc_oracle_embed(const EncOracle *o, int group, int layer) {
    if (layer < 0 || layer >= (group ? o->n_ac : o->n_sem)) return NULL;
    return group ? o->ac[layer].embed : o->sem[layer].embed;
}

/* Intermediate stage (numbering above enc_oracle_n_stages). */
int enc_oracle_stage(EncOracle *o, const float *audio, int64_t n, int stage, float *out, size_t cap, int *C, int64_t *L) {
    o->err[0] = 0;
    Tap tp = {stage, out, cap, 0, 0, 0};
    float *h = NULL;
    int64_t T = 0;
    if (encoder_forward(o, audio, n, &h, &T, &tp)) return -1;
    int base = 2 + 2 * o->n_blocks + o->n_layers + 1;
    if (stage == base || stage == base + 1) {
        const int D = o->cb_dim;
        float *ps = (float *)malloc(sizeof(float) * D * T), *pa = (float *)malloc(sizeof(float) * D * T);
        int64_t *codes = (int64_t *)malloc(sizeof(int64_t) * enc_oracle_use_q(o) * T);
        memset(pa, 0, sizeof(float) * D * T);
        int rc = quantize(o, h, T, codes, ps, pa);
        if (!rc) tap(&tp, stage, stage == base ? ps : pa, D, T);
        free(ps); free(pa); free(codes);
        if (rc) { free(h); return rc; }
    }
    free(h);
    if (!tp.hit) { snprintf(o->err, sizeof o->err, "no such stage %d", stage); return -1; }
    *C = tp.C; *L = tp.L;
    if ((size_t)tp.C * tp.L > cap) { snprintf(o->err, sizeof o->err, "stage buffer too small"); return -2; }
    return 0;
}

/* resample (725-741): linear interpolation evaluated in f64, cast to f32. */
int64_t enc_oracle_resample(const float *x, int64_t n, uint32_t from_rate, uint32_t to_rate, float *out, int64_t cap) {
    double ratio = (double)from_rate / (double)to_rate;
    int64_t out_len = (int64_t)((double)n / ratio);
    if (!out) return out_len;
    if (out_len > cap) return -1;
    for (int64_t i = 0; i < out_len; i++) {
        double pos = (double)i * ratio;
        int64_t idx = (int64_t)pos;
        double frac = pos - (double)idx;
        double v;
        if (idx + 1 < n) v = (double)x[idx] * (1.0 - frac) + (double)x[idx + 1] * frac;
        else v = (double)x[idx < n - 1 ? idx : n - 1];
        out[i] = (float)v;
    }
    return out_len;
}
