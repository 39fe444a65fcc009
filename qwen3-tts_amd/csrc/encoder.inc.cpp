// encoder.inc.cpp -- host side of the speech-tokenizer ENCODER path, compiled as part of hipvoc.cpp
// (it shares Ctx, Loader and PlanBuilder).  C ABI: include/hipvoc_encoder.h.
//
// What this mirrors in the reference (/root/reference/src/speech_tokenizer.rs):
//   hipvoc_enc_load          <- SpeechTokenizer::load        617-641 (+ SEANetEncoder::load 178-232, TransformerLayer::load
//                                                            407-424, EuclideanCodebook::load 482-491, ...)
//   hipvoc_enc_encode        <- SpeechTokenizer::encode      646-675
//   hipvoc_resample_linear   <- resample                     725-741
// Dataflow on the GPU, channel-first [C][utterance][t] like the vocoder:
//   audio -> conv k7 -> 4 x { x + conv1(elu(conv3(elu(x)))) ; elu ; conv(k=2r, stride r) } -> elu -> conv k3
//         -> 8 x { LayerNorm, QKV, RoPE sliding-window attention, O, layer scale, +; LayerNorm, fc1, tanh-GELU, fc2, scale, + }
//         -> conv(k=4, stride 2) -> RVQ encode (1 semantic + 15 acoustic codebooks of the 32).
// Every conv / linear is conv_gemm (MFMA); a strided conv runs as a 2-tap conv over the space-to-depth form of its
// input (launch_s2d); the nearest-codeword search is rvq_encode_kernel (bit-exact contract with the encoder test oracle).

namespace hipvoc {

struct EncModel {
    int n_filters = 0, dim = 0, n_layers = 0, ff = 0, n_heads = 8, context = 250, n_use = 16, ds_stride = 2;
    float rope_max_period = 10000.f, ln_eps = 1e-5f;
    int conv0_taps = 7;
    int cb_size = 0, D = 0, n_sem = 0, n_ac = 0;      // n_sem / n_ac: layers actually evaluated (the first n_use)
    Gemm conv0, fin, ds;
    struct Blk { Gemm c1, c2, down; int C = 0, ratio = 0; };
    std::vector<Blk> blocks;
    struct Layer { float *ln1w, *ln1b, *ln2w, *ln2b, *ls1, *ls2; Gemm qkv, o, fc1, fc2; };
    std::vector<Layer> layers;
    float *wp_t = nullptr, *emb_t = nullptr, *e_sq = nullptr;
    float *cos_t = nullptr, *sin_t = nullptr;
    int rope_cap = 0;
    std::map<std::pair<int, long>, std::unique_ptr<Plan>> plans;
    long long *d_codes = nullptr; size_t d_codes_cap = 0;
    long long *cur_codes_out = nullptr;      // per-call binding: where the search writes (null: d_codes)
    double macs_per_sample = 0;

    int64_t n_frames(int64_t n) const {
        for (const Blk &b : blocks) n /= b.ratio;       // each strided conv floors (pad = k - s, no right pad)
        return n / ds_stride;
    }
    int samples_per_frame() const {
        int s = ds_stride;
        for (const Blk &b : blocks) s *= b.ratio;
        return s;
    }
};

static void enc_release(Ctx *c) {
    if (!c->enc) return;
    if (c->enc->d_codes) (void)hipFree(c->enc->d_codes);
    delete c->enc;
    c->enc = nullptr;
}

static EncModel *enc_of(Ctx *c) { return c->enc; }

// CausalConv1d with kernel 2r and stride r as a 2-tap GEMM over the space-to-depth input:
// out[co][q] = sum_{tap,ci,j} W[co][ci][tap*r + j] * x[ci][(q - 1 + tap)*r + j]   (left pad r = one whole block)
static Gemm enc_strided(Loader &L, const std::string &wname, const std::string &bias_name, int r) {
    Gemm g;
    std::vector<float> w;
    std::vector<int64_t> sh;
    if (!L.get(wname, w, sh)) return g;
    if (sh.size() != 3 || sh[2] != 2 * r) {
        L.err = wname + ": strided conv needs kernel == 2*stride";
        L.ok = false;
        return g;
    }
    int Co = (int)sh[0], Ci = (int)sh[1], K = (int)sh[2];
    g.Cin = Ci * r; g.N = Co; g.TN = pick_tn(Co); g.NP = rup(Co, 32 * g.TN);
    g.ntaps = 2; g.shift0 = -1; g.step = 1; g.S = 1;
    int CinP = rup(g.Cin, 32);
    std::vector<float> p((size_t)2 * CinP * g.NP, 0.0f);
    for (int co = 0; co < Co; co++)
        for (int ci = 0; ci < Ci; ci++)
            for (int tap = 0; tap < 2; tap++)
                for (int j = 0; j < r; j++)
                    p[((size_t)tap * CinP + ci * r + j) * g.NP + co] = w[((size_t)co * Ci + ci) * K + tap * r + j];
    L.finish(g, p, CinP);
    if (!bias_name.empty() && L.st.has(bias_name)) g.bias = L.vec(bias_name, Co);
    return g;
}

static Gemm enc_conv(Loader &L, const std::string &prefix, int dil = 1) {
    const std::string b = prefix + ".bias";
    return L.conv({prefix + ".weight"}, L.st.has(b) ? b : std::string(), dil);     // bias is optional (speech_tokenizer.rs:56)
}

// The first conv (1 input channel, K taps, causal): W'[j][co] = W[co][0][j] over the K shifted copies of the waveform.
static Gemm enc_conv0(Loader &L, const std::string &prefix) {
    Gemm g;
    std::vector<float> w;
    std::vector<int64_t> sh;
    if (!L.get(prefix + ".weight", w, sh)) return g;
    if (sh.size() != 3 || sh[1] != 1) {
        L.err = prefix + ".weight: expected [C, 1, K]";
        L.ok = false;
        return g;
    }
    const int Co = (int)sh[0], K = (int)sh[2];
    g.Cin = K; g.N = Co; g.TN = pick_tn(Co); g.NP = rup(Co, 32 * g.TN);
    g.ntaps = 1; g.shift0 = 0; g.step = 0; g.S = 1;
    const int CinP = rup(K, 32);
    std::vector<float> p((size_t)CinP * g.NP, 0.0f);
    for (int co = 0; co < Co; co++)
        for (int j = 0; j < K; j++) p[(size_t)j * g.NP + co] = w[(size_t)co * K + j];
    L.finish(g, p, CinP);
    if (L.st.has(prefix + ".bias")) g.bias = L.vec(prefix + ".bias", Co);
    return g;
}

static bool enc_load_model(Ctx *c, EncModel *m, const char *path, std::string &err) {
    Loader L;
    L.c = c;
    if (!L.st.open(path, err)) return false;
    auto &md = L.st.metadata;
    if (md.count("n_heads")) m->n_heads = atoi(md["n_heads"].c_str());
    if (md.count("context")) m->context = atoi(md["context"].c_str());
    if (md.count("n_use_codebooks")) m->n_use = atoi(md["n_use_codebooks"].c_str());
    if (md.count("rope_max_period")) m->rope_max_period = (float)atof(md["rope_max_period"].c_str());
    if (m->n_heads < 1 || m->context < 1 || m->n_use < 1) { err = "bad metadata"; return false; }
    const std::string E = "encoder.encoder.layers.";
    const StTensor *w0 = L.st.find(E + "0.conv.weight");
    if (!w0 || w0->shape.size() != 3 || w0->shape[1] != 1 || w0->shape[0] < 1 || w0->shape[2] < 1) { err = "missing/invalid " + E + "0.conv.weight"; return false; }
    m->n_filters = (int)w0->shape[0];
    m->conv0 = enc_conv0(L, E + "0.conv");      // 1-tap GEMM over the tap stack of the waveform (launch_tap_stack)
    m->conv0_taps = (int)w0->shape[2];
    int idx = 1, C = m->n_filters;
    while (L.st.has(E + fmt("%d.block.1.conv.weight", idx))) {       // SEANetEncoder::load 187-215
        EncModel::Blk b;
        b.C = C;
        const StTensor *dw = L.st.find(E + fmt("%d.conv.weight", idx + 2));
        if (!dw || dw->shape.size() != 3 || dw->shape[2] < 2 || dw->shape[2] % 2 || dw->shape[1] != C) { err = "missing/invalid down conv"; return false; }
        b.ratio = (int)dw->shape[2] / 2;
        b.c1 = enc_conv(L, E + fmt("%d.block.1.conv", idx));        // dilation 1 (193)
        b.c2 = enc_conv(L, E + fmt("%d.block.3.conv", idx));
        b.down = enc_strided(L, E + fmt("%d.conv.weight", idx + 2), E + fmt("%d.conv.bias", idx + 2), b.ratio);
        if (!L.ok) { err = L.err; return false; }
        if (b.c1.Cin != C || b.c2.N != C || b.c2.Cin != b.c1.N || b.c2.ntaps != 1) { err = "residual block: unexpected shapes"; return false; }
        C = b.down.N;
        m->blocks.push_back(b);
        idx += 3;
    }
    if (m->blocks.empty()) { err = "no encoder blocks found"; return false; }
    m->fin = enc_conv(L, E + fmt("%d.conv", idx + 1));              // 218-225
    if (!L.ok) { err = L.err; return false; }
    if (m->fin.Cin != C) { err = "final conv: unexpected input channels"; return false; }
    m->dim = m->fin.N;
    const int D = m->dim;
    if (D % m->n_heads || (D / m->n_heads) % 2) { err = "dimension not divisible into even heads"; return false; }
    const std::string T = "encoder.encoder_transformer.layers.";
    for (int i = 0; L.st.has(T + fmt("%d.self_attn.q_proj.weight", i)); i++) {
        const std::string p = T + fmt("%d.", i);
        EncModel::Layer ly;
        ly.ln1w = L.vec(p + "input_layernorm.weight", D);
        ly.ln1b = L.vec(p + "input_layernorm.bias", D);
        ly.ln2w = L.vec(p + "post_attention_layernorm.weight", D);
        ly.ln2b = L.vec(p + "post_attention_layernorm.bias", D);
        ly.ls1 = L.vec(p + "self_attn_layer_scale.scale", D);
        ly.ls2 = L.vec(p + "mlp_layer_scale.scale", D);
        ly.qkv = L.conv({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, "", 1);
        ly.o = L.conv({p + "self_attn.o_proj.weight"}, "", 1);
        ly.fc1 = L.conv({p + "mlp.fc1.weight"}, "", 1);
        ly.fc2 = L.conv({p + "mlp.fc2.weight"}, "", 1);
        if (!L.ok) { err = L.err; return false; }
        if (ly.qkv.N != 3 * D || ly.qkv.Cin != D || ly.o.N != D || ly.fc2.N != D || ly.fc2.Cin != ly.fc1.N) { err = "transformer layer: unexpected shapes"; return false; }
        m->ff = ly.fc1.N;
        m->layers.push_back(ly);
    }
    m->n_layers = (int)m->layers.size();
    {
        const StTensor *dw = L.st.find("encoder.downsample.conv.weight");
        if (!dw || dw->shape.size() != 3 || dw->shape[2] < 2 || dw->shape[2] % 2) { err = "missing/invalid encoder.downsample.conv.weight"; return false; }
        m->ds_stride = (int)dw->shape[2] / 2;                       // kernel = 2 * stride (596)
        m->ds = enc_strided(L, "encoder.downsample.conv.weight", "encoder.downsample.conv.bias", m->ds_stride);
        if (!L.ok) { err = L.err; return false; }
        if (m->ds.N != D || m->ds.Cin != D * m->ds_stride) { err = "downsample conv: unexpected shape"; return false; }
    }
    // quantiser: only the first n_use codebooks are ever kept (664-667); later layers do not influence earlier ones
    const std::string Q = "encoder.quantizer.";
    const char *grp[2] = {"semantic_residual_vector_quantizer.", "acoustic_residual_vector_quantizer."};
    int have[2] = {0, 0};
    for (int g = 0; g < 2; g++)
        while (L.st.has(Q + grp[g] + fmt("layers.%d.codebook.embed_sum", have[g]))) have[g]++;
    if (!have[0]) { err = "no semantic codebook"; return false; }
    m->n_sem = std::min(have[0], m->n_use);
    m->n_ac = std::min(have[1], m->n_use - m->n_sem);
    const int nq = m->n_sem + m->n_ac;
    std::vector<float> emb_t, e_sq, wp_t;
    for (int q = 0; q < nq; q++) {
        const int g = q < m->n_sem ? 0 : 1, l = g ? q - m->n_sem : q;
        const std::string p = Q + grp[g] + fmt("layers.%d.codebook.", l);
        std::vector<float> es, cu;
        std::vector<int64_t> sh, sh2;
        if (!L.get(p + "embed_sum", es, sh) || !L.get(p + "cluster_usage", cu, sh2)) { err = L.err; return false; }
        if (sh.size() != 2 || (int64_t)cu.size() != sh[0]) { err = p + ": bad shapes"; return false; }
        if (!q) { m->cb_size = (int)sh[0]; m->D = (int)sh[1]; emb_t.resize((size_t)nq * m->D * m->cb_size); e_sq.resize((size_t)nq * m->cb_size); }
        if (sh[0] != m->cb_size || sh[1] != m->D) { err = p + ": codebooks differ in shape"; return false; }
        for (int cidx = 0; cidx < m->cb_size; cidx++) {
            // embed = embed_sum / cluster_usage (482-491: plain division), |e|^2 = ascending sum of rounded squares;
            // volatile keeps the host compiler from contracting the square into the sum (the oracle's form)
            volatile float s = 0.0f;
            for (int d = 0; d < m->D; d++) {
                volatile float e = es[(size_t)cidx * m->D + d] / cu[cidx];
                volatile float sq = e * e;
                s = s + sq;
                emb_t[((size_t)q * m->D + d) * m->cb_size + cidx] = e;
            }
            e_sq[(size_t)q * m->cb_size + cidx] = s;
        }
    }
    wp_t.assign((size_t)2 * D * m->D, 0.0f);
    for (int g = 0; g < 2; g++) {
        if (g && !m->n_ac) break;
        std::vector<float> w;
        std::vector<int64_t> sh;
        if (!L.get(Q + grp[g] + "input_proj.weight", w, sh)) { err = L.err; return false; }
        if ((int64_t)w.size() != (int64_t)m->D * D) { err = std::string(grp[g]) + "input_proj.weight: wrong size"; return false; }
        for (int d = 0; d < m->D; d++)
            for (int k = 0; k < D; k++) wp_t[((size_t)g * D + k) * m->D + d] = w[(size_t)d * D + k];
    }
    m->emb_t = L.upload(emb_t);
    m->e_sq = L.upload(e_sq);
    m->wp_t = L.upload(wp_t);
    if (!L.ok) { err = L.err; return false; }
    // algorithmic MACs per input sample (for throughput lines)
    {
        double macs = 0, rate = 1.0;
        macs += (double)m->conv0.N * m->conv0.Cin * m->conv0.ntaps;      // (Cin = taps here)
        for (const EncModel::Blk &b : m->blocks) {
            macs += rate * ((double)b.c1.N * b.c1.Cin * b.c1.ntaps + (double)b.c2.N * b.c2.Cin);
            rate /= b.ratio;
            macs += rate * (double)b.down.N * b.down.Cin * 2;
        }
        macs += rate * (double)m->fin.N * m->fin.Cin * m->fin.ntaps;
        for (const EncModel::Layer &ly : m->layers)
            macs += rate * ((double)ly.qkv.N * D + (double)D * D + 2.0 * ly.fc1.N * D);
        rate /= m->ds_stride;
        macs += rate * ((double)m->ds.N * m->ds.Cin * 2 + 2.0 * m->D * D + (double)nq * m->cb_size * m->D);
        m->macs_per_sample = macs;
    }
    return true;
}

enum EncBuf { EB_AUDIO = 100, EB_A, EB_B, EB_M, EB_SD, EB_XN, EB_QKV, EB_AO, EB_FF, EB_LAT, EB_PROJ };

// RoPE tables (apply_rope 311-344): inv_freq in f64 cast to f32, angle = f32 product, cosf / sinf
static bool enc_rope(Ctx *c, EncModel *m, int T) {
    if (T <= m->rope_cap) return true;
    int cap = std::max(T, std::max(512, m->rope_cap * 2));
    const int hd = m->dim / m->n_heads, half = hd / 2;
    std::vector<float> cs((size_t)cap * half), sn((size_t)cap * half);
    for (int t = 0; t < cap; t++)
        for (int i = 0; i < half; i++) {
            float inv = (float)(1.0 / pow((double)m->rope_max_period, 2.0 * (double)i / (double)hd));
            float ang = (float)t * inv;
            cs[(size_t)t * half + i] = cosf(ang);
            sn[(size_t)t * half + i] = sinf(ang);
        }
    Loader L;
    L.c = c;
    float *dc = L.upload(cs), *ds = L.upload(sn);      // older plans keep the older (smaller) tables
    if (!L.ok) { c->err = L.err; return false; }
    m->cos_t = dc; m->sin_t = ds; m->rope_cap = cap;
    return true;
}

// Plan for G utterances of n samples each (shorter ones are zero-padded at the END by the caller: every layer is
// causal, so their frames are unaffected).  Stage numbering = the encoder test oracle.
// Production plans write the input of every strided conv straight from its producer's epilogue (ConvGemmArgs::s2d);
// debug plans (stage taps want the raw tensors) write the raw tensor and convert it in a separate pass (s2d_kernel).
static Plan *enc_get_plan(Ctx *c, EncModel *m, int G, long n, bool debug) {
    auto key = std::make_pair(debug ? -G : G, n);
    auto it = m->plans.find(key);
    if (it != m->plans.end()) return it->second.get();
    // one plan per (batch, length): a service that sees arbitrary clip lengths must not grow this without bound.  A plan
    // is only host-side launch closures (its launches are already enqueued when a later call gets here), so dropping
    // the cache is safe; the workspace buffers stay.
    if (m->plans.size() >= 128) m->plans.clear();
    std::unique_ptr<Plan> plan(new Plan);
    plan->G = G; plan->T = (int)m->n_frames(n);
    PlanBuilder b{c, plan.get(), G, plan->T};
    // sizes
    size_t big = 0, sd = 0, mid = 0;
    {
        long L = n;
        big = (size_t)std::max(m->n_filters, m->conv0_taps) * rupl(L, 4);      // (the tap stack lives in a ping-pong buffer)
        for (const EncModel::Blk &bl : m->blocks) {
            mid = std::max(mid, (size_t)bl.c1.N * rupl(L, 4));
            long Lq = L / bl.ratio;
            sd = std::max(sd, (size_t)bl.C * bl.ratio * rupl(Lq, 4));
            big = std::max(big, (size_t)bl.down.N * rupl(Lq, 4));
            L = Lq;
        }
        big = std::max(big, (size_t)m->dim * rupl(L, 4));
        sd = std::max(sd, (size_t)m->dim * m->ds_stride * rupl(L / m->ds_stride, 4));
    }
    auto grow = [&](int id, size_t floats) {      // geometric growth: stale buffers of older plans stay valid
        auto f = c->bufs.find(id);
        if (f != c->bufs.end() && f->second.second >= floats + 64) return f->second.first;
        return b.buf(id, floats + floats / 2);
    };
    const int Lp0 = (int)rupl(n, 4);
    float *audio = grow(EB_AUDIO, (size_t)G * Lp0);
    float *A = grow(EB_A, big * G), *B = grow(EB_B, big * G), *M = grow(EB_M, mid * G), *SD = grow(EB_SD, sd * G);
    if (b.fail || !audio || !A || !B || !M || !SD) return nullptr;
    plan->enc_audio = audio;
    int st = 0;
    int L = (int)n, Lp = Lp0;
    float *h = A, *o = B;
    {
        // taps of the waveform as channels (into the other ping-pong buffer), then the first conv as a 1-tap GEMM
        const float *x = audio;
        float *y = o;
        const int K = m->conv0_taps, GG = G, LL = L, LLp = Lp;
        b.add("enc_tap_stack", "tap_stack_kernel", 0, [=](hipStream_t s) { return launch_tap_stack(x, y, K, GG, LL, LLp, s); });
    }
    b.gemm("enc_conv0", m->conv0, o, L, Lp, h, Lp, nullptr, nullptr, nullptr, ACT_NONE);
    b.mark(st++, h, m->conv0.N, L, Lp);
    for (const EncModel::Blk &bl : m->blocks) {
        // ResidualBlock::forward (139-149); ELU is applied where the value is consumed / produced
        b.gemm("enc_res_conv3", bl.c1, h, L, Lp, M, Lp, nullptr, nullptr, nullptr, ACT_ELU, nullptr, nullptr, nullptr, nullptr, 1);
        const int r = bl.ratio, Lq = L / r, Lqp = rup(std::max(Lq, 1), 4);
        if (!debug) {
            b.gemm("enc_res_conv1", bl.c2, M, L, Lp, SD, Lqp, nullptr, h, nullptr, ACT_ELU, nullptr, nullptr, nullptr, nullptr, 0, r);
            st++;
        } else {
            b.gemm("enc_res_conv1", bl.c2, M, L, Lp, h, Lp, nullptr, h, nullptr, ACT_NONE);
            b.mark(st++, h, bl.C, L, Lp);
            const float *x = h;
            float *y = SD;
            int C = bl.C, GG = G, LL = L, LLp = Lp;
            b.add("enc_s2d", "s2d_kernel", 0, [=](hipStream_t s) { return launch_s2d(x, y, C, GG, LL, LLp, r, Lq, Lqp, 1, s); });
        }
        b.gemm("enc_down", bl.down, SD, Lq, Lqp, o, Lqp, nullptr, nullptr, nullptr, ACT_NONE);
        std::swap(h, o);
        L = Lq; Lp = Lqp;
        b.mark(st++, h, bl.down.N, L, Lp);
    }
    b.gemm("enc_final_conv", m->fin, h, L, Lp, o, Lp, nullptr, nullptr, nullptr, ACT_NONE, nullptr, nullptr, nullptr, nullptr, 1);
    std::swap(h, o);
    b.mark(st++, h, m->dim, L, Lp);
    const int D = m->dim;
    const int dsr = m->ds_stride, dsT = L / dsr, dsTp = rup(std::max(dsT, 1), 4);
    bool ds_fused = false;
    if (m->n_layers) {
        if (!enc_rope(c, m, L)) return nullptr;
        float *xn = grow(EB_XN, (size_t)D * G * Lp), *qkv = grow(EB_QKV, (size_t)3 * D * G * Lp);
        float *ao = grow(EB_AO, (size_t)D * G * Lp), *ffb = grow(EB_FF, (size_t)m->ff * G * Lp);
        if (b.fail || !xn || !qkv || !ao || !ffb) return nullptr;
        const long P = (long)G * Lp;
        const int nh = m->n_heads, hd = D / nh, win = m->context, T = L, Tp = Lp;
        const float *ct = m->cos_t, *sn = m->sin_t;
        const float eps = m->ln_eps;
        for (const EncModel::Layer &ly : m->layers) {
            float *hh = h;
            b.add("enc_layernorm", layernorm_kernel_name(D), 0, [=](hipStream_t s) { return launch_layernorm(hh, xn, ly.ln1w, ly.ln1b, D, P, eps, s); });
            b.gemm("enc_qkv", ly.qkv, xn, T, Tp, qkv, Tp, nullptr, nullptr, nullptr, ACT_NONE);
            b.add("enc_attention", attention_kernel_name(hd, T, win), (double)G * nh * T * std::min(T, win) * hd, [=](hipStream_t s) {
                return launch_attention(qkv, ao, ct, sn, nh, hd, G, T, Tp, win, s);
            });
            b.gemm("enc_o_proj", ly.o, ao, T, Tp, h, Tp, nullptr, h, ly.ls1, ACT_NONE);
            b.add("enc_layernorm", layernorm_kernel_name(D), 0, [=](hipStream_t s) { return launch_layernorm(hh, xn, ly.ln2w, ly.ln2b, D, P, eps, s); });
            b.gemm("enc_fc1", ly.fc1, xn, T, Tp, ffb, Tp, nullptr, nullptr, nullptr, ACT_GELU_TANH);
            if (!debug && &ly == &m->layers.back()) {      // the last layer's output only feeds the stride-2 conv
                b.gemm("enc_fc2", ly.fc2, ffb, T, Tp, SD, dsTp, nullptr, h, ly.ls2, ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0, dsr);
                ds_fused = true;
                st++;
            } else {
                b.gemm("enc_fc2", ly.fc2, ffb, T, Tp, h, Tp, nullptr, h, ly.ls2, ACT_NONE);
                b.mark(st++, h, D, T, Tp);
            }
        }
    }
    const int r = dsr, T = dsT, Tp = dsTp;
    if (!ds_fused) {
        const float *x = h;
        float *y = SD;
        int GG = G, LL = L, LLp = Lp;
        b.add("enc_s2d", "s2d_kernel", 0, [=](hipStream_t s) { return launch_s2d(x, y, D, GG, LL, LLp, r, T, Tp, 0, s); });
    }
    float *lat = grow(EB_LAT, (size_t)D * G * Tp), *proj = grow(EB_PROJ, (size_t)2 * m->D * G * Tp);
    if (b.fail || !lat || !proj) return nullptr;
    b.gemm("enc_downsample", m->ds, SD, T, Tp, lat, Tp, nullptr, nullptr, nullptr, ACT_NONE);
    b.mark(st++, lat, D, T, Tp);
    {
        RvqEncArgs a{};
        a.h = lat; a.wp_t = m->wp_t; a.emb_t = m->emb_t; a.e_sq = m->e_sq; a.proj = proj;
        a.dim = D; a.D = m->D; a.cb_size = m->cb_size; a.n_sem = m->n_sem; a.n_ac = m->n_ac; a.T = T; a.Lp = Tp; a.G = G;
        EncModel *mm = m;
        const int nq = m->n_sem + m->n_ac;
        b.add("enc_rvq_encode", "rvq_encode_kernel", (double)G * T * ((double)nq * m->cb_size * m->D + 2.0 * m->D * D),
              [a, mm](hipStream_t s) {
                  RvqEncArgs aa = a;
                  aa.codes = mm->cur_codes_out ? mm->cur_codes_out : mm->d_codes;
                  return launch_rvq_encode(aa, s);
              });
        b.mark(st++, proj, m->D, T, Tp);                          // semantic projection
    }
    if (b.fail) return nullptr;
    Plan *raw = plan.get();
    m->plans[key] = std::move(plan);
    return raw;
}

// d_audio != null: B equal-length clips already in HBM (row stride d_stride floats), codes to d_codes_out
static int enc_run(Ctx *c, EncModel *m, const float *const *audio, const int64_t *n_samples, int B, int stop_stage, const Step **stopped,
                   Plan **plan_out, const float *d_audio_in = nullptr, size_t d_stride = 0, long long *d_codes_out = nullptr) {
    long n = 0;
    for (int i = 0; i < B; i++) {
        if (n_samples[i] < 0 || (!d_audio_in && !audio[i])) { c->err = "bad argument"; return HIPVOC_E_ARG; }
        n = std::max<long>(n, (long)n_samples[i]);
    }
    if (m->n_frames(n) < 1) { c->err = fmt("audio too short: %ld samples give no frame (one frame = %d samples)", n, m->samples_per_frame()); return HIPVOC_E_ARG; }
    // the kernels index positions of one tensor row-set with 32-bit ints (flat 1x1 convs see B * padded length columns)
    if (n > 0x7ffffff0L / 4 || (long)B * rupl(n, 4) > 0x7fffff00L) { c->err = "batch too long for one pass: split it"; return HIPVOC_E_ARG; }
    Plan *p = enc_get_plan(c, m, B, n, stop_stage >= 0);
    if (!p) { if (c->err.empty()) c->err = "workspace allocation failed"; return HIPVOC_E_HIP; }
    const int Lp0 = (int)rupl(n, 4);
    float *d_audio = p->enc_audio;
    if (d_audio_in) {
        HIPCHK(hipMemcpy2DAsync(d_audio, (size_t)Lp0 * sizeof(float), d_audio_in, d_stride * sizeof(float), (size_t)n * sizeof(float), B,
                                hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIPCHK(hipMemsetAsync(d_audio, 0, (size_t)B * Lp0 * sizeof(float), c->stream));
        for (int i = 0; i < B; i++)
            if (n_samples[i])
                HIPCHK(hipMemcpyAsync(d_audio + (size_t)i * Lp0, audio[i], (size_t)n_samples[i] * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    m->cur_codes_out = d_codes_out;
    const size_t ncodes = (size_t)B * p->T * (m->n_sem + m->n_ac);
    if (!d_codes_out && m->d_codes_cap < ncodes) {
        if (m->d_codes) HIPCHK(hipFree(m->d_codes));
        m->d_codes = nullptr; m->d_codes_cap = 0;
        HIPCHK(hipMalloc((void **)&m->d_codes, ncodes * sizeof(long long)));
        m->d_codes_cap = ncodes;
    }
    int rc = run_plan(c, p, nullptr, nullptr, 0, stop_stage, stopped);
    if (rc) return rc;
    if (plan_out) *plan_out = p;
    return HIPVOC_OK;
}

}  // namespace hipvoc

using namespace hipvoc;

extern "C" {

#define ENC(h)                                                                                  \
    CTX(h)                                                                                      \
    EncModel *m = enc_of(c);                                                                    \
    if (!m) { c->err = "not an encoder handle"; return HIPVOC_E_ARG; }

static void *enc_load_impl(const char *weights_path, int device_id) {
    auto fail = [](const std::string &m) -> void * {
        set_load_error(m.c_str());
        return nullptr;
    };
    if (!weights_path) return fail("weights_path is null");
    int ndev = hipvoc_device_count();
    if (ndev <= 0) return fail("no HIP device available (libhipvoc has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(fmt("device %d out of range (%d devices)", device_id, ndev));
    if (hipSetDevice(device_id) != hipSuccess) return fail("hipSetDevice failed");
    std::unique_ptr<Ctx> c(new Ctx);
    c->device = device_id;
    c->is_encoder = true;
    if (const char *pf = getenv("HIPVOC_PROFILE")) c->profile = atoi(pf) != 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    std::unique_ptr<EncModel> m(new EncModel);
    std::string err;
    if (!enc_load_model(c.get(), m.get(), weights_path, err)) return fail(err);
    if (hipDeviceSynchronize() != hipSuccess) return fail("device error after load");
    c->enc = m.release();
    return c.release();
}

void *hipvoc_enc_load(const char *weights_path, int device_id) {
    try {
        return enc_load_impl(weights_path, device_id);
    } catch (const std::bad_alloc &) {
        set_load_error("hipvoc_enc_load: out of host memory");
    } catch (const std::exception &e) {
        set_load_error(e.what());
    } catch (...) {
        set_load_error("hipvoc_enc_load: unknown C++ exception");
    }
    return nullptr;
}

int hipvoc_enc_n_codebooks(void *h) {
    Ctx *c = (Ctx *)h;
    EncModel *m = c ? enc_of(c) : nullptr;
    return m ? m->n_sem + m->n_ac : HIPVOC_E_ARG;
}
int hipvoc_enc_samples_per_frame(void *h) {
    Ctx *c = (Ctx *)h;
    EncModel *m = c ? enc_of(c) : nullptr;
    return m ? m->samples_per_frame() : HIPVOC_E_ARG;
}
int64_t hipvoc_enc_n_frames(void *h, int64_t n_samples) {
    Ctx *c = (Ctx *)h;
    EncModel *m = c ? enc_of(c) : nullptr;
    return m && n_samples >= 0 ? m->n_frames(n_samples) : HIPVOC_E_ARG;
}
double hipvoc_enc_macs_per_sample(void *h) {
    Ctx *c = (Ctx *)h;
    EncModel *m = c ? enc_of(c) : nullptr;
    return m ? m->macs_per_sample : -1.0;
}

int hipvoc_enc_encode_batch(void *h, const float *const *audio, const int64_t *n_samples, int B, int64_t *const *codes,
                            const size_t *codes_cap, int64_t *n_frames) {
    return guarded(h, [&]() -> int {
        ENC(h)
        if (!audio || !n_samples || !codes || !codes_cap || !n_frames || B < 1) { c->err = "bad argument"; return HIPVOC_E_ARG; }
        const int nq = m->n_sem + m->n_ac;
        for (int i = 0; i < B; i++) {
            if (n_samples[i] < 0) { c->err = "bad argument"; return HIPVOC_E_ARG; }
            n_frames[i] = m->n_frames(n_samples[i]);
            if (n_frames[i] < 1) { c->err = fmt("audio %d too short: %lld samples give no frame", i, (long long)n_samples[i]); return HIPVOC_E_ARG; }
            if ((size_t)n_frames[i] * nq > codes_cap[i]) { c->err = "output capacity too small"; return HIPVOC_E_CAP; }
        }
        Plan *p = nullptr;
        int rc = enc_run(c, m, audio, n_samples, B, -1, nullptr, &p);
        if (rc) return rc;
        std::vector<long long> host((size_t)B * p->T * nq);
        HIPCHK(hipMemcpyAsync(host.data(), m->d_codes, host.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < B; i++)      // frames past an utterance's own length come from its zero padding: dropped
            for (size_t j = 0; j < (size_t)n_frames[i] * nq; j++) codes[i][j] = host[(size_t)i * p->T * nq + j];
        return HIPVOC_OK;
    });
}

int hipvoc_enc_encode(void *h, const float *audio, int64_t n_samples, int64_t *codes, size_t codes_cap, int64_t *n_frames) {
    const float *ap[1] = {audio};
    int64_t *cp[1] = {codes};
    int64_t nf = 0;
    int rc = hipvoc_enc_encode_batch(h, ap, &n_samples, 1, cp, &codes_cap, &nf);
    if (n_frames) *n_frames = nf;
    return rc;
}

int hipvoc_enc_encode_device(void *h, const float *d_audio, int B, int64_t n_samples, size_t stride, int64_t *d_codes) {
    return guarded(h, [&]() -> int {
        ENC(h)
        if (!d_audio || !d_codes || B < 1 || n_samples < 1 || stride < (size_t)n_samples) { c->err = "bad argument"; return HIPVOC_E_ARG; }
        std::vector<int64_t> ns((size_t)B, n_samples);
        std::vector<const float *> none((size_t)B, nullptr);
        return enc_run(c, m, none.data(), ns.data(), B, -1, nullptr, nullptr, d_audio, stride, (long long *)d_codes);
    });
}

int hipvoc_enc_num_stages(void *h) {
    Ctx *c = (Ctx *)h;
    EncModel *m = c ? enc_of(c) : nullptr;
    return m ? 2 + 2 * (int)m->blocks.size() + m->n_layers + 3 : HIPVOC_E_ARG;
}

int hipvoc_enc_debug_stage(void *h, const float *audio, int64_t n_samples, int stage, float *out, size_t out_cap, int *Cc, int64_t *Ll) {
    return guarded(h, [&]() -> int {
        ENC(h)
        const int nst = hipvoc_enc_num_stages(h);
        if (!audio || !out || stage < 0 || stage >= nst) { c->err = "no such stage"; return HIPVOC_E_STAGE; }
        const Step *st = nullptr;
        const int want = stage == nst - 1 ? nst - 2 : stage;       // both projections are produced by the last step
        const float *ap[1] = {audio};
        int rc = enc_run(c, m, ap, &n_samples, 1, want, &st, nullptr);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!st) { c->err = "stage not reached"; return HIPVOC_E_STAGE; }
        const float *src = st->out;
        int C = st->C, L = st->L, Lp = st->Lp;
        if (stage == nst - 1) src += (size_t)m->D * Lp;            // acoustic projection follows the semantic one (G = 1)
        if ((size_t)C * L > out_cap) { c->err = "output capacity too small"; return HIPVOC_E_CAP; }
        HIPCHK(hipMemcpy2D(out, (size_t)L * sizeof(float), src, (size_t)Lp * sizeof(float), (size_t)L * sizeof(float), C, hipMemcpyDeviceToHost));
        if (Cc) *Cc = C;
        if (Ll) *Ll = L;
        return HIPVOC_OK;
    });
}

// resample (speech_tokenizer.rs:725-741): linear interpolation in f64, cast to f32.  Host code (the reference does it
// on the host before the tensor exists); out == null returns the output length.
int64_t hipvoc_resample_linear(const float *x, int64_t n, uint32_t from_rate, uint32_t to_rate, float *out, int64_t cap) {
    if (!x || n < 0 || !from_rate || !to_rate) return HIPVOC_E_ARG;
    const double ratio = (double)from_rate / (double)to_rate;
    const int64_t out_len = (int64_t)((double)n / ratio);
    if (!out) return out_len;
    if (out_len > cap) return HIPVOC_E_CAP;
    for (int64_t i = 0; i < out_len; i++) {
        const double pos = (double)i * ratio;
        const int64_t idx = (int64_t)pos;
        const double frac = pos - (double)idx;
        double v;
        if (idx + 1 < n) v = (double)x[idx] * (1.0 - frac) + (double)x[idx + 1] * frac;
        else v = (double)x[idx < n - 1 ? idx : n - 1];
        out[i] = (float)v;
    }
    return out_len;
}

}  // extern "C"
