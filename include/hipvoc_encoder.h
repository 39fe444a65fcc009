/*
 * hipvoc_encoder.h -- C ABI of the MI355X speech-tokenizer ENCODER path in libhipvoc:
 * 24 kHz mono waveform -> [n_frames, 16] codec tokens at 12.5 Hz, the inverse direction of hipvoc.h.
 *
 * Each entry point names the reference interface it replaces (/root/reference/src/speech_tokenizer.rs).  Same
 * conventions as hipvoc.h: opaque handle, NULL on load failure (text via hipvoc_last_error(NULL)), int 0 / negative
 * HIPVOC_E_* codes, caller-allocated outputs, no exceptions across the boundary, hipSetDevice per call.  A handle is
 * released with hipvoc_free(); hipvoc_last_error(), hipvoc_set_profile() and hipvoc_last_profile() work on it; the
 * vocoder entry points refuse it.  There is no CPU fallback: without a HIP device load fails.
 */
#ifndef HIPVOC_ENCODER_H
#define HIPVOC_ENCODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SpeechTokenizer::load (speech_tokenizer.rs:617-641): a safetensors file holding the `encoder.*` tensors of
 * speech_tokenizer/model.safetensors (F32 / F16 / BF16 storage).  The architecture is derived from tensor shapes;
 * the constants shapes cannot tell (speech_tokenizer.rs:25-29) may be overridden by safetensors metadata keys
 * n_heads, context, n_use_codebooks, rope_max_period. */
void *hipvoc_enc_load(const char *weights_path, int device_id);

/* Codebooks per frame in the output: min(16, available) (speech_tokenizer.rs:664-667). */
int hipvoc_enc_n_codebooks(void *h);
/* Input samples per output frame (1920 for the 12.5 Hz tokenizer). */
int hipvoc_enc_samples_per_frame(void *h);
/* Frames produced for n_samples: every strided conv floors (CausalConv1d::forward, 65-79). */
int64_t hipvoc_enc_n_frames(void *h, int64_t n_samples);
/* Algorithmic multiply-accumulates per input sample (throughput lines). */
double hipvoc_enc_macs_per_sample(void *h);

/* SpeechTokenizer::encode (speech_tokenizer.rs:646-675): audio[n_samples] (24 kHz mono f32, host) ->
 * codes[n_frames][n_codebooks] (i64, host, row-major: what voice files and Request::Vocode carry).
 * HIPVOC_E_ARG when the clip gives no frame, HIPVOC_E_CAP when codes_cap (in elements) is too small. */
int hipvoc_enc_encode(void *h, const float *audio, int64_t n_samples, int64_t *codes, size_t codes_cap, int64_t *n_frames);

/* B clips in one pass (no counterpart in the reference, which encodes one clip per call): shorter clips are
 * zero-padded at the end; every layer is causal, so each clip's codes are bit-identical to encoding it alone. */
int hipvoc_enc_encode_batch(void *h, const float *const *audio, const int64_t *n_samples, int B, int64_t *const *codes,
                            const size_t *codes_cap, int64_t *n_frames);

/* Device-resident form (throughput path; asynchronous on the handle's stream, complete after hipvoc_sync()):
 * B clips of exactly n_samples each at d_audio + i*stride (floats, device memory) -> d_codes[B][n_frames][n_codebooks]
 * (i64, device memory).  No counterpart in the reference. */
int hipvoc_enc_encode_device(void *h, const float *d_audio, int B, int64_t n_samples, size_t stride, int64_t *d_codes);

/* Parity hooks (tests only): stage numbering of oracle/enc_oracle.c (0 initial conv; 1+2i residual block i; 2+2i
 * down conv i; then the final conv, the transformer layers, the downsample conv, the semantic and the acoustic
 * input projections); out[C][L] row-major. */
int hipvoc_enc_num_stages(void *h);
int hipvoc_enc_debug_stage(void *h, const float *audio, int64_t n_samples, int stage, float *out, size_t out_cap, int *C, int64_t *L);

/* resample (speech_tokenizer.rs:725-741): linear interpolation evaluated in f64; host code, as in the reference.
 * out == NULL returns the output length; otherwise the number of samples written or a negative error. */
int64_t hipvoc_resample_linear(const float *x, int64_t n, uint32_t from_rate, uint32_t to_rate, float *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
