/*
 * hipvoc.h -- C ABI of the MI355X-native vocoder backend (libhipvoc.so / libhipvoc.a).
 *
 * This is the drop-in boundary behind qwen3-tts' Vocoder worker role: the functions below are what
 * a `#[cfg(feature = "hip-vocoder")] VocoderBackend::Hip` arm in src/worker/vocoder.rs binds over
 * FFI (INTEGRATION.md shows the Rust side).  Conventions mirror the reference's existing native
 * shim, cpp/code_pred_ggml_shim.cpp:11-75 as bound by src/worker/code_predictor_ggml.rs:15-26:
 * opaque `void*` handle, NULL on load failure, `int` 0 on success / negative on error,
 * caller-allocated outputs, explicit free.  No function aborts or throws across the boundary.
 *
 * Threading (src/worker/mod.rs:567-577,628-631): calls on one handle are serialised by the
 * caller (the worker holds a tokio Mutex); a handle may be used from different threads over time.
 * The library keeps no thread-local state (but the text of a failed load) and sets the HIP device on every call.
 *
 * Host entry points (hipvoc_run / decode / decode_batch / vocode_pcm16*) take and return HOST memory and are complete on
 * return.  They allocate nothing: pinned staging, device buffers, events and a copy stream for two passes in flight are
 * created once in hipvoc_load; pass k's samples leave the GPU as one copy while pass k+1 computes.
 * Device entry points (hipvoc_decode_units_device*) enqueue on the handle's own NON-BLOCKING stream, which has no implicit
 * ordering with the null stream or the caller's streams: d_codes must be complete on the device before the call, and
 * d_out is valid (and d_codes reusable) only after hipvoc_sync(h) -- or order against hipvoc_stream(h) with events.
 */
#ifndef HIPVOC_H
#define HIPVOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (negative) */
#define HIPVOC_OK 0
#define HIPVOC_E_ARG (-1)       /* bad argument (null pointer, n_frames < 0, ...) */
#define HIPVOC_E_CODE (-2)      /* a code index outside [0, codebook_size) */
#define HIPVOC_E_CAP (-3)       /* output capacity too small */
#define HIPVOC_E_STAGE (-4)     /* no such debug stage */
#define HIPVOC_E_HIP (-5)       /* HIP runtime / launch failure (text in hipvoc_last_error) */
#define HIPVOC_E_MODEL (-6)     /* weights file unreadable / unsupported architecture */
#define HIPVOC_E_SCRATCH (-7)   /* hipvoc_decode_units_device_noalloc: the pass does not fit the static scratch */
#define HIPVOC_E_NOMEM (-8)     /* host allocation failed inside the library (caught at the boundary, never thrown across it) */
#define HIPVOC_E_INTERNAL (-9)  /* any other C++ exception caught at the boundary (text in hipvoc_last_error) */

#define HIPVOC_DTYPE_F32 0
#define HIPVOC_DTYPE_BF16 1     /* bf16 conv-GEMM mode: operands of every conv / linear / transposed conv rounded to bf16
                                   (RNE), fp32 accumulate; RVQ, norms, attention and SnakeBeta arithmetic stay fp32 (SnakeBeta's
                                   sine by the hardware instruction).  From round 3 the decoder blocks' residual stream
                                   (every block but the first) and the final C->1 conv's input are rounded to bf16 as well
                                   where they are stored (env HIPVOC_RES_BF16=0 HIPVOC_FIN_BF16=0: fp32, +0.3 dB, +8 % time).
                                   Tolerance: SNR >= 30 dB, max-abs 3e-2 against the fp32 path on the seeded weight families
                                   (measured 32.4 - 33.7 dB); see DESIGN.md 7 / 7d and tests -- not the fp32 gates. */

#define HIPVOC_DTYPE_BF16X3 2   /* split-operand mode: every conv-GEMM operand is carried as two bf16 planes (hi + mid, 16
                                   mantissa bits) and every product is three bf16 MFMAs with fp32 accumulate; everything
                                   else as in the fp32 mode.  Meets the fp32 mode's waveform gates (max-abs 1e-3, SNR 60 dB
                                   vs the fp32 CPU path; measured margin in DESIGN.md section 7b), not bit-equal to it. */

#define HIPVOC_SAMPLES_PER_TOKEN 1920   /* src/worker/vocoder.rs:6, src/protocol.rs:122 */

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int hipvoc_device_count(void);

/*
 * Replaces OnnxVocoder::load (src/worker/vocoder.rs:137-162) + Vocoder::load (vocoder.rs:21-57).
 * weights_path: a .safetensors file holding the upstream `decoder.*` tensors of the Qwen3-TTS
 * speech tokenizer (F32/F16/BF16); architecture is derived from tensor shapes, the few
 * hyper-parameters shapes cannot tell come from the file's __metadata__ (defaults: 16 heads,
 * window 72, rope theta 1e4, rms eps 1e-5; keys num_attention_heads, sliding_window, rope_theta, rms_norm_eps).
 * transconv_trim_right_only=1 selects the other trim policy of the decoder blocks' transposed convs: k - s samples off
 * the right only (exactly 1920 samples per frame) instead of off both sides (1920*T - 555 per unit, the default, which
 * follows the published sibling decoder).  Which one the reference's shipped vocoder.onnx has cannot be read offline.
 * unit_frames: frames per decode unit = the reference's `max_tokens` (ONNX input dim 1; 64).
 * Returns NULL on failure; hipvoc_last_error(NULL) then holds the reason.
 */
void *hipvoc_load(const char *weights_path, int device_id, int dtype, int unit_frames);

/*
 * Replaces find_vocoder_model (src/worker/mod.rs:792-813): the first existing candidate under models_dir (or its
 * "vocoder" sub-directory, the role directory download::ensure_models returns, src/download.rs:43-94), in this order:
 * vocoder.safetensors, vocoder_64.safetensors, vocoder_256.safetensors, model.safetensors,
 * speech_tokenizer/model.safetensors.  Returns 0 and the path in path_out, or HIPVOC_E_MODEL with the reason in
 * path_out ("No vocoder model found in <dir> ...", as the reference words it).  Host code only; no GPU needed.
 */
int hipvoc_find_vocoder_model(const char *models_dir, char *path_out, size_t cap);

/* `pub max_tokens` of the reference's Vocoder (vocoder.rs:17). */
int hipvoc_max_tokens(void *h);

/* Codebooks per frame the loaded decoder expects (16 for the shipped model: the `16` of src/worker/vocoder.rs:62,69-72 and
 * src/worker/mod.rs:513), derived from the weights file. */
int hipvoc_n_codebooks(void *h);

/* Samples one full unit produces: out_len(unit_frames) = 1920*unit_frames - 555 for the shipped
 * topology (length of OnnxVocoder::run's output). */
int64_t hipvoc_unit_samples(void *h);

/*
 * Replaces OnnxVocoder::run (vocoder.rs:164-172): one unit of exactly max_tokens frames
 * (codes row-major [max_tokens, 16] i64, as fed to the ONNX session) -> hipvoc_unit_samples() f32.
 */
int hipvoc_run(void *h, const int64_t *codes, int max_tokens, float *out, size_t out_cap, size_t *n_out);

/*
 * Replaces the body of Vocoder::synthesize (vocoder.rs:60-85): splits n_frames into independent
 * max_tokens-frame units, pads the last with code 0, decodes ALL units in one batched pass and
 * keeps min(chunk_len*1920, unit_samples) samples of each, concatenated.  codes: [n_frames,16].
 */
int hipvoc_decode(void *h, const int64_t *codes, int n_frames, float *out, size_t out_cap, size_t *n_out);

/*
 * Batched form (many utterances per request).  codes: the B sequences
 * concatenated ([sum n_frames, 16]); sequence b's samples are written at out + out_off[b]
 * (capacity n_frames[b]*1920 each) and its sample count to n_out[b].  Every sequence is chunked
 * exactly like hipvoc_decode, so results equal B separate calls bit for bit.
 */
int hipvoc_decode_batch(void *h, const int64_t *codes, const int32_t *n_frames, int B, float *out,
                        const size_t *out_off, size_t *n_out);

/*
 * Replaces Worker::handle_vocode's payload handling (src/worker/mod.rs:507-531): codes arrive as
 * i64 little-endian bytes ([n_frames,16]), audio leaves as i16 little-endian PCM,
 * (s*32767.0).clamp(-32768.0,32767.0) as i16, converted on the GPU.  pcm capacity in samples.
 */
int hipvoc_vocode_pcm16(void *h, const uint8_t *codes_le, int n_frames, int16_t *pcm, size_t pcm_cap,
                        size_t *n_samples);

/* Batched form of hipvoc_vocode_pcm16 (the VocodeBatch request): codes_le = the B sequences' i64-LE bytes
 * concatenated; sequence b's PCM goes to pcm + out_off[b] (capacity n_frames[b]*1920 samples), its count to n_samples[b]. */
int hipvoc_vocode_pcm16_batch(void *h, const uint8_t *codes_le, const int32_t *n_frames, int B, int16_t *pcm,
                              const size_t *out_off, size_t *n_samples);

/*
 * Device-resident batched units (what bench.py times): d_codes [n_units, max_tokens, 16] i64 and
 * d_out [n_units, unit_stride] f32 are DEVICE pointers (unit_stride >= hipvoc_unit_samples, in
 * floats).  Work is enqueued on the handle's stream; call hipvoc_sync to wait and collect
 * deferred errors (out-of-range codes are detected on the device).
 * A call of more than hipvoc_group units runs as several passes.  In the bf16 mode (env HIPVOC_LANES, default 2 there,
 * 1 in fp32 / bf16x3) those passes alternate between the handle's stream and a second stream of the handle with a
 * workspace of its own, so that one pass's tail overlaps the next one's head; the second stream starts behind everything
 * already enqueued on the handle's stream and the handle's stream ends behind it, so the ordering contract above
 * (hipvoc_sync, or events on hipvoc_stream) is unchanged, and the samples are those of one lane bit for bit.
 */
int hipvoc_decode_units_device(void *h, const int64_t *d_codes, int n_units, float *d_out, size_t unit_stride);
int hipvoc_sync(void *h);

/*
 * The entry point that allocates nothing.  It has NO counterpart in the reference; it is for callers that cannot call
 * hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments,
 * kernels, launch plan and results (bit for bit) as hipvoc_decode_units_device, but the activation workspace is a
 * file-scope `static __device__` array of HIPVOC_NOALLOC_SCRATCH_BYTES (384 MiB) that each GPU the library runs on
 * holds until the process ends, instead of the handle's hipMalloc'ed buffers: no hipMalloc, hipFree, event or stream
 * creation happens in the call (the first call for a given n_units builds its launch plan, host memory only).
 * Limit: the pass must fit the array -- with the full-size decoder, 2 units of 64 frames (one 10 s utterance) in
 * fp32 and bf16, 4 in bf16x3; never more than 128 units.  Larger requests fail with HIPVOC_E_SCRATCH and the
 * needed size in hipvoc_last_error; nothing is launched.  The array keeps the last call's contents and is NOT
 * cleared: nothing in it has to start at zero (every byte a result depends on is written by the pass's own kernels
 * first).  All handles on one GPU share that GPU's array.  The library orders their passes ON THE DEVICE: each call
 * makes its stream wait for the previous no-alloc pass on that GPU (one event per GPU, created by hipvoc_load) and holds
 * a per-GPU host lock while it enqueues, so two handles (the worker's --contexts 2) called one after the other without a
 * sync in between do not overlap in the array.  What still races: reading d_out, or reusing d_codes, before the
 * handle's own stream has finished (hipvoc_sync), as with hipvoc_decode_units_device.  Work is enqueued on the handle's
 * stream like hipvoc_decode_units_device.
 */
#define HIPVOC_NOALLOC_SCRATCH_BYTES (384u << 20)
int hipvoc_decode_units_device_noalloc(void *h, const int64_t *d_codes, int n_units, float *d_out, size_t unit_stride);

/* The hipStream_t the handle launches on (for hipEvent timing by the caller). */
void *hipvoc_stream(void *h);

/* Units decoded per pass (workspace is sized for this many; default 64, env HIPVOC_GROUP). */
int hipvoc_group_units(void *h);

/* Lanes of the handle (1 or 2): passes of one call that may be in flight side by side, each with a stream and a workspace of its
 * own (see hipvoc_decode_units_device; env HIPVOC_LANES, read by hipvoc_load).  No counterpart in the reference: a report for
 * benchmarks and capacity planning (a second lane is a second workspace). */
int hipvoc_lanes(void *h);

/* Algorithmic work of one unit, recomputed from the loaded tensor shapes:
 * multiply-accumulates of all GEMM-shaped stages. */
int64_t hipvoc_unit_macs(void *h);

/*
 * Diagnostics for parity tests: output of intermediate stage `stage` for ONE unit of n_frames
 * frames (no chunking; 1 <= n_frames <= max_tokens), channel-first [C][L] into out.
 * Stage numbering = oracle/voc_oracle.c decoder_forward: 0 RVQ dequant, 1 pre_conv, 2 input_proj,
 * 3.. transformer layers, then output_proj, upsample (2 each), decoder[0], 4 per decoder block,
 * last = waveform.  Stage 1000 = raw dequant sums [2*256][T] (the bit-exact contract).
 */
int hipvoc_num_stages(void *h);
int hipvoc_debug_stage(void *h, const int64_t *codes, int n_frames, int stage, float *out, size_t out_cap,
                       int *C, int64_t *L);

/* Per-step-family device time of the LAST pass (one group of units) when profiling is on
 * (hipvoc_set_profile(h,1) or env HIPVOC_PROFILE=1): hipEvents are recorded on the handle's stream
 * around every launch and the stream is synchronised after the pass.  names[i] is
 * "<step>|<kernel symbol>", ms[i] the summed device time, flops[i] the summed algorithmic FLOPs
 * (2 x MAC), launches[i] the launch count.  Arrays hold up to cap entries; returns the count. */
int hipvoc_set_profile(void *h, int on);
int hipvoc_last_profile(void *h, const char **names, float *ms, double *flops, int *launches, int cap);

/* Text of the last error on this handle, or (h == NULL) of the CALLING THREAD's last failed hipvoc_load /
 * hipvoc_enc_load: per thread, so concurrent loads do not race.  The pointer stays valid until the next failing call
 * on that handle / that thread's next failed load. */
const char *hipvoc_last_error(void *h);

void hipvoc_free(void *h);

#ifdef __cplusplus
}
#endif
#endif
