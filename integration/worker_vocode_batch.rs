// Source of the `hip-vocoder` feature's Rust side.  Written against the reference tree; NOT compiled
// here (this image has no cargo / rustc).  Three fragments: the worker's VocodeBatch handler, model discovery, download.

// ---- src/worker/mod.rs, Worker::dispatch (mod.rs:260-294): one more arm, beside `Request::Vocode` (mod.rs:288) ---------
//     Request::VocodeBatch { codes, n_tokens } => self.handle_vocode_batch(&codes, &n_tokens),
// (dispatch is a sync `fn dispatch(&mut self, req: Request) -> Result<Response>`; every handler returns
//  `Result<Response>` built by `ok(ResponseData::..)`, mod.rs:702-708.)

impl Worker {
    /// Several sequences, one batched GPU decode.  Per sequence exactly what handle_vocode
    /// (mod.rs:507-531) does: LE bytes -> i64, synthesize, f32 -> i16 -> LE bytes; a sequence with fewer code bytes than
    /// n_tokens*16*8 is an error response instead of the silent truncation of mod.rs:513-516's chunks_exact.
    fn handle_vocode_batch(&mut self, codes: &[Vec<u8>], n_tokens: &[usize]) -> Result<Response> {
        let voc = match &mut self.state {                                   // as handle_vocode, mod.rs:508-511
            WorkerState::Vocoder(vs) => &mut vs.vocoder,
            _ => anyhow::bail!("This worker does not have a vocoder"),
        };
        if codes.len() != n_tokens.len() {
            anyhow::bail!("VocodeBatch: {} code buffers but {} token counts", codes.len(), n_tokens.len());
        }
        let mut seqs: Vec<Vec<i64>> = Vec::with_capacity(codes.len());
        for (bytes, &n) in codes.iter().zip(n_tokens) {
            let need = n.checked_mul(16 * 8).ok_or_else(|| anyhow::anyhow!("VocodeBatch: n_tokens overflows"))?;
            if bytes.len() < need {
                anyhow::bail!("VocodeBatch: codes shorter than n_tokens*16");
            }
            seqs.push(bytes[..need].chunks_exact(8)
                .map(|c| i64::from_le_bytes([c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]]))
                .collect());
        }
        let waves = voc.synthesize_batch(&seqs, n_tokens)?;                 // vocoder.rs: below and hip_vocoder.rs
        let mut audio = Vec::with_capacity(waves.len());
        let mut n_samples = Vec::with_capacity(waves.len());
        for w in waves {
            let pcm: Vec<i16> = w.iter().map(|&s| (s * 32767.0).clamp(-32768.0, 32767.0) as i16).collect();   // mod.rs:521-524
            n_samples.push(pcm.len());
            audio.push(pcm.iter().flat_map(|s| s.to_le_bytes()).collect::<Vec<u8>>());
        }
        ok(ResponseData::VocodeBatch { audio, n_samples })
    }
}

// src/worker/vocoder.rs, impl Vocoder: the batched form.  Backends without a batch entry point run the ordinary loop.
impl Vocoder {
    pub fn synthesize_batch(&mut self, seqs: &[Vec<i64>], n_tokens: &[usize]) -> Result<Vec<Vec<f32>>> {
        #[cfg(feature = "hip-vocoder")]
        if let VocoderBackend::Hip(v) = &mut self.backend {
            return v.synthesize_batch(seqs, n_tokens);
        }
        seqs.iter().zip(n_tokens).map(|(c, &n)| self.synthesize(c, n)).collect()
    }
}

// ---- src/worker/mod.rs:792-813, find_vocoder_model: the hip-vocoder candidates, first ------------------------------------
fn find_vocoder_model(models_dir: &Path) -> Result<PathBuf> {
    let mut candidates = Vec::new();

    // HIP backend: the upstream speech-tokenizer checkpoint (decoder.* tensors), no ONNX export step.  Same order as
    // hipvoc_find_vocoder_model (include/hipvoc.h), which a C host uses instead.
    #[cfg(feature = "hip-vocoder")]
    {
        candidates.push(models_dir.join("vocoder.safetensors"));
        candidates.push(models_dir.join("vocoder_64.safetensors"));
        candidates.push(models_dir.join("vocoder_256.safetensors"));
        candidates.push(models_dir.join("model.safetensors"));
        candidates.push(models_dir.join("speech_tokenizer").join("model.safetensors"));
    }

    // RKNN models (only when feature enabled)
    #[cfg(feature = "rknn-vocoder")]
    {
        candidates.push(models_dir.join("vocoder.rknn"));
        candidates.push(models_dir.join("vocoder_64.rknn"));
        candidates.push(models_dir.join("vocoder_256.rknn"));
    }

    // ONNX models (default, noise-free)
    candidates.push(models_dir.join("vocoder.onnx"));
    candidates.push(models_dir.join("vocoder_64.onnx"));
    candidates.push(models_dir.join("vocoder_256.onnx"));

    candidates
        .iter()
        .find(|p| p.exists())
        .cloned()
        .ok_or_else(|| anyhow::anyhow!("No vocoder model found in {}", models_dir.display()))
}

// ---- src/download.rs:35-39: which file the vocoder role fetches ------------------------------------------------------
#[cfg(feature = "hip-vocoder")]
const VOCODER_FILES: &[&str] = &["vocoder/vocoder.safetensors"];

#[cfg(all(feature = "rknn-vocoder", not(feature = "hip-vocoder")))]
const VOCODER_FILES: &[&str] = &["vocoder/vocoder.rknn"];

#[cfg(not(any(feature = "rknn-vocoder", feature = "hip-vocoder")))]
const VOCODER_FILES: &[&str] = &["vocoder/vocoder.onnx"];
