// Source of the `hip-vocoder` Cargo feature's Rust side.  Written against the reference tree;
// NOT compiled here (this image has no cargo / rustc).

// goes into src/worker/vocoder.rs
#[cfg(feature = "hip-vocoder")]
mod hip {
    use anyhow::{bail, Result};
    use std::ffi::{c_char, c_int, c_void, CStr, CString};
    extern "C" {
        fn hipvoc_load(path: *const c_char, device: c_int, dtype: c_int, unit_frames: c_int) -> *mut c_void;
        fn hipvoc_max_tokens(h: *mut c_void) -> c_int;
        fn hipvoc_decode(h: *mut c_void, codes: *const i64, n_frames: c_int,
                         out: *mut f32, cap: usize, n_out: *mut usize) -> c_int;
        fn hipvoc_decode_batch(h: *mut c_void, codes: *const i64, n_frames: *const i32, b: c_int,
                               out: *mut f32, out_off: *const usize, n_out: *mut usize) -> c_int;
        fn hipvoc_last_error(h: *mut c_void) -> *const c_char;
        fn hipvoc_free(h: *mut c_void);
    }
    pub struct HipVocoder { h: *mut c_void, pub max_tokens: usize }
    unsafe impl Send for HipVocoder {}
    unsafe impl Sync for HipVocoder {}
    fn err(h: *mut c_void) -> String {
        unsafe { CStr::from_ptr(hipvoc_last_error(h)).to_string_lossy().into_owned() }
    }
    impl HipVocoder {
        pub fn load(path: &std::path::Path) -> Result<Self> {
            let dev: i32 = std::env::var("QWEN3_TTS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let p = CString::new(path.to_str().unwrap_or(""))?;
            let h = unsafe { hipvoc_load(p.as_ptr(), dev, 0 /* f32 */, 64) };
            if h.is_null() { bail!("hip vocoder: {}", err(std::ptr::null_mut())); }
            let max_tokens = unsafe { hipvoc_max_tokens(h) } as usize;
            Ok(Self { h, max_tokens })
        }
        /// Whole request in one call: the library applies the 64-frame chunking of
        /// Vocoder::synthesize itself and batches the units on the GPU.
        pub fn synthesize(&mut self, codes: &[i64], n_tokens: usize) -> Result<Vec<f32>> {
            // the reference panics on the slice codes[pos*16..] (vocoder.rs:69-72); never hand C a short buffer
            if codes.len() < n_tokens * 16 || n_tokens > c_int::MAX as usize {
                bail!("hip vocoder: {} codes given, n_tokens*16 = {} needed", codes.len(), n_tokens * 16);
            }
            let mut out = vec![0f32; n_tokens * super::SAMPLES_PER_TOKEN];
            let mut n = 0usize;
            let rc = unsafe { hipvoc_decode(self.h, codes.as_ptr(), n_tokens as c_int,
                                            out.as_mut_ptr(), out.len(), &mut n) };
            if rc != 0 { bail!("hip vocoder: {} (code {})", err(self.h), rc); }
            out.truncate(n);
            Ok(out)
        }
    }
    impl HipVocoder {
        /// Many sequences, one batched decode (Request::VocodeBatch): bit-identical to separate `synthesize` calls.
        pub fn synthesize_batch(&mut self, seqs: &[Vec<i64>], n_tokens: &[usize]) -> Result<Vec<Vec<f32>>> {
            if seqs.len() != n_tokens.len() { bail!("hip vocoder: {} sequences, {} token counts", seqs.len(), n_tokens.len()); }
            let mut flat: Vec<i64> = Vec::new();
            let mut nf: Vec<i32> = Vec::with_capacity(seqs.len());
            let mut off: Vec<usize> = Vec::with_capacity(seqs.len());
            let mut total = 0usize;
            for (c, &n) in seqs.iter().zip(n_tokens) {
                if c.len() < n * 16 || n > i32::MAX as usize { bail!("hip vocoder: codes shorter than n_tokens*16"); }
                flat.extend_from_slice(&c[..n * 16]);
                nf.push(n as i32);
                off.push(total);
                total += n * super::SAMPLES_PER_TOKEN;
            }
            let mut out = vec![0f32; total.max(1)];
            let mut n_out = vec![0usize; seqs.len().max(1)];
            if flat.is_empty() { flat.push(0); }
            let rc = unsafe { hipvoc_decode_batch(self.h, flat.as_ptr(), nf.as_ptr(), seqs.len() as c_int,
                                                  out.as_mut_ptr(), off.as_ptr(), n_out.as_mut_ptr()) };
            if rc != 0 { bail!("hip vocoder: {} (code {})", err(self.h), rc); }
            Ok((0..seqs.len()).map(|b| out[off[b]..off[b] + n_out[b]].to_vec()).collect())
        }
    }
    impl Drop for HipVocoder { fn drop(&mut self) { unsafe { hipvoc_free(self.h) } } }
}

enum VocoderBackend {
    Onnx(OnnxVocoder),
    #[cfg(feature = "rknn-vocoder")] Rknn(RknnVocoder),
    #[cfg(feature = "hip-vocoder")] Hip(hip::HipVocoder),
}
// Vocoder::load: `if path_str.ends_with(".safetensors")` -> VocoderBackend::Hip (before the ".onnx" check)
// Vocoder::synthesize: `if let VocoderBackend::Hip(v) = &mut self.backend { return v.synthesize(codes, n_tokens); }`
//   placed before the chunk loop; the ONNX/RKNN arms keep the per-chunk loop unchanged.
