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
    /// One libhipvoc handle per GPU listed in QWEN3_TTS_HIP_DEVICES (default: QWEN3_TTS_HIP_DEVICE, else 0).
    pub struct HipVocoder { hs: Vec<*mut c_void>, next: usize, pub max_tokens: usize }
    unsafe impl Send for HipVocoder {}
    unsafe impl Sync for HipVocoder {}
    /// A handle on its way to a scoped worker thread (the library takes hipSetDevice per call and keeps no thread-local state).
    #[derive(Clone, Copy)]
    struct Handle(*mut c_void);
    unsafe impl Send for Handle {}
    fn err(h: *mut c_void) -> String {
        unsafe { CStr::from_ptr(hipvoc_last_error(h)).to_string_lossy().into_owned() }
    }
    /// QWEN3_TTS_VOCODER_DTYPE = fp32 | bf16 | bf16x3 (env handling as mod.rs:815-833 does for ORT_DYLIB_PATH).
    /// fp32 is the reference's precision and the default; bf16 carries its own stated tolerance;
    /// bf16x3 meets the fp32 gates on the bf16 matrix cores (section 7b).
    fn dtype_from_env() -> Result<c_int> {
        match std::env::var("QWEN3_TTS_VOCODER_DTYPE").ok().as_deref().map(str::trim) {
            None | Some("") | Some("fp32") | Some("f32") => Ok(0),
            Some("bf16") => Ok(1),
            Some("bf16x3") => Ok(2),
            Some(other) => bail!("QWEN3_TTS_VOCODER_DTYPE={}: expected fp32, bf16 or bf16x3", other),
        }
    }
    /// QWEN3_TTS_HIP_DEVICES = "0,1,2,3" (one handle per entry; an entry may repeat); else QWEN3_TTS_HIP_DEVICE; else 0.
    fn devices_from_env() -> Result<Vec<c_int>> {
        if let Ok(list) = std::env::var("QWEN3_TTS_HIP_DEVICES") {
            let devs: Result<Vec<c_int>, _> = list.split(',').map(|t| t.trim().parse::<c_int>()).collect();
            return match devs {
                Ok(d) if !d.is_empty() && d.iter().all(|&x| x >= 0) => Ok(d),
                _ => bail!("QWEN3_TTS_HIP_DEVICES={}: expected a comma-separated list of device indices", list),
            };
        }
        Ok(vec![std::env::var("QWEN3_TTS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)])
    }
    fn decode_batch_on(h: *mut c_void, seqs: &[&Vec<i64>], n_tokens: &[usize]) -> Result<Vec<Vec<f32>>> {
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
        let rc = unsafe { hipvoc_decode_batch(h, flat.as_ptr(), nf.as_ptr(), seqs.len() as c_int,
                                              out.as_mut_ptr(), off.as_ptr(), n_out.as_mut_ptr()) };
        if rc != 0 { bail!("hip vocoder: {} (code {})", err(h), rc); }
        Ok((0..seqs.len()).map(|b| out[off[b]..off[b] + n_out[b]].to_vec()).collect())
    }
    impl HipVocoder {
        pub fn load(path: &std::path::Path) -> Result<Self> {
            let dtype = dtype_from_env()?;
            let devs = devices_from_env()?;
            let p = CString::new(path.to_str().unwrap_or(""))?;
            let mut hs = Vec::with_capacity(devs.len());
            for &dev in &devs {
                let h = unsafe { hipvoc_load(p.as_ptr(), dev, dtype, 64) };
                if h.is_null() {
                    let e = err(std::ptr::null_mut());
                    for &g in &hs { unsafe { hipvoc_free(g) } }
                    bail!("hip vocoder (device {}): {}", dev, e);
                }
                hs.push(h);
            }
            let max_tokens = unsafe { hipvoc_max_tokens(hs[0]) } as usize;
            Ok(Self { hs, next: 0, max_tokens })
        }
        /// Whole request in one call: the library applies the 64-frame chunking of
        /// Vocoder::synthesize itself and batches the units on the GPU.  With several devices, requests rotate over them.
        pub fn synthesize(&mut self, codes: &[i64], n_tokens: usize) -> Result<Vec<f32>> {
            // the reference panics on the slice codes[pos*16..] (vocoder.rs:69-72); never hand C a short buffer
            if codes.len() < n_tokens * 16 || n_tokens > c_int::MAX as usize {
                bail!("hip vocoder: {} codes given, n_tokens*16 = {} needed", codes.len(), n_tokens * 16);
            }
            let h = self.hs[self.next % self.hs.len()];
            self.next = self.next.wrapping_add(1);
            let mut out = vec![0f32; n_tokens * super::SAMPLES_PER_TOKEN];
            let mut n = 0usize;
            let rc = unsafe { hipvoc_decode(h, codes.as_ptr(), n_tokens as c_int,
                                            out.as_mut_ptr(), out.len(), &mut n) };
            if rc != 0 { bail!("hip vocoder: {} (code {})", err(h), rc); }
            out.truncate(n);
            Ok(out)
        }
        /// Many sequences, one batched decode per device (Request::VocodeBatch): bit-identical to separate `synthesize`
        /// calls.  With several devices the sequences are dealt out in contiguous runs balanced by frame count, one scoped
        /// thread per device (units are independent: no exchange between devices), and returned in request order.
        pub fn synthesize_batch(&mut self, seqs: &[Vec<i64>], n_tokens: &[usize]) -> Result<Vec<Vec<f32>>> {
            if seqs.len() != n_tokens.len() { bail!("hip vocoder: {} sequences, {} token counts", seqs.len(), n_tokens.len()); }
            let nd = self.hs.len().min(seqs.len().max(1));
            if nd <= 1 {
                let refs: Vec<&Vec<i64>> = seqs.iter().collect();
                return decode_batch_on(self.hs[0], &refs, n_tokens);
            }
            // contiguous runs with about total/nd frames each
            let total: usize = n_tokens.iter().sum();
            let mut cuts = vec![0usize];
            let mut acc = 0usize;
            for (i, &n) in n_tokens.iter().enumerate() {
                acc += n;
                let k = cuts.len();                                  // the boundary that closes run k-1
                if k < nd && acc * nd >= k * total.max(1) { cuts.push(i + 1); }
            }
            while cuts.len() < nd { cuts.push(seqs.len()); }
            cuts.push(seqs.len());
            let handles: Vec<Handle> = self.hs.iter().map(|&h| Handle(h)).collect();
            let parts: Vec<Result<Vec<Vec<f32>>>> = std::thread::scope(|sc| {
                let joins: Vec<_> = (0..nd).map(|d| {
                    let (lo, hi) = (cuts[d], cuts[d + 1]);
                    let h = handles[d];
                    let refs: Vec<&Vec<i64>> = seqs[lo..hi].iter().collect();
                    let nt = &n_tokens[lo..hi];
                    sc.spawn(move || { let h = h; decode_batch_on(h.0, &refs, nt) })
                }).collect();
                joins.into_iter().map(|j| j.join().unwrap_or_else(|_| Err(anyhow::anyhow!("hip vocoder: worker thread panicked")))).collect()
            });
            let mut out = Vec::with_capacity(seqs.len());
            for p in parts { out.extend(p?); }
            Ok(out)
        }
    }
    impl Drop for HipVocoder { fn drop(&mut self) { for &h in &self.hs { unsafe { hipvoc_free(h) } } } }
}

enum VocoderBackend {
    Onnx(OnnxVocoder),
    #[cfg(feature = "rknn-vocoder")] Rknn(RknnVocoder),
    #[cfg(feature = "hip-vocoder")] Hip(hip::HipVocoder),
}
// Vocoder::load: `if path_str.ends_with(".safetensors")` -> VocoderBackend::Hip (before the ".onnx" check)
// Vocoder::synthesize: `if let VocoderBackend::Hip(v) = &mut self.backend { return v.synthesize(codes, n_tokens); }`
//   placed before the chunk loop; the ONNX/RKNN arms keep the per-chunk loop unchanged.
