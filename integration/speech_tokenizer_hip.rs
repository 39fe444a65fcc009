// Source of the `hip-vocoder` Cargo feature's Rust side.  Written against the reference tree;
// NOT compiled here (this image has no cargo / rustc).

// replaces the candle implementation in src/speech_tokenizer.rs under the feature (encode_wav / resample stay)
use anyhow::Result;
use std::ffi::{c_char, c_int, c_void, CStr, CString};

fn last_error(h: *mut c_void) -> String {
    unsafe { CStr::from_ptr(hipvoc_last_error(h)).to_string_lossy().into_owned() }
}

extern "C" {
    fn hipvoc_enc_load(path: *const c_char, device_id: c_int) -> *mut c_void;
    fn hipvoc_enc_n_frames(h: *mut c_void, n_samples: i64) -> i64;
    fn hipvoc_enc_n_codebooks(h: *mut c_void) -> c_int;
    fn hipvoc_enc_encode(h: *mut c_void, audio: *const f32, n_samples: i64, codes: *mut i64, cap: usize, n_frames: *mut i64) -> c_int;
    fn hipvoc_last_error(h: *mut c_void) -> *const c_char;
    fn hipvoc_free(h: *mut c_void);
}

pub struct SpeechTokenizer { h: *mut c_void }
unsafe impl Send for SpeechTokenizer {}

impl SpeechTokenizer {
    pub fn load(model_path: &str) -> Result<Self> {
        let p = CString::new(model_path)?;
        let h = unsafe { hipvoc_enc_load(p.as_ptr(), 0) };
        if h.is_null() { anyhow::bail!("hipvoc_enc_load: {}", last_error(std::ptr::null_mut())); }
        Ok(Self { h })
    }
    pub fn encode(&self, audio: &[f32]) -> Result<Vec<Vec<i64>>> {
        let nq = unsafe { hipvoc_enc_n_codebooks(self.h) } as usize;
        let t = unsafe { hipvoc_enc_n_frames(self.h, audio.len() as i64) }.max(0) as usize;
        let mut flat = vec![0i64; t.max(1) * nq];
        let mut n = 0i64;
        let rc = unsafe { hipvoc_enc_encode(self.h, audio.as_ptr(), audio.len() as i64, flat.as_mut_ptr(), flat.len(), &mut n) };
        if rc != 0 { anyhow::bail!("hipvoc_enc_encode: {}", last_error(self.h)); }
        Ok(flat[..n as usize * nq].chunks(nq).map(|r| r.to_vec()).collect())      // Vec<Vec<i64>> as at :671
    }
}
impl Drop for SpeechTokenizer { fn drop(&mut self) { unsafe { hipvoc_free(self.h) } } }
