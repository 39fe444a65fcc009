"""ctypes binding of the encoder entry points of libhipvoc.so (include/hipvoc_encoder.h).

Host-side mirror of the reference's SpeechTokenizer (src/speech_tokenizer.rs:608-741): ``HipSpeechTokenizer.load(path)``,
``.encode(audio) -> [n_tokens, 16] int64``, ``resample(samples, from_rate, to_rate)``; ``encode_samples`` does what
``encode_wav`` does after the file is read (mono mix, resample to 24 kHz; speech_tokenizer.rs:698-720).

There is NO fallback: if the shared library is missing or no HIP device is present, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np

from .hipvoc import HipVocError, load_library

SAMPLE_RATE = 24000             # speech_tokenizer.rs:15

ENC_EXPORTS = [
    "hipvoc_enc_load", "hipvoc_enc_n_codebooks", "hipvoc_enc_samples_per_frame", "hipvoc_enc_n_frames",
    "hipvoc_enc_macs_per_sample", "hipvoc_enc_encode", "hipvoc_enc_encode_batch", "hipvoc_enc_encode_device",
    "hipvoc_enc_num_stages",
    "hipvoc_enc_debug_stage", "hipvoc_resample_linear",
]

_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, sz = C.c_void_p, C.c_size_t
        L.hipvoc_enc_load.restype = vp
        L.hipvoc_enc_load.argtypes = [C.c_char_p, C.c_int]
        L.hipvoc_enc_n_codebooks.argtypes = [vp]
        L.hipvoc_enc_samples_per_frame.argtypes = [vp]
        L.hipvoc_enc_n_frames.restype = C.c_int64
        L.hipvoc_enc_n_frames.argtypes = [vp, C.c_int64]
        L.hipvoc_enc_macs_per_sample.restype = C.c_double
        L.hipvoc_enc_macs_per_sample.argtypes = [vp]
        L.hipvoc_enc_encode.argtypes = [vp, vp, C.c_int64, vp, sz, C.POINTER(C.c_int64)]
        L.hipvoc_enc_encode_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
        L.hipvoc_enc_encode_device.argtypes = [vp, vp, C.c_int, C.c_int64, sz, vp]
        L.hipvoc_enc_num_stages.argtypes = [vp]
        L.hipvoc_enc_debug_stage.argtypes = [vp, vp, C.c_int64, C.c_int, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.hipvoc_resample_linear.restype = C.c_int64
        L.hipvoc_resample_linear.argtypes = [vp, C.c_int64, C.c_uint32, C.c_uint32, vp, C.c_int64]
        _bound = True
    return L


def resample(samples, from_rate: int, to_rate: int) -> np.ndarray:
    """Linear resampling as the reference does it (speech_tokenizer.rs:725-741)."""
    x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    L = _lib()
    n = int(L.hipvoc_resample_linear(x.ctypes.data, x.size, from_rate, to_rate, None, 0))
    if n < 0:
        raise HipVocError(n, "resample: bad argument")
    out = np.empty(n, dtype=np.float32)
    L.hipvoc_resample_linear(x.ctypes.data, x.size, from_rate, to_rate, out.ctypes.data, out.size)
    return out


def read_wav(path: str):
    """Minimal RIFF/WAVE reader for what hound hands the reference (speech_tokenizer.rs:681-696): PCM integer of 8 / 16 /
    24 / 32 bits (scaled by 1 / 2^(bits-1); 8-bit files are unsigned, offset 128, as hound reads them) and IEEE float32.
    -> (interleaved float32 samples, sample_rate, channels).  Raises ValueError on anything else."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"Failed to open WAV file: {path}")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise ValueError(f"Failed to open WAV file: {path} (missing fmt / data chunk)")
    tag, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                      # WAVE_FORMAT_EXTENSIBLE: the real tag leads the sub-format GUID
        tag = struct.unpack("<H", fmt[24:26])[0]
    if channels < 1 or rate < 1:
        raise ValueError("WAV: bad channel count / sample rate")
    if tag == 3 and bits == 32:
        x = np.frombuffer(payload[: len(payload) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == 1 and bits in (8, 16, 24, 32):
        scale = np.float32(1 << (bits - 1))
        if bits == 8:
            v = np.frombuffer(payload, dtype=np.uint8).astype(np.int32) - 128
        elif bits == 16:
            v = np.frombuffer(payload[: len(payload) // 2 * 2], dtype="<i2").astype(np.int32)
        elif bits == 32:
            v = np.frombuffer(payload[: len(payload) // 4 * 4], dtype="<i4")
        else:
            b = np.frombuffer(payload[: len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = (v.astype(np.float32) / scale).astype(np.float32)      # `s as f32 / max_val`
    else:
        raise ValueError(f"WAV: unsupported sample format (tag {tag}, {bits} bits)")
    return x, int(rate), int(channels)


class HipSpeechTokenizer:
    """SpeechTokenizer: 24 kHz mono waveform -> codec tokens on one MI355X (speech_tokenizer.rs:608-675)."""

    def __init__(self, handle, lib):
        self._h = handle
        self._L = lib
        self.n_codebooks = int(lib.hipvoc_enc_n_codebooks(handle))
        self.samples_per_frame = int(lib.hipvoc_enc_samples_per_frame(handle))
        self.macs_per_sample = float(lib.hipvoc_enc_macs_per_sample(handle))

    @classmethod
    def load(cls, model_path: str, device: int = 0) -> "HipSpeechTokenizer":
        L = _lib()
        h = L.hipvoc_enc_load(os.fsencode(model_path), device)
        if not h:
            raise HipVocError(-6, L.hipvoc_last_error(None).decode())
        return cls(h, L)

    def close(self):
        if self._h:
            self._L.hipvoc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise HipVocError(rc, self._L.hipvoc_last_error(self._h).decode())

    def n_frames(self, n_samples: int) -> int:
        return int(self._L.hipvoc_enc_n_frames(self._h, n_samples))

    def encode(self, audio) -> np.ndarray:
        """audio: 24 kHz mono f32 -> codes [n_tokens, n_codebooks] int64 (SpeechTokenizer::encode, 646-675)."""
        a = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        T = max(self.n_frames(a.size), 1)
        out = np.empty((T, self.n_codebooks), dtype=np.int64)
        n = C.c_int64(0)
        self._check(self._L.hipvoc_enc_encode(self._h, a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value]

    def encode_batch(self, clips: Sequence[np.ndarray]) -> List[np.ndarray]:
        """Several clips in one pass; each result equals encode() of that clip alone, bit for bit."""
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in clips]
        B = len(arrs)
        if B == 0:
            return []
        outs = [np.empty((max(self.n_frames(a.size), 1), self.n_codebooks), dtype=np.int64) for a in arrs]
        ap = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        ns = (C.c_int64 * B)(*[a.size for a in arrs])
        cp = (C.c_void_p * B)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * B)(*[o.size for o in outs])
        nf = (C.c_int64 * B)()
        self._check(self._L.hipvoc_enc_encode_batch(self._h, ap, ns, B, cp, caps, nf))
        return [o[: nf[i]] for i, o in enumerate(outs)]

    def encode_device(self, d_audio_ptr: int, B: int, n_samples: int, stride: int, d_codes_ptr: int):
        """B equal-length clips resident in HBM -> codes in HBM (asynchronous; .sync() to wait)."""
        self._check(self._L.hipvoc_enc_encode_device(self._h, d_audio_ptr, B, n_samples, stride, d_codes_ptr))

    def sync(self):
        self._check(self._L.hipvoc_sync(self._h))

    def encode_samples(self, samples, sample_rate: int, channels: int = 1) -> np.ndarray:
        """What encode_wav does once the file is decoded (698-720): interleaved channels are averaged to mono, other
        rates are resampled linearly to 24 kHz."""
        x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        if channels > 1:
            x = x[: x.size // channels * channels].reshape(-1, channels)
            x = (x.sum(axis=1, dtype=np.float32) / np.float32(channels)).astype(np.float32)
        if sample_rate != SAMPLE_RATE:
            x = resample(x, sample_rate, SAMPLE_RATE)
        return self.encode(x)

    def encode_wav(self, wav_path: str) -> np.ndarray:
        """SpeechTokenizer::encode_wav (speech_tokenizer.rs:680-721): RIFF/WAVE file -> codes.  Integer samples are
        divided by 2^(bits-1) (687-691), float samples taken as they are (693-695); then mono mix and resampling as in
        encode_samples."""
        samples, rate, channels = read_wav(wav_path)
        return self.encode_samples(samples, rate, channels)

    # ---- parity hooks ----------------------------------------------------------------------
    @property
    def n_stages(self) -> int:
        return int(self._L.hipvoc_enc_num_stages(self._h))

    def stage(self, audio, stage: int) -> np.ndarray:
        a = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        out = np.empty(max(a.size * 64, 1 << 16), dtype=np.float32)
        c, l = C.c_int(0), C.c_int64(0)
        self._check(self._L.hipvoc_enc_debug_stage(self._h, a.ctypes.data, a.size, stage, out.ctypes.data, out.size,
                                                   C.byref(c), C.byref(l)))
        return out[: c.value * l.value].reshape(c.value, l.value).copy()

    def set_profile(self, on: bool):
        self._check(self._L.hipvoc_set_profile(self._h, 1 if on else 0))

    def last_profile(self):
        cap = 128
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        ln = (C.c_int * cap)()
        n = self._L.hipvoc_last_profile(self._h, names, ms, fl, ln, cap)
        return {names[i].decode(): (float(ms[i]), float(fl[i]), int(ln[i])) for i in range(min(n, cap))}
