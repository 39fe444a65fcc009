"""ctypes binding of libhipvoc.so (include/hipvoc.h) -- the same C ABI the Rust host binds.

Host-side mirror of the reference's Vocoder interface (src/worker/vocoder.rs:15-86):
``HipVocoder.load(path)``, ``.synthesize(codes, n_tokens)``, ``.max_tokens`` and of
``Worker::handle_vocode`` (src/worker/mod.rs:507-531): ``.handle_vocode(codes_bytes, n_tokens)``.

There is NO fallback: if the shared library is missing or no HIP device is present, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES_PER_TOKEN = 1920        # src/worker/vocoder.rs:6
SAMPLE_RATE = 24000             # src/protocol.rs:121

EXPORTS = [
    "hipvoc_device_count", "hipvoc_find_vocoder_model", "hipvoc_load", "hipvoc_max_tokens", "hipvoc_n_codebooks", "hipvoc_vocode_pcm16_batch", "hipvoc_unit_samples", "hipvoc_run",
    "hipvoc_decode", "hipvoc_decode_batch", "hipvoc_vocode_pcm16", "hipvoc_decode_units_device", "hipvoc_decode_units_device_noalloc",
    "hipvoc_sync",
    "hipvoc_stream", "hipvoc_group_units", "hipvoc_lanes", "hipvoc_unit_macs", "hipvoc_num_stages", "hipvoc_debug_stage",
    "hipvoc_set_profile", "hipvoc_last_profile", "hipvoc_last_error", "hipvoc_free",
]


class HipVocError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hipvoc error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # HIPVOC_LIBRARY: another build of the same library (e.g. the timing-ablation build, make ABLATE=1)
    return os.environ.get("HIPVOC_LIBRARY") or os.path.join(_HERE, "libhipvoc.so")


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise HipVocError(-6, f"{p} not built (run __graft_entry__.build() / make -C qwen3-tts_amd/csrc)")
    L = C.CDLL(p)
    vp, i64p, sz = C.c_void_p, C.c_void_p, C.c_size_t
    L.hipvoc_device_count.restype = C.c_int
    L.hipvoc_find_vocoder_model.argtypes = [C.c_char_p, C.c_char_p, sz]
    L.hipvoc_load.restype = vp
    L.hipvoc_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.hipvoc_max_tokens.argtypes = [vp]
    L.hipvoc_n_codebooks.argtypes = [vp]
    L.hipvoc_vocode_pcm16_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    L.hipvoc_unit_samples.restype = C.c_int64
    L.hipvoc_unit_samples.argtypes = [vp]
    L.hipvoc_run.argtypes = [vp, i64p, C.c_int, vp, sz, C.POINTER(sz)]
    L.hipvoc_decode.argtypes = [vp, i64p, C.c_int, vp, sz, C.POINTER(sz)]
    L.hipvoc_decode_batch.argtypes = [vp, i64p, vp, C.c_int, vp, vp, vp]
    L.hipvoc_vocode_pcm16.argtypes = [vp, vp, C.c_int, vp, sz, C.POINTER(sz)]
    L.hipvoc_decode_units_device.argtypes = [vp, vp, C.c_int, vp, sz]
    L.hipvoc_decode_units_device_noalloc.argtypes = [vp, vp, C.c_int, vp, sz]
    L.hipvoc_sync.argtypes = [vp]
    L.hipvoc_stream.restype = vp
    L.hipvoc_stream.argtypes = [vp]
    L.hipvoc_group_units.argtypes = [vp]
    L.hipvoc_lanes.argtypes = [vp]
    L.hipvoc_unit_macs.restype = C.c_int64
    L.hipvoc_unit_macs.argtypes = [vp]
    L.hipvoc_num_stages.argtypes = [vp]
    L.hipvoc_debug_stage.argtypes = [vp, i64p, C.c_int, C.c_int, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.hipvoc_set_profile.argtypes = [vp, C.c_int]
    L.hipvoc_last_profile.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.hipvoc_last_error.restype = C.c_char_p
    L.hipvoc_last_error.argtypes = [vp]
    L.hipvoc_free.argtypes = [vp]
    _lib = L
    return L


DTYPE_F32, DTYPE_BF16, DTYPE_BF16X3 = 0, 1, 2      # include/hipvoc.h: HIPVOC_DTYPE_*


def find_vocoder_model(models_dir: str) -> str:
    """find_vocoder_model (src/worker/mod.rs:792-813) for the hip-vocoder backend: first existing safetensors candidate."""
    L = load_library()
    buf = C.create_string_buffer(4096)
    rc = L.hipvoc_find_vocoder_model(os.fsencode(models_dir), buf, len(buf))
    if rc != 0:
        raise HipVocError(rc, buf.value.decode() or "no vocoder model found")
    return os.fsdecode(buf.value)


class HipVocoder:
    """Vocoder: converts codec tokens -> audio waveform on one MI355X (vocoder.rs:14-18)."""

    def __init__(self, handle, lib):
        self._h = handle
        self._L = lib
        self.max_tokens = int(lib.hipvoc_max_tokens(handle))
        self.n_codebooks = int(lib.hipvoc_n_codebooks(handle))      # 16 for the shipped decoder (vocoder.rs:62)

    @classmethod
    def load(cls, model_path: str, device: int = 0, unit_frames: int = 64, dtype: int = 0) -> "HipVocoder":
        L = load_library()
        if os.path.isdir(model_path):                      # a models directory: the reference's discovery (mod.rs:792-813)
            model_path = find_vocoder_model(model_path)
        h = L.hipvoc_load(os.fsencode(model_path), device, dtype, unit_frames)
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

    # ---- reference-shaped API ------------------------------------------------------------
    def synthesize(self, codes, n_tokens: Optional[int] = None) -> np.ndarray:
        """codes [n_tokens,16] i64 (or flat) -> f32 samples (Vocoder::synthesize, vocoder.rs:60-85)."""
        codes = np.ascontiguousarray(codes, dtype=np.int64).reshape(-1)
        if n_tokens is None:
            n_tokens = codes.size // self.n_codebooks
        # the reference panics on the slice codes[pos*16..(pos+chunk_len)*16] (vocoder.rs:69-72); here it is an error
        if n_tokens < 0 or codes.size < n_tokens * self.n_codebooks:
            raise HipVocError(-1, f"codes holds {codes.size} values, n_tokens*{self.n_codebooks} = {n_tokens * self.n_codebooks} needed")
        out = np.empty(max(n_tokens, 1) * SAMPLES_PER_TOKEN, dtype=np.float32)
        n = C.c_size_t(0)
        self._check(self._L.hipvoc_decode(self._h, codes.ctypes.data, n_tokens, out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value]

    def run(self, codes) -> np.ndarray:
        """One padded unit [max_tokens,16] -> all unit samples (OnnxVocoder::run, vocoder.rs:164-172)."""
        codes = np.ascontiguousarray(codes, dtype=np.int64).reshape(-1)
        if codes.size < self.max_tokens * self.n_codebooks:
            raise HipVocError(-1, f"run() needs max_tokens*{self.n_codebooks} = {self.max_tokens * self.n_codebooks} codes, got {codes.size}")
        out = np.empty(self.unit_samples, dtype=np.float32)
        n = C.c_size_t(0)
        self._check(self._L.hipvoc_run(self._h, codes.ctypes.data, self.max_tokens, out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value]

    def handle_vocode(self, codes_bytes: bytes, n_tokens: int) -> Tuple[bytes, int]:
        """i64-LE code bytes -> (i16-LE PCM bytes, n_samples)  (Worker::handle_vocode, mod.rs:507-531)."""
        buf = np.frombuffer(codes_bytes, dtype=np.uint8)
        if n_tokens < 0 or buf.size < n_tokens * self.n_codebooks * 8:
            raise HipVocError(-1, f"codes buffer shorter than n_tokens*{self.n_codebooks}*8 bytes")
        pcm = np.empty(max(n_tokens, 1) * SAMPLES_PER_TOKEN, dtype="<i2")
        n = C.c_size_t(0)
        self._check(self._L.hipvoc_vocode_pcm16(self._h, buf.ctypes.data, n_tokens, pcm.ctypes.data, pcm.size, C.byref(n)))
        return pcm[: n.value].tobytes(), int(n.value)

    def handle_vocode_batch(self, codes_bytes: Sequence[bytes], n_tokens: Sequence[int]) -> List[Tuple[bytes, int]]:
        """The VocodeBatch request: per item what handle_vocode returns, from one batched decode with the
        f32 -> i16 conversion on the GPU (hipvoc_vocode_pcm16_batch)."""
        B = len(n_tokens)
        if len(codes_bytes) != B:
            raise HipVocError(-1, "codes and n_tokens differ in length")
        nq8 = self.n_codebooks * 8
        for cb, n in zip(codes_bytes, n_tokens):
            if n < 0 or len(cb) < n * nq8:
                raise HipVocError(-1, f"codes buffer shorter than n_tokens*{self.n_codebooks}*8 bytes")
        nf = np.array(list(n_tokens), dtype=np.int32)
        blob = np.frombuffer(b"".join(bytes(cb[: n * nq8]) for cb, n in zip(codes_bytes, n_tokens)) or b"\0", dtype=np.uint8)
        caps = nf.astype(np.int64) * SAMPLES_PER_TOKEN
        off = np.zeros(max(B, 1), dtype=np.uint64)
        if B:
            off[1:B] = np.cumsum(caps)[:-1]
        pcm = np.empty(max(int(caps.sum()), 1), dtype="<i2")
        n_out = np.zeros(max(B, 1), dtype=np.uint64)
        self._check(self._L.hipvoc_vocode_pcm16_batch(self._h, blob.ctypes.data, nf.ctypes.data, B, pcm.ctypes.data,
                                                      off.ctypes.data, n_out.ctypes.data))
        return [(pcm[int(off[b]): int(off[b]) + int(n_out[b])].tobytes(), int(n_out[b])) for b in range(B)]

    def synthesize_batch(self, seqs: Sequence[np.ndarray]) -> List[np.ndarray]:
        """Many utterances in one call (hipvoc_decode_batch)."""
        arrs = [np.ascontiguousarray(s, dtype=np.int64).reshape(-1, self.n_codebooks) for s in seqs]
        B = len(arrs)
        nf = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        codes = np.concatenate(arrs, axis=0) if B else np.zeros((0, self.n_codebooks), dtype=np.int64)
        caps = nf.astype(np.int64) * SAMPLES_PER_TOKEN
        off = np.zeros(B, dtype=np.uint64)
        if B:
            off[1:] = np.cumsum(caps)[:-1]
        out = np.empty(max(int(caps.sum()), 1), dtype=np.float32)
        n_out = np.zeros(max(B, 1), dtype=np.uint64)
        self._check(self._L.hipvoc_decode_batch(self._h, codes.ctypes.data, nf.ctypes.data, B, out.ctypes.data,
                                                off.ctypes.data, n_out.ctypes.data))
        return [out[int(off[b]): int(off[b]) + int(n_out[b])].copy() for b in range(B)]

    # ---- device-resident / diagnostics ---------------------------------------------------
    @property
    def unit_samples(self) -> int:
        return int(self._L.hipvoc_unit_samples(self._h))

    @property
    def group_units(self) -> int:
        return int(self._L.hipvoc_group_units(self._h))

    @property
    def lanes(self) -> int:
        """Passes of one call in flight side by side (hipvoc_lanes; 2 in the bf16 mode unless HIPVOC_LANES says otherwise)."""
        return int(self._L.hipvoc_lanes(self._h))

    @property
    def unit_macs(self) -> int:
        return int(self._L.hipvoc_unit_macs(self._h))

    @property
    def stream(self) -> int:
        return int(self._L.hipvoc_stream(self._h) or 0)

    @property
    def n_stages(self) -> int:
        return int(self._L.hipvoc_num_stages(self._h))

    def decode_units_device(self, d_codes_ptr: int, n_units: int, d_out_ptr: int, unit_stride: int):
        self._check(self._L.hipvoc_decode_units_device(self._h, d_codes_ptr, n_units, d_out_ptr, unit_stride))

    def decode_units_device_noalloc(self, d_codes_ptr: int, n_units: int, d_out_ptr: int, unit_stride: int):
        """The entry point that allocates nothing: workspace in the library's static device array (include/hipvoc.h)."""
        self._check(self._L.hipvoc_decode_units_device_noalloc(self._h, d_codes_ptr, n_units, d_out_ptr, unit_stride))

    def sync(self):
        self._check(self._L.hipvoc_sync(self._h))

    def stage(self, codes, stage: int, cap_floats: int) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64).reshape(-1, self.n_codebooks)
        out = np.empty(cap_floats, dtype=np.float32)
        c, l = C.c_int(0), C.c_int64(0)
        self._check(self._L.hipvoc_debug_stage(self._h, codes.ctypes.data, codes.shape[0], stage, out.ctypes.data,
                                               out.size, C.byref(c), C.byref(l)))
        return out[: c.value * l.value].reshape(c.value, l.value).copy()

    def set_profile(self, on: bool):
        self._check(self._L.hipvoc_set_profile(self._h, 1 if on else 0))

    def last_profile(self) -> Dict[str, Tuple[float, float, int]]:
        """{"<step>|<kernel symbol>": (ms, algorithmic flops, launches)} of the last pass."""
        cap = 128
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        ln = (C.c_int * cap)()
        n = self._L.hipvoc_last_profile(self._h, names, ms, fl, ln, cap)
        return {names[i].decode(): (float(ms[i]), float(fl[i]), int(ln[i])) for i in range(min(n, cap))}
