"""ctypes front-end of oracle/libenc_oracle.so (CPU restatement of the speech-tokenizer encoder, oracle/enc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
(qwen3-tts_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libenc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "enc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libenc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
        L.enc_oracle_new.restype = vp
        L.enc_oracle_free.argtypes = [vp]
        L.enc_oracle_error.restype = C.c_char_p
        L.enc_oracle_error.argtypes = [vp]
        L.enc_oracle_set_param.argtypes = [vp, C.c_char_p, C.c_double]
        L.enc_oracle_set_tensor.argtypes = [vp, C.c_char_p, vp, i64p, C.c_int]
        L.enc_oracle_finalize.argtypes = [vp]
        L.enc_oracle_n_stages.argtypes = [vp]
        L.enc_oracle_use_q.argtypes = [vp]
        L.enc_oracle_dim.argtypes = [vp]
        L.enc_oracle_cb_size.argtypes = [vp]
        L.enc_oracle_cb_dim.argtypes = [vp]
        L.enc_oracle_n_frames.restype = C.c_int64
        L.enc_oracle_n_frames.argtypes = [vp, C.c_int64]
        L.enc_oracle_encode.argtypes = [vp, vp, C.c_int64, vp, C.c_size_t, i64p]
        L.enc_oracle_quantize.argtypes = [vp, vp, C.c_int64, vp]
        L.enc_oracle_distances.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.enc_oracle_embed.restype = C.POINTER(C.c_float)
        L.enc_oracle_embed.argtypes = [vp, C.c_int, C.c_int]
        L.enc_oracle_stage.argtypes = [vp, vp, C.c_int64, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int), i64p]
        L.enc_oracle_resample.restype = C.c_int64
        L.enc_oracle_resample.argtypes = [vp, C.c_int64, C.c_uint32, C.c_uint32, vp, C.c_int64]
        _lib = L
    return _lib


class EncOracle:
    """CPU encoder built from a dict name -> float32 ndarray (the names speech_tokenizer.rs loads)."""

    def __init__(self, weights: Dict[str, np.ndarray], params: Optional[Dict[str, float]] = None):
        self._L = lib()
        self._h = self._L.enc_oracle_new()
        self._keep = []
        for k, v in (params or {}).items():
            if self._L.enc_oracle_set_param(self._h, k.encode(), float(v)):
                raise RuntimeError(f"unknown oracle parameter {k}")
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            self._keep.append(a)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            if self._L.enc_oracle_set_tensor(self._h, name.encode(), a.ctypes.data, shp, a.ndim):
                raise RuntimeError(f"set_tensor({name}) failed")
        if self._L.enc_oracle_finalize(self._h):
            raise RuntimeError("encoder oracle finalize: " + self._err())

    def _err(self) -> str:
        return self._L.enc_oracle_error(self._h).decode()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.enc_oracle_free(self._h)
            self._h = None

    @property
    def n_stages(self) -> int:
        return int(self._L.enc_oracle_n_stages(self._h))

    @property
    def use_q(self) -> int:
        return int(self._L.enc_oracle_use_q(self._h))

    @property
    def dim(self) -> int:
        return int(self._L.enc_oracle_dim(self._h))

    def n_frames(self, n_samples: int) -> int:
        return int(self._L.enc_oracle_n_frames(self._h, n_samples))

    def encode(self, audio: np.ndarray) -> np.ndarray:
        audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        T = max(self.n_frames(audio.size), 0)
        out = np.empty((max(T, 1), self.use_q), dtype=np.int64)
        n = C.c_int64(0)
        if self._L.enc_oracle_encode(self._h, audio.ctypes.data, audio.size, out.ctypes.data, out.size, C.byref(n)):
            raise RuntimeError("encoder oracle encode: " + self._err())
        return out[: n.value]

    def quantize(self, latent: np.ndarray) -> np.ndarray:
        """latent [dim, T] -> codes [T, use_q]: the integer part alone."""
        h = np.ascontiguousarray(latent, dtype=np.float32)
        assert h.ndim == 2 and h.shape[0] == self.dim
        out = np.empty((h.shape[1], self.use_q), dtype=np.int64)
        if self._L.enc_oracle_quantize(self._h, h.ctypes.data, h.shape[1], out.ctypes.data):
            raise RuntimeError("encoder oracle quantize: " + self._err())
        return out

    def distances(self, group: int, layer: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(int(self._L.enc_oracle_cb_size(self._h)), dtype=np.float32)
        if self._L.enc_oracle_distances(self._h, group, layer, x.ctypes.data, out.ctypes.data):
            raise RuntimeError("no such codebook")
        return out

    def embed(self, group: int, layer: int) -> np.ndarray:
        p = self._L.enc_oracle_embed(self._h, group, layer)
        if not p:
            raise RuntimeError("no such codebook")
        n, d = int(self._L.enc_oracle_cb_size(self._h)), int(self._L.enc_oracle_cb_dim(self._h))
        return np.ctypeslib.as_array(p, shape=(n, d)).copy()

    def stage(self, audio: np.ndarray, stage: int) -> np.ndarray:
        audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        cap = max(audio.size * 64, 1 << 16)
        out = np.empty(cap, dtype=np.float32)
        c, l = C.c_int(0), C.c_int64(0)
        if self._L.enc_oracle_stage(self._h, audio.ctypes.data, audio.size, stage, out.ctypes.data, out.size,
                                    C.byref(c), C.byref(l)):
            raise RuntimeError("encoder oracle stage: " + self._err())
        return out[: c.value * l.value].reshape(c.value, l.value).copy()


def set_threads(n: int) -> int:
    return int(lib().enc_oracle_set_threads(int(n)))


def resample(x: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = int(lib().enc_oracle_resample(x.ctypes.data, x.size, from_rate, to_rate, None, 0))
    out = np.empty(max(n, 0), dtype=np.float32)
    lib().enc_oracle_resample(x.ctypes.data, x.size, from_rate, to_rate, out.ctypes.data, out.size)
    return out
