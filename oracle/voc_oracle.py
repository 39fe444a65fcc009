"""ctypes front-end of oracle/libvoc_oracle.so (the CPU restatement, oracle/voc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product (qwen3-tts_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvoc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "voc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libvoc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.voc_oracle_new.restype = C.c_void_p
        L.voc_oracle_free.argtypes = [C.c_void_p]
        L.voc_oracle_error.restype = C.c_char_p
        L.voc_oracle_error.argtypes = [C.c_void_p]
        L.voc_oracle_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.voc_oracle_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.voc_oracle_finalize.argtypes = [C.c_void_p]
        L.voc_oracle_out_len.restype = C.c_int64
        L.voc_oracle_out_len.argtypes = [C.c_void_p, C.c_int]
        L.voc_oracle_n_stages.argtypes = [C.c_void_p]
        L.voc_oracle_run_unit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.voc_oracle_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.voc_oracle_stage_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.voc_oracle_synthesize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t)]
        L.voc_oracle_transconv.restype = C.c_int64
        L.voc_oracle_transconv.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]
        L.voc_oracle_pcm16.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.voc_oracle_codes_from_le.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


class Oracle:
    """CPU vocoder built from a dict name -> float32 ndarray (upstream ``decoder.*`` names)."""

    def __init__(self, weights: Dict[str, np.ndarray], params: Optional[Dict[str, float]] = None):
        self._L = lib()
        self._h = self._L.voc_oracle_new()
        self._keep = []
        for k, v in (params or {}).items():
            self._L.voc_oracle_set_param(self._h, k.encode(), float(v))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            self._keep.append(a)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self._L.voc_oracle_set_tensor(self._h, name.encode(), a.ctypes.data, shp, a.ndim)
            if rc:
                raise RuntimeError(f"set_tensor({name}) failed")
        if self._L.voc_oracle_finalize(self._h):
            raise RuntimeError("oracle finalize: " + self._err())

    def _err(self) -> str:
        return self._L.voc_oracle_error(self._h).decode()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.voc_oracle_free(self._h)
            self._h = None

    def out_len(self, frames: int) -> int:
        return int(self._L.voc_oracle_out_len(self._h, frames))

    @property
    def n_stages(self) -> int:
        return int(self._L.voc_oracle_n_stages(self._h))

    def run_unit(self, codes: np.ndarray) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        T = codes.shape[0]
        out = np.empty(self.out_len(T), dtype=np.float32)
        n = C.c_size_t(0)
        rc = self._L.voc_oracle_run_unit(self._h, codes.ctypes.data, T, out.ctypes.data, out.size, C.byref(n))
        if rc:
            raise RuntimeError("oracle run_unit: " + self._err())
        return out[: n.value]

    def stage(self, codes: np.ndarray, stage: int) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        T = codes.shape[0]
        c, l = C.c_int(0), C.c_int64(0)
        if self._L.voc_oracle_stage_shape(self._h, T, stage, C.byref(c), C.byref(l)):
            raise RuntimeError(f"no such stage {stage}")
        out = np.empty(c.value * l.value, dtype=np.float32)
        rc = self._L.voc_oracle_stage(self._h, codes.ctypes.data, T, stage, out.ctypes.data, out.size,
                                      C.byref(c), C.byref(l))
        if rc:
            raise RuntimeError("oracle stage: " + self._err())
        return out[: c.value * l.value].reshape(c.value, l.value).copy()

    def synthesize(self, codes: np.ndarray, max_tokens: int = 64) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        n_tokens = codes.shape[0]
        out = np.empty(max(n_tokens, 1) * 1920, dtype=np.float32)
        n = C.c_size_t(0)
        rc = self._L.voc_oracle_synthesize(self._h, codes.ctypes.data, n_tokens, max_tokens, out.ctypes.data,
                                           out.size, C.byref(n))
        if rc:
            raise RuntimeError("oracle synthesize: " + self._err())
        return out[: n.value]


def transconv(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int, right_only: bool) -> np.ndarray:
    """x [Cin, L], w [Cin, Cout, K] (ConvTranspose1d layout), b [Cout] -> y [Cout, Lout] under the given trim policy."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    cin, L = x.shape
    _, cout, K = w.shape
    bp = None if bb is None else bb.ctypes.data
    Lo = int(lib().voc_oracle_transconv(x.ctypes.data, cin, L, w.ctypes.data, bp, cout, K, stride, int(right_only), None))
    y = np.empty((cout, Lo), dtype=np.float32)
    lib().voc_oracle_transconv(x.ctypes.data, cin, L, w.ctypes.data, bp, cout, K, stride, int(right_only), y.ctypes.data)
    return y


def set_threads(n: int) -> int:
    return int(lib().voc_oracle_set_threads(int(n)))


def pcm16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.size, dtype=np.int16)
    lib().voc_oracle_pcm16(x.ctypes.data, x.size, out.ctypes.data)
    return out


def codes_from_le(b: bytes) -> np.ndarray:
    n = len(b) // 8
    out = np.empty(n, dtype=np.int64)
    buf = (C.c_char * len(b)).from_buffer_copy(b)
    lib().voc_oracle_codes_from_le(buf, n, out.ctypes.data)
    return out
