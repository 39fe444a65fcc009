"""Voice-profile files as code sources for the vocoder path (host-side mirror of the reference loader).

Follows /root/reference/src/voice_loader.rs:7-60: a voice file carries the reference speaker's codec tokens as
[n_tokens, 16]; the loader returns them flattened row-major (frame-major) plus the optional reference text.
  .json  {"ref_text": str?, "codec_tokens": [[16 ints] * n]}            (voice_loader.rs:10-17)
  .npy   int64 2-D array; a [16, N] array with N > 16 is transposed     (voice_loader.rs:18-29)
  .pt    refused: the reference reads it through a pickle reader (voice_loader.rs:30-57); this package loads
         nothing that needs unpickling.
These are the only realistic (non-uniform) code inputs a deployment has without running the talker, so the wire /
worker tests use them as request payloads.
"""
from __future__ import annotations

import json
import os
from typing import Optional, Tuple

import numpy as np

N_Q = 16


def load_voice_file(path: str) -> Tuple[np.ndarray, Optional[str]]:
    """-> (codes int64 [n, 16], ref_text or None).  Raises ValueError on malformed content, as the reference bails."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".json":
        with open(path, "r", encoding="utf-8") as f:
            data = json.load(f)
        if not isinstance(data, dict) or "codec_tokens" not in data:
            raise ValueError("Invalid codec_tokens in voice JSON")
        ref_text = data.get("ref_text")
        if not isinstance(ref_text, str):
            ref_text = None
        rows = data["codec_tokens"]
        if not isinstance(rows, list) or any(not isinstance(r, list) for r in rows):
            raise ValueError("Invalid codec_tokens in voice JSON")
        flat = [v for r in rows for v in r]           # `.into_iter().flatten()`: row lengths are not checked there
        if any(isinstance(v, bool) or not isinstance(v, int) for v in flat):
            raise ValueError("Invalid codec_tokens in voice JSON")
        if len(flat) % N_Q:
            raise ValueError("voice JSON: token count is not a multiple of 16")
        return np.asarray(flat, dtype=np.int64).reshape(-1, N_Q), ref_text
    if ext == ".npy":
        arr = np.load(path, allow_pickle=False)
        if arr.ndim != 2 or arr.dtype != np.int64:
            raise ValueError("voice .npy must be a 2-D int64 array")
        if arr.shape[0] == N_Q and arr.shape[1] > N_Q:   # saved as [16, N]
            arr = arr.T
        flat = np.ascontiguousarray(arr).reshape(-1)
        if flat.size % N_Q:
            raise ValueError("voice .npy: token count is not a multiple of 16")
        return flat.reshape(-1, N_Q), None
    if ext == ".pt":
        raise ValueError("voice .pt files need an unpickler and are not loaded here; convert to .json or .npy")
    raise ValueError(f"Unsupported voice format: {path}")


def save_voice_json(path: str, codes: np.ndarray, ref_text: Optional[str] = None) -> None:
    codes = np.asarray(codes, dtype=np.int64).reshape(-1, N_Q)
    obj = {"codec_tokens": codes.tolist()}
    if ref_text is not None:
        obj["ref_text"] = ref_text
    with open(path, "w", encoding="utf-8") as f:
        json.dump(obj, f)
