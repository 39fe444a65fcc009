#!/usr/bin/env python3
"""Generates the golden fixtures in this directory (run in the CPU container only).

Source of truth: tf_structural.py (the published transformers building blocks, assembled as the
Qwen3-TTS 12 Hz tokenizer decoder, seeded synthetic weights).  The fixtures hold DATA only: codes,
expected waveforms and strided probes of every intermediate stage.  Weights are not stored: they are
regenerated from qwen3-tts_amd/synth.py (config name + seed).

    python tests/golden/make_golden.py

The reference itself has no fixture for this path, so these vectors pin the
oracle to the published layer definitions, not to the shipped vocoder.onnx ("parity unpinned" there).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "qwen3-tts_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
sys.modules["synth"] = synth
spec.loader.exec_module(synth)
import tf_structural as tfs  # noqa: E402

PROBES = 1536


def probe_idx(n):
    if n <= PROBES:
        return np.arange(n)
    return np.unique(np.linspace(0, n - 1, PROBES).astype(np.int64))


def unit_fixture(name, cfg_name, T, wseed, cseed):
    cfg = getattr(synth, cfg_name)
    W = synth.make_weights(cfg, wseed)
    codes = synth.make_codes(T, cseed, cfg)
    if name.endswith("edge"):
        codes[0, :] = 0
        codes[-1, :] = cfg.codebook_size - 1
        codes[1, 0] = 3            # cluster_usage == 0 row (clamp path)
    model = tfs.build(cfg, W)
    import torch
    with torch.no_grad():
        stages = model(torch.from_numpy(codes))
    out = {"cfg": np.array(cfg_name), "wseed": np.array(wseed), "codes": codes, "wav": stages[-1][0],
           "n_stages": np.array(len(stages))}
    for i, s in enumerate(stages):
        flat = s.reshape(-1)
        idx = probe_idx(flat.size)
        out[f"s{i}_shape"] = np.array(s.shape)
        out[f"s{i}_idx"] = idx
        out[f"s{i}_val"] = flat[idx]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "stages", len(stages), "wav", stages[-1].shape)
    return cfg, W, model


def chunk_fixture(name, cfg_name, n_tokens, max_tokens, wseed, cseed):
    """Vocoder::synthesize restated over the TF model (vocoder.rs:60-85)."""
    import torch
    cfg = getattr(synth, cfg_name)
    W = synth.make_weights(cfg, wseed)
    model = tfs.build(cfg, W)
    codes = synth.make_codes(n_tokens, cseed, cfg)
    chunks = []
    for start in range(0, n_tokens, max_tokens):
        end = min(start + max_tokens, n_tokens)
        padded = np.zeros((max_tokens, cfg.num_quantizers), dtype=np.int64)      # pad = code index 0
        padded[: end - start] = codes[start:end]
        with torch.no_grad():
            wav = model(torch.from_numpy(padded))[-1][0]
        keep = min((end - start) * 1920, wav.size)
        chunks.append(wav[:keep])
    wav = np.concatenate(chunks)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=np.array(cfg_name), wseed=np.array(wseed),
                        codes=codes, max_tokens=np.array(max_tokens), wav=wav)
    print(name, "n_tokens", n_tokens, "max_tokens", max_tokens, "samples", wav.size)


def pcm_fixture():
    """(s*32767.0).clamp(-32768.0, 32767.0) as i16 (mod.rs:521-524): truncation toward zero."""
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 1.00003, -1.00003, -1.0001, 3.0, -3.0, 1e-5, -1e-5,
                  0.999984, -0.999984, 3.0518e-5, -3.0518e-5, 6.1e-5, 0.25, -0.75, 0.1], dtype=np.float32)
    rng = np.random.default_rng(7)
    x = np.concatenate([x, rng.uniform(-1.1, 1.1, 2000).astype(np.float32)])
    v = x.astype(np.float32) * np.float32(32767.0)
    v = np.clip(v, np.float32(-32768.0), np.float32(32767.0))
    y = np.trunc(v).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "pcm16.npz"), x=x, y=y)
    print("pcm16", x.size)


if __name__ == "__main__":
    unit_fixture("small_T5", "SMALL", 5, 0, 1000)
    unit_fixture("small_T3_edge", "SMALL", 3, 1, 1001)
    unit_fixture("odd_T9", "ODD", 9, 0, 1002)           # sliding window 5 < T: the mask binds
    unit_fixture("full_T3", "FULL", 3, 0, 1003)
    chunk_fixture("small_chunks_10x4", "SMALL", 10, 4, 0, 1004)   # 2 full units + 2-frame tail
    chunk_fixture("small_chunks_8x4", "SMALL", 8, 4, 0, 1005)     # exact multiple
    pcm_fixture()
