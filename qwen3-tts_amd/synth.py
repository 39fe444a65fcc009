"""Seeded synthetic weights and code sequences for the vocoder path.

No real checkpoint exists offline, so tests and bench.py run on random-init weights
of the Qwen3-TTS 12 Hz tokenizer-decoder architecture.  Tensor names mirror the upstream
``speech_tokenizer.decoder`` state-dict (what scripts/export_vocoder.py:25-27 wraps), so a real
``model.safetensors`` holding ``decoder.*`` tensors loads through the same path.

Unlike a fresh nn.Module init (alpha=beta=0, gamma=1e-6, layer-scale 0.01: SURVEY.md 8d) every
parameter here is drawn so that it matters numerically (random SnakeBeta alpha/beta, O(0.1) layer
scales, a few cluster_usage entries under the 1e-5 clamp), while gains keep the waveform inside
(-1, 1) so the final clamp does not hide errors.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, asdict
from typing import Dict, Tuple

import numpy as np


@dataclass(frozen=True)
class DecoderConfig:
    codebook_size: int = 2048
    codebook_dim: int = 512          # RVQ output dim; codebook rows are codebook_dim // 2 wide
    num_quantizers: int = 16
    latent_dim: int = 1024
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 16
    head_dim: int = 64
    intermediate_size: int = 1024
    sliding_window: int = 72
    rope_theta: float = 10000.0
    rms_norm_eps: float = 1e-5
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    decoder_dim: int = 1536

    def out_len(self, frames: int) -> int:
        n = frames
        for r in self.upsampling_ratios:
            n *= r
        for r in self.upsample_rates:
            n = (n - 1) * r
        return n

    def metadata(self) -> Dict[str, str]:
        return {
            "num_attention_heads": str(self.num_attention_heads),
            "sliding_window": str(self.sliding_window),
            "rope_theta": repr(float(self.rope_theta)),
            "rms_norm_eps": repr(float(self.rms_norm_eps)),
        }


FULL = DecoderConfig()
# Same topology (so 1 frame is still 1920 samples, protocol.rs:122), narrow channels: CPU-test size.
SMALL = DecoderConfig(codebook_size=64, codebook_dim=64, latent_dim=64, hidden_size=64, num_hidden_layers=2,
                      num_attention_heads=2, head_dim=32, intermediate_size=96, decoder_dim=512)
# Odd widths: channel counts that are not multiples of the kernel tiles (exercises the masks).
ODD = DecoderConfig(codebook_size=50, codebook_dim=40, latent_dim=72, hidden_size=48, num_hidden_layers=1,
                    num_attention_heads=3, head_dim=16, intermediate_size=80, decoder_dim=208,
                    sliding_window=5)


def make_weights(cfg: DecoderConfig = FULL, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    W: Dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def uniform(shape, lo, hi):
        return rng.uniform(lo, hi, size=shape).astype(np.float32)

    half = cfg.codebook_dim // 2
    nrest = cfg.num_quantizers - 1

    def codebook(prefix, std):
        usage = uniform((cfg.codebook_size,), 0.5, 50.0)
        usage[:: max(7, cfg.codebook_size // 9)] = 1e-7          # below the 1e-5 clamp
        usage[3] = 0.0
        emb = normal((cfg.codebook_size, half), std)
        W[prefix + "._codebook.cluster_usage"] = usage
        W[prefix + "._codebook.embedding_sum"] = (emb * np.maximum(usage, 1e-5)[:, None]).astype(np.float32)

    codebook("decoder.quantizer.rvq_first.vq.layers.0", 1.0)
    for i in range(nrest):
        codebook(f"decoder.quantizer.rvq_rest.vq.layers.{i}", 1.0 / np.sqrt(max(nrest, 1)))
    for g in ("rvq_first", "rvq_rest"):
        W[f"decoder.quantizer.{g}.input_proj.weight"] = normal((half, cfg.codebook_dim, 1), cfg.codebook_dim ** -0.5)
        W[f"decoder.quantizer.{g}.output_proj.weight"] = normal((cfg.codebook_dim, half, 1), (2 * half) ** -0.5)

    def conv(prefix, cout, cin, k, gain=1.0):
        W[prefix + ".weight"] = normal((cout, cin, k), gain * (cin * k) ** -0.5)
        W[prefix + ".bias"] = normal((cout,), 0.02)

    def linear(prefix, cout, cin, bias=True, gain=1.0):
        W[prefix + ".weight"] = normal((cout, cin), gain * cin ** -0.5)
        if bias:
            W[prefix + ".bias"] = normal((cout,), 0.02)

    conv("decoder.pre_conv.conv", cfg.latent_dim, cfg.codebook_dim, 3)
    pt = "decoder.pre_transformer"
    linear(pt + ".input_proj", cfg.hidden_size, cfg.latent_dim)
    qkv = cfg.num_attention_heads * cfg.head_dim
    for i in range(cfg.num_hidden_layers):
        p = f"{pt}.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj"):
            linear(f"{p}.self_attn.{n}", qkv, cfg.hidden_size, bias=False, gain=1.5)
        linear(f"{p}.self_attn.o_proj", cfg.hidden_size, qkv, bias=False)
        linear(f"{p}.mlp.gate_proj", cfg.intermediate_size, cfg.hidden_size, bias=False)
        linear(f"{p}.mlp.up_proj", cfg.intermediate_size, cfg.hidden_size, bias=False)
        linear(f"{p}.mlp.down_proj", cfg.hidden_size, cfg.intermediate_size, bias=False)
        W[f"{p}.input_layernorm.weight"] = (1.0 + normal((cfg.hidden_size,), 0.1)).astype(np.float32)
        W[f"{p}.post_attention_layernorm.weight"] = (1.0 + normal((cfg.hidden_size,), 0.1)).astype(np.float32)
        W[f"{p}.self_attn_layer_scale.scale"] = uniform((cfg.hidden_size,), 0.05, 0.4)
        W[f"{p}.mlp_layer_scale.scale"] = uniform((cfg.hidden_size,), 0.05, 0.4)
    W[pt + ".norm.weight"] = (1.0 + normal((cfg.hidden_size,), 0.1)).astype(np.float32)
    linear(pt + ".output_proj", cfg.latent_dim, cfg.hidden_size)

    C = cfg.latent_dim
    for i, r in enumerate(cfg.upsampling_ratios):
        p = f"decoder.upsample.{i}"
        W[p + ".0.conv.weight"] = normal((C, C, r), C ** -0.5)
        W[p + ".0.conv.bias"] = normal((C,), 0.02)
        W[p + ".1.dwconv.conv.weight"] = normal((C, 1, 7), 7 ** -0.5)
        W[p + ".1.dwconv.conv.bias"] = normal((C,), 0.02)
        W[p + ".1.norm.weight"] = (1.0 + normal((C,), 0.1)).astype(np.float32)
        W[p + ".1.norm.bias"] = normal((C,), 0.05)
        linear(p + ".1.pwconv1", 4 * C, C)
        linear(p + ".1.pwconv2", C, 4 * C)
        W[p + ".1.gamma"] = uniform((C,), 0.05, 0.4)

    conv("decoder.decoder.0.conv", cfg.decoder_dim, C, 7)
    cin = cfg.decoder_dim
    for b, r in enumerate(cfg.upsample_rates):
        p = f"decoder.decoder.{b + 1}.block"
        cout = cin // 2
        W[p + ".0.alpha"] = normal((cin,), 0.3)
        W[p + ".0.beta"] = normal((cin,), 0.3)
        W[p + ".1.conv.weight"] = normal((cin, cout, 2 * r), 0.7 * (2 * cin) ** -0.5)
        W[p + ".1.conv.bias"] = normal((cout,), 0.02)
        for u in range(3):
            q = f"{p}.{u + 2}"
            W[q + ".act1.alpha"] = normal((cout,), 0.3)
            W[q + ".act1.beta"] = normal((cout,), 0.3)
            conv(q + ".conv1.conv", cout, cout, 7, gain=0.8)
            W[q + ".act2.alpha"] = normal((cout,), 0.3)
            W[q + ".act2.beta"] = normal((cout,), 0.3)
            conv(q + ".conv2.conv", cout, cout, 1, gain=0.35)
        cin = cout
    nb = len(cfg.upsample_rates)
    W[f"decoder.decoder.{nb + 1}.alpha"] = normal((cin,), 0.3)
    W[f"decoder.decoder.{nb + 1}.beta"] = normal((cin,), 0.3)
    conv(f"decoder.decoder.{nb + 2}.conv", 1, cin, 7, gain=0.12)
    return W


def make_codes(n_frames: int, seed: int, cfg: DecoderConfig = FULL) -> np.ndarray:
    """Synthetic code sequence [n_frames, n_q] int64 (distribution of scripts/export_vocoder.py:34)."""
    return np.random.default_rng(seed).integers(0, cfg.codebook_size, size=(n_frames, cfg.num_quantizers),
                                                dtype=np.int64)


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round to nearest even."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def save_safetensors(path: str, tensors: Dict[str, np.ndarray], metadata: Dict[str, str] | None = None,
                     dtype: str = "F32") -> None:
    """Minimal safetensors writer (8-byte LE header length, JSON header, raw LE data).  dtype: the storage
    type of every tensor, "F32", "F16" or "BF16" (what upstream checkpoints come in)."""
    enc = {"F32": lambda a: np.ascontiguousarray(a, dtype="<f4").tobytes(),
           "F16": lambda a: np.ascontiguousarray(a, dtype=np.float32).astype("<f2").tobytes(),
           "BF16": lambda a: to_bf16_bits(a).astype("<u2").tobytes()}[dtype]
    item = {"F32": 4, "F16": 2, "BF16": 2}[dtype]
    header = {}
    if metadata:
        header["__metadata__"] = dict(metadata)
    off = 0
    names = sorted(tensors)
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        header[n] = {"dtype": dtype, "shape": list(a.shape), "data_offsets": [off, off + a.size * item]}
        off += a.size * item
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for n in names:
            f.write(enc(tensors[n]))


def config_dict(cfg: DecoderConfig) -> dict:
    return asdict(cfg)
