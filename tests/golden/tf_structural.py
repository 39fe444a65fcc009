"""Structural oracle (container-only, test-only): the decoder assembled from the *published* building
blocks in the locally installed `transformers` package (NOT part of the reference):

  Qwen3OmniMoeCausalConvNet / CausalTransConvNet / ConvNeXtBlock / Code2WavTransformerModel /
  Code2WavDecoderBlock / SnakeBeta   transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3180-3684
  MimiSplitResidualVectorQuantizer   transformers/models/mimi/modeling_mimi.py:964-1137

assembled the way the Qwen3-TTS 12 Hz tokenizer decoder is (quantizer -> pre_conv -> input_proj ->
transformer -> output_proj -> upsample -> decoder -> clamp), and loaded with
the seeded synthetic weights of qwen3-tts_amd/synth.py.  Used only by make_golden.py to pin
oracle/voc_oracle.c and to emit the fixtures in this directory; never runs on the GPU box.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


def build(cfg, weights):
    from transformers.models.mimi.configuration_mimi import MimiConfig
    from transformers.models.mimi.modeling_mimi import MimiSplitResidualVectorQuantizer
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig

    half = cfg.codebook_dim // 2
    mcfg = MimiConfig(codebook_size=cfg.codebook_size, codebook_dim=half,
                      vector_quantization_hidden_dimension=half, hidden_size=cfg.codebook_dim,
                      num_quantizers=cfg.num_quantizers, num_semantic_quantizers=1)
    c2w = Qwen3OmniMoeCode2WavConfig(
        codebook_size=cfg.codebook_size, hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_attention_heads, sliding_window=cfg.sliding_window,
        intermediate_size=cfg.intermediate_size, rms_norm_eps=cfg.rms_norm_eps,
        num_hidden_layers=cfg.num_hidden_layers, num_quantizers=cfg.num_quantizers,
        upsample_rates=tuple(cfg.upsample_rates), upsampling_ratios=tuple(cfg.upsampling_ratios),
        decoder_dim=cfg.decoder_dim, rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta})
    object.__setattr__(c2w, "head_dim", cfg.head_dim)
    c2w._attn_implementation = "eager"

    class Decoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.quantizer = MimiSplitResidualVectorQuantizer(mcfg)
            self.pre_conv = M.Qwen3OmniMoeCausalConvNet(cfg.codebook_dim, cfg.latent_dim, 3)
            self.pre_transformer = M.Qwen3OmniMoeCode2WavTransformerModel._from_config(c2w)
            self.input_proj = nn.Linear(cfg.latent_dim, cfg.hidden_size)
            self.output_proj = nn.Linear(cfg.hidden_size, cfg.latent_dim)
            self.upsample = nn.ModuleList([
                nn.ModuleList([M.Qwen3OmniMoeCausalTransConvNet(cfg.latent_dim, cfg.latent_dim, f, f),
                               M.Qwen3OmniMoeConvNeXtBlock(cfg.latent_dim)]) for f in cfg.upsampling_ratios])
            dec = [M.Qwen3OmniMoeCausalConvNet(cfg.latent_dim, cfg.decoder_dim, 7)]
            for i in range(len(cfg.upsample_rates)):
                dec.append(M.Qwen3OmniMoeCode2WavDecoderBlock(c2w, i))
            out_dim = cfg.decoder_dim // 2 ** len(cfg.upsample_rates)
            dec += [M.Qwen3OmniMoeSnakeBeta(out_dim), M.Qwen3OmniMoeCausalConvNet(out_dim, 1, 7)]
            self.decoder = nn.ModuleList(dec)

        def forward(self, codes_tq):                     # [T, n_q] wire order (vocoder.rs:166)
            stages = []
            codes = codes_tq.t().unsqueeze(0)            # [1, n_q, T]
            h = self.quantizer.decode(codes)             # [1, 512, T]
            stages.append(h)
            h = self.pre_conv(h)
            stages.append(h)
            h = self.input_proj(h.transpose(1, 2))
            stages.append(h.transpose(1, 2))
            tm = self.pre_transformer
            pos = torch.arange(h.shape[1])[None]
            pe = tm.rotary_emb(h, pos)
            from transformers.masking_utils import create_sliding_window_causal_mask
            mask = create_sliding_window_causal_mask(config=c2w, inputs_embeds=h, attention_mask=None,
                                                     past_key_values=None, position_ids=pos)
            for layer in tm.layers:
                h = layer(h, attention_mask=mask, position_embeddings=pe, position_ids=pos)
                stages.append(h.transpose(1, 2))
            h = tm.norm(h)
            h = self.output_proj(h).permute(0, 2, 1)
            stages.append(h)
            for blocks in self.upsample:
                for blk in blocks:
                    h = blk(h)
                    stages.append(h)
            h = self.decoder[0](h)
            stages.append(h)
            for blk in self.decoder[1:-2]:
                for sub in blk.block:
                    h = sub(h)
                    if not isinstance(sub, M.Qwen3OmniMoeSnakeBeta):
                        stages.append(h)
            h = self.decoder[-2](h)
            h = self.decoder[-1](h)
            h = h.clamp(min=-1, max=1)
            stages.append(h)
            return [s[0].detach().numpy().astype(np.float32) for s in stages]

    m = Decoder().eval()
    sd = {}
    for k, v in weights.items():
        t = torch.from_numpy(np.asarray(v, dtype=np.float32))
        if not k.startswith("decoder."):
            continue
        k = k[len("decoder."):]
        if ".input_proj.weight" in k and k.startswith("quantizer."):
            continue
        k = k.replace("quantizer.rvq_first.vq.", "quantizer.semantic_residual_vector_quantizer.")
        k = k.replace("quantizer.rvq_rest.vq.", "quantizer.acoustic_residual_vector_quantizer.")
        k = k.replace("quantizer.rvq_first.", "quantizer.semantic_residual_vector_quantizer.")
        k = k.replace("quantizer.rvq_rest.", "quantizer.acoustic_residual_vector_quantizer.")
        k = k.replace("._codebook.embedding_sum", ".codebook.embed_sum").replace("._codebook.", ".codebook.")
        k = k.replace("pre_transformer.input_proj.", "input_proj.").replace("pre_transformer.output_proj.", "output_proj.")
        sd[k] = t
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "initialized" not in k and "input_proj" not in k.split("quantizer.")[-1][:60]
               or k.startswith(("input_proj", "output_proj"))]
    missing = [k for k in missing if not k.endswith(".initialized")
               and not (k.startswith("quantizer.") and ".input_proj." in k)]
    if missing or unexpected:
        raise RuntimeError(f"state-dict mismatch: missing={missing[:8]} unexpected={unexpected[:8]}")
    return m


def run(cfg, weights, codes):
    m = build(cfg, weights)
    with torch.no_grad():
        return m(torch.from_numpy(np.asarray(codes, dtype=np.int64)))
