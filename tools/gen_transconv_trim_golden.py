#!/usr/bin/env python3
"""Known answers for the two trans-conv trim policies from transformers (run in the build container only):
  both sides  : Qwen3OmniMoeCausalTransConvNet   (the sibling decoder's block, TF:...qwen3_omni_moe:3216-3228)
  right only  : MimiConvTranspose1d, causal, trim_right_ratio = 1 (TF:models/mimi/modeling_mimi.py)
-> tests/golden/transconv_trim.npz (inputs, weights, both outputs)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as qm
from transformers.models.mimi import modeling_mimi as mm, configuration_mimi as cm

torch.manual_seed(0)
out = {}
for name, (cin, cout, s, L) in {"a": (6, 4, 4, 9), "b": (5, 7, 3, 1), "c": (8, 8, 5, 12)}.items():
    K = 2 * s
    x = torch.randn(1, cin, L)
    w = torch.randn(cin, cout, K) * 0.3
    b = torch.randn(cout) * 0.1
    both = qm.Qwen3OmniMoeCausalTransConvNet(cin, cout, K, s)
    both.conv.weight.data.copy_(w); both.conv.bias.data.copy_(b)
    right = mm.MimiConvTranspose1d(cm.MimiConfig(use_causal_conv=True, trim_right_ratio=1.0), cin, cout, K, s, bias=True)
    right.conv.weight.data.copy_(w); right.conv.bias.data.copy_(b)
    with torch.no_grad():
        yb = both(x)[0].numpy(); yr = right(x)[0].numpy()
    assert yb.shape == (cout, (L - 1) * s) and yr.shape == (cout, L * s), (yb.shape, yr.shape)
    for k, v in (("x", x[0].numpy()), ("w", w.numpy()), ("b", b.numpy()), ("both", yb), ("right", yr)):
        out[f"{name}_{k}"] = v
    out[f"{name}_stride"] = np.array(s)
path = os.path.join(ROOT, "tests", "golden", "transconv_trim.npz")
np.savez_compressed(path, **out)
print("wrote", path)
