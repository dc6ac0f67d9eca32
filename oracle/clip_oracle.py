"""ORACLE — test infrastructure only.  CPU fp32 restatement of the CLIP text encoder forward that
``StableDiffusionPipeline.encode_prompt`` runs for /root/reference/inference_ID-Booth.py:138 (and
train_ID-Booth.py:476-491): ``text_encoder(input_ids, attention_mask=None)[0]`` = last_hidden_state after
the final LayerNorm.  Restates transformers 4.34.1 ``models/clip/modeling_clip.py`` (CLIPTextEmbeddings,
CLIPAttention with the causal mask, CLIPMLP with exact GELU, CLIPEncoderLayer pre-LN, final_layer_norm).

PINNED: unlike the diffusion path, this restatement IS checked against a real upstream implementation —
``transformers.CLIPTextModel`` (installed here, v5.x, same arithmetic) built from a config with the same
synthetic weights (tests/test_clip_cpu.py).  Only tests/, smoke() and bench's cpu_baseline may import this."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from faceposegenerator_amd import spec as S

SD = Dict[str, torch.Tensor]


def clip_text_forward(sd: SD, cfg: S.ClipTextConfig, input_ids: torch.Tensor) -> torch.Tensor:
    """input_ids [B, L] int64 -> last_hidden_state [B, L, hidden] fp32."""
    b, n = input_ids.shape
    d, heads = cfg.hidden_size, cfg.num_attention_heads
    hd = d // heads
    p = "text_model."
    x = sd[p + "embeddings.token_embedding.weight"][input_ids] + sd[p + "embeddings.position_embedding.weight"][:n][None]
    causal = torch.full((n, n), float("-inf")).triu(1)
    for i in range(cfg.num_hidden_layers):
        lp = f"{p}encoder.layers.{i}."
        h = F.layer_norm(x, (d,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], cfg.layer_norm_eps)
        q = F.linear(h, sd[lp + "self_attn.q_proj.weight"], sd[lp + "self_attn.q_proj.bias"]) * hd ** -0.5
        k = F.linear(h, sd[lp + "self_attn.k_proj.weight"], sd[lp + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[lp + "self_attn.v_proj.weight"], sd[lp + "self_attn.v_proj.bias"])
        q, k, v = (t.view(b, n, heads, hd).transpose(1, 2) for t in (q, k, v))
        w = torch.softmax(q @ k.transpose(-1, -2) + causal, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(b, n, d)
        x = x + F.linear(a, sd[lp + "self_attn.out_proj.weight"], sd[lp + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (d,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], cfg.layer_norm_eps)
        h = F.gelu(F.linear(h, sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"])
    return F.layer_norm(x, (d,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], cfg.layer_norm_eps)
