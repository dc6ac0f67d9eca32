"""CLIP text encoder on the gfx950 kernels (the step immediately before the sampler).

Replaces ``text_encoder(input_ids, attention_mask=None)[0]`` of ``StableDiffusionPipeline.encode_prompt``
(/root/reference/inference_ID-Booth.py:138 implicitly; same API at train_ID-Booth.py:476-491): transformers
``CLIPTextModel`` as SD-2.1-base ships it — token + position embeddings, 23 pre-LN layers (causal self-attention,
16 heads x 64, exact-GELU MLP 1024 -> 4096 -> 1024), final LayerNorm, last_hidden_state.

Kernels reused from the sampler: ``idb_layernorm``, ``idb_gemm`` (fused Q|K|V projection with bias, GELU epilogue,
residual epilogue), ``idb_attention`` (causal flag); new: ``idb_embed_tokens``.  ~45 GFLOP per prompt, once per call.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch

from . import _lib as L
from . import spec as S
from .engine import HipEngine, _stream


class ClipTextEncoder:
    def __init__(self, eng: HipEngine, cfg: S.ClipTextConfig, sd: Dict[str, torch.Tensor]):
        if cfg.hidden_size // cfg.num_attention_heads != 64 or cfg.hidden_size % 64:
            raise ValueError("the attention kernel needs head_dim 64 (CLIP-H: 1024 / 16)")
        if cfg.hidden_act != "gelu":
            raise ValueError(f"hidden_act {cfg.hidden_act!r} unsupported (SD-2.x text encoder uses exact GELU)")
        self.eng, self.cfg = eng, cfg
        self.config = SimpleNamespace(**cfg.__dict__)
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in sd.items()}
        w, p = {}, "text_model."
        w["tok"] = eng._f32(sd[p + "embeddings.token_embedding.weight"])
        w["pos"] = eng._f32(sd[p + "embeddings.position_embedding.weight"])
        for i in range(cfg.num_hidden_layers):
            lp = f"{p}encoder.layers.{i}."
            for n in ("1", "2"):
                w[f"{i}.ln{n}.g"] = eng._f32(sd[lp + f"layer_norm{n}.weight"])
                w[f"{i}.ln{n}.b"] = eng._f32(sd[lp + f"layer_norm{n}.bias"])
            w[f"{i}.qkv.w"] = eng._pack_mat(torch.cat([sd[lp + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0))
            w[f"{i}.qkv.b"] = eng._f32(torch.cat([sd[lp + f"self_attn.{n}_proj.bias"] for n in "qkv"], dim=0))
            w[f"{i}.o.w"] = eng._pack_mat(sd[lp + "self_attn.out_proj.weight"])
            w[f"{i}.o.b"] = eng._f32(sd[lp + "self_attn.out_proj.bias"])
            w[f"{i}.fc1.w"] = eng._pack_mat(sd[lp + "mlp.fc1.weight"])
            w[f"{i}.fc1.b"] = eng._f32(sd[lp + "mlp.fc1.bias"])
            w[f"{i}.fc2.w"] = eng._pack_mat(sd[lp + "mlp.fc2.weight"])
            w[f"{i}.fc2.b"] = eng._f32(sd[lp + "mlp.fc2.bias"])
        w["lnf.g"] = eng._f32(sd[p + "final_layer_norm.weight"])
        w["lnf.b"] = eng._f32(sd[p + "final_layer_norm.bias"])
        self.w = w
        torch.cuda.synchronize(eng.device)

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids [B, L] int64 (host or device) -> last_hidden_state [B, L, hidden] fp32 on the device."""
        eng, cfg, w = self.eng, self.cfg, self.w
        if input_ids.ndim != 2 or input_ids.shape[1] > cfg.max_position_embeddings:
            raise ValueError(f"input_ids must be [B, L <= {cfg.max_position_embeddings}], got {tuple(input_ids.shape)}")
        if int(input_ids.min()) < 0 or int(input_ids.max()) >= cfg.vocab_size:
            raise ValueError("token id out of range")
        b, n = input_ids.shape
        d, heads, f = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        ids = input_ids.to(device=eng.device, dtype=torch.int64).contiguous()
        eng.arena.reset()
        eng._pinned.clear()
        rows = b * n
        x = eng.arena.alloc((rows, d), eng.tdt)
        L.check(eng.lib.idb_embed_tokens(ids.data_ptr(), w["tok"].data_ptr(), w["pos"].data_ptr(), x.data_ptr(), b, n, d, eng.dt,
                                         _stream()), "idb_embed_tokens")
        for i in range(cfg.num_hidden_layers):
            h = eng.layernorm(x, rows, d, w[f"{i}.ln1.g"], w[f"{i}.ln1.b"])
            qkv = eng.linear(h, w[f"{i}.qkv.w"], 3 * d, d, bias=w[f"{i}.qkv.b"])
            eng.arena.free(h)
            p = qkv.data_ptr()
            o = eng.attention(qkv, 3 * d, p + 2 * d, p + 4 * d, 3 * d, b, heads, n, n, n, causal=True)
            eng.arena.free(qkv)
            x2 = eng.linear(o, w[f"{i}.o.w"], d, d, bias=w[f"{i}.o.b"], residual=x)
            eng.arena.free(o)
            eng.arena.free(x)
            h = eng.layernorm(x2, rows, d, w[f"{i}.ln2.g"], w[f"{i}.ln2.b"])
            m = eng.linear(h, w[f"{i}.fc1.w"], f, d, bias=w[f"{i}.fc1.b"], act=1)
            eng.arena.free(h)
            x = eng.linear(m, w[f"{i}.fc2.w"], d, f, bias=w[f"{i}.fc2.b"], residual=x2)
            eng.arena.free(m)
            eng.arena.free(x2)
        y = eng.layernorm(x, rows, d, w["lnf.g"], w["lnf.b"])
        eng.arena.free(x)
        out = torch.empty((b, n, d), dtype=torch.float32, device=eng.device)
        L.check(eng.lib.idb_nhwc_to_nchw_f32(y.data_ptr(), out.data_ptr(), rows, 1, d, eng.dt, _stream()), "idb_nhwc_to_nchw_f32")
        eng.arena.free(y)
        return out

    def __call__(self, input_ids, attention_mask=None, **kw):
        """``text_encoder(ids)[0]`` form (train_ID-Booth.py:484-489)."""
        if attention_mask is not None:
            raise ValueError("attention_mask is not supported (SD-2.1 text encoder config has no use_attention_mask)")
        out = self.encode(input_ids)
        return (out,)
