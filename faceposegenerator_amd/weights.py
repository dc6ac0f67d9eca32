"""Weights: seeded synthetic factory and local-directory I/O in the diffusers on-disk layout.

The reference loads weights by model NAME (`inference_ID-Booth.py:103`) and LoRA checkpoints from
`Trained_LoRA_Models/<model>/<ID>/checkpoint-31-6400` (`inference_ID-Booth.py:52-54,98,107`);
neither is available offline, so benchmarks and tests use seeded synthetic tensors of the exact
published shapes, and real weights can be read from a LOCAL directory only.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict
from typing import Dict, Optional, Tuple

import torch

from . import spec as S

SD = Dict[str, torch.Tensor]


def _gen(seed: int, index: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + index * 7919 + 17) % (2 ** 63 - 1))
    return g


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, rich: bool = True) -> SD:
    """N(0, 1/fan_in) matrices; with ``rich`` non-trivial biases and norm affine parameters so
    that every epilogue path is exercised.  Each tensor has its own generator (seed, index in
    sorted key order), so generation order does not matter and results are version-stable
    (torch's CPU mt19937 stream)."""
    sd: SD = {}
    for i, name in enumerate(sorted(shapes)):
        shape = shapes[name]
        g = _gen(seed, i)
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = torch.randn(shape, generator=g, dtype=torch.float32) * (fan_in ** -0.5)
        elif name.endswith(".weight"):               # norm gamma
            sd[name] = (0.8 + 0.4 * torch.rand(shape, generator=g)) if rich else torch.ones(shape)
        else:                                        # biases / norm beta
            sd[name] = 0.05 * torch.randn(shape, generator=g) if rich else torch.zeros(shape)
    return sd


def synth_unet(cfg: S.UNetConfig = S.SD21_UNET, seed: int = 1234, calibrated: bool = False) -> SD:
    sd = synth_state_dict(S.unet_param_shapes(cfg), seed)
    return calibrate_unet(sd, cfg) if calibrated else sd


CALIB_K = 2.0           # value of the constant companion channels of a pass-through lane pair
CALIB_OUT_SCALE = 0.3   # scale of the random part of conv_out: eps = x_t + 0.3 * conv_out_random(...) (std ~0.14 at full size)


def calibrate_unet(sd: SD, cfg: S.UNetConfig, out_scale: float = CALIB_OUT_SCALE) -> SD:
    """Seeded N(0, 1/fan_in) weights make eps uncorrelated with x_t, and a 30-step DDPM chain then blows the latents up by
    1/sqrt(alpha_bar_T) (std 1.4 -> 20): the regime a trained denoiser never visits (SURVEY.md §7 step 1 asks for activations that
    stay O(1)).  A trained eps-model's output is x_t plus a context-dependent correction; this edit gives the synthetic network that
    shape, by construction and without running it:

      * a pass-through LANE PAIR per latent channel c: two channels of conv_in carry +x_c and -x_c (identity taps, no bias); the
        same channels of the LAST up resnet (which receives conv_in's output as its skip) copy them from the skip (conv2 and
        shortcut rows replaced), and the following transformer's proj_out adds nothing to them;
      * the remaining channels of the pair's GroupNorm group(s) carry the constants +-K along the same path, so that conv_norm_out
        sees groups whose mean and variance are fixed by construction (x_t has unit variance, +x and -x cancel in the mean):
        h+ = (x - mu) / sd = -h-;
      * conv_norm_out: gamma 1, beta 0 on the lanes, gamma = beta = 0 on the constants; SiLU then gives silu(h) and silu(-h), whose
        DIFFERENCE is exactly h (sigmoid(h) + sigmoid(-h) = 1): no offset to ride on, so 16-bit storage keeps its relative precision;
      * conv_out: random part scaled by ``out_scale`` (and its expectation removed from the bias: a constant term would be summed by
        all 30 steps), plus +sd / -sd on the centre taps of the pair, so that eps_c = x_c + out_scale * (random network).

    Every other weight — all 16 transformer blocks, the other 21 resnets, every LoRA-affected matrix — stays the seeded random tensor."""
    import math
    sd = dict(sd)
    lc, c0, G = cfg.in_channels, cfg.block_out_channels[0], cfg.norm_num_groups
    cpg = c0 // G
    g = S.unet_graph(cfg)
    last = g.up[-1]["resnets"][-1]
    if last.skip_channels != c0 or last.cout != c0:
        raise ValueError("calibrate_unet: the last up resnet must take conv_in's output as its skip")
    xch = last.cin - last.skip_channels
    K = CALIB_K
    if cpg == 2 and 2 * lc <= G:            # groups {+x, +K} and {-x, -K}: h- = -h+ with mu = K / 2, sd^2 = 1/2 + K^2/4
        plus = [4 * c for c in range(lc)]
        minus = [4 * c + 2 for c in range(lc)]
        consts = {**{4 * c + 1: K for c in range(lc)}, **{4 * c + 3: -K for c in range(lc)}}
        mu, sdev = K / 2.0, math.sqrt(0.5 + K * K / 4.0)
    elif cpg >= 4 and cpg % 2 == 0 and lc <= G:   # one group {+x, -x, (cpg-2)/2 x (+K), (cpg-2)/2 x (-K)}: mu = 0
        plus = [c * cpg for c in range(lc)]
        minus = [c * cpg + 1 for c in range(lc)]
        consts = {c * cpg + j: (K if j % 2 == 0 else -K) for c in range(lc) for j in range(2, cpg)}
        mu, sdev = 0.0, math.sqrt((2.0 + (cpg - 2) * K * K) / cpg)
    else:
        raise ValueError("calibrate_unet needs 2 or an even number >= 4 of channels per GroupNorm group at the first level")
    designed = plus + minus + sorted(consts)

    def edit(name, fn):
        t = sd[name].clone()
        fn(t)
        sd[name] = t

    def conv_in_w(w):
        w[designed] = 0.0
        for c in range(lc):
            w[plus[c], c, 1, 1] = 1.0
            w[minus[c], c, 1, 1] = -1.0

    def conv_in_b(b):
        b[designed] = 0.0
        for ch, k in consts.items():
            b[ch] = k

    edit("conv_in.weight", conv_in_w)
    edit("conv_in.bias", conv_in_b)
    edit(last.name + ".conv2.weight", lambda w: w.__setitem__(designed, 0.0))
    edit(last.name + ".conv2.bias", lambda b: b.__setitem__(designed, 0.0))

    def shortcut_w(w):
        w[designed] = 0.0
        for ch in designed:
            w[ch, xch + ch, 0, 0] = 1.0

    edit(last.name + ".conv_shortcut.weight", shortcut_w)
    edit(last.name + ".conv_shortcut.bias", lambda b: b.__setitem__(designed, 0.0))
    if g.up[-1]["attns"]:
        a = g.up[-1]["attns"][-1].name
        edit(a + ".proj_out.weight", lambda w: w.__setitem__(designed, 0.0))
        edit(a + ".proj_out.bias", lambda b: b.__setitem__(designed, 0.0))

    def norm_w(w):
        w[plus + minus] = 1.0
        w[sorted(consts)] = 0.0

    edit("conv_norm_out.weight", norm_w)
    edit("conv_norm_out.bias", lambda b: b.__setitem__(designed, 0.0))

    # expectation of the random part: E[silu(gamma z + beta)] per channel for z ~ N(0,1) (what conv_norm_out emits) times the summed taps
    z = torch.linspace(-6.0, 6.0, 2401, dtype=torch.float64)
    pdf = torch.exp(-0.5 * z * z)
    pdf = pdf / pdf.sum()
    gam, bet = sd["conv_norm_out.weight"].double(), sd["conv_norm_out.bias"].double()
    mean_act = (torch.nn.functional.silu(gam[:, None] * z[None, :] + bet[:, None]) * pdf[None, :]).sum(dim=1)      # [C]
    w_rand = sd["conv_out.weight"].double() * out_scale
    w_rand[:, designed] = 0.0                          # the random part does not read the designed channels
    dc = (w_rand.sum(dim=(2, 3)) * mean_act[None, :]).sum(dim=1).float()                                          # [out_channels]

    def conv_out_w(w):
        w *= out_scale
        w[:, designed] = 0.0
        for c in range(lc):
            w[c, plus[c], 1, 1] = sdev             # eps_c = sd * (silu(h+) - silu(h-)) = sd * h+ = x_c - mu
            w[c, minus[c], 1, 1] = -sdev

    def conv_out_b(b):
        b *= out_scale
        b += mu - dc

    edit("conv_out.weight", conv_out_w)
    edit("conv_out.bias", conv_out_b)
    return sd


def synth_vae(cfg: S.VAEConfig = S.SD21_VAE, seed: int = 4321) -> SD:
    return synth_state_dict(S.vae_decoder_param_shapes(cfg), seed)


def synth_vae_encoder(cfg: S.VAEConfig = S.SD21_VAE, seed: int = 4322) -> SD:
    return synth_state_dict(S.vae_encoder_param_shapes(cfg), seed)


def synth_clip(cfg: S.ClipTextConfig = S.SD21_CLIP, seed: int = 99) -> SD:
    sd = synth_state_dict(S.clip_text_param_shapes(cfg), seed)
    g = _gen(seed, 100003)
    # embeddings are looked up, not multiplied: unit-variance rows like a trained table after scaling
    sd["text_model.embeddings.token_embedding.weight"] = 0.5 * torch.randn(
        sd["text_model.embeddings.token_embedding.weight"].shape, generator=g)
    sd["text_model.embeddings.position_embedding.weight"] = 0.5 * torch.randn(
        sd["text_model.embeddings.position_embedding.weight"].shape, generator=g)
    return sd


def synth_lora(cfg: S.UNetConfig = S.SD21_UNET, seed: int = 1, rank: int = 4, dialect: str = "diffusers") -> SD:
    """A ~ N(0, 1/r) (PEFT 'gaussian' init, train_ID-Booth.py:675); B ~ N(0, 0.02) so that the
    branch is non-zero.  Keys in the dialect the reference writes (train_ID-Booth.py:705)."""
    shapes = S.lora_param_shapes(cfg, rank)
    sd: SD = {}
    for i, name in enumerate(sorted(shapes)):
        g = _gen(seed + 77, i)
        std = (1.0 / rank) if ".down." in name else 0.02
        t = torch.randn(shapes[name], generator=g, dtype=torch.float32) * std
        if dialect == "peft":
            name = name.replace(".lora.down.weight", ".lora_A.weight").replace(".lora.up.weight", ".lora_B.weight")
        sd[name] = t
    return sd


def normalize_lora_keys(raw: SD) -> SD:
    """Both dialects -> ``<module path>.lora_A/lora_B.weight`` without the ``unet.`` prefix
    (what diffusers' ``load_lora_weights`` does before injecting PEFT layers)."""
    out: SD = {}
    for k, v in raw.items():
        if k.startswith("text_encoder."):
            continue      # no text-encoder LoRA in the reference runs (configs/config_train_SD21.py:72)
        if k.startswith("unet."):
            k = k[len("unet."):]
        k = k.replace(".lora.down.weight", ".lora_A.weight").replace(".lora.up.weight", ".lora_B.weight")
        k = k.replace(".lora_A.default.weight", ".lora_A.weight").replace(".lora_B.default.weight", ".lora_B.weight")
        out[k] = v.float()
    return out


# --------------------------------------------------------------------------------------
# On-disk layout: model dir + LoRA checkpoint dir
# --------------------------------------------------------------------------------------
def save_model_dir(root: str, unet_sd: SD, vae_sd: SD, ucfg: S.UNetConfig, vcfg: S.VAEConfig,
                   scfg: S.SchedulerConfig = S.SD21_SCHED) -> None:
    from safetensors.torch import save_file
    for sub in ("unet", "vae", "scheduler"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    with open(os.path.join(root, "model_index.json"), "w") as f:
        json.dump({"_class_name": "StableDiffusionPipeline",
                   "unet": ["diffusers", "UNet2DConditionModel"],
                   "vae": ["diffusers", "AutoencoderKL"],
                   "scheduler": ["diffusers", "PNDMScheduler"]}, f, indent=1)
    ucfg_d = {"_class_name": "UNet2DConditionModel", "in_channels": ucfg.in_channels,
              "out_channels": ucfg.out_channels, "sample_size": ucfg.sample_size,
              "block_out_channels": list(ucfg.block_out_channels),
              "attention_head_dim": list(ucfg.num_heads),
              "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in ucfg.down_has_attn],
              "up_block_types": ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in ucfg.up_has_attn],
              "layers_per_block": ucfg.layers_per_block, "cross_attention_dim": ucfg.cross_attention_dim,
              "norm_num_groups": ucfg.norm_num_groups, "norm_eps": ucfg.norm_eps,
              "use_linear_projection": True, "act_fn": "silu", "flip_sin_to_cos": True, "freq_shift": 0}
    with open(os.path.join(root, "unet", "config.json"), "w") as f:
        json.dump(ucfg_d, f, indent=1)
    vcfg_d = {"_class_name": "AutoencoderKL", "latent_channels": vcfg.latent_channels,
              "out_channels": vcfg.out_channels, "block_out_channels": list(vcfg.block_out_channels),
              "layers_per_block": vcfg.layers_per_block, "norm_num_groups": vcfg.norm_num_groups,
              "scaling_factor": vcfg.scaling_factor, "act_fn": "silu"}
    with open(os.path.join(root, "vae", "config.json"), "w") as f:
        json.dump(vcfg_d, f, indent=1)
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(asdict(scfg), _class_name="PNDMScheduler", trained_betas=None), f, indent=1)
    save_file({k: v.contiguous() for k, v in unet_sd.items()},
              os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in vae_sd.items()},
              os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))


def _read_json(path: str) -> dict:
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    with open(path) as f:
        return json.load(f)


def load_unet_config(root: str) -> S.UNetConfig:
    d = _read_json(os.path.join(root, "unet", "config.json"))
    boc = tuple(d["block_out_channels"])
    heads = d.get("attention_head_dim", 8)
    heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * len(boc)
    return S.UNetConfig(in_channels=d.get("in_channels", 4), out_channels=d.get("out_channels", 4),
                        sample_size=d.get("sample_size", 64), block_out_channels=boc, num_heads=heads,
                        down_has_attn=tuple(t.startswith("CrossAttn") for t in d["down_block_types"]),
                        layers_per_block=d.get("layers_per_block", 2),
                        cross_attention_dim=d.get("cross_attention_dim", 1024),
                        norm_num_groups=d.get("norm_num_groups", 32), norm_eps=d.get("norm_eps", 1e-5),
                        time_proj_dim=boc[0])


def load_vae_config(root: str) -> S.VAEConfig:
    d = _read_json(os.path.join(root, "vae", "config.json"))
    return S.VAEConfig(latent_channels=d.get("latent_channels", 4), out_channels=d.get("out_channels", 3),
                       block_out_channels=tuple(d["block_out_channels"]),
                       layers_per_block=d.get("layers_per_block", 2),
                       norm_num_groups=d.get("norm_num_groups", 32),
                       scaling_factor=d.get("scaling_factor", 0.18215))


def load_scheduler_config(root: str, subfolder: Optional[str] = "scheduler") -> S.SchedulerConfig:
    p = os.path.join(root, subfolder) if subfolder else root
    d = _read_json(os.path.join(p, "scheduler_config.json"))
    # DDPMScheduler.from_pretrained on a PNDM config keeps the shared keys and takes DDPM-class
    # defaults for the rest.
    return S.SchedulerConfig(num_train_timesteps=d.get("num_train_timesteps", 1000),
                             beta_start=d.get("beta_start", 0.00085), beta_end=d.get("beta_end", 0.012),
                             beta_schedule=d.get("beta_schedule", "scaled_linear"),
                             prediction_type=d.get("prediction_type", "epsilon"),
                             clip_sample=d.get("clip_sample", False),
                             steps_offset=d.get("steps_offset", 1))


def _load_safetensors(path: str) -> SD:
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    from safetensors.torch import load_file
    return {k: v.float() for k, v in load_file(path).items()}


def load_unet_weights(root: str) -> SD:
    return _load_safetensors(os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))


def load_vae_decoder_weights(root: str) -> SD:
    raw = _load_safetensors(os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    out: SD = {}
    for k, v in raw.items():
        if k.startswith("encoder.") or k.startswith("quant_conv."):
            continue
        if ".attentions.0." in k:                       # legacy names, Appendix B
            for old, new in S.VAE_LEGACY_ATTN_KEYS.items():
                k = k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
            if k.endswith(".weight") and v.ndim == 4:   # very old checkpoints store 1x1 convs
                v = v[:, :, 0, 0]
        out[k] = v
    return out


def load_vae_encoder_weights(root: str) -> SD:
    """``encoder.*`` and ``quant_conv.*`` of <root>/vae (AutoencoderKL.encode, train_ID-Booth.py:1001)."""
    raw = _load_safetensors(os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    out: SD = {}
    for k, v in raw.items():
        if not (k.startswith("encoder.") or k.startswith("quant_conv.")):
            continue
        if ".attentions.0." in k:
            for old, new in S.VAE_LEGACY_ATTN_KEYS.items():
                k = k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
            if k.endswith(".weight") and v.ndim == 4:
                v = v[:, :, 0, 0]
        out[k] = v
    if not out:
        raise FileNotFoundError(f"{root}/vae holds no encoder weights")
    return out


def load_text_encoder(root: str):
    """(ClipTextConfig, state dict with ``text_model.`` keys) from <root>/text_encoder, or None if absent."""
    d = os.path.join(root, "text_encoder")
    if not os.path.isfile(os.path.join(d, "config.json")):
        return None
    c = _read_json(os.path.join(d, "config.json"))
    if c.get("hidden_act", "gelu") not in ("gelu",):
        raise ValueError(f"text encoder hidden_act {c.get('hidden_act')!r} is not supported (SD-2.x uses exact GELU)")
    cfg = S.ClipTextConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                           num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                           max_position_embeddings=c.get("max_position_embeddings", 77), vocab_size=c["vocab_size"],
                           layer_norm_eps=c.get("layer_norm_eps", 1e-5), bos_token_id=c.get("bos_token_id", 49406),
                           eos_token_id=c.get("eos_token_id", 49407), pad_token_id=c.get("pad_token_id", 0) or 0)
    raw = _load_safetensors(os.path.join(d, "model.safetensors"))
    sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in raw.items()
          if "text_projection" not in k and "position_ids" not in k}
    return cfg, sd


def save_text_encoder(root: str, cfg: S.ClipTextConfig, sd: SD) -> None:
    from safetensors.torch import save_file
    d = os.path.join(root, "text_encoder")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(asdict(cfg), _class_name="CLIPTextModel", architectures=["CLIPTextModel"], projection_dim=512), f, indent=1)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "model.safetensors"))


def save_lora(path_or_dir: str, lora_sd: SD, weight_name: str = "pytorch_lora_weights.safetensors") -> str:
    from safetensors.torch import save_file
    path = path_or_dir
    if not path.endswith(".safetensors"):
        os.makedirs(path_or_dir, exist_ok=True)
        path = os.path.join(path_or_dir, weight_name)
    save_file({k: v.contiguous() for k, v in lora_sd.items()}, path)
    return path


def load_lora(path_or_dir: str, weight_name: str = "pytorch_lora_weights.safetensors") -> Tuple[SD, Dict[str, float]]:
    """Returns (normalized tensors, per-module alpha).  No alpha entries => alpha = rank => scale 1
    (what the reference's checkpoints contain)."""
    path = path_or_dir
    if os.path.isdir(path_or_dir):
        path = os.path.join(path_or_dir, weight_name)
    raw = _load_safetensors(path)
    alphas = {k[len("unet."):-len(".alpha")] if k.startswith("unet.") else k[:-len(".alpha")]: float(v)
              for k, v in raw.items() if k.endswith(".alpha")}
    tensors = normalize_lora_keys({k: v for k, v in raw.items() if not k.endswith(".alpha")})
    return tensors, alphas
