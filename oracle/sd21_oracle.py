"""ORACLE — test infrastructure only.  CPU fp32 restatement of the ID-Booth sampling path.

PARITY UNPINNED: the reference (`/root/reference`) has no tests, golden vectors or fixtures
for this path, and the arithmetic lives in un-vendored third-party packages that
are not installed and cannot be fetched here: ``diffusers==0.32.2`` (requirements.txt:4),
``transformers==4.34.1`` (requirements.txt:5), ``peft`` (unpinned, requirements.txt:6).  This
file restates their published algorithms in plain ``torch.nn.functional`` ops; it is anchored on
the reference's own call sites and on the structural invariants of SURVEY.md Appendix C/F
(parameter counts, scheduler known-answers, skip-tensor shapes), which `tests/` checks.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s ``cpu_baseline`` leg may import this
module.  The product path (`faceposegenerator_amd.engine`) never does.

What each function follows (reference call site -> upstream symbol restated):
  * ``unet_forward``      inference_ID-Booth.py:138, train_ID-Booth.py:1040-1046
                          -> diffusers ``UNet2DConditionModel.forward`` (unet_2d_condition.py),
                          ``ResnetBlock2D`` (resnet.py), ``Transformer2DModel``, ``BasicTransformerBlock``,
                          ``GEGLU``, ``Attention``/``AttnProcessor2_0``, ``get_timestep_embedding``
  * ``lora`` handling     inference_ID-Booth.py:107, train_ID-Booth.py:672-678
                          -> peft ``lora.Linear.forward``: y = W x + (alpha/r) * B(A x)
  * ``ddpm_*``            inference_ID-Booth.py:104, train_ID-Booth.py:1081
                          -> diffusers ``DDPMScheduler`` (scheduling_ddpm.py)
  * ``vae_decode``        train_ID-Booth.py:410-412 -> ``AutoencoderKL.decode`` / ``Decoder`` (vae.py)
  * ``sample``            inference_ID-Booth.py:138 -> ``StableDiffusionPipeline.__call__``
  * ``postprocess_*``     inference_ID-Booth.py:139-144 -> ``VaeImageProcessor.postprocess`` and
                          torchvision ``save_image``'s uint8 quantisation
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from faceposegenerator_amd import spec as S

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# Error-attribution hook (tests/tools only; None = the plain fp32 restatement).  When set to ``f(kind, tensor) -> tensor`` it is
# applied at the points where the HIP engine rounds to its operand dtype, so that the CPU oracle can emulate the product's
# rounding class by class: kind "res" = residual-stream tensors (conv_in / resnet / transformer / down- and up-sample outputs
# and the h0..h3 stream inside a transformer block), "act" = every other stored activation (norm outputs, conv1 output,
# q/k/v, attention output, GEGLU product).  Weight rounding is emulated by rounding the state dict before the call.
ROUND = None


def _r(kind: str, t: Tensor) -> Tensor:
    return t if ROUND is None else ROUND(kind, t)


# Second hook, for the fp8 path (BASELINE configs[4]): applied INSTEAD of ROUND to the GroupNorm+SiLU output that feeds a 3x3 conv of a
# UNet ResnetBlock2D (the HIP engine writes that tensor as e4m3 straight from fp32), called as ROUND_CONV_IN(tensor, scale) with the
# layer's activation scale.  ``fp8_quantize`` / ``fp8_act_scale`` / ``fp8_weights`` are the emulation of the engine's quantisers:
# (8 max|gamma| + max|beta|) / 448 per GroupNorm layer for the activations, absmax / 448 per output channel for the weights.
ROUND_CONV_IN = None


def fp8_act_scale(sd: SD, norm: str) -> float:
    return float(8.0 * sd[norm + ".weight"].abs().max() + sd[norm + ".bias"].abs().max()) / 448.0


def fp8_quantize(t: Tensor, scale: float) -> Tensor:
    return (t * (1.0 / scale)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale


def fp8_weights(sd: SD) -> SD:
    """State dict with the conv1 / conv2 weights of every UNet ResnetBlock2D replaced by their dequantised e4m3 form."""
    out = dict(sd)
    for k, v in sd.items():
        if ".resnets." in k and k.endswith((".conv1.weight", ".conv2.weight")):
            s_ = v.flatten(1).abs().amax(1) / 448.0
            s_ = torch.where(s_ > 0, s_, torch.ones_like(s_))
            out[k] = (v * (1.0 / s_)[:, None, None, None]).to(torch.float8_e4m3fn).float() * s_[:, None, None, None]
    return out


# ----------------------------------------------------------------------------------------
# embeddings  (diffusers models/embeddings.py: get_timestep_embedding, flip_sin_to_cos=True,
# downscale_freq_shift=0, max_period=10000)
# ----------------------------------------------------------------------------------------
def timestep_embedding(timesteps: Tensor, dim: int) -> Tensor:
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)   # flipped: [cos | sin]


# ----------------------------------------------------------------------------------------
# LoRA  (peft tuners/lora/layer.py Linear.forward; scale = alpha / r = 1 for the reference)
# ----------------------------------------------------------------------------------------
def _linear(sd: SD, name: str, x: Tensor, lora: Optional[SD] = None, lora_scale: float = 1.0) -> Tensor:
    y = F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))
    if lora is not None:
        a = lora.get(name + ".lora_A.weight")
        if a is not None:
            b = lora[name + ".lora_B.weight"]
            y = y + lora_scale * F.linear(F.linear(x, a), b)
    return y


def normalize_lora_keys(raw: SD) -> SD:
    """Accept both dialects: ``unet.<path>.lora.down/up.weight`` (written by
    the reference, train_ID-Booth.py:705) and PEFT ``<path>.lora_A/lora_B.weight``."""
    out: SD = {}
    for k, v in raw.items():
        if k.startswith("unet."):
            k = k[len("unet."):]
        k = k.replace(".lora.down.weight", ".lora_A.weight").replace(".lora.up.weight", ".lora_B.weight")
        k = k.replace(".lora_A.default.weight", ".lora_A.weight").replace(".lora_B.default.weight", ".lora_B.weight")
        out[k] = v.float()
    return out


def merge_lora(sd: SD, lora: SD, scale: float = 1.0) -> SD:
    """W' = W + scale * B A (the merged form; must equal the unmerged form up to fp32 rounding)."""
    out = dict(sd)
    for k in lora:
        if k.endswith(".lora_A.weight"):
            base = k[: -len(".lora_A.weight")]
            out[base + ".weight"] = sd[base + ".weight"] + scale * lora[base + ".lora_B.weight"] @ lora[k]
    return out


# ----------------------------------------------------------------------------------------
# UNet building blocks
# ----------------------------------------------------------------------------------------
def resnet_block(sd: SD, p: str, x: Tensor, temb: Optional[Tensor], groups: int, eps: float) -> Tensor:
    fp8_in = ROUND_CONV_IN is not None and temb is not None          # UNet resnets only (the VAE's have no temb)
    h = F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = ROUND_CONV_IN(F.silu(h), fp8_act_scale(sd, p + ".norm1")) if fp8_in else _r("act", F.silu(h))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    if temb is not None:
        t = F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
        h = h + t[:, :, None, None]
    h = _r("act", h)
    h = F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = ROUND_CONV_IN(F.silu(h), fp8_act_scale(sd, p + ".norm2")) if fp8_in else _r("act", F.silu(h))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return _r("res", x + h)


def attention(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, lora: Optional[SD]) -> Tensor:
    b, n, c = x.shape
    q = _r("act", _linear(sd, p + ".to_q", x, lora))
    k = _r("act", _linear(sd, p + ".to_k", ctx, lora))
    v = _r("act", _linear(sd, p + ".to_v", ctx, lora))
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)          # scale 1/sqrt(d), no mask, no dropout
    o = _r("act", o.transpose(1, 2).reshape(b, n, c))
    return _linear(sd, p + ".to_out.0", o, lora)


def transformer_block(sd: SD, p: str, h: Tensor, ctx: Tensor, heads: int, lora: Optional[SD]) -> Tensor:
    c = h.shape[-1]
    n1 = _r("act", F.layer_norm(h, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5))
    h = _r("res", attention(sd, p + ".attn1", n1, n1, heads, lora) + h)
    n2 = _r("act", F.layer_norm(h, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5))
    h = _r("res", attention(sd, p + ".attn2", n2, ctx, heads, lora) + h)
    n3 = _r("act", F.layer_norm(h, (c,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5))
    proj = F.linear(n3, sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"])
    val, gate = proj.chunk(2, dim=-1)                    # GEGLU: first half value, second half gate
    ff = F.linear(_r("act", val * F.gelu(gate)), sd[p + ".ff.net.2.weight"], sd[p + ".ff.net.2.bias"])
    return _r("res", ff + h)


def transformer_2d(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, groups: int,
                   lora: Optional[SD]) -> Tensor:
    b, c, hh, ww = x.shape
    h = _r("act", F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6))
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h = _r("res", F.linear(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"]))
    h = transformer_block(sd, p + ".transformer_blocks.0", h, ctx, heads, lora)
    h = F.linear(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return _r("res", h + x)


def unet_forward(sd: SD, cfg: S.UNetConfig, sample: Tensor, timestep, ctx: Tensor,
                 lora: Optional[SD] = None, taps: Optional[dict] = None) -> Tensor:
    """eps_theta(x_t, t, c).  sample [B,4,H,W] fp32, timestep scalar or [B], ctx [B,L,cross_dim].

    ``taps`` (optional dict) receives named intermediate activations for per-module parity tests; with
    ``taps["__blocks__"] = True`` also the output of every resnet / transformer / down- / up-sample block under its
    module name (the per-block error attribution of tests/test_parity_gpu.py).
    """
    g = S.unet_graph(cfg)
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    b = sample.shape[0]
    t = torch.as_tensor(timestep)
    if t.ndim == 0:
        t = t[None].expand(b)
    temb = timestep_embedding(t, cfg.time_proj_dim)
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    if taps is not None:
        taps["temb"] = temb

    def tap(name: str, t: Tensor) -> Tensor:
        if taps is not None and taps.get("__blocks__"):
            taps[name] = t
        return t

    ctx = _r("act", ctx)
    h = tap("conv_in", _r("res", F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)))
    skips: List[Tensor] = [h]
    for blk in g.down:
        for j, r in enumerate(blk["resnets"]):
            h = tap(r.name, resnet_block(sd, r.name, h, temb, G, eps))
            if blk["attns"]:
                a = blk["attns"][j]
                h = tap(a.name, transformer_2d(sd, a.name, h, ctx, a.heads, G, lora))
            skips.append(h)
        if blk["down"]:
            h = tap(blk["down"], _r("res", F.conv2d(h, sd[blk["down"] + ".weight"], sd[blk["down"] + ".bias"], stride=2, padding=1)))
            skips.append(h)
    if taps is not None:
        taps["down_out"] = h
    h = tap(g.mid["resnets"][0].name, resnet_block(sd, g.mid["resnets"][0].name, h, temb, G, eps))
    a = g.mid["attn"]
    h = tap(a.name, transformer_2d(sd, a.name, h, ctx, a.heads, G, lora))
    h = tap(g.mid["resnets"][1].name, resnet_block(sd, g.mid["resnets"][1].name, h, temb, G, eps))
    if taps is not None:
        taps["mid_out"] = h
    for blk in g.up:
        for j, r in enumerate(blk["resnets"]):
            h = torch.cat([h, skips.pop()], dim=1)
            h = tap(r.name, resnet_block(sd, r.name, h, temb, G, eps))
            if blk["attns"]:
                a = blk["attns"][j]
                h = tap(a.name, transformer_2d(sd, a.name, h, ctx, a.heads, G, lora))
        if blk["up"]:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = tap(blk["up"], _r("res", F.conv2d(h, sd[blk["up"] + ".weight"], sd[blk["up"] + ".bias"], padding=1)))
    assert not skips
    if taps is not None:
        taps["up_out"] = h
    h = F.group_norm(h, G, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)
    h = _r("act", F.silu(h))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# ----------------------------------------------------------------------------------------
# VAE decoder  (diffusers autoencoder_kl.py decode -> vae.py Decoder)
# ----------------------------------------------------------------------------------------
def vae_attention(sd: SD, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    b, c, hh, ww = x.shape
    h = F.group_norm(x, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    h = h.view(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]   # one head, d=C
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def vae_decode(sd: SD, cfg: S.VAEConfig, z: Tensor) -> Tensor:
    """z is the already-unscaled latent (caller divides by scaling_factor, as the reference does at
    train_ID-Booth.py:410)."""
    g = S.vae_graph(cfg)
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = resnet_block(sd, "decoder.mid_block.resnets.0", h, None, G, eps)
    h = vae_attention(sd, "decoder.mid_block.attentions.0", h, G, eps)
    h = resnet_block(sd, "decoder.mid_block.resnets.1", h, None, G, eps)
    for blk in g.up:
        for name, _, _ in blk["resnets"]:
            h = resnet_block(sd, name, h, None, G, eps)
        if blk["up"]:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[blk["up"] + ".weight"], sd[blk["up"] + ".bias"], padding=1)
    h = F.group_norm(h, G, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps)
    h = F.silu(h)
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def vae_encode(sd: SD, cfg: S.VAEConfig, x: Tensor) -> Tuple[Tensor, Tensor]:
    """AutoencoderKL.encode(x).latent_dist parameters (mean, logvar), NCHW — diffusers Encoder.forward + quant_conv +
    DiagonalGaussianDistribution (logvar clamped to [-30, 20]); called at train_ID-Booth.py:1001.  x: [B,3,H,W] in [-1,1].
    Downsample2D of the encoder uses padding 0 with F.pad(x, (0,1,0,1)) before the stride-2 conv."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for blk in S.vae_encoder_blocks(cfg):
        for name, _, _ in blk["resnets"]:
            h = resnet_block(sd, name, h, None, G, eps)
        if blk["down"]:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0.0)
            h = F.conv2d(h, sd[blk["down"] + ".weight"], sd[blk["down"] + ".bias"], stride=2)
    h = resnet_block(sd, "encoder.mid_block.resnets.0", h, None, G, eps)
    h = vae_attention(sd, "encoder.mid_block.attentions.0", h, G, eps)
    h = resnet_block(sd, "encoder.mid_block.resnets.1", h, None, G, eps)
    h = F.silu(F.group_norm(h, G, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    m = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_latent_sample(mean: Tensor, logvar: Tensor, noise: Optional[Tensor], scaling_factor: float = 1.0) -> Tensor:
    """latent_dist.sample() (noise given) or .mode() (noise None), times scaling_factor (train_ID-Booth.py:1001-1002)."""
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return z * scaling_factor


def postprocess_np(image: Tensor) -> Tensor:
    """VaeImageProcessor.postprocess(output_type='np'): NHWC float32 in [0,1]."""
    return (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous()

def to_uint8(image01_nhwc: Tensor) -> Tensor:
    """torchvision.utils.save_image quantisation: mul(255).add_(0.5).clamp_(0,255).to(uint8)."""
    return image01_nhwc.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------
# DDPM scheduler  (diffusers scheduling_ddpm.py)
# ----------------------------------------------------------------------------------------
def ddpm_tables(cfg: S.SchedulerConfig = S.SD21_SCHED) -> Tensor:
    assert cfg.beta_schedule == "scaled_linear"
    betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, cfg.num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_timesteps(n: int, cfg: S.SchedulerConfig = S.SD21_SCHED) -> List[int]:
    assert cfg.timestep_spacing == "leading"
    ratio = cfg.num_train_timesteps // n
    return [int(round(i * ratio)) + cfg.steps_offset for i in range(n - 1, -1, -1)]


def ddpm_step(alphas_cumprod: Tensor, timesteps: Sequence[int], t: int, model_output: Tensor,
              sample: Tensor, noise: Optional[Tensor], prediction_type: str = "epsilon") -> Tuple[Tensor, Tensor]:
    """One reverse step; returns (prev_sample, pred_original_sample).  ``noise`` is the N(0,1)
    draw the scheduler makes when t > 0 (the caller draws it so RNG order stays explicit)."""
    idx = list(timesteps).index(int(t))
    prev_t = timesteps[idx + 1] if idx + 1 < len(timesteps) else -1
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    if prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    elif prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
    else:
        raise ValueError(prediction_type)
    c_x0 = (a_prev ** 0.5 * cur_b) / b_t
    c_x = cur_a ** 0.5 * b_prev / b_t
    prev = c_x0 * x0 + c_x * sample
    if t > 0:
        var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)       # fixed_small
        prev = prev + var ** 0.5 * noise
    return prev, x0


def ddpm_coefficients(alphas_cumprod: Tensor, timesteps: Sequence[int], t: int):
    """(alpha_bar_t, c_x0, c_x, sigma) — the Appendix C known-answer quantities."""
    idx = list(timesteps).index(int(t))
    prev_t = timesteps[idx + 1] if idx + 1 < len(timesteps) else -1
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
    return float(a_t), float(a_prev ** 0.5 * cur_b / b_t), float(cur_a ** 0.5 * b_prev / b_t), float(var ** 0.5)


# ----------------------------------------------------------------------------------------
# Whole sampler  (StableDiffusionPipeline.__call__)
# ----------------------------------------------------------------------------------------
def draw_noise(generator: torch.Generator, batch: int, steps: int, latent_hw: Tuple[int, int],
               channels: int = 4) -> Tensor:
    """RNG order of the upstream pipeline with a CPU generator: initial latents first, then one
    draw per step including the last (t=1 > 0).  Returns [steps+1, B, C, H, W]."""
    shape = (batch, channels, latent_hw[0], latent_hw[1])
    return torch.stack([torch.randn(shape, generator=generator, dtype=torch.float32)
                        for _ in range(steps + 1)])


def sample(unet_sd: SD, ucfg: S.UNetConfig, prompt_embeds: Tensor, negative_prompt_embeds: Tensor,
           noise: Tensor, num_inference_steps: int, guidance_scale: float,
           lora: Optional[SD] = None, sched: S.SchedulerConfig = S.SD21_SCHED,
           trace: Optional[list] = None) -> Tensor:
    """Returns final latents [B,4,h,w].  noise = draw_noise(...) stack.  ``trace`` collects
    (eps_uncond, eps_cond, latents_after_step) per step for teacher-forced parity tests."""
    ac = ddpm_tables(sched)
    ts = ddpm_timesteps(num_inference_steps, sched)
    latents = noise[0] * 1.0                                   # init_noise_sigma = 1 for DDPM
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds])   # uncond FIRST
    for i, t in enumerate(ts):
        x_in = torch.cat([latents] * 2)                        # scale_model_input = identity
        eps = unet_forward(unet_sd, ucfg, x_in, t, ctx, lora)
        e_u, e_c = eps.chunk(2)
        eps_g = e_u + guidance_scale * (e_c - e_u)
        latents, _ = ddpm_step(ac, ts, t, eps_g, latents, noise[i + 1], sched.prediction_type)
        if trace is not None:
            trace.append((e_u.clone(), e_c.clone(), latents.clone()))
    return latents


# ----------------------------------------------------------------------------------------
# DPM-Solver++ (2M) — the validation sampler of train_ID-Booth.py:155 (diffusers scheduling_dpmsolver_multistep.py with
# its defaults on SD-2.1's scheduler config: dpmsolver++, order 2, midpoint, lower_order_final, final sigma zero,
# "leading" spacing).  Restated from the published algorithm (Lu et al. 2022, "DPM-Solver++", Alg. 2) in upstream's
# sigma parameterisation; PARITY UNPINNED like the rest of this file, except for the analytic checks in
# tests/test_oracle_cpu.py (order 1 == DDIM, last step returns the x0 prediction).
# ----------------------------------------------------------------------------------------
def dpmpp_timesteps_sigmas(n: int, cfg: S.SchedulerConfig = S.SD21_SCHED) -> Tuple[List[int], Tensor]:
    ac = ddpm_tables(cfg).double()
    ratio = cfg.num_train_timesteps // (n + 1)
    ts = [int(round(i * ratio)) + cfg.steps_offset for i in range(n + 1)][::-1][:-1]
    sig = ((1 - ac) / ac) ** 0.5
    sigmas = torch.cat([sig[torch.tensor(ts)], torch.zeros(1, dtype=torch.float64)]).float()
    return ts, sigmas


def dpmpp_2m_sample(unet_sd: SD, ucfg: S.UNetConfig, prompt_embeds: Tensor, negative_prompt_embeds: Tensor, latents: Tensor,
                    num_inference_steps: int, guidance_scale: float, lora: Optional[SD] = None,
                    sched: S.SchedulerConfig = S.SD21_SCHED, solver_order: int = 2, model=None) -> Tensor:
    """``model(x, t) -> guided model output`` may replace the UNet (analytic tests)."""
    ts, sigmas = dpmpp_timesteps_sigmas(num_inference_steps, sched)
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if model is None else None

    def conv(sigma):
        alpha_t = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha_t, sigma * alpha_t

    x = latents * 1.0
    x0_hist: List[Tensor] = []
    for i, t in enumerate(ts):
        if model is None:
            eps = unet_forward(unet_sd, ucfg, torch.cat([x] * 2), t, ctx, lora)
            e_u, e_c = eps.chunk(2)
            out = e_u + guidance_scale * (e_c - e_u)
        else:
            out = model(x, t)
        alpha_s0, sigma_s0 = conv(sigmas[i])
        x0 = alpha_s0 * x - sigma_s0 * out if sched.prediction_type == "v_prediction" else (x - sigma_s0 * out) / alpha_s0
        x0_hist.append(x0)
        alpha_t, sigma_t = conv(sigmas[i + 1])
        lam_t = torch.log(alpha_t) - torch.log(sigma_t)
        lam_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lam_t - lam_s0
        last = i == len(ts) - 1
        if solver_order == 1 or i == 0 or last:
            x = (sigma_t / sigma_s0) * x - (alpha_t * (torch.exp(-h) - 1.0)) * x0
        else:
            alpha_s1, sigma_s1 = conv(sigmas[i - 1])
            lam_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
            h_0 = lam_s0 - lam_s1
            r0 = h_0 / h
            d0, d1 = x0_hist[-1], (1.0 / r0) * (x0_hist[-1] - x0_hist[-2])
            x = (sigma_t / sigma_s0) * x - (alpha_t * (torch.exp(-h) - 1.0)) * d0 - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * d1
    return x


def decode_to_images(vae_sd: SD, vcfg: S.VAEConfig, latents: Tensor) -> Tensor:
    """latents -> NHWC float32 [0,1] (what ``output_type='np'`` returns)."""
    return postprocess_np(vae_decode(vae_sd, vcfg, latents / vcfg.scaling_factor))
