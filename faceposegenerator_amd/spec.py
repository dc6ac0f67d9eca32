"""Graph specification of the ID-Booth sampling path (SD-2.1 UNet + VAE decoder) as data.

The reference selects these models by name at ``inference_ID-Booth.py:103-104`` and never
spells the graph out; the shapes below restate the diffusers-layout configs of
``stabilityai/stable-diffusion-2-1-base`` and are validated by the
published parameter counts (Appendix F):

    UNet 865,910,724   VAE decoder + post_quant_conv 49,490,199   rank-4 LoRA 829,952

Tensor names follow the diffusers state-dict layout so that a user-supplied local model
directory can be loaded unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

Shape = Tuple[int, ...]


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 64
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # diffusers calls this "attention_head_dim" but for SD-2.x these are head COUNTS
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_proj_dim: int = 320          # sinusoidal width = block_out_channels[0]
    prediction_type: str = "epsilon"  # "v_prediction" for SD-2.1 768 (BASELINE config 5)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def up_has_attn(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_has_attn))


@dataclass(frozen=True)
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


@dataclass(frozen=True)
class SchedulerConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    prediction_type: str = "epsilon"
    variance_type: str = "fixed_small"
    clip_sample: bool = False
    steps_offset: int = 1
    timestep_spacing: str = "leading"


SD21_UNET = UNetConfig()
SD21_VAE = VAEConfig()
SD21_SCHED = SchedulerConfig()

# Reduced-width graph with the same topology (head_dim 64, 32 groups, every C a multiple of 64):
# used by CPU tests and kernel-level GPU tests where the full 866 M-parameter model is too slow
# for the oracle.  Not a benchmark configuration.
TINY_UNET = UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4),
                       cross_attention_dim=128, time_proj_dim=64)
TINY_VAE = VAEConfig(block_out_channels=(64, 64, 128, 128))


# --------------------------------------------------------------------------------------
# UNet
# --------------------------------------------------------------------------------------
def _resnet(prefix: str, cin: int, cout: int, temb: int | None, out: Dict[str, Shape]) -> None:
    out[f"{prefix}.norm1.weight"] = (cin,)
    out[f"{prefix}.norm1.bias"] = (cin,)
    out[f"{prefix}.conv1.weight"] = (cout, cin, 3, 3)
    out[f"{prefix}.conv1.bias"] = (cout,)
    if temb is not None:
        out[f"{prefix}.time_emb_proj.weight"] = (cout, temb)
        out[f"{prefix}.time_emb_proj.bias"] = (cout,)
    out[f"{prefix}.norm2.weight"] = (cout,)
    out[f"{prefix}.norm2.bias"] = (cout,)
    out[f"{prefix}.conv2.weight"] = (cout, cout, 3, 3)
    out[f"{prefix}.conv2.bias"] = (cout,)
    if cin != cout:
        out[f"{prefix}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[f"{prefix}.conv_shortcut.bias"] = (cout,)


def _transformer(prefix: str, c: int, ctx: int, out: Dict[str, Shape]) -> None:
    out[f"{prefix}.norm.weight"] = (c,)
    out[f"{prefix}.norm.bias"] = (c,)
    out[f"{prefix}.proj_in.weight"] = (c, c)
    out[f"{prefix}.proj_in.bias"] = (c,)
    b = f"{prefix}.transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        out[f"{b}.{n}.weight"] = (c,)
        out[f"{b}.{n}.bias"] = (c,)
    for attn, kdim in (("attn1", c), ("attn2", ctx)):
        out[f"{b}.{attn}.to_q.weight"] = (c, c)
        out[f"{b}.{attn}.to_k.weight"] = (c, kdim)
        out[f"{b}.{attn}.to_v.weight"] = (c, kdim)
        out[f"{b}.{attn}.to_out.0.weight"] = (c, c)
        out[f"{b}.{attn}.to_out.0.bias"] = (c,)
    out[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{b}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{b}.ff.net.2.bias"] = (c,)
    out[f"{prefix}.proj_out.weight"] = (c, c)
    out[f"{prefix}.proj_out.bias"] = (c,)


@dataclass
class ResnetSpec:
    name: str
    cin: int
    cout: int
    side_div: int          # spatial side = sample_size // side_div
    skip_channels: int = 0  # >0: input is cat([h, skip]) with the last `skip_channels` from the skip


@dataclass
class AttnSpec:
    name: str
    channels: int
    heads: int
    side_div: int


@dataclass
class UNetGraph:
    """Execution-ordered description of the UNet; shared by the oracle and the HIP engine."""
    cfg: UNetConfig
    down: List[dict] = field(default_factory=list)   # per block: resnets, attns, downsample name
    mid: dict = field(default_factory=dict)
    up: List[dict] = field(default_factory=list)
    skip_channels: List[int] = field(default_factory=list)
    skip_side_div: List[int] = field(default_factory=list)


def unet_graph(cfg: UNetConfig = SD21_UNET) -> UNetGraph:
    g = UNetGraph(cfg)
    boc = cfg.block_out_channels
    nb = len(boc)
    div = 1
    g.skip_channels.append(boc[0])
    g.skip_side_div.append(1)
    out_ch = boc[0]
    for i in range(nb):
        in_ch, out_ch = out_ch, boc[i]
        blk = {"resnets": [], "attns": [], "down": None}
        for j in range(cfg.layers_per_block):
            blk["resnets"].append(ResnetSpec(f"down_blocks.{i}.resnets.{j}",
                                             in_ch if j == 0 else out_ch, out_ch, div))
            if cfg.down_has_attn[i]:
                blk["attns"].append(AttnSpec(f"down_blocks.{i}.attentions.{j}", out_ch,
                                             cfg.num_heads[i], div))
            g.skip_channels.append(out_ch)
            g.skip_side_div.append(div)
        if i != nb - 1:
            blk["down"] = f"down_blocks.{i}.downsamplers.0.conv"
            blk["down_channels"] = out_ch
            div *= 2
            g.skip_channels.append(out_ch)
            g.skip_side_div.append(div)
        g.down.append(blk)
    c = boc[-1]
    g.mid = {"resnets": [ResnetSpec("mid_block.resnets.0", c, c, div),
                         ResnetSpec("mid_block.resnets.1", c, c, div)],
             "attn": AttnSpec("mid_block.attentions.0", c, cfg.num_heads[-1], div)}
    rev = list(reversed(boc))
    rev_heads = list(reversed(cfg.num_heads))
    skips = list(g.skip_channels)
    out_ch = rev[0]
    for i in range(nb):
        prev, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, nb - 1)]
        blk = {"resnets": [], "attns": [], "up": None}
        for j in range(cfg.layers_per_block + 1):
            skip = skips.pop()
            assert skip == (in_ch if j == cfg.layers_per_block else out_ch)
            r_in = prev if j == 0 else out_ch
            blk["resnets"].append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", r_in + skip, out_ch,
                                             div, skip_channels=skip))
            if cfg.up_has_attn[i]:
                blk["attns"].append(AttnSpec(f"up_blocks.{i}.attentions.{j}", out_ch,
                                             rev_heads[i], div))
        if i != nb - 1:
            blk["up"] = f"up_blocks.{i}.upsamplers.0.conv"
            blk["up_channels"] = out_ch
            div //= 2
        g.up.append(blk)
    assert not skips and div == 1
    return g


def unet_param_shapes(cfg: UNetConfig = SD21_UNET) -> Dict[str, Shape]:
    g = unet_graph(cfg)
    out: Dict[str, Shape] = {}
    c0 = cfg.block_out_channels[0]
    te = cfg.time_embed_dim
    out["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    out["conv_in.bias"] = (c0,)
    out["time_embedding.linear_1.weight"] = (te, cfg.time_proj_dim)
    out["time_embedding.linear_1.bias"] = (te,)
    out["time_embedding.linear_2.weight"] = (te, te)
    out["time_embedding.linear_2.bias"] = (te,)
    for blk in g.down:
        for j, r in enumerate(blk["resnets"]):
            _resnet(r.name, r.cin, r.cout, te, out)
            if blk["attns"]:
                a = blk["attns"][j]
                _transformer(a.name, a.channels, cfg.cross_attention_dim, out)
        if blk["down"]:
            ch = blk["down_channels"]
            out[blk["down"] + ".weight"] = (ch, ch, 3, 3)
            out[blk["down"] + ".bias"] = (ch,)
    _resnet(g.mid["resnets"][0].name, g.mid["resnets"][0].cin, g.mid["resnets"][0].cout, te, out)
    _transformer(g.mid["attn"].name, g.mid["attn"].channels, cfg.cross_attention_dim, out)
    _resnet(g.mid["resnets"][1].name, g.mid["resnets"][1].cin, g.mid["resnets"][1].cout, te, out)
    for blk in g.up:
        for j, r in enumerate(blk["resnets"]):
            _resnet(r.name, r.cin, r.cout, te, out)
            if blk["attns"]:
                a = blk["attns"][j]
                _transformer(a.name, a.channels, cfg.cross_attention_dim, out)
        if blk["up"]:
            ch = blk["up_channels"]
            out[blk["up"] + ".weight"] = (ch, ch, 3, 3)
            out[blk["up"] + ".bias"] = (ch,)
    out["conv_norm_out.weight"] = (c0,)
    out["conv_norm_out.bias"] = (c0,)
    out["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    out["conv_out.bias"] = (cfg.out_channels,)
    return out


def unet_attention_modules(cfg: UNetConfig = SD21_UNET) -> List[AttnSpec]:
    g = unet_graph(cfg)
    mods: List[AttnSpec] = []
    for blk in g.down:
        mods += blk["attns"]
    mods.append(g.mid["attn"])
    for blk in g.up:
        mods += blk["attns"]
    return mods


# --------------------------------------------------------------------------------------
# LoRA (train_ID-Booth.py:672-678: r=4, alpha=4, to_q/to_k/to_v/to_out.0 of attn1 and attn2)
# --------------------------------------------------------------------------------------
LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


def lora_param_shapes(cfg: UNetConfig = SD21_UNET, rank: int = 4) -> Dict[str, Shape]:
    """Keys in the diffusers dialect written by the reference (train_ID-Booth.py:705,716-720)."""
    out: Dict[str, Shape] = {}
    for a in unet_attention_modules(cfg):
        for attn in ("attn1", "attn2"):
            for t in LORA_TARGETS:
                cin = cfg.cross_attention_dim if (attn == "attn2" and t in ("to_k", "to_v")) \
                    else a.channels
                base = f"unet.{a.name}.transformer_blocks.0.{attn}.{t}"
                out[f"{base}.lora.down.weight"] = (rank, cin)
                out[f"{base}.lora.up.weight"] = (a.channels, rank)
    return out


# --------------------------------------------------------------------------------------
# VAE decoder
# --------------------------------------------------------------------------------------
@dataclass
class VAEGraph:
    cfg: VAEConfig
    mid_channels: int
    up: List[dict]


def vae_graph(cfg: VAEConfig = SD21_VAE) -> VAEGraph:
    rev = list(reversed(cfg.block_out_channels))
    up = []
    out_ch = rev[0]
    for i, ch in enumerate(rev):
        prev, out_ch = out_ch, ch
        blk = {"resnets": [], "up": None, "channels": out_ch}
        for j in range(cfg.layers_per_block + 1):
            blk["resnets"].append((f"decoder.up_blocks.{i}.resnets.{j}",
                                   prev if j == 0 else out_ch, out_ch))
        if i != len(rev) - 1:
            blk["up"] = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        up.append(blk)
    return VAEGraph(cfg, rev[0], up)


def vae_decoder_param_shapes(cfg: VAEConfig = SD21_VAE) -> Dict[str, Shape]:
    g = vae_graph(cfg)
    out: Dict[str, Shape] = {}
    lc, cm = cfg.latent_channels, g.mid_channels
    out["post_quant_conv.weight"] = (lc, lc, 1, 1)
    out["post_quant_conv.bias"] = (lc,)
    out["decoder.conv_in.weight"] = (cm, lc, 3, 3)
    out["decoder.conv_in.bias"] = (cm,)
    _resnet("decoder.mid_block.resnets.0", cm, cm, None, out)
    a = "decoder.mid_block.attentions.0"
    out[f"{a}.group_norm.weight"] = (cm,)
    out[f"{a}.group_norm.bias"] = (cm,)
    for t in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{t}.weight"] = (cm, cm)
        out[f"{a}.{t}.bias"] = (cm,)
    _resnet("decoder.mid_block.resnets.1", cm, cm, None, out)
    for blk in g.up:
        for name, cin, cout in blk["resnets"]:
            _resnet(name, cin, cout, None, out)
        if blk["up"]:
            ch = blk["channels"]
            out[blk["up"] + ".weight"] = (ch, ch, 3, 3)
            out[blk["up"] + ".bias"] = (ch,)
    c_last = cfg.block_out_channels[0]
    out["decoder.conv_norm_out.weight"] = (c_last,)
    out["decoder.conv_norm_out.bias"] = (c_last,)
    out["decoder.conv_out.weight"] = (cfg.out_channels, c_last, 3, 3)
    out["decoder.conv_out.bias"] = (cfg.out_channels,)
    return out


def vae_encoder_blocks(cfg: VAEConfig = SD21_VAE) -> List[dict]:
    """diffusers Encoder.down_blocks (DownEncoderBlock2D x4): layers_per_block resnets each, a stride-2 Downsample2D
    (padding 0 + F.pad (0,1,0,1)) after every block but the last (AutoencoderKL.encode, train_ID-Booth.py:1001)."""
    blocks, prev = [], cfg.block_out_channels[0]
    for i, ch in enumerate(cfg.block_out_channels):
        blk = {"resnets": [], "down": None, "channels": ch}
        for j in range(cfg.layers_per_block):
            blk["resnets"].append((f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch))
        if i != len(cfg.block_out_channels) - 1:
            blk["down"] = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        blocks.append(blk)
        prev = ch
    return blocks


def vae_encoder_param_shapes(cfg: VAEConfig = SD21_VAE) -> Dict[str, Shape]:
    out: Dict[str, Shape] = {}
    lc, c0, cm = cfg.latent_channels, cfg.block_out_channels[0], cfg.block_out_channels[-1]
    out["encoder.conv_in.weight"] = (c0, cfg.out_channels, 3, 3)          # in_channels == out_channels == 3 (RGB)
    out["encoder.conv_in.bias"] = (c0,)
    for blk in vae_encoder_blocks(cfg):
        for name, cin, cout in blk["resnets"]:
            _resnet(name, cin, cout, None, out)
        if blk["down"]:
            out[blk["down"] + ".weight"] = (blk["channels"], blk["channels"], 3, 3)
            out[blk["down"] + ".bias"] = (blk["channels"],)
    _resnet("encoder.mid_block.resnets.0", cm, cm, None, out)
    a = "encoder.mid_block.attentions.0"
    out[f"{a}.group_norm.weight"] = (cm,)
    out[f"{a}.group_norm.bias"] = (cm,)
    for t in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{t}.weight"] = (cm, cm)
        out[f"{a}.{t}.bias"] = (cm,)
    _resnet("encoder.mid_block.resnets.1", cm, cm, None, out)
    out["encoder.conv_norm_out.weight"] = (cm,)
    out["encoder.conv_norm_out.bias"] = (cm,)
    out["encoder.conv_out.weight"] = (2 * lc, cm, 3, 3)                    # double_z: mean and log-variance
    out["encoder.conv_out.bias"] = (2 * lc,)
    out["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    out["quant_conv.bias"] = (2 * lc,)
    return out


# Legacy VAE attention key names used by SD-2.x hub checkpoints.
VAE_LEGACY_ATTN_KEYS = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def count_params(shapes: Dict[str, Shape]) -> int:
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


# --------------------------------------------------------------------------------------
# Algorithmic work: MACs of one UNet forward and one VAE decode
# --------------------------------------------------------------------------------------
def unet_macs(cfg: UNetConfig = SD21_UNET, latent_side: int | None = None, ctx_len: int = 77) -> int:
    side0 = latent_side or cfg.sample_size
    g = unet_graph(cfg)
    te = cfg.time_embed_dim
    macs = 0

    def conv(cin, cout, side, k=3):
        return cin * cout * k * k * side * side

    def resnet(r: ResnetSpec):
        s = side0 // r.side_div
        m = conv(r.cin, r.cout, s) + conv(r.cout, r.cout, s) + te * r.cout
        if r.cin != r.cout:
            m += conv(r.cin, r.cout, s, 1)
        return m

    def attn(a: AttnSpec):
        s = side0 // a.side_div
        n, c = s * s, a.channels
        m = 2 * n * c * c                      # proj_in / proj_out
        m += 4 * n * c * c + 2 * n * n * c     # self: q,k,v,out + QK^T + PV
        m += 2 * n * c * c + 2 * ctx_len * cfg.cross_attention_dim * c + 2 * n * ctx_len * c
        m += n * c * 8 * c + n * 4 * c * c     # GEGLU in + FF out
        return m

    macs += conv(cfg.in_channels, cfg.block_out_channels[0], side0)
    macs += cfg.time_proj_dim * te + te * te
    for blk in g.down:
        for j, r in enumerate(blk["resnets"]):
            macs += resnet(r)
            if blk["attns"]:
                macs += attn(blk["attns"][j])
        if blk["down"]:
            s = side0 // blk["resnets"][0].side_div // 2
            macs += conv(blk["down_channels"], blk["down_channels"], s)
    macs += resnet(g.mid["resnets"][0]) + attn(g.mid["attn"]) + resnet(g.mid["resnets"][1])
    for blk in g.up:
        for j, r in enumerate(blk["resnets"]):
            macs += resnet(r)
            if blk["attns"]:
                macs += attn(blk["attns"][j])
        if blk["up"]:
            s = side0 // blk["resnets"][0].side_div * 2
            macs += conv(blk["up_channels"], blk["up_channels"], s)
    macs += conv(cfg.block_out_channels[0], cfg.out_channels, side0)
    return macs


def vae_decode_macs(cfg: VAEConfig = SD21_VAE, latent_side: int = 64) -> int:
    g = vae_graph(cfg)
    cm = g.mid_channels
    s = latent_side
    macs = cfg.latent_channels ** 2 * s * s + cfg.latent_channels * cm * 9 * s * s

    def resnet(cin, cout, side):
        m = (cin * cout + cout * cout) * 9 * side * side
        if cin != cout:
            m += cin * cout * side * side
        return m

    n = s * s
    macs += 2 * resnet(cm, cm, s) + 4 * n * cm * cm + 2 * n * n * cm
    for blk in g.up:
        for _, cin, cout in blk["resnets"]:
            macs += resnet(cin, cout, s)
        if blk["up"]:
            s *= 2
            macs += blk["channels"] ** 2 * 9 * s * s
    macs += cfg.block_out_channels[0] * cfg.out_channels * 9 * s * s
    return macs


# --------------------------------------------------------------------------------------
# CLIP text encoder: transformers CLIPTextModel as SD-2.1-base ships it
# (text_encoder/config.json: 23 pre-LN layers, width 1024, 16 heads, MLP 4096, exact GELU, causal mask)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ClipTextConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 23
    num_attention_heads: int = 16
    max_position_embeddings: int = 77
    vocab_size: int = 49408
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5
    bos_token_id: int = 49406
    eos_token_id: int = 49407
    pad_token_id: int = 0                 # SD-2.x tokenizer pads with id 0 ("!")


SD21_CLIP = ClipTextConfig()
TINY_CLIP = ClipTextConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                           vocab_size=1000, bos_token_id=998, eos_token_id=999)


def clip_text_param_shapes(cfg: ClipTextConfig = SD21_CLIP) -> Dict[str, Shape]:
    """Keys as transformers 4.34.1 writes them (``text_model.`` prefix)."""
    d, f = cfg.hidden_size, cfg.intermediate_size
    out: Dict[str, Shape] = {
        "text_model.embeddings.token_embedding.weight": (cfg.vocab_size, d),
        "text_model.embeddings.position_embedding.weight": (cfg.max_position_embeddings, d),
        "text_model.final_layer_norm.weight": (d,), "text_model.final_layer_norm.bias": (d,),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[f"{p}.self_attn.{n}.weight"] = (d, d)
            out[f"{p}.self_attn.{n}.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            out[f"{p}.{n}.weight"] = (d,)
            out[f"{p}.{n}.bias"] = (d,)
        out[f"{p}.mlp.fc1.weight"] = (f, d)
        out[f"{p}.mlp.fc1.bias"] = (f,)
        out[f"{p}.mlp.fc2.weight"] = (d, f)
        out[f"{p}.mlp.fc2.bias"] = (d,)
    return out
