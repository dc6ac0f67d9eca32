"""API mirror of ``diffusers.StableDiffusionPipeline`` for the calls the reference makes.

Reference usage reproduced (same names / kwargs / error behaviour):
  * ``StableDiffusionPipeline.from_pretrained(path, torch_dtype=...)``, ``.to(device)``
    (inference_ID-Booth.py:103) — LOCAL directory only, model names cannot be resolved offline;
  * ``pipe.scheduler = DDPMScheduler.from_pretrained(path, subfolder="scheduler")`` (:104);
  * ``pipe.load_lora_weights(dir)`` (:107), ``pipe.set_progress_bar_config(disable=True)`` (:108);
  * ``pipe(prompt=..., negative_prompt=..., output_type="np", generator=..., num_inference_steps=30,
    guidance_scale=5.0, width=512, height=512).images`` (:138);
  * ``prompt_embeds`` / ``negative_prompt_embeds`` kwargs (train_ID-Booth.py:1221-1224);
  * ``pipe.unet(x, t, ehs, return_dict=False)[0]`` (train_ID-Booth.py:1040-1046), ``pipe.vae.decode(z).sample``
    (:412), ``pipe.vae.config.scaling_factor`` (:1002).

Text conditioning: when the local model directory holds ``text_encoder/`` and ``tokenizer/`` the
pipeline encodes string prompts itself — tokenisation is upstream's own host-side ``transformers.CLIPTokenizer`` (pure
string processing, needs the vocab/merges files of the model directory), the 23-layer CLIP text model runs on the HIP
kernels (``text_encoder.ClipTextEncoder``).  Without them pass ``prompt_embeds`` / ``negative_prompt_embeds``.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, List, Optional, Union

import torch

from . import spec as S
from . import weights as W
from .scheduler import DDPMScheduler, DPMSolverMultistepScheduler


class _range:
    """roctx range around a pipeline stage (rocprofv3 --marker-trace shows text-encode / sampling loop / VAE decode);
    on only with IDB_ROCTX=1 so that the default path makes no extra calls."""
    _on = os.environ.get("IDB_ROCTX") == "1"

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if self._on:
            try:
                torch.cuda.nvtx.range_push(self.name)        # roctxRangePush on ROCm builds of torch
            except Exception:
                type(self)._on = False
        return self

    def __exit__(self, *exc):
        if self._on:
            try:
                torch.cuda.nvtx.range_pop()
            except Exception:
                type(self)._on = False
        return False


class StableDiffusionPipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class _LatentDist:
    """DiagonalGaussianDistribution as the reference uses it: ``vae.encode(x).latent_dist.sample()`` (train_ID-Booth.py:1001)."""

    def __init__(self, engine, image):
        self._engine, self._image = engine, image
        _, self.mean, self.logvar = engine.vae_encode(image, None)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        # upstream randn_tensor: a CPU generator draws on the CPU, then the sample moves to the device
        if generator is not None and generator.device.type == "cpu":
            noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32).to(self.mean.device)
        else:
            noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32, device=self.mean.device)
        return self._engine.vae_encode(self._image, noise)[0]

    def mode(self):
        return self.mean


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class UNetHandle:
    """``pipe.unet``: callable like UNet2DConditionModel for the kwargs the reference uses."""

    def __init__(self, pipe: "StableDiffusionPipeline"):
        self._pipe = pipe
        self.config = SimpleNamespace(in_channels=pipe.unet_config.in_channels, sample_size=pipe.unet_config.sample_size,
                                      time_cond_proj_dim=None, cross_attention_dim=pipe.unet_config.cross_attention_dim)

    def __call__(self, sample, timestep, encoder_hidden_states, class_labels=None, return_dict: bool = True, **kw):
        if class_labels is not None:
            raise ValueError("class_labels are not supported (SD-2.1 has num_class_embeds = null)")
        out = self._pipe._engine().unet_forward(sample, timestep, encoder_hidden_states)
        return (out,) if not return_dict else SimpleNamespace(sample=out)


class VAEHandle:
    """``pipe.vae``: ``decode(z).sample`` and ``config.scaling_factor``."""

    def __init__(self, pipe: "StableDiffusionPipeline"):
        self._pipe = pipe
        self.config = SimpleNamespace(scaling_factor=pipe.vae_config.scaling_factor,
                                      latent_channels=pipe.vae_config.latent_channels)

    def decode(self, z, return_dict: bool = True, **kw):
        out = self._pipe._engine().vae_decode(z)
        return (out,) if not return_dict else _DecoderOutput(out)

    def encode(self, x, return_dict: bool = True, **kw):
        """``vae.encode(pixel_values).latent_dist`` (train_ID-Booth.py:1001); x: [B,3,H,W] in [-1,1]."""
        eng = self._pipe._engine()
        if not getattr(eng, "has_vae_encoder", False):
            sd = self._pipe.vae_encoder_state_dict()
            if sd is None:
                raise FileNotFoundError("this pipeline was built without VAE encoder weights")
            eng.pack_vae_encoder(sd)
        dist = _LatentDist(eng, x)
        return (dist,) if not return_dict else _EncoderOutput(dist)


class StableDiffusionPipeline:
    def __init__(self, unet_config: S.UNetConfig, vae_config: S.VAEConfig, unet_sd, vae_sd,
                 scheduler: Optional[DDPMScheduler] = None, torch_dtype=None, encode_prompt: Optional[Callable] = None,
                 text_encoder=None, tokenizer=None):
        self.unet_config, self.vae_config = unet_config, vae_config
        self._unet_sd, self._vae_sd = unet_sd, vae_sd
        self.scheduler = scheduler or DDPMScheduler(S.SchedulerConfig(prediction_type=unet_config.prediction_type))
        # f16 is the reference's operand dtype (torch_dtype=torch.float16, inference_ID-Booth.py:103) and the default here: same MFMA
        # rate as bf16, 11 instead of 8 significant bits; bf16 stays selectable
        self.dtype_name = {None: "f16", torch.bfloat16: "bf16", torch.float16: "f16", "bf16": "bf16", "f16": "f16", "fp8": "fp8",
                           getattr(torch, "float8_e4m3fn", "fp8"): "fp8"}.get(torch_dtype)
        if self.dtype_name is None:
            raise ValueError(f"torch_dtype {torch_dtype} unsupported: use torch.bfloat16, torch.float16 or 'fp8' (e4m3 resnet convs)")
        self.device = torch.device("cpu")
        self._eng = None
        self._lora = None
        self._progress = {}
        self.encode_prompt_fn = encode_prompt
        self._text = text_encoder            # (ClipTextConfig, state dict) or None
        self._text_eng = None
        self.tokenizer = tokenizer
        self.use_graph = True
        self.vae_chunk = 4
        self.unet = UNetHandle(self)
        self.vae = VAEHandle(self)
        self.vae_scale_factor = 2 ** (len(vae_config.block_out_channels) - 1)
        self._vae_encoder = None             # state dict, or a directory to read it from on first vae.encode()

    def vae_encoder_state_dict(self):
        if isinstance(self._vae_encoder, str):
            self._vae_encoder = W.load_vae_encoder_weights(self._vae_encoder)
        return self._vae_encoder

    def set_vae_encoder_weights(self, sd) -> None:
        """Attach AutoencoderKL encoder weights (``encoder.*``, ``quant_conv.*``) to a pipeline built from state dicts."""
        self._vae_encoder = sd

    # ---- construction -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=None, **kw) -> "StableDiffusionPipeline":
        root = pretrained_model_name_or_path
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root!r} is not a local directory; model names cannot be resolved offline "
                                    f"(expected the diffusers layout: model_index.json, unet/, vae/, scheduler/)")
        ucfg, vcfg = W.load_unet_config(root), W.load_vae_config(root)
        scfg = W.load_scheduler_config(root)
        ucfg = S.UNetConfig(**{**ucfg.__dict__, "prediction_type": scfg.prediction_type})
        text = W.load_text_encoder(root)
        tokenizer = None
        if text is not None and os.path.isdir(os.path.join(root, "tokenizer")):
            from transformers import CLIPTokenizer           # host-side string -> ids, as the reference uses it
            tokenizer = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
        pipe = cls(ucfg, vcfg, W.load_unet_weights(root), W.load_vae_decoder_weights(root), DDPMScheduler(scfg), torch_dtype,
                   text_encoder=text, tokenizer=tokenizer)
        pipe._vae_encoder = root             # encoder.* / quant_conv.* are read only if vae.encode() is called
        return pipe

    @classmethod
    def from_synthetic(cls, unet_config: S.UNetConfig = S.SD21_UNET, vae_config: S.VAEConfig = S.SD21_VAE,
                       seed: int = 1234, torch_dtype=None) -> "StableDiffusionPipeline":
        """Seeded random weights of the published shapes (there is no network for checkpoints)."""
        return cls(unet_config, vae_config, W.synth_unet(unet_config, seed), W.synth_vae(vae_config, seed + 1),
                   torch_dtype=torch_dtype)

    def to(self, device) -> "StableDiffusionPipeline":
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("this pipeline runs on an MI355X only (device must be 'cuda:N'); there is no CPU path")
        if self._eng is not None and device != self.device:
            self._eng = None
        self.device = device
        return self

    def _engine(self):
        if self._eng is None:
            if self.device.type != "cuda":
                raise RuntimeError("call .to('cuda:0') first: the sampler has no CPU path")
            from .engine import HipEngine          # raises loudly if the HIP library is missing
            self._eng = HipEngine(self.unet_config, self.vae_config, self._unet_sd, self._vae_sd, self.device, self.dtype_name)
            if self._lora is not None:
                self._eng.set_lora_groups(*self._lora)
        return self._eng

    @property
    def text_encoder(self):
        """``pipe.text_encoder(ids)[0]`` (train_ID-Booth.py:484-489); None when the model dir has no text encoder."""
        if self._text is None:
            return None
        if self._text_eng is None:
            from .text_encoder import ClipTextEncoder
            self._text_eng = ClipTextEncoder(self._engine(), *self._text)
        return self._text_eng

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance: bool = True):
        """Upstream ``encode_prompt``: 77-token max-length padding + truncation, ``text_encoder(ids)[0]``; the
        unconditional prompt defaults to "" (uncond embeddings come FIRST when concatenated by the sampler)."""
        te = self.text_encoder
        if te is None or self.tokenizer is None:
            raise NotImplementedError(
                "string prompts need text_encoder/ and tokenizer/ in the local model directory (or encode_prompt=callable); "
                "otherwise pass `prompt_embeds`/`negative_prompt_embeds` (train_ID-Booth.py:1221-1224 style)")
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        n_max = self.tokenizer.model_max_length if getattr(self.tokenizer, "model_max_length", 77) <= 77 else 77

        def ids_of(texts):
            enc = self.tokenizer(texts, padding="max_length", max_length=n_max, truncation=True, return_tensors="pt")
            return enc.input_ids

        pe = te.encode(ids_of(prompts))
        ne = None
        if do_classifier_free_guidance:
            if negative_prompt is None:
                negs = [""] * len(prompts)
            elif isinstance(negative_prompt, str):
                negs = [negative_prompt] * len(prompts)
            else:
                negs = list(negative_prompt)
                if len(negs) != len(prompts):
                    raise ValueError(f"`negative_prompt` has batch size {len(negs)}, but `prompt` has batch size {len(prompts)}.")
            ne = te.encode(ids_of(negs))
        return pe, ne

    def set_progress_bar_config(self, **kwargs) -> None:
        self._progress = dict(kwargs)

    # ---- LoRA -------------------------------------------------------------------------
    def load_lora_weights(self, pretrained_model_name_or_path_or_dict, weight_name: str = "pytorch_lora_weights.safetensors",
                          **kw) -> None:
        src = pretrained_model_name_or_path_or_dict
        if isinstance(src, (list, tuple)):
            # one adapter set per GROUP of the batch (group g = samples [g*B/G, (g+1)*B/G) of every call): a mixed-identity batch in
            # one sampler call (BASELINE configs[2]: 8 identities x 8 prompts); each entry is what the single form accepts
            sets = []
            for one in src:
                if isinstance(one, dict):
                    sets.append((W.normalize_lora_keys(one), {}))
                else:
                    if not os.path.exists(one):
                        raise FileNotFoundError(f"LoRA checkpoint {one!r} not found (local paths only)")
                    sets.append(W.load_lora(one, weight_name))
            if not sets or any(not t for t, _ in sets):
                raise ValueError("no UNet LoRA tensors found in a checkpoint of the list")
            self._lora = ([t for t, _ in sets], 1.0, sets[0][1])
            if self._eng is not None:
                self._eng.set_lora_groups(*self._lora)
            return
        if isinstance(src, dict):
            tensors, alphas = W.normalize_lora_keys(src), {}
        else:
            if not os.path.exists(src):
                raise FileNotFoundError(f"LoRA checkpoint {src!r} not found (local paths only)")
            tensors, alphas = W.load_lora(src, weight_name)
        if not tensors:
            raise ValueError("no UNet LoRA tensors found in the checkpoint")
        self._lora = ([tensors], 1.0, alphas)
        if self._eng is not None:
            self._eng.set_lora_groups(*self._lora)

    def unload_lora_weights(self) -> None:
        self._lora = None
        if self._eng is not None:
            self._eng.set_lora_groups([None])

    # ---- sampling ---------------------------------------------------------------------
    def check_inputs(self, prompt, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, "
                             f"but got {tuple(prompt_embeds.shape)} != {tuple(negative_prompt_embeds.shape)}.")

    def _encode(self, prompt, negative_prompt, do_cfg):
        if self.encode_prompt_fn is not None:
            return self.encode_prompt_fn(prompt, negative_prompt, do_cfg)
        return self.encode_prompt(prompt, negative_prompt, do_cfg)

    def prepare_noise(self, batch: int, steps: int, height: int, width: int, generator) -> torch.Tensor:
        """RNG ORDER of the upstream pipeline (randn_tensor draws on the generator's device): initial latents, then one draw
        per step including the last (t=1 > 0).  A list of generators gives one stream per sample, as upstream.  Returns
        [steps+1, B, C, h, w] fp32 (host).  Only the order of draws is the reference's: the reference seeds a
        ``torch.Generator(device='cuda:0')`` (inference_ID-Booth.py:47,111) whose Philox stream, drawn in fp16 by diffusers,
        is a different sequence of numbers from the CPU mt19937 fp32 stream used here — the same seed does not give the
        same image as the reference's CUDA run (no CUDA generator exists on this platform to compare with)."""
        lc = self.unet_config.in_channels
        h, w_ = height // self.vae_scale_factor, width // self.vae_scale_factor
        if isinstance(generator, (list, tuple)):
            if len(generator) != batch:
                raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                                 f"effective batch size of {batch}.")
            draws = []
            for _ in range(steps + 1):
                draws.append(torch.cat([torch.randn((1, lc, h, w_), generator=g, device=g.device, dtype=torch.float32).cpu()
                                        for g in generator]))
            return torch.stack(draws)
        gdev = generator.device if generator is not None else torch.device("cpu")
        return torch.stack([torch.randn((batch, lc, h, w_), generator=generator, device=gdev, dtype=torch.float32).cpu()
                            for _ in range(steps + 1)])

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                 num_images_per_prompt: int = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: str = "pil", return_dict: bool = True, noise: Optional[torch.Tensor] = None, **kw):
        height = height or self.unet_config.sample_size * self.vae_scale_factor
        width = width or self.unet_config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if output_type not in ("np", "pt", "latent", "pil", "uint8"):
            raise ValueError(f"output_type {output_type!r} not supported")
        do_cfg = guidance_scale > 1.0
        if prompt_embeds is None:
            with _range("idb:text_encoder"):
                prompt_embeds, negative_prompt_embeds = self._encode(prompt, negative_prompt, do_cfg)
        if do_cfg and negative_prompt_embeds is None:
            raise ValueError("guidance_scale > 1 needs `negative_prompt_embeds` (or a negative prompt with a text encoder)")
        if num_images_per_prompt != 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
            if negative_prompt_embeds is not None:
                negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        B = prompt_embeds.shape[0]
        eng = self._engine()
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps.tolist()
        multistep = isinstance(sch, DPMSolverMultistepScheduler)         # deterministic solver: initial latents only
        if noise is None:
            if latents is not None and not multistep:
                # upstream prepare_latents draws nothing when `latents` is given: the generator's first draw is step 0's noise
                step_noise = self.prepare_noise(B, num_inference_steps - 1, height, width, generator)
                noise = torch.cat([(latents.float().cpu() * sch.init_noise_sigma)[None], step_noise])
                latents = None
            else:
                noise = self.prepare_noise(B, 0 if multistep else num_inference_steps, height, width, generator)
        if latents is not None:
            noise = noise.clone()
            noise[0] = latents.float().cpu() * sch.init_noise_sigma
        if multistep:
            noise = noise[:1]
            coefs = torch.tensor([list(sch.step_coefficients(i)) + [float(guidance_scale)] for i in range(len(timesteps))],
                                 dtype=torch.float32)
        else:
            coefs = torch.tensor([list(sch.step_coefficients(t)) + [float(guidance_scale)] for t in timesteps], dtype=torch.float32)
        with _range("idb:sampling_loop"):
            lat = eng.sample(prompt_embeds, negative_prompt_embeds if do_cfg else None, noise.to(self.device), timesteps,
                             coefs.to(self.device), vpred=(sch.config.prediction_type == "v_prediction"),
                             use_graph=self.use_graph, multistep=multistep)
        if output_type == "latent":
            images = lat
        else:
            with _range("idb:vae_decode_postprocess"):
                img01, u8 = eng.decode_images(lat, chunk=self.vae_chunk, want_u8=True)
            if output_type == "np":
                images = img01.cpu().numpy()                       # NHWC float32 in [0,1]
            elif output_type == "pt":
                images = img01.permute(0, 3, 1, 2).contiguous()
            elif output_type == "uint8":
                images = u8
            else:
                from PIL import Image
                images = [Image.fromarray(a) for a in u8.cpu().numpy()]
        if not return_dict:
            return (images, None)
        return StableDiffusionPipelineOutput(images=images, nsfw_content_detected=None)
