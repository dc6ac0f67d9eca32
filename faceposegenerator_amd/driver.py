"""Generation driver: the reference's prompt/seed policy, identity-sharded multi-GPU execution with one
all-gather of the decoded images, and the PNG/JPG sink.

Restates ``/root/reference/inference_ID-Booth.py`` (a top-level script without functions):
  * prompt vocabulary and combination list            :17-45
  * ``set_seed(0)`` -> ``random.seed``                :48,67   (accelerate.utils.set_seed seeds python/numpy/torch)
  * natural-sorted identity folders                   :71-73, utils/sorting_utils.py:4-13
  * per identity ``random.sample(combos, 21)``        :94      (one permutation per ID, shared by the 3 models)
  * per (identity, model) generator seeded with the identity index   :111
  * prompt assembly incl. the ``random.choice`` pose coin            :113-134
  * output naming and the 63-tile comparison sheet                   :56-61,100,142-156

Multi-GPU: work items are independent 30-step chains; identities are dealt round-robin to
ranks (one LoRA set resident per GPU at a time), every rank decodes its own images and ONE all-gather of
fixed-size uint8 tensors (RCCL over xGMI; gloo on CPU in tests) collects them.  No other collective.
"""
from __future__ import annotations

import os
import random
import re
from dataclasses import dataclass, field
from itertools import product
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

NEGATIVE_PROMPT = ("cartoon, cgi, render, illustration, painting, drawing, black and white, "
                   "bad body proportions, landscape")                       # inference_ID-Booth.py:81
ORIGINAL_PROMPT = "face portrait photo of sks person"                      # :82
BACKGROUNDS = ["", "forest", "city street", "beach", "office", "bus", "laboratory", "factory",
               "construction site", "hospital", "night club"]              # :17
AGE_PHASES = ["", "young", "middle-aged", "old"]                           # :20
MODELS_TO_TEST = ["DreamBooth", "PortraitBooth", "ID-Booth"]               # :53
CHECKPOINT = "checkpoint-31-6400"                                          # :54
ARCH = "stable-diffusion-2-1-base"                                         # :63-65


def atoi(text: str):
    return int(text) if text.isdigit() else text


def natural_keys(text: str):
    """Human sort key: ID_2 < ID_10 (utils/sorting_utils.py:7-13)."""
    return [atoi(c) for c in re.split(r"(\d+)", text)]


@dataclass
class PolicyConfig:
    num_prompts: int = 21
    num_samples_per_prompt: int = 1
    add_gender: bool = True
    add_pose: bool = True
    add_age: bool = False
    add_background: bool = True
    do_not_use_negative_prompt: bool = False
    seed: int = 0
    guidance_scale: float = 5.0
    num_inference_steps: int = 30
    width: int = 512
    height: int = 512
    models_to_test: Tuple[str, ...] = tuple(MODELS_TO_TEST)
    checkpoint: str = CHECKPOINT

    def output_folder(self) -> str:                                        # :56-61
        out = "Generated_Samples/FacePortrait_Photo_21"
        if self.add_gender: out += "_Gender"
        if self.add_pose: out += "_Pose"
        if self.add_age: out += "_Age"
        if self.add_background: out += "_Background"
        if self.do_not_use_negative_prompt: out += "_NoNegPrompt"
        return out


def prompt_combinations(cfg: PolicyConfig) -> list:
    """inference_ID-Booth.py:33-45."""
    bgs = [f"{b} background" if b != "" else "" for b in BACKGROUNDS]
    if cfg.add_age and cfg.add_background:
        return list(product(AGE_PHASES, bgs))
    if cfg.add_background:
        if cfg.num_prompts == 100:
            return list(bgs[1:] * 10)
        return list([""] + bgs[1:] * 2)
    if cfg.add_age:
        return list(AGE_PHASES * 6)
    return list([""] * cfg.num_prompts)


@dataclass
class WorkItem:
    id_number: int            # index of the identity in natural-sorted order = generator seed (:111)
    which_id: str
    model_name: str
    prompt_index: int         # i
    sample_index: int         # j
    prompt: str
    stream_offset: int        # how many pipeline calls precede this one on the (identity, model) generator

    def file_name(self) -> str:
        return f"{self.prompt_index}_{self.sample_index}_{self.prompt}.png"


def build_work_list(ids: Sequence[str], genders: Optional[Dict[str, str]], cfg: PolicyConfig = PolicyConfig()) -> List[WorkItem]:
    """The exact prompt stream of the reference loop (:86-138), consuming python's global-style RNG in the
    same order: one ``random.sample`` per identity, one ``random.choice`` per (model, prompt) when add_pose."""
    rng = random.Random()
    rng.seed(cfg.seed)                                   # set_seed(seed) -> random.seed(seed)
    combos = prompt_combinations(cfg)
    ids = sorted([i for i in ids if ".json" not in i], key=natural_keys)
    items: List[WorkItem] = []
    for id_number, which_id in enumerate(ids):
        gender = None
        if cfg.add_gender:
            if genders is None or which_id not in genders:
                raise KeyError(f"gender of {which_id!r} missing (the reference reads tufts_gender_dict.json, :76-78)")
            gender = {"M": "male", "F": "female"}.get(genders[which_id], genders[which_id])
        all_prompts_for_id = rng.sample(combos, cfg.num_prompts)
        for model_name in cfg.models_to_test:
            call = 0
            for i in range(cfg.num_prompts):
                additions = all_prompts_for_id[i]
                prompt = ORIGINAL_PROMPT
                if cfg.add_age:
                    if isinstance(additions, str):
                        age_insert = additions
                    else:
                        age_insert = additions[0]
                        additions = additions[1:]
                    if age_insert != "":
                        prompt = prompt.replace(" sks person", f" {age_insert} sks person")
                if cfg.add_gender:
                    prompt = prompt.replace(" sks person", f" {gender} sks person")
                if cfg.add_pose and rng.choice([True, False]):
                    prompt = prompt.replace("portrait", "side-portrait")
                if cfg.add_background:
                    if isinstance(additions, str):
                        prompt += f", {additions}"                      # trailing ", " when the entry is "" (:129-130)
                    else:
                        for addition in additions:
                            if addition != "":
                                prompt += f", {addition}"
                for j in range(cfg.num_samples_per_prompt):
                    items.append(WorkItem(id_number, which_id, model_name, i, j, prompt, call))
                    call += 1
    return items


def draw_noise_sequential(seed: int, n_calls: int, steps: int, latent_shape: Tuple[int, int, int],
                          first_call: int = 0) -> torch.Tensor:
    """Noise for ``n_calls`` consecutive batch-1 pipeline calls that share one generator (:111 is outside the
    prompt loop): call k draws its initial latents then one tensor per step.  Returns [steps+1, n_calls, C, h, w],
    i.e. the layout the batched sampler consumes, with the reference's per-call stream order preserved.
    ``first_call`` skips the draws of earlier calls (work split across batches)."""
    g = torch.Generator().manual_seed(seed)
    c, h, w = latent_shape
    out = torch.empty((steps + 1, n_calls, c, h, w), dtype=torch.float32)
    for k in range(first_call + n_calls):
        for s in range(steps + 1):
            t = torch.randn((1, c, h, w), generator=g, dtype=torch.float32)
            if k >= first_call:
                out[s, k - first_call] = t[0]
    return out


# ---------------------------------------------------------------------------------------------------
# sharding + the one collective
# ---------------------------------------------------------------------------------------------------
def shard_identities(n_ids: int, rank: int, world: int) -> List[int]:
    """Identity indices owned by `rank` (round-robin: keeps one LoRA set resident per GPU per sub-batch)."""
    return list(range(rank, n_ids, world))


def shard_work(items: Sequence[WorkItem], rank: int, world: int) -> List[WorkItem]:
    return [it for it in items if it.id_number % world == rank]


def all_gather_images(local: torch.Tensor, counts: Sequence[int], group=None, force: bool = False) -> torch.Tensor:
    """One all-gather of uint8 images.  `local` is [n_local, H, W, 3]; `counts[r]` is rank r's true count.
    Ranks pad to max(counts) (equal counts are required by the collective); the pad is dropped afterwards.
    Returns [sum(counts), H, W, 3] in rank order on every rank."""
    import torch.distributed as dist
    world = len(counts)
    if (world == 1 and not force) or not (dist.is_available() and dist.is_initialized()):
        return local                                    # force: run the collective at world size 1 too (plumbing test)
    cap = max(counts)
    shape = (cap,) + tuple(local.shape[1:])
    buf = torch.zeros(shape, dtype=torch.uint8, device=local.device)
    buf[: local.shape[0]] = local
    gathered = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    parts = [gathered[r * cap: r * cap + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def synthetic_embed_fn(cross_dim: int, n_ctx: int = 77) -> Callable[[Sequence[str]], torch.Tensor]:
    """Stand-in for the CLIP text encoder (next row): a deterministic N(0,1) embedding per
    prompt string (real last_hidden_state is O(1) after the final LayerNorm)."""
    import zlib

    def fn(prompts: Sequence[str]) -> torch.Tensor:
        out = []
        for p in prompts:
            g = torch.Generator().manual_seed(zlib.crc32(p.encode("utf-8")))
            out.append(torch.randn(n_ctx, cross_dim, generator=g))
        return torch.stack(out)
    return fn


def plan_calls(items: Sequence[WorkItem], max_batch: int = 64, group_identities: int = 1) -> List[List[List[WorkItem]]]:
    """The pipeline calls one rank makes for its `items`: a list of calls, each a list of per-identity item lists (one entry unless
    identities are grouped).  Items of one (identity, model) pair keep the reference's generator stream order and are cut into
    sub-batches of at most `max_batch`.  group_identities = G > 1: sub-batches of the SAME model and the same length from up to G
    different identities share one call (one merged LoRA set per group of the batch, `load_lora_weights([...])`), as long as
    G * length <= max_batch — BASELINE configs[2]'s 8 identities x 8 prompts become one batch-64 call."""
    pairs: Dict[Tuple[int, str], List[WorkItem]] = {}
    for it in items:
        pairs.setdefault((it.id_number, it.model_name), []).append(it)
    subs: List[List[WorkItem]] = []
    for group_items in pairs.values():
        for b0 in range(0, len(group_items), max_batch):
            sub = group_items[b0:b0 + max_batch]
            assert [it.stream_offset for it in sub] == list(range(sub[0].stream_offset, sub[0].stream_offset + len(sub)))
            subs.append(sub)
    if group_identities <= 1:
        return [[sub] for sub in subs]
    calls: List[List[List[WorkItem]]] = []
    pending: Dict[Tuple[str, int, int], List[List[WorkItem]]] = {}        # (model, length, first stream offset) -> sub-batches waiting
    for sub in subs:
        key = (sub[0].model_name, len(sub), sub[0].stream_offset)
        grp = pending.setdefault(key, [])
        grp.append(sub)
        if len(grp) == group_identities or (len(grp) + 1) * len(sub) > max_batch:
            calls.append(pending.pop(key))
    calls += [g for g in pending.values()]
    return calls


def generate(pipe, items: Sequence[WorkItem], embed_fn: Callable, cfg: PolicyConfig = PolicyConfig(),
             lora_for: Optional[Callable[[str, str], object]] = None, rank: int = 0, world: int = 1,
             max_batch: int = 64, group=None, group_identities: int = 1) -> Tuple[torch.Tensor, List[WorkItem]]:
    """Run this rank's share of `items` and all-gather the uint8 images.  Items of one (identity, model) pair are
    batched into single pipeline calls (B up to `max_batch`) with the reference's generator stream order; with
    group_identities = G > 1 (and `lora_for`) up to G identities of the same model share one call, each group of the batch with
    its own merged LoRA set (``plan_calls``).  `lora_for(model_name, which_id)` returns what ``load_lora_weights`` accepts (path
    or state dict).  Returns (images [len(items), H, W, 3] uint8 in rank-major work order, the items in that order)."""
    mine = shard_work(items, rank, world)
    order: List[WorkItem] = []
    chunks: List[torch.Tensor] = []
    neg = NEGATIVE_PROMPT if not cfg.do_not_use_negative_prompt else ""
    lat_shape = (pipe.unet_config.in_channels, cfg.height // pipe.vae_scale_factor, cfg.width // pipe.vae_scale_factor)
    G = group_identities if lora_for is not None else 1
    for call in plan_calls(mine, max_batch, G):
        if lora_for is not None:
            sets = [lora_for(sub[0].model_name, sub[0].which_id) for sub in call]
            pipe.load_lora_weights(sets if len(sets) > 1 else sets[0])
        noise = torch.cat([draw_noise_sequential(sub[0].id_number, len(sub), cfg.num_inference_steps, lat_shape, sub[0].stream_offset)
                           for sub in call], dim=1)
        flat = [it for sub in call for it in sub]
        pe = embed_fn([it.prompt for it in flat])
        ne = embed_fn([neg] * len(flat))
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=cfg.num_inference_steps,
                   guidance_scale=cfg.guidance_scale, height=cfg.height, width=cfg.width, output_type="uint8", noise=noise)
        chunks.append(out.images)
        order += flat
    if chunks:
        local = torch.cat(chunks)
    else:
        local = torch.empty((0, cfg.height, cfg.width, 3), dtype=torch.uint8, device=pipe.device)
    counts = [len(shard_work(items, r, world)) for r in range(world)]
    images = all_gather_images(local, counts, group)
    all_order: List[WorkItem] = []
    for r in range(world):
        if r == rank:
            all_order += order
        else:
            all_order += [it for call in plan_calls(shard_work(items, r, world), max_batch, G) for sub in call for it in sub]
    return images, all_order


# ---------------------------------------------------------------------------------------------------
# sink (inference_ID-Booth.py:139-156; torchvision.utils.save_image / make_grid(nrow, padding=0))
# ---------------------------------------------------------------------------------------------------
def _write_png(job) -> str:
    from PIL import Image
    arr, pth = job
    Image.fromarray(arr).save(pth)
    return pth


def save_outputs(images_u8: torch.Tensor, items: Sequence[WorkItem], root: str, cfg: PolicyConfig = PolicyConfig(),
                 workers: Optional[int] = None) -> List[str]:
    """The sink of inference_ID-Booth.py:139-156: one PNG per sample (same folder and file names) and one comparison JPG per
    identity.  PNG encoding (zlib: ~10-30 ms per 512x512 image on one core, the bottleneck above ~50 images/s) runs on a thread
    pool — PIL's encoders release the GIL — so that the host keeps up with the GPU; `workers` = 1 is the sequential reference."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    paths = []
    arr = images_u8.cpu().numpy()
    by_id: Dict[str, List[int]] = {}
    jobs = []
    for k, it in enumerate(items):
        d = os.path.join(root, cfg.output_folder(), it.model_name, f"{it.which_id}_{cfg.checkpoint}_{ARCH}")
        os.makedirs(d, exist_ok=True)
        pth = os.path.join(d, it.file_name())
        jobs.append((arr[k], pth))
        paths.append(pth)
        by_id.setdefault(it.which_id, []).append(k)
    if workers is None:
        try:
            workers = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            workers = min(16, os.cpu_count() or 1)
    if workers <= 1 or len(jobs) <= 1:
        for j in jobs:
            _write_png(j)
    else:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(_write_png, jobs))
    nrow = cfg.num_prompts * cfg.num_samples_per_prompt
    for which_id, idx in by_id.items():
        comp = os.path.join(root, cfg.output_folder(), "Comparison")
        os.makedirs(comp, exist_ok=True)
        tiles = arr[idx]
        rows = [tiles[r:r + nrow] for r in range(0, len(tiles), nrow)]
        h, w = tiles.shape[1:3]
        sheet = Image.new("RGB", (w * nrow, h * len(rows)))
        for r, row in enumerate(rows):
            for c, tile in enumerate(row):
                sheet.paste(Image.fromarray(tile), (c * w, r * h))
        pth = os.path.join(comp, f"{which_id}_{cfg.checkpoint}_{ARCH}_{cfg.guidance_scale}.jpg")
        sheet.save(pth)
        paths.append(pth)
    return paths
