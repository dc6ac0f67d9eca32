"""Headline parity: BASELINE configs[1] — SD-2.1-base shapes + rank-4 LoRA, batch 1, 64x64 latents, 30 DDPM steps, CFG 5.0 —
against the committed oracle trajectory tests/golden/sd21_config1.npz (tests/golden/make_golden.py config1), plus the
per-block error attribution of one CFG forward against the oracle's block taps.

What is compared (measured values are printed, recorded in DESIGN.md section 2, and bounded at <= 1.5x the measurement):
  * free-running: latents after each of the 30 steps and the final latents (max-abs, rel-RMS), decoded uint8 image (PSNR);
  * teacher-forced: eps (uncond, cond) of the steps stored in the fixture, the golden latents of the step before as input;
  * per block: output of every resnet / transformer / down- / up-sample block of the step-0 CFG forward, HIP engine vs the
    fp32 oracle run here on the host CPU, next to the SAME comparison for the oracle with the engine's rounding points
    emulated (oracle.ROUND): the engine must sit in the error class of its operand dtype, block by block.

north_star states "latents within 1e-2 max-abs of the CPU fp32 reference".  Round 3: the fixture is made with CALIBRATED synthetic
weights (weights.calibrate_unet: eps = x_t + a network-dependent correction, as a trained eps-model's output is), so the latents
stay O(1) over the 30 steps (std 1.0 -> 0.4, max < 5; tests/test_oracle_cpu.py asserts it) and the ABSOLUTE figures below can be
read against the 1e-2 bound.  Rounds 1-2 used plain N(0, 1/fan_in) weights, whose chain blows the latents up to std 20; their
figures (f16 0.157, bf16 1.21 max-abs) are in DESIGN.md's round history.  The bounds below are 1.5x the values measured on MI355X.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_NAMES = ["conv_in.weight", "mid_block.resnets.0.conv1.weight",
            "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_k.weight"]

# (final latents max-abs, final latents rel-RMS, teacher-forced eps rel-RMS, min PSNR dB of the decoded uint8 image)
# = 1.5x (PSNR: -2 dB) the values measured on MI355X with the calibrated fixture: f16 3.01e-3 / 1.34e-3 /
# 5.98e-4 / 57.9 dB (emulated 16-bit-storage oracle: 2.66e-3 / 1.24e-3), bf16 1.86e-2 / 9.89e-3 / 4.84e-3 / 46.5 dB (emulated 2.19e-2)
BOUNDS = {"f16": (4.6e-3, 2.1e-3, 9.0e-4, 55.9), "bf16": (2.8e-2, 1.5e-2, 7.3e-3, 44.4)}
NORTH_STAR_MAX_ABS = 1e-2          # "latents within 1e-2 max-abs of CPU fp32 reference": asserted for the default dtype (f16)
# per-block rel-RMS error of the step-0 forward: the engine must stay within 1.25x the emulated-rounding oracle's error of the
# same block (measured: 0.995-1.013x on all 45 block outputs, both dtypes)
BLOCK_FACTOR = 1.25


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())), float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def cfg1(lib):
    from faceposegenerator_amd import spec as S, weights as W
    path = os.path.join(GOLD, "sd21_config1.npz")
    gold = np.load(path)
    useed, vseed, lseed, batch, side, steps, eseed, nseed, calibrated = gold["meta"].tolist()
    assert calibrated == 1, "sd21_config1.npz must be the calibrated-weights fixture (tests/golden/make_golden.py config1)"
    usd, vsd = W.synth_unet(S.SD21_UNET, useed, calibrated=True), W.synth_vae(S.SD21_VAE, vseed)
    assert np.allclose([float(usd[n].double().sum()) for n in FP_NAMES + ["conv_out.weight", "conv_in.weight"]], gold["unet_fingerprint"], rtol=0, atol=1e-7), \
        "synthetic weights differ from the ones the golden vectors were made with"
    lora_raw = W.synth_lora(S.SD21_UNET, lseed)
    g = torch.Generator().manual_seed(eseed)
    pe = torch.randn(batch, 77, 1024, generator=g)
    ne = torch.randn(batch, 77, 1024, generator=g)
    gen = torch.Generator().manual_seed(nseed)
    noise = torch.stack([torch.randn((batch, 4, side, side), generator=gen, dtype=torch.float32) for _ in range(steps + 1)])
    assert np.array_equal(noise.flatten()[:4].numpy(), gold["noise_first4"])
    return dict(gold=gold, usd=usd, vsd=vsd, lora_raw=lora_raw, pe=pe, ne=ne, noise=noise, steps=steps, side=side)


def _pipe(c, dtype):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    pipe = StableDiffusionPipeline(S.SD21_UNET, S.SD21_VAE, c["usd"], c["vsd"], torch_dtype=dtype).to(DEV)
    pipe.load_lora_weights(c["lora_raw"])
    return pipe


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_config1_30_steps_against_golden(cfg1, dtype):
    c, gold = cfg1, cfg1["gold"]
    b_max, b_rel, b_eps, b_psnr = BOUNDS[dtype]
    pipe = _pipe(c, dtype)
    eng = pipe._engine()
    steps, side = c["steps"], c["side"]
    ref_steps = gold["latents_per_step"]                     # [30, 1, 4, 64, 64]
    # ---- free-running, eager with a per-step trace
    sch = pipe.scheduler
    sch.set_timesteps(steps)
    ts = sch.timesteps.tolist()
    assert ts == gold["timesteps"].tolist()
    coefs = torch.tensor([list(sch.step_coefficients(t)) + [5.0] for t in ts], dtype=torch.float32)
    trace = []
    lat = eng.sample(c["pe"], c["ne"], c["noise"].to(DEV), ts, coefs.to(DEV), use_graph=False, trace=trace)
    print(f"[{dtype}] configs[1] free-running, per step (latents rel-RMS / max-abs vs the fp32 oracle; |ref| std):")
    for i in (0, 1, 2, 4, 9, 14, 19, 24, 27, 28, 29):
        r, m = _rel(trace[i][1].cpu().numpy(), ref_steps[i])
        print(f"    step {i:2d} t={ts[i]:3d}: {r:.3e} / {m:.3e}   ({ref_steps[i].std():.2f})")
    r, m = _rel(lat.cpu().numpy(), gold["final_latents"])
    sd = float(gold["final_latents"].std())
    print(f"[{dtype}] configs[1] FINAL latents after 30 steps: max-abs {m:.4e}  rel-RMS {r:.4e}  (|ref| std {sd:.3f}, "
          f"max-abs / std = {m / sd:.3e}; north_star bound 1e-2 max-abs: {'met' if m <= 1e-2 else 'NOT met'})")
    assert m < b_max and r < b_rel
    if dtype == "f16":
        assert m <= NORTH_STAR_MAX_ABS, "the default dtype must meet north_star's absolute bound on the calibrated network"
        assert max(_rel(trace[i][1].cpu().numpy(), ref_steps[i])[1] for i in range(steps)) <= NORTH_STAR_MAX_ABS      # at EVERY step
    # ---- the graph-replayed product path gives the same latents as the eager trace
    out = pipe(prompt_embeds=c["pe"], negative_prompt_embeds=c["ne"], num_inference_steps=steps, guidance_scale=5.0,
               height=side * 8, width=side * 8, output_type="latent", noise=c["noise"])
    assert torch.equal(out.images, lat), "HIP-graph replay differs from the eager loop"
    # ---- decoded image of the free-running latents vs the oracle's image
    _, u8 = eng.decode_images(lat)
    d = u8.cpu().numpy().astype(np.float64) - gold["image_u8"].astype(np.float64)
    mse = float((d ** 2).mean())
    psnr = 10 * math.log10(255.0 ** 2 / mse) if mse > 0 else float("inf")
    print(f"[{dtype}] configs[1] decoded 512x512 image: PSNR {psnr:.2f} dB, uint8 max diff {int(np.abs(d).max())}, "
          f"{100 * float((np.abs(d) <= 2).mean()):.2f} % of pixels within 2 levels")
    assert psnr > b_psnr
    # ---- teacher-forced eps at the stored steps
    worst = 0.0
    for j, i in enumerate(gold["eps_steps"].tolist()):
        x_in = c["noise"][0] if i == 0 else torch.from_numpy(ref_steps[i - 1])
        eps = pipe.unet(torch.cat([x_in, x_in]).to(DEV), ts[i], torch.cat([c["ne"], c["pe"]]).to(DEV), return_dict=False)[0].cpu()
        r_u, m_u = _rel(eps[0:1].numpy(), gold["eps_uncond"][j])
        r_c, m_c = _rel(eps[1:2].numpy(), gold["eps_cond"][j])
        worst = max(worst, r_u, r_c)
        print(f"[{dtype}] teacher-forced step {i:2d} t={ts[i]:3d}: eps rel-RMS uncond {r_u:.3e} cond {r_c:.3e}  max-abs {max(m_u, m_c):.3e}")
    assert worst < b_eps
    del pipe, eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_per_block_error_attribution(cfg1, dtype):
    """One CFG forward (step 0 of configs[1]): every block's output, engine vs fp32 oracle, beside the oracle with the
    engine's rounding points emulated on the CPU.  ~3 oracle forwards on the host cores."""
    from faceposegenerator_amd import spec as S
    from oracle import sd21_oracle as O
    c = cfg1
    ucfg = S.SD21_UNET
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    x0 = c["noise"][0]
    xin, ctx = torch.cat([x0, x0]), torch.cat([c["ne"], c["pe"]])
    t = int(c["gold"]["timesteps"][0])
    merged = O.merge_lora(c["usd"], O.normalize_lora_keys(c["lora_raw"]))
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    with torch.no_grad():
        ref = {"__blocks__": True}
        eps_ref = O.unet_forward(merged, ucfg, xin, t, ctx, taps=ref)
        wsd = {k: (v.to(tdt).float() if (v.ndim >= 2 and not k.startswith(("conv_in.", "time_embedding.")) and ".time_emb_proj." not in k)
                   else v) for k, v in merged.items()}
        emu = {"__blocks__": True}
        O.ROUND = lambda kind, z: z.to(tdt).float()
        try:
            eps_emu = O.unet_forward(wsd, ucfg, xin, t, ctx, taps=emu)
        finally:
            O.ROUND = None
    pipe = _pipe(c, dtype)
    eng = pipe._engine()
    eng.taps = {}
    eps = pipe.unet(xin.to(DEV), t, ctx.to(DEV), return_dict=False)[0].cpu()
    taps, eng.taps = eng.taps, None
    print(f"[{dtype}] step-0 CFG forward, rel-RMS error per block output: HIP engine | emulated-rounding oracle")
    bad = []
    for name in [k for k in ref if k not in ("__blocks__", "temb", "down_out", "mid_out", "up_out")]:
        r = ref[name]                                            # NCHW fp32
        got = taps[name].cpu().reshape(r.shape[0], r.shape[2], r.shape[3], r.shape[1]).permute(0, 3, 1, 2)
        e_gpu = float((got - r).norm() / r.norm())
        e_emu = float((emu[name] - r).norm() / r.norm())
        print(f"    {name:45s} {e_gpu:.3e} | {e_emu:.3e}")
        if e_gpu > BLOCK_FACTOR * e_emu + 1e-5:
            bad.append((name, e_gpu, e_emu))
    e_gpu, e_emu = float((eps - eps_ref).norm() / eps_ref.norm()), float((eps_emu - eps_ref).norm() / eps_ref.norm())
    print(f"    {'eps (conv_out)':45s} {e_gpu:.3e} | {e_emu:.3e}")
    assert not bad, f"blocks outside {BLOCK_FACTOR}x the operand-dtype error class: {bad}"
    assert e_gpu < BLOCK_FACTOR * e_emu
    del pipe, eng
    torch.cuda.empty_cache()
