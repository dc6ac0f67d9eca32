"""CPU tests (-m "not gpu"): the oracle against the known answers and structural invariants of SURVEY.md
Appendix C/E/F, the committed golden vectors, and the host-side mirrors (scheduler, weights I/O, pipeline
argument checks).  No GPU compute happens here."""
import math
import os

import numpy as np
import pytest
import torch

from faceposegenerator_amd import spec as S, weights as W
from faceposegenerator_amd.scheduler import DDPMScheduler
from oracle import sd21_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- Appendix F: structural invariants --------------------------------------------------------------
def test_published_parameter_counts():
    assert S.count_params(S.unet_param_shapes()) == 865_910_724
    assert S.count_params(S.vae_decoder_param_shapes()) == 49_490_199
    lora = S.lora_param_shapes()
    assert S.count_params(lora) == 829_952 and len(lora) == 256


def test_algorithmic_work_matches_baseline_md():
    assert S.unet_macs() == 402_128_732_160
    assert S.vae_decode_macs() == 1_257_259_466_752
    per_image = 60 * 2 * S.unet_macs() + 2 * S.vae_decode_macs()
    assert abs(per_image / 1e12 - 50.77) < 0.01
    assert abs(2 * S.unet_macs(latent_side=96) / 1e9 - 2149.11) < 0.5       # 768x768 config


def test_unet_graph_shape_invariants():
    g = S.unet_graph()
    assert g.skip_channels == [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
    assert [64 // d for d in g.skip_side_div] == [64, 64, 64, 32, 32, 32, 16, 16, 16, 8, 8, 8]
    resnets = [r for b in g.down + g.up for r in b["resnets"]] + g.mid["resnets"]
    assert len(resnets) == 22 and sum(r.cin != r.cout for r in resnets) == 14
    assert len(S.unet_attention_modules()) == 16
    assert sum(1 for b in g.down if b["down"]) == 3 and sum(1 for b in g.up if b["up"]) == 3
    up_in = [r.cin for b in g.up for r in b["resnets"]]
    assert up_in == [2560, 2560, 2560, 2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]


# ---- Appendix C: scheduler / embedding known answers -----------------------------------------------
def test_scheduler_known_answers():
    ac = O.ddpm_tables()
    for idx, val in ((0, 0.99914998), (1, 0.99829602), (500, 0.27633247), (999, 0.00466010)):
        assert abs(ac[idx].item() - val) < 5e-8
    assert O.ddpm_timesteps(4) == [751, 501, 251, 1]
    ts30 = O.ddpm_timesteps(30)
    assert ts30 == [1 + 33 * k for k in range(29, -1, -1)]
    kat4 = {751: (0.05571898, 0.44282490, 0.34559989, 0.78244293), 501: (0.27499884, 0.66816682, 0.28924572, 0.51690716),
            251: (0.67215115, 0.99565125, 0.00426475, 0.04120697), 1: (0.99829602, 1.0, 0.0, 1.0e-10)}
    for t, want in kat4.items():
        got = O.ddpm_coefficients(ac, O.ddpm_timesteps(4), t)
        assert all(abs(a - b) < 2e-7 for a, b in zip(got, want)), (t, got)
    got = O.ddpm_coefficients(ac, ts30, 958)
    assert all(abs(a - b) < 2e-7 for a, b in zip(got, (0.00753477, 0.03215177, 0.83015662, 0.55242407)))


def test_sinusoid_and_rng_known_answers():
    te = O.timestep_embedding(torch.tensor([958]), 320)[0]
    want = [-0.98279631, 0.93290263, 0.99485564, 0.18469287, -0.36012876, 0.10130244]
    assert all(abs(te[i].item() - w) < 2e-6 for i, w in zip([0, 1, 159, 160, 161, 319], want))
    g = torch.Generator().manual_seed(0)
    first = torch.randn((1, 4, 64, 64), generator=g).flatten()[:4]
    assert torch.allclose(first, torch.tensor([-1.12583983, -1.15236020, -0.25057858, -0.43387881]), atol=1e-7)


def test_rng_draw_order_is_latents_then_one_per_step():
    n = O.draw_noise(torch.Generator().manual_seed(3), 2, 30, (8, 8))
    assert n.shape == (31, 2, 4, 8, 8)
    g = torch.Generator().manual_seed(3)
    for i in range(31):
        assert torch.equal(n[i], torch.randn((2, 4, 8, 8), generator=g))


# ---- host mirror of the scheduler vs the oracle ----------------------------------------------------
def test_host_scheduler_matches_oracle():
    sch = DDPMScheduler()
    assert sch.init_noise_sigma == 1.0 and sch.config.prediction_type == "epsilon" and len(sch) == 1000
    for n in (4, 30, 50):
        sch.set_timesteps(n)
        assert sch.timesteps.tolist() == O.ddpm_timesteps(n)
    sch.set_timesteps(30)
    ac, ts = O.ddpm_tables(), O.ddpm_timesteps(30)
    g = torch.Generator().manual_seed(1)
    x, e, nz = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    for t in (958, 463, 1):
        ref_prev, ref_x0 = O.ddpm_step(ac, ts, t, e, x, nz)
        out = sch.step(e, t, x, variance_noise=nz)
        assert torch.allclose(out.prev_sample, ref_prev, atol=2e-6, rtol=1e-6)
        assert torch.allclose(out.pred_original_sample, ref_x0, atol=2e-5, rtol=1e-6)
        assert sch.step(e, t, x, variance_noise=nz, return_dict=False)[0].shape == x.shape
    # generator-driven noise: same stream as torch.randn on that generator
    out = sch.step(e, 958, x, generator=torch.Generator().manual_seed(9))
    ref, _ = O.ddpm_step(ac, ts, 958, e, x, torch.randn(x.shape, generator=torch.Generator().manual_seed(9)))
    assert torch.allclose(out.prev_sample, ref, atol=2e-6)
    # v-prediction branch (BASELINE config 5)
    sv = DDPMScheduler(S.SchedulerConfig(prediction_type="v_prediction"))
    sv.set_timesteps(30)
    ref_prev, _ = O.ddpm_step(ac, ts, 463, e, x, nz, "v_prediction")
    assert torch.allclose(sv.step(e, 463, x, variance_noise=nz).prev_sample, ref_prev, atol=2e-6)
    # training-side helpers
    t = torch.tensor([10, 500])
    noisy = sch.add_noise(x, nz, t)
    assert torch.allclose(noisy[1], ac[500] ** 0.5 * x[1] + (1 - ac[500]) ** 0.5 * nz[1])
    vel = sch.get_velocity(x, nz, t)
    assert torch.allclose(vel[0], ac[10] ** 0.5 * nz[0] - (1 - ac[10]) ** 0.5 * x[0])
    with pytest.raises(ValueError):
        DDPMScheduler(S.SchedulerConfig(beta_schedule="linear"))


# ---- oracle self-consistency -------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny():
    return W.synth_unet(S.TINY_UNET, 7), W.synth_vae(S.TINY_VAE, 8), W.synth_lora(S.TINY_UNET, 3)


def test_lora_merged_equals_unmerged_and_both_dialects(tiny):
    usd, _, lora_raw = tiny
    g = torch.Generator().manual_seed(1)
    x, ctx = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 77, 128, generator=g)
    lora = O.normalize_lora_keys(lora_raw)
    assert len(lora) == 256 and all(k.endswith((".lora_A.weight", ".lora_B.weight")) for k in lora)
    peft = O.normalize_lora_keys(W.synth_lora(S.TINY_UNET, 3, dialect="peft"))
    assert peft.keys() == lora.keys() and all(torch.equal(peft[k], lora[k]) for k in lora)
    assert W.normalize_lora_keys(lora_raw).keys() == lora.keys()
    with torch.no_grad():
        a = O.unet_forward(usd, S.TINY_UNET, x, 500, ctx, lora)
        b = O.unet_forward(O.merge_lora(usd, lora), S.TINY_UNET, x, 500, ctx, None)
        c = O.unet_forward(usd, S.TINY_UNET, x, 500, ctx, None)
    assert (a - b).abs().max() < 2e-5 and (a - c).abs().max() > 1e-3


def test_oracle_is_deterministic_and_batch_consistent(tiny):
    usd, _, _ = tiny
    g = torch.Generator().manual_seed(2)
    x, ctx = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 77, 128, generator=g)
    with torch.no_grad():
        a = O.unet_forward(usd, S.TINY_UNET, x, torch.tensor([900, 20]), ctx)
        b = O.unet_forward(usd, S.TINY_UNET, x, torch.tensor([900, 20]), ctx)
        c = O.unet_forward(usd, S.TINY_UNET, x[1:], 20, ctx[1:])
    assert torch.equal(a, b) and (a[1:] - c).abs().max() < 1e-5


def test_postprocess_and_uint8_quantisation():
    x = torch.tensor([[[[-3.0, -1.0, 0.0, 0.999, 1.0, 7.0]]]])
    img = O.postprocess_np(x)
    assert torch.allclose(img.flatten(), torch.tensor([0.0, 0.0, 0.5, 0.9995, 1.0, 1.0]))
    assert O.to_uint8(img.clone()).flatten().tolist() == [0, 0, 128, 255, 255, 255]


# ---- committed golden vectors -------------------------------------------------------------------------
def test_oracle_reproduces_tiny_golden_trajectory(tiny):
    usd, vsd, lora_raw = tiny
    gold = np.load(os.path.join(GOLD, "tiny_trajectory.npz"))
    useed, vseed, lseed, batch, side, steps, eseed, nseed = gold["meta"].tolist()
    assert (useed, vseed, lseed) == (7, 8, 3)
    names = ["conv_in.weight", "mid_block.resnets.0.conv1.weight",
             "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_k.weight"]
    assert np.allclose([float(usd[n].double().sum()) for n in names], gold["unet_fingerprint"], rtol=0, atol=1e-9)
    g = torch.Generator().manual_seed(eseed)
    pe = torch.randn(batch, 77, 128, generator=g)
    ne = torch.randn(batch, 77, 128, generator=g)
    noise = O.draw_noise(torch.Generator().manual_seed(nseed), batch, steps, (side, side))
    assert np.array_equal(noise.flatten()[:4].numpy(), gold["noise_first4"])
    trace = []
    with torch.no_grad():
        lat = O.sample(usd, S.TINY_UNET, pe, ne, noise, steps, float(gold["guidance_scale"]),
                       lora=O.normalize_lora_keys(lora_raw), trace=trace)
        img = O.decode_to_images(vsd, S.TINY_VAE, lat)
    assert gold["timesteps"].tolist() == [751, 501, 251, 1]
    # same machine/torch build reproduces bit-for-bit; allow fp32 reassociation noise across CPUs
    assert np.abs(lat.numpy() - gold["final_latents"]).max() < 5e-4
    assert np.abs(torch.stack([t[2] for t in trace]).numpy() - gold["latents_per_step"]).max() < 5e-4
    assert np.abs(O.to_uint8(img.clone()).numpy().astype(int) - gold["image_u8"].astype(int)).max() <= 1


def test_full_size_golden_fixture_is_wellformed():
    gold = np.load(os.path.join(GOLD, "sd21_config0.npz"))
    assert gold["final_latents"].shape == (1, 4, 64, 64) and gold["latents_per_step"].shape == (4, 1, 4, 64, 64)
    assert gold["image_u8"].shape == (1, 512, 512, 3) and gold["image_u8"].dtype == np.uint8
    assert gold["timesteps"].tolist() == [751, 501, 251, 1]
    assert np.isfinite(gold["final_latents"]).all() and 0.3 < gold["final_latents"].std() < 10


# ---- weights I/O and pipeline argument checks (host logic) --------------------------------------------
def test_model_dir_roundtrip_and_lora_files(tmp_path, tiny):
    usd, vsd, lora_raw = tiny
    root = str(tmp_path / "model")
    W.save_model_dir(root, usd, vsd, S.TINY_UNET, S.TINY_VAE)
    assert W.load_unet_config(root).block_out_channels == S.TINY_UNET.block_out_channels
    assert W.load_unet_config(root).num_heads == S.TINY_UNET.num_heads
    assert W.load_vae_config(root).block_out_channels == S.TINY_VAE.block_out_channels
    assert W.load_scheduler_config(root).steps_offset == 1
    back = W.load_unet_weights(root)
    assert back.keys() == usd.keys() and all(torch.equal(back[k], usd[k]) for k in usd)
    vback = W.load_vae_decoder_weights(root)
    assert vback.keys() == vsd.keys()
    sch = DDPMScheduler.from_pretrained(root, subfolder="scheduler")
    sch.set_timesteps(30)
    assert sch.timesteps[0].item() == 958
    # LoRA checkpoint directory in the layout the reference loads (inference_ID-Booth.py:98,107)
    ck = str(tmp_path / "Trained_LoRA_Models" / "ID-Booth" / "ID_1" / "checkpoint-31-6400")
    W.save_lora(ck, lora_raw)
    tensors, alphas = W.load_lora(ck)
    assert alphas == {} and len(tensors) == 256
    # legacy VAE attention key names (SD-2.x hub files)
    legacy = {}
    for k, v in vsd.items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
        legacy[k] = v
    from safetensors.torch import save_file
    save_file(legacy, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    assert W.load_vae_decoder_weights(root).keys() == vsd.keys()
    with pytest.raises(FileNotFoundError):
        W.load_unet_weights(str(tmp_path / "nope"))


def test_pipeline_argument_checks_without_gpu(tmp_path, tiny):
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    usd, vsd, _ = tiny
    with pytest.raises(FileNotFoundError):
        StableDiffusionPipeline.from_pretrained("stabilityai/stable-diffusion-2-1-base")
    root = str(tmp_path / "m")
    W.save_model_dir(root, usd, vsd, S.TINY_UNET, S.TINY_VAE)
    pipe = StableDiffusionPipeline.from_pretrained(root, torch_dtype=torch.float16)
    assert pipe.dtype_name == "f16" and pipe.vae.config.scaling_factor == 0.18215 and pipe.vae_scale_factor == 8
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler = DDPMScheduler.from_pretrained(root, subfolder="scheduler")
    pe = torch.zeros(1, 77, 128)
    with pytest.raises(ValueError):
        pipe.check_inputs(None, 100, 128, None, pe, pe)
    with pytest.raises(ValueError):
        pipe.check_inputs("a", 128, 128, None, pe, None)
    with pytest.raises(ValueError):
        pipe.check_inputs(None, 128, 128, None, None, None)
    with pytest.raises(ValueError):
        pipe.check_inputs(None, 128, 128, None, pe, torch.zeros(1, 76, 128))
    with pytest.raises(ValueError):
        StableDiffusionPipeline(S.TINY_UNET, S.TINY_VAE, usd, vsd, torch_dtype=torch.float32)
    with pytest.raises(FileNotFoundError):
        pipe.load_lora_weights(str(tmp_path / "missing"))
    n = pipe.prepare_noise(2, 3, 128, 128, torch.Generator().manual_seed(4))
    assert n.shape == (4, 2, 4, 16, 16) and torch.equal(n, O.draw_noise(torch.Generator().manual_seed(4), 2, 3, (16, 16)))
    with pytest.raises(ValueError):
        pipe.prepare_noise(2, 3, 128, 128, [torch.Generator().manual_seed(1)])


def test_vae_encoder_spec_and_oracle():
    """AutoencoderKL.encode.  Parameter counts (encoder 34,163,592 + quant_conv 72; the whole SD-2.1
    AutoencoderKL is 83,653,863), output geometry, the bottom/right padding of the stride-2 convs, and the sample/mode tail."""
    import torch.nn.functional as F
    from faceposegenerator_amd import spec as S, weights as W
    from oracle import sd21_oracle as O
    enc = S.count_params(S.vae_encoder_param_shapes(S.SD21_VAE))
    assert enc == 34_163_664 and enc + S.count_params(S.vae_decoder_param_shapes(S.SD21_VAE)) == 83_653_863
    cfg = S.TINY_VAE
    sd = W.synth_vae_encoder(cfg, 21)
    x = torch.rand(2, 3, 32, 40, generator=torch.Generator().manual_seed(1)) * 2 - 1
    mean, logvar = O.vae_encode(sd, cfg, x)
    assert mean.shape == (2, cfg.latent_channels, 4, 5) and logvar.shape == mean.shape
    assert logvar.min().item() >= -30.0 and logvar.max().item() <= 20.0
    # shifting the image by one pixel down/right changes what the bottom/right zero padding sees: the stride-2 convs are not
    # the symmetric-padding ones (a symmetric variant would give a different result on the same input)
    name = S.vae_encoder_blocks(cfg)[0]["down"]
    h = torch.randn(1, cfg.block_out_channels[0], 8, 8, generator=torch.Generator().manual_seed(2))
    a = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[name + ".weight"], sd[name + ".bias"], stride=2)
    b = F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], stride=2, padding=1)
    assert a.shape == b.shape == (1, cfg.block_out_channels[0], 4, 4) and (a - b).abs().max().item() > 1e-2
    noise = torch.randn(mean.shape, generator=torch.Generator().manual_seed(3))
    z = O.vae_latent_sample(mean, logvar, noise, cfg.scaling_factor)
    assert torch.allclose(z, (mean + (0.5 * logvar).exp() * noise) * cfg.scaling_factor)
    assert torch.equal(O.vae_latent_sample(mean, logvar, None), mean)


def test_dpm_solver_pp_scheduler_and_oracle():
    """Validation sampler of train_ID-Booth.py:155.  Analytic pins (the upstream module is not importable): (1) timesteps of
    "leading" spacing with N+1 intervals; (2) a first-order DPM-Solver++ step is the deterministic DDIM step
    x' = sqrt(abar') x0 + sqrt(1-abar') eps; (3) the last step (final sigma 0) returns the x0 prediction; (4) the host scheduler's
    fused-step coefficients reproduce the oracle's trajectory with an arbitrary model, orders 1 and 2, eps- and v-prediction."""
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.scheduler import DPMSolverMultistepScheduler, DDPMScheduler
    from oracle import sd21_oracle as O
    ts, sigmas = O.dpmpp_timesteps_sigmas(25)
    assert ts[:3] == [951, 913, 875] and ts[-1] == 1 + 38 and len(ts) == 25 and sigmas[-1].item() == 0.0
    sch = DPMSolverMultistepScheduler.from_config(DDPMScheduler().config, variance_type="fixed_small")
    sch.set_timesteps(25)
    assert sch.timesteps.tolist() == ts and torch.allclose(sch.sigmas, sigmas)
    ac = O.ddpm_tables().double()
    g = torch.Generator().manual_seed(0)
    x, eps = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    # (2) first step (order 1 by construction) against DDIM between ts[0] and ts[1]
    a_s, s_s, c_0, c_x, c_1 = sch.step_coefficients(0)
    x0 = (x - s_s * eps) / a_s
    ddim = ac[ts[1]].sqrt().float() * x0 + (1 - ac[ts[1]]).sqrt().float() * eps
    assert c_1 == 0.0 and torch.allclose(c_0 * x0 + c_x * x, ddim, atol=2e-5)
    # (3) last step
    a_s, s_s, c_0, c_x, c_1 = sch.step_coefficients(24)
    assert c_x == 0.0 and c_1 == 0.0 and abs(c_0 - 1.0) < 1e-6
    # (4) coefficient form == oracle loop, with a made-up nonlinear "model"
    for pred in ("epsilon", "v_prediction"):
        cfg = S.SchedulerConfig(prediction_type=pred)
        for order in (1, 2):
            model = lambda xx, t: torch.tanh(xx * 0.7 + t / 1000.0) - 0.1 * xx
            ref = O.dpmpp_2m_sample(None, None, None, None, x, 10, 1.0, sched=cfg, solver_order=order, model=model)
            s2 = DPMSolverMultistepScheduler(cfg, solver_order=order)
            s2.set_timesteps(10)
            cur = x.clone()
            for t in s2.timesteps.tolist():
                cur = s2.step(model(cur, t), t, cur).prev_sample
            assert torch.allclose(cur, ref, rtol=1e-4, atol=1e-4), (pred, order, (cur - ref).abs().max())
    with pytest.raises(ValueError):
        s2.step(x, 5, x)


# ---- the oracle's topology, independently of faceposegenerator_amd.spec -------------------------------
# SURVEY.md Appendix A.1 written out by hand: the module sequence of UNet2DConditionModel for SD-2.1 (block types
# [CrossAttnDownBlock2D x3, DownBlock2D] / mid / [UpBlock2D, CrossAttnUpBlock2D x3], layers_per_block 2).  "R" = ResnetBlock2D,
# "T" = Transformer2DModel (heads from the per-level head counts), "D" / "U" = down- / up-sampler conv, "S" = push skip,
# "C" = concatenate the popped skip.  The oracle (and the product) walk spec.unet_graph; this literal shares nothing with it.
_A1_LITERAL = (
    [("S",)] +
    [("R", "down_blocks.0.resnets.0"), ("T", "down_blocks.0.attentions.0", 0), ("S",),
     ("R", "down_blocks.0.resnets.1"), ("T", "down_blocks.0.attentions.1", 0), ("S",), ("D", "down_blocks.0.downsamplers.0.conv"), ("S",),
     ("R", "down_blocks.1.resnets.0"), ("T", "down_blocks.1.attentions.0", 1), ("S",),
     ("R", "down_blocks.1.resnets.1"), ("T", "down_blocks.1.attentions.1", 1), ("S",), ("D", "down_blocks.1.downsamplers.0.conv"), ("S",),
     ("R", "down_blocks.2.resnets.0"), ("T", "down_blocks.2.attentions.0", 2), ("S",),
     ("R", "down_blocks.2.resnets.1"), ("T", "down_blocks.2.attentions.1", 2), ("S",), ("D", "down_blocks.2.downsamplers.0.conv"), ("S",),
     ("R", "down_blocks.3.resnets.0"), ("S",), ("R", "down_blocks.3.resnets.1"), ("S",),
     ("R", "mid_block.resnets.0"), ("T", "mid_block.attentions.0", 3), ("R", "mid_block.resnets.1"),
     ("C",), ("R", "up_blocks.0.resnets.0"), ("C",), ("R", "up_blocks.0.resnets.1"), ("C",), ("R", "up_blocks.0.resnets.2"),
     ("U", "up_blocks.0.upsamplers.0.conv"),
     ("C",), ("R", "up_blocks.1.resnets.0"), ("T", "up_blocks.1.attentions.0", 2),
     ("C",), ("R", "up_blocks.1.resnets.1"), ("T", "up_blocks.1.attentions.1", 2),
     ("C",), ("R", "up_blocks.1.resnets.2"), ("T", "up_blocks.1.attentions.2", 2), ("U", "up_blocks.1.upsamplers.0.conv"),
     ("C",), ("R", "up_blocks.2.resnets.0"), ("T", "up_blocks.2.attentions.0", 1),
     ("C",), ("R", "up_blocks.2.resnets.1"), ("T", "up_blocks.2.attentions.1", 1),
     ("C",), ("R", "up_blocks.2.resnets.2"), ("T", "up_blocks.2.attentions.2", 1), ("U", "up_blocks.2.upsamplers.0.conv"),
     ("C",), ("R", "up_blocks.3.resnets.0"), ("T", "up_blocks.3.attentions.0", 0),
     ("C",), ("R", "up_blocks.3.resnets.1"), ("T", "up_blocks.3.attentions.1", 0),
     ("C",), ("R", "up_blocks.3.resnets.2"), ("T", "up_blocks.3.attentions.2", 0)])


def _unet_forward_from_literal(sd, heads_per_level, time_proj_dim, sample, timestep, ctx, groups=32, eps=1e-5):
    import torch.nn.functional as F
    temb = O.timestep_embedding(torch.as_tensor(timestep)[None].expand(sample.shape[0]), time_proj_dim)
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = []
    for op in _A1_LITERAL:
        if op[0] == "S":
            skips.append(h)
        elif op[0] == "C":
            h = torch.cat([h, skips.pop()], dim=1)
        elif op[0] == "R":
            h = O.resnet_block(sd, op[1], h, temb, groups, eps)
        elif op[0] == "T":
            h = O.transformer_2d(sd, op[1], h, ctx, heads_per_level[op[2]], groups, None)
        elif op[0] == "D":
            h = F.conv2d(h, sd[op[1] + ".weight"], sd[op[1] + ".bias"], stride=2, padding=1)
        else:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[op[1] + ".weight"], sd[op[1] + ".bias"], padding=1)
    assert not skips
    h = F.silu(F.group_norm(h, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def test_oracle_topology_equals_hand_written_appendix_a1(tiny):
    """The oracle's graph walk (spec.unet_graph, shared with the product) against the hand-written module sequence: same
    numbers, and the literal consumes every weight tensor of the state dict exactly as named in SURVEY.md Appendix B."""
    usd, _, _ = tiny
    g = torch.Generator().manual_seed(21)
    x, ctx = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 77, 128, generator=g)
    with torch.no_grad():
        a = O.unet_forward(usd, S.TINY_UNET, x, 437, ctx)
        b = _unet_forward_from_literal(usd, S.TINY_UNET.num_heads, S.TINY_UNET.time_proj_dim, x, 437, ctx)
    assert torch.equal(a, b)
    named = {op[1] for op in _A1_LITERAL if len(op) > 1}
    rest = [k for k in usd if not any(k.startswith(m + ".") for m in named)
            and not k.startswith(("conv_in.", "conv_out.", "conv_norm_out.", "time_embedding."))]
    assert not rest, f"weights the hand-written topology never touches: {rest[:5]}"
    assert sum(1 for op in _A1_LITERAL if op[0] == "R") == 22 and sum(1 for op in _A1_LITERAL if op[0] == "T") == 16
    assert sum(1 for op in _A1_LITERAL if op[0] == "S") == 12 == sum(1 for op in _A1_LITERAL if op[0] == "C")


def test_rounding_hook_is_identity_by_default_and_emulates_storage(tiny):
    usd, _, _ = tiny
    g = torch.Generator().manual_seed(22)
    x, ctx = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 77, 128, generator=g)
    with torch.no_grad():
        ref = O.unet_forward(usd, S.TINY_UNET, x, 500, ctx)
        kinds = []
        O.ROUND = lambda kind, t: (kinds.append(kind), t)[1]
        try:
            same = O.unet_forward(usd, S.TINY_UNET, x, 500, ctx)
            O.ROUND = lambda kind, t: t.half().float()
            f16 = O.unet_forward(usd, S.TINY_UNET, x, 500, ctx)
        finally:
            O.ROUND = None
    assert torch.equal(ref, same) and set(kinds) == {"act", "res"}
    rel = ((f16 - ref).norm() / ref.norm()).item()
    assert 1e-4 < rel < 1e-2


def test_config1_golden_fixture_is_wellformed():
    """BASELINE configs[1] trajectory (30 steps, LoRA): structure + the first step reproduced by the oracle on the reduced checks
    the CPU suite can afford (scheduler tables and RNG stream; the full trajectory is regenerated by make_golden.py config1)."""
    gold = np.load(os.path.join(GOLD, "sd21_config1.npz"))
    assert gold["meta"].tolist() == [1234, 1235, 1, 1, 64, 30, 2024, 0, 1]          # last entry: calibrated synthetic weights
    assert gold["latents_per_step"].shape == (30, 1, 4, 64, 64) and gold["final_latents"].shape == (1, 4, 64, 64)
    # activations stay O(1) through the 30 steps (rounds 1-2: std 1.4 -> 20, max 77 on uncalibrated weights)
    stds = gold["latents_per_step"].std(axis=(1, 2, 3, 4))
    assert 0.25 < stds.min() and stds.max() < 1.3 and np.abs(gold["latents_per_step"]).max() < 6.0, (stds, np.abs(gold["latents_per_step"]).max())
    assert np.abs(gold["latents_per_step"].mean(axis=(1, 2, 3, 4))).max() < 0.2
    for e in (gold["eps_uncond"], gold["eps_cond"]):                                # eps has about unit variance at every stored step
        assert 0.3 < e.std(axis=(1, 2, 3, 4)).min() and e.std(axis=(1, 2, 3, 4)).max() < 1.3
    assert np.array_equal(gold["latents_per_step"][-1], gold["final_latents"])
    assert gold["timesteps"].tolist() == O.ddpm_timesteps(30) == [1 + 33 * k for k in range(29, -1, -1)]
    assert gold["eps_steps"].tolist() == [0, 4, 9, 14, 19, 24, 29] and gold["eps_cond"].shape == (7, 1, 4, 64, 64)
    assert gold["image_u8"].shape == (1, 512, 512, 3) and np.isfinite(gold["latents_per_step"]).all()
    # step 0 is a closed form of the stored eps: latents_0 = ddpm_step(noise_0, eps_guided, noise_1)
    noise = O.draw_noise(torch.Generator().manual_seed(0), 1, 30, (64, 64))
    assert np.array_equal(noise.flatten()[:4].numpy(), gold["noise_first4"])
    e_u, e_c = torch.from_numpy(gold["eps_uncond"][0]), torch.from_numpy(gold["eps_cond"][0])
    prev, _ = O.ddpm_step(O.ddpm_tables(), O.ddpm_timesteps(30), 958, e_u + 5.0 * (e_c - e_u), noise[0], noise[1])
    assert np.abs(prev.numpy() - gold["latents_per_step"][0]).max() < 1e-5


def test_calibrated_weights_keep_latents_of_order_one(tiny):
    """weights.calibrate_unet: a 30-step CFG-5 DDPM chain on the reduced graph stays O(1) with the calibrated
    weights and blows up with the plain N(0, 1/fan_in) ones; the edit touches only the lane rows of five modules."""
    usd, _, lora_raw = tiny
    cal = W.calibrate_unet(usd, S.TINY_UNET)
    changed = sorted(k for k in usd if not torch.equal(usd[k], cal[k]))
    last = S.unet_graph(S.TINY_UNET).up[-1]
    assert changed == sorted(["conv_in.weight", "conv_in.bias", "conv_out.weight", "conv_out.bias", "conv_norm_out.weight", "conv_norm_out.bias",
                              last["resnets"][-1].name + ".conv2.weight", last["resnets"][-1].name + ".conv2.bias",
                              last["resnets"][-1].name + ".conv_shortcut.weight", last["resnets"][-1].name + ".conv_shortcut.bias",
                              last["attns"][-1].name + ".proj_out.weight", last["attns"][-1].name + ".proj_out.bias"])
    assert sum(v.numel() for v in cal.values()) == sum(v.numel() for v in usd.values())
    g = torch.Generator().manual_seed(2024)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    noise = O.draw_noise(torch.Generator().manual_seed(0), 1, 30, (16, 16))
    lora = O.normalize_lora_keys(lora_raw)
    with torch.no_grad():
        x = noise[0]
        eps = O.unet_forward(cal, S.TINY_UNET, x, 958, pe, lora)
        slope = ((eps * x).sum() / (x * x).sum()).item()
        assert 0.9 < slope < 1.1 and (eps - slope * x).std().item() < 0.4           # eps = x_t + a smaller network-dependent part
        lat_cal = O.sample(cal, S.TINY_UNET, pe, ne, noise, 30, 5.0, lora=lora)
        lat_raw = O.sample(usd, S.TINY_UNET, pe, ne, noise, 30, 5.0, lora=lora)
    assert lat_cal.std().item() < 2.0 and lat_cal.abs().max().item() < 8.0 and abs(lat_cal.mean().item()) < 0.6
    assert lat_raw.std().item() > 8.0                                                # what rounds 1-2 measured parity on
