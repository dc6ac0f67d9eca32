"""GPU parity against the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the
CPU fp32 oracle).  The oracle itself is NOT run here for the full-size case: the 866 M-parameter synthetic weights are
regenerated from their seed (fingerprints checked) and the HIP path is compared with the stored oracle outputs.

Stated tolerances (north_star: "within stated fp16 tolerance"; operands bf16 = 8 significant bits, f16 = 11):
  * one CFG UNet forward at full size (teacher-forced, step 0): rel-RMS of eps <= 2.5e-2 (bf16) / 4e-3 (f16)
  * free-running 4-step trajectory (CFG 5 amplifies eps error ~9x, 1/sqrt(abar_751) ~4.2x): rel-RMS of the final
    latents <= 8e-2 (bf16) / 1.2e-2 (f16)
  * VAE decode of the golden latents: |uint8 difference| <= 6 levels (bf16) / 2 (f16) on >= 99.5 % of pixels
Measured values are printed (and recorded in DESIGN.md)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_NAMES = ["conv_in.weight", "mid_block.resnets.0.conv1.weight",
            "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_k.weight"]


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())), float(np.abs(a - b).max())


def _inputs(gold, cross_dim):
    useed, vseed, lseed, batch, side, steps, eseed, nseed = gold["meta"].tolist()
    g = torch.Generator().manual_seed(eseed)
    pe = torch.randn(batch, 77, cross_dim, generator=g)
    ne = torch.randn(batch, 77, cross_dim, generator=g)
    noise = torch.stack([torch.randn((batch, 4, side, side), generator=gen, dtype=torch.float32)
                         for gen in [torch.Generator().manual_seed(nseed)] for _ in range(steps + 1)])
    assert np.array_equal(noise.flatten()[:4].numpy(), gold["noise_first4"])
    return pe, ne, noise, steps, side


@pytest.fixture(scope="module")
def full_models(lib):
    from faceposegenerator_amd import spec as S, weights as W
    gold = np.load(os.path.join(GOLD, "sd21_config0.npz"))
    usd, vsd = W.synth_unet(S.SD21_UNET, 1234), W.synth_vae(S.SD21_VAE, 1235)
    assert np.allclose([float(usd[n].double().sum()) for n in FP_NAMES], gold["unet_fingerprint"], rtol=0, atol=1e-7), \
        "synthetic weights differ from the ones the golden vectors were made with"
    return gold, usd, vsd


@pytest.mark.parametrize("dtype,tol_fw,tol_traj,tol_px", [("bf16", 2.5e-2, 8e-2, 6), ("f16", 4e-3, 1.2e-2, 2)])
def test_full_size_config0_against_golden(full_models, dtype, tol_fw, tol_traj, tol_px):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    gold, usd, vsd = full_models
    pipe = StableDiffusionPipeline(S.SD21_UNET, S.SD21_VAE, usd, vsd, torch_dtype=dtype).to(DEV)
    pe, ne, noise, steps, side = _inputs(gold, 1024)
    # (1) teacher-forced first step: eps of the CFG pair on the golden initial latents
    x0 = noise[0]
    eps = pipe.unet(torch.cat([x0, x0]).to(DEV), int(gold["timesteps"][0]), torch.cat([ne, pe]).to(DEV), return_dict=False)[0].cpu()
    r_u, m_u = _rel(eps[0:1].numpy(), gold["eps_uncond"][0])
    r_c, m_c = _rel(eps[1:2].numpy(), gold["eps_cond"][0])
    print(f"[{dtype}] full-size CFG forward: rel-rms uncond {r_u:.3e} cond {r_c:.3e}; max-abs {max(m_u, m_c):.3e}")
    assert max(r_u, r_c) < tol_fw
    # (2) free-running 4 steps (BASELINE configs[0]) through the pipeline API, HIP graph on
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=steps, guidance_scale=float(gold["guidance_scale"]),
               height=side * 8, width=side * 8, output_type="latent", noise=noise)
    r, m = _rel(out.images.cpu().numpy(), gold["final_latents"])
    print(f"[{dtype}] full-size 4-step trajectory: latents rel-rms {r:.3e} max-abs {m:.3e} (|ref| std {gold['final_latents'].std():.2f})")
    assert r < tol_traj
    # (3) VAE decode + postprocess + uint8 of the GOLDEN latents
    img01, u8 = pipe._engine().decode_images(torch.from_numpy(gold["final_latents"]).to(DEV))
    diff = np.abs(u8.cpu().numpy().astype(int) - gold["image_u8"].astype(int))
    frac = float((diff <= tol_px).mean())
    print(f"[{dtype}] full-size VAE decode: uint8 max diff {diff.max()}, {100 * frac:.3f} % of pixels within {tol_px}")
    assert frac >= 0.995
    del pipe
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,tol", [("bf16", 8e-2), ("f16", 1.2e-2)])
def test_tiny_trajectory_against_golden(lib, dtype, tol):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    gold = np.load(os.path.join(GOLD, "tiny_trajectory.npz"))
    pipe = StableDiffusionPipeline(S.TINY_UNET, S.TINY_VAE, W.synth_unet(S.TINY_UNET, 7), W.synth_vae(S.TINY_VAE, 8),
                                   torch_dtype=dtype).to(DEV)
    pipe.load_lora_weights(W.synth_lora(S.TINY_UNET, 3))
    pe, ne, noise, steps, side = _inputs(gold, 128)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=steps, guidance_scale=5.0, height=side * 8,
               width=side * 8, output_type="latent", noise=noise)
    r, m = _rel(out.images.cpu().numpy(), gold["final_latents"])
    print(f"[{dtype}] tiny 4-step trajectory: rel-rms {r:.3e} max-abs {m:.3e}")
    assert r < tol
