"""GPU parity of the training-side row: AutoencoderKL.encode — the encoder conv stack with its
bottom/right-padded stride-2 convolutions (idb_gemm pad_mode 1), quant_conv folded into conv_out, and the
DiagonalGaussianDistribution sample/mode tail (idb_vae_sample) — against the CPU fp32 oracle (oracle.vae_encode), through
the ``vae.encode(x).latent_dist`` API the reference calls at train_ID-Booth.py:1001-1002.  Parity with upstream diffusers is
unpinned like the rest of the oracle (diffusers is not importable here)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"bf16": (8e-2, 2e-2), "f16": (1.5e-2, 3e-3)}       # max-abs on O(1) moments, rel-RMS (see test_engine_gpu.py)


def _stats(got, ref):
    d = (got.float().cpu() - ref.float()).abs()
    return d.max().item(), (d.pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt()).item()


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng(request, lib):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    return HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, request.param)


@pytest.mark.parametrize("b,h,w_,cin,cout,tile", [(2, 16, 16, 64, 128, 0), (1, 32, 32, 128, 128, 0), (3, 8, 12, 64, 64, 0), (1, 64, 64, 128, 160, 8),
                                                  (2, 16, 16, 64, 128, 7), (1, 16, 16, 128, 64, 14)])
def test_conv_stride2_bottom_right_padding(eng, b, h, w_, cin, cout, tile):
    """Downsample2D(padding=0) of the VAE encoder: F.pad(x, (0,1,0,1)) then a stride-2 3x3 conv."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, cin, h, w_, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
    bias = torch.randn(cout, generator=g)
    xq = x.to(DEV).to(eng.tdt)
    wq = w.to(DEV).to(eng.tdt)
    ref = F.conv2d(F.pad(xq.float(), (0, 1, 0, 1)), wq.float(), bias.to(DEV), stride=2)
    ref_sym = F.conv2d(xq.float(), wq.float(), bias.to(DEV), stride=2, padding=1)
    x_nhwc = xq.permute(0, 2, 3, 1).reshape(b * h * w_, cin).contiguous()
    wp = eng._pack_conv(w)
    out = eng.gemm([(x_nhwc, cin, 9, h, w_, 0)], wp, cout, b, h // 2, w_ // 2, bias=bias.to(DEV), stride=2, pad_mode=1, tile=tile)
    torch.cuda.synchronize()
    got = out.float().view(b, h // 2, w_ // 2, cout).permute(0, 3, 1, 2)
    tol = (2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9) * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol
    assert (got - ref_sym).abs().max().item() > 10 * tol            # and it is NOT the symmetric-padding result


def test_vae_sample_kernel(eng):
    from faceposegenerator_amd import _lib as L
    from oracle import sd21_oracle as O
    g = torch.Generator().manual_seed(3)
    b, c, hw = 3, 4, 80
    mom = torch.randn(b, hw, 2 * c, generator=g) * 12.0               # log-variances beyond [-30, 20] get clamped
    noise = torch.randn(b, c, 8, 10, generator=g)
    lat = torch.empty(b, c, 8, 10, device=DEV)
    mean, logvar = torch.empty_like(lat), torch.empty_like(lat)
    st = torch.cuda.current_stream().cuda_stream
    mom_d, noise_d = mom.to(DEV), noise.to(DEV)
    L.check(eng.lib.idb_vae_sample(mom_d.data_ptr(), noise_d.data_ptr(), 0.18215, lat.data_ptr(), mean.data_ptr(), logvar.data_ptr(),
                                   b, c, hw, st))
    m_ref = mom.view(b, 8, 10, 2 * c).permute(0, 3, 1, 2)
    mean_ref, lv_ref = m_ref[:, :c], m_ref[:, c:].clamp(-30.0, 20.0)
    ref = O.vae_latent_sample(mean_ref, lv_ref, noise, 0.18215)
    torch.cuda.synchronize()
    assert torch.equal(mean.cpu(), mean_ref.contiguous()) and torch.equal(logvar.cpu(), lv_ref.contiguous())
    assert torch.allclose(lat.cpu(), ref, rtol=2e-6, atol=1e-6)
    L.check(eng.lib.idb_vae_sample(mom_d.data_ptr(), None, 2.0, lat.data_ptr(), None, None, b, c, hw, st))
    torch.cuda.synchronize()
    assert torch.equal(lat.cpu(), (mean_ref * 2.0).contiguous())


def test_tiny_vae_encode_matches_oracle(eng):
    from faceposegenerator_amd import spec as S, weights as W
    from oracle import sd21_oracle as O
    cfg = S.TINY_VAE
    sd = W.synth_vae_encoder(cfg, 21)
    eng.pack_vae_encoder(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(3, 3, 128, 64, generator=g) * 2 - 1             # 16x8 latents (the mid attention needs h*w % 64 == 0)
    noise = torch.randn(3, cfg.latent_channels, 16, 8, generator=g)
    mean_ref, lv_ref = O.vae_encode(sd, cfg, x)
    lat, mean, logvar = eng.vae_encode(x.to(DEV), noise.to(DEV), scale=cfg.scaling_factor, chunk=2)
    mx, rel = _stats(torch.cat([mean, logvar], 1), torch.cat([mean_ref, lv_ref], 1))
    print(f"[{eng.dtype_name}] tiny VAE encode moments: max-abs {mx:.3e} rel-rms {rel:.3e}")
    assert mx < TOL[eng.dtype_name][0] * max(1.0, mean_ref.abs().max().item()) and rel < TOL[eng.dtype_name][1]
    ref = O.vae_latent_sample(mean_ref, lv_ref, noise, cfg.scaling_factor)
    assert _stats(lat, ref)[1] < 2 * TOL[eng.dtype_name][1]
    mode = eng.vae_encode(x.to(DEV), None, chunk=2)[0]              # same chunking = same GEMM plans: bit-identical
    assert torch.equal(mode, mean)
    # another chunking changes M, hence tile / split-K plans and where the GroupNorm statistics are summed: same values to rounding
    mode4 = eng.vae_encode(x.to(DEV), None, chunk=4)[0]
    assert _stats(mode4, mean.cpu())[1] < TOL[eng.dtype_name][1]
    with pytest.raises(ValueError):
        eng.vae_encode(torch.zeros(1, 3, 60, 64), None)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_size_vae_encode_and_api(lib, dtype):
    """SD-2.1 AutoencoderKL encoder shapes (34,163,664 parameters with quant_conv) on one 512x512 image, through
    ``pipe.vae.encode(x).latent_dist.sample(generator) * pipe.vae.config.scaling_factor`` (train_ID-Booth.py:1001-1002)."""
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    cfg = S.SD21_VAE
    esd = W.synth_vae_encoder(cfg, 4322)
    pipe = StableDiffusionPipeline(S.TINY_UNET, cfg, W.synth_unet(S.TINY_UNET, 7), W.synth_vae(cfg, 1235), torch_dtype=dtype).to(DEV)
    with pytest.raises(FileNotFoundError):
        pipe.vae.encode(torch.zeros(1, 3, 64, 64))
    pipe.set_vae_encoder_weights(esd)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    with torch.no_grad():
        mean_ref, lv_ref = O.vae_encode(esd, cfg, x)
    dist = pipe.vae.encode(x.to(DEV)).latent_dist
    mx, rel = _stats(torch.cat([dist.mean, dist.logvar], 1), torch.cat([mean_ref, lv_ref], 1))
    print(f"[{dtype}] full-size VAE encode (512x512): moments max-abs {mx:.3e} rel-rms {rel:.3e} (|ref| max {mean_ref.abs().max():.2f})")
    assert rel < TOL[dtype][1] and mx < TOL[dtype][0] * max(1.0, mean_ref.abs().max().item())
    lat = dist.sample(torch.Generator().manual_seed(77)) * pipe.vae.config.scaling_factor
    noise = torch.randn(mean_ref.shape, generator=torch.Generator().manual_seed(77))
    ref = O.vae_latent_sample(mean_ref, lv_ref, noise, cfg.scaling_factor)
    assert lat.shape == (1, 4, 64, 64) and _stats(lat, ref)[1] < 2 * TOL[dtype][1]
    assert torch.equal(dist.mode(), dist.mean)
    del pipe
    torch.cuda.empty_cache()
