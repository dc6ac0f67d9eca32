"""idb_hconv — GroupNorm(+SiLU) applied inside the consuming conv3x3 / 1x1 (ResnetBlock2D norm1+conv1, norm2+conv2+shortcut,
Transformer2DModel norm+proj_in) — against plain torch fp32 ops on the same operand-dtype-rounded inputs: group_norm -> silu ->
(rounded to the operand dtype, as the kernel keeps the normalised patch in LDS in that dtype) -> conv2d, plus the epilogue terms.
Same tolerance as the other kernel tests: a few output ulps (2^-7 relative for bf16, 2^-9 for f16)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng(request, lib):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    return HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, request.param)


def _tol(eng):
    return 2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _check(out, ref, tol, what):
    err = (out.float() - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err <= tol * max(1.0, mag), f"{what}: max err {err:.4e} vs tol {tol * max(1.0, mag):.4e} (|ref| max {mag:.3f})"


@pytest.mark.parametrize("b,h,w_,c0,c1,cout,split_k,silu", [
    (2, 8, 8, 128, 0, 128, 0, True), (2, 8, 8, 128, 0, 128, 1, True), (1, 16, 16, 64, 64, 160, 0, True), (2, 16, 16, 128, 0, 320, 2, True),
    (1, 32, 32, 320, 0, 320, 0, True), (1, 64, 64, 320, 0, 320, 0, True), (1, 64, 64, 64, 0, 128, 1, True), (8, 4, 4, 64, 0, 128, 0, True),
    (2, 32, 32, 192, 128, 640, 4, True), (4, 8, 8, 256, 0, 256, 3, False), (2, 16, 16, 320, 0, 64, 5, True), (1, 16, 16, 640, 320, 160, 8, True)])
def test_hconv_groupnorm_silu_conv3x3(eng, b, h, w_, c0, c1, cout, split_k, silu):
    G, eps = 32, 1e-5
    c = c0 + c1
    x = (_rand((b, c, h, w_), 1, 1.5) + 0.3).to(eng.tdt)
    gamma, beta = 1.0 + 0.2 * _rand((c,), 2), 0.1 * _rand((c,), 3)
    wt = _rand((cout, c, 3, 3), 4, (9 * c) ** -0.5)
    bias, sb = _rand((cout,), 5), _rand((b, cout), 6)
    res = _rand((b, cout, h, w_), 7).to(eng.tdt)
    y = F.group_norm(x.float(), G, gamma, beta, eps)
    y = (F.silu(y) if silu else y).to(eng.tdt).float()
    ref = F.conv2d(y, wt.to(eng.tdt).float(), bias, padding=1) + sb[:, :, None, None] + res.float()
    xa = _nhwc(x[:, :c0])
    xb = _nhwc(x[:, c0:]) if c1 else None
    segs = [(xa, c0, xb, c1, 9)]
    wp = eng._pack_conv(wt)
    assert eng.hconv_supported(segs, wp, cout, b, h, w_, G)
    part, chunks = eng.gn_statistics(xa, c0, xb, c1, b, h * w_, G)
    out = eng.hconv(segs, wp, cout, b, h, w_, gn=(part, chunks, G, eps, gamma, beta, silu), bias=bias, sbias=(sb, 0, cout),
                    residual=_nhwc(res), gn_stats=G if cout % G == 0 and cout // G >= 2 else 0, split_k=split_k)
    torch.cuda.synchronize()
    _check(out.view(b, h, w_, cout).permute(0, 3, 1, 2), ref, _tol(eng), f"hconv {b}x{h}x{w_} {c0}+{c1}->{cout} sk={split_k}")
    st = getattr(out, "_gn", None)
    if st is not None:                          # statistics of the ROUNDED output, per (sample, 64-row chunk, group)
        o = out.float().view(b, h * w_ // 64, 64, G, cout // G)
        ref_s = torch.stack([o.sum(dim=(2, 4)), (o * o).sum(dim=(2, 4))], dim=-1)
        got = st[0].view(b, h * w_ // 64, G, 2)
        assert torch.allclose(got, ref_s, rtol=2e-4, atol=2e-2), (got - ref_s).abs().max().item()


@pytest.mark.parametrize("b,h,w_,ca,cb,cout,split_k", [(2, 8, 8, 128, 64, 128, 0), (1, 32, 32, 320, 320, 320, 0), (2, 16, 16, 256, 0, 128, 3),
                                                      (1, 64, 64, 64, 64, 320, 2), (1, 16, 16, 128, 0, 192, 1)])
def test_hconv_conv2_with_fused_shortcut(eng, b, h, w_, ca, cb, cout, split_k):
    """norm2 + SiLU + conv2 over h1, plus the 1x1 conv_shortcut over the RAW cat[xa, xb] as a second K segment."""
    G, eps = 32, 1e-5
    h1 = (_rand((b, cout, h, w_), 10, 2.0) - 0.5).to(eng.tdt)
    xa, xb = _rand((b, ca, h, w_), 11).to(eng.tdt), (_rand((b, cb, h, w_), 12).to(eng.tdt) if cb else None)
    gamma, beta = 1.0 + 0.2 * _rand((cout,), 13), 0.1 * _rand((cout,), 14)
    w2 = _rand((cout, cout, 3, 3), 15, (9 * cout) ** -0.5)
    ws = _rand((cout, ca + cb, 1, 1), 16, (ca + cb) ** -0.5)
    bias = _rand((cout,), 17)
    y = F.silu(F.group_norm(h1.float(), G, gamma, beta, eps)).to(eng.tdt).float()
    xcat = torch.cat([xa, xb], 1) if cb else xa
    ref = F.conv2d(y, w2.to(eng.tdt).float(), None, padding=1) + F.conv2d(xcat.float(), ws.to(eng.tdt).float(), bias)
    wp = torch.cat([eng._pack_conv(w2), eng._pack_mat(ws.reshape(cout, ca + cb))], dim=1).contiguous()
    h1n, xan, xbn = _nhwc(h1), _nhwc(xa), (_nhwc(xb) if cb else None)
    segs = [(h1n, cout, None, 0, 9), (xan, ca, xbn, cb, 1)]
    assert eng.hconv_supported(segs, wp, cout, b, h, w_, G)
    part, chunks = eng.gn_statistics(h1n, cout, None, 0, b, h * w_, G)
    out = eng.hconv(segs, wp, cout, b, h, w_, gn=(part, chunks, G, eps, gamma, beta, True), bias=bias, split_k=split_k)
    torch.cuda.synchronize()
    _check(out.view(b, h, w_, cout).permute(0, 3, 1, 2), ref, _tol(eng), "hconv conv2+shortcut")


@pytest.mark.parametrize("b,h,w_,c,split_k", [(2, 16, 16, 128, 0), (1, 64, 64, 320, 0), (2, 8, 8, 256, 2), (1, 32, 32, 640, 1)])
def test_hconv_norm_proj_in(eng, b, h, w_, c, split_k):
    """Transformer2DModel: GroupNorm(eps 1e-6, no activation) + Linear proj_in as a 1x1 segment."""
    G, eps = 32, 1e-6
    x = (_rand((b, c, h, w_), 20, 1.3) + 0.2).to(eng.tdt)
    gamma, beta = 1.0 + 0.2 * _rand((c,), 21), 0.1 * _rand((c,), 22)
    wt, bias = _rand((c, c), 23, c ** -0.5), _rand((c,), 24)
    y = F.group_norm(x.float(), G, gamma, beta, eps).to(eng.tdt).float()
    ref = F.conv2d(y, wt.to(eng.tdt).float()[:, :, None, None], bias)
    xn = _nhwc(x)
    segs = [(xn, c, None, 0, 1)]
    wp = eng._pack_mat(wt)
    assert eng.hconv_supported(segs, wp, c, b, h, w_, G)
    part, chunks = eng.gn_statistics(xn, c, None, 0, b, h * w_, G)
    out = eng.hconv(segs, wp, c, b, h, w_, gn=(part, chunks, G, eps, gamma, beta, False), bias=bias, split_k=split_k)
    torch.cuda.synchronize()
    _check(out.view(b, h, w_, c).permute(0, 3, 1, 2), ref, _tol(eng), "hconv norm+proj_in")


def test_hconv_statistics_from_the_producing_gemm(eng):
    """The statistics input may come from the conv that produced the tensor (idb_gemm_desc.gn_partials): same result as from the
    statistics launch, bit for bit (both sum the rounded tensor per 64-row chunk in a fixed order ... up to fp32 association)."""
    b, h, w_, cin, c = 2, 16, 16, 128, 256
    G, eps = 32, 1e-5
    x0 = _nhwc(_rand((b, cin, h, w_), 30).to(eng.tdt))
    w0 = eng._pack_conv(_rand((c, cin, 3, 3), 31, (9 * cin) ** -0.5))
    mid = eng.gemm([(x0, cin, 9, h, w_, 0)], w0, c, b, h, w_, gn_stats=G, gn_stats_always=True)
    assert getattr(mid, "_gn", None) is not None
    gamma, beta = 1.0 + 0.2 * _rand((c,), 32), 0.1 * _rand((c,), 33)
    wp = eng._pack_conv(_rand((c, c, 3, 3), 34, (9 * c) ** -0.5))
    part_a, chunks_a = eng.gn_statistics(mid, c, None, 0, b, h * w_, G)          # from the producing launch
    assert chunks_a == h * w_ // 64
    out_a = eng.hconv([(mid, c, None, 0, 9)], wp, c, b, h, w_, gn=(part_a, chunks_a, G, eps, gamma, beta, True))
    part_b, chunks_b = eng.gn_statistics(mid, c, None, 0, b, h * w_, G)          # _gn consumed: a statistics launch
    out_b = eng.hconv([(mid, c, None, 0, 9)], wp, c, b, h, w_, gn=(part_b, chunks_b, G, eps, gamma, beta, True))
    torch.cuda.synchronize()
    assert (out_a.float() - out_b.float()).abs().max().item() <= _tol(eng) * out_b.float().abs().max().item()


def test_hconv_rejects_shapes_it_does_not_tile(eng):
    x = _nhwc(_rand((2, 64, 13, 11), 40).to(eng.tdt))
    w = eng._pack_conv(_rand((64, 64, 3, 3), 41))
    assert not eng.hconv_supported([(x, 64, None, 0, 9)], w, 64, 2, 13, 11, 32)      # 13x11: not whole rows of 128 pixels
    x = _nhwc(_rand((1, 64, 96, 96), 42).to(eng.tdt))
    assert not eng.hconv_supported([(x, 64, None, 0, 9)], w, 64, 1, 96, 96, 32)      # 96-wide maps (768^2 config)
    x = _nhwc(_rand((3, 64, 8, 8), 43).to(eng.tdt))
    assert not eng.hconv_supported([(x, 64, None, 0, 9)], w, 64, 3, 8, 8, 32)        # 3 samples of 64 pixels: not a multiple of 128


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_unet_and_vae_with_the_fused_path_switched_on(lib, dtype):
    """The engine's opt-in policy (IDB_HCONV=1): every ResnetBlock2D and Transformer2DModel.norm+proj_in of the reduced graph through
    idb_hconv, against the CPU oracle with the single-forward tolerance of tests/test_engine_gpu.py, and against the default path."""
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    ucfg, vcfg = S.TINY_UNET, S.TINY_VAE
    usd, vsd = W.synth_unet(ucfg, 7), W.synth_vae(vcfg, 8)
    pipe = StableDiffusionPipeline(ucfg, vcfg, usd, vsd, torch_dtype=dtype).to(DEV)
    eng = pipe._engine()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 4, 16, 16, generator=g)                       # 8 samples: the 4x4 level tiles as 8 whole samples
    ctx = torch.randn(8, 77, ucfg.cross_attention_dim, generator=g)
    ref = O.unet_forward(usd, ucfg, x, 501, ctx)
    base = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    n_base = eng.last_forward_launches
    eng._use_hconv = True
    got = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    n_fused = eng.last_forward_launches
    mx_tol, rel_tol = {"bf16": (6e-2, 2e-2), "f16": (1e-2, 3e-3)}[dtype]
    err = (got - ref).abs().max().item()
    rel = ((got - ref).norm() / ref.norm()).item()
    print(f"[{dtype}] fused-GroupNorm UNet forward: max-abs {err:.3e} rel-rms {rel:.3e}; launches {n_base} -> {n_fused}")
    assert err < mx_tol and rel < rel_tol
    assert (got - base).abs().max().item() < mx_tol and n_fused < n_base
    z = torch.randn(2, 4, 16, 16, generator=g) * 3
    ref_v = O.vae_decode(vsd, vcfg, z)
    got_v = pipe.vae.decode(z.to(DEV)).sample.cpu()
    assert ((got_v - ref_v).norm() / ref_v.norm()).item() < rel_tol
