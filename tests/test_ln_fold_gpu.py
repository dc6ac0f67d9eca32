"""LayerNorm folded into the consuming projection (idb_gemm_desc.row_stats_out / ln_*): the producer GEMM's epilogue emits per-row
partial sums of its rounded output, the consumer multiplies the RAW rows with gamma-scaled weights and applies
rstd (acc - mean u) + v in its epilogue.  Checked against torch fp32 LayerNorm -> Linear (and GEGLU) and against the unfolded
engine path; the folded form skips one rounding (the LayerNorm output), so it is compared with the exact fp32 result."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=["f16", "bf16"])
def eng(request, lib):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    return HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, request.param)


def _tol(eng):
    return 2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("m,c,n_out,geglu", [(512, 320, 960, False), (8192, 320, 320, False), (2048, 640, 1920, False), (300, 128, 256, False),
                                             (512, 1280, 3840, False), (4096, 320, 2560, True), (1024, 640, 5120, True), (200, 128, 1024, True),
                                             # persistent-variant consumers: ragged M; 4 and 20 partials per row (> 8: the looped loads)
                                             (4000, 320, 2560, True), (16384, 640, 5120, True), (512, 1280, 10240, True)])
def test_producer_row_stats_and_folded_consumer(eng, m, c, n_out, geglu):
    # producer: h = a Wp^T + b + res  (attn.to_out + residual), emitting the row statistics of the rounded h
    a = _rand((m, c), 1).to(eng.tdt)
    wp = _rand((c, c), 2, c ** -0.5).to(eng.tdt)
    bp, res = _rand((c,), 3), (_rand((m, c), 4, 3.0) + 0.7).to(eng.tdt)
    h = eng.linear(a, wp, c, c, bias=bp, residual=res, row_stats=True)
    rs = getattr(h, "_rs", None)
    assert rs is not None, "this shape's plan should run the LDS-staged epilogue"
    torch.cuda.synchronize()
    part = rs[0].view(m, rs[1], 2).double()
    hf = h.double()
    assert torch.allclose(part[:, :, 0].sum(1), hf.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[:, :, 1].sum(1), (hf * hf).sum(1), rtol=1e-5, atol=1e-3)
    # consumer: LayerNorm(h) W^T (+ bias) [GEGLU]
    gamma, beta = 1.0 + 0.3 * _rand((c,), 5), 0.2 * _rand((c,), 6)
    w = _rand((n_out, c), 7, c ** -0.5)
    bias = _rand((n_out,), 8)
    ln = F.layer_norm(h.float(), (c,), gamma, beta, 1e-5)
    proj = ln @ w.t() + bias
    if geglu:
        val, gate = proj.chunk(2, dim=-1)
        ref = val * F.gelu(gate)
        perm = eng._geglu_perm(n_out).to(DEV)
        wln = eng._pack_mat(w * gamma[None, :], geglu=True)
        u, v, b_ = wln.float().sum(1).contiguous(), (w @ beta)[perm].contiguous(), bias[perm].contiguous()
    else:
        ref = proj
        wln = eng._pack_mat(w * gamma[None, :])
        u, v, b_ = wln.float().sum(1).contiguous(), (w @ beta).contiguous(), bias
    # the layer's bias rides in ln_v (idb_gemm rejects bias + ln_stats)
    out = eng.linear(h, wln, n_out, c, geglu=geglu, ln=(rs[0], rs[1], u, (v + b_).contiguous(), 1e-5), flags=256)   # bit 8: also in the persistent variant
    assert out is not None
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    tol = 2 * _tol(eng) * max(1.0, ref.abs().max().item())
    assert err <= tol, f"folded LN m={m} c={c} n={n_out} geglu={geglu}: {err:.4e} vs {tol:.4e}"
    # the unfolded path on the same inputs is no closer to the exact result
    t = eng.layernorm(h, m, c, gamma, beta)
    old = eng.linear(t, eng._pack_mat(w, geglu=geglu), n_out, c, bias=b_, geglu=geglu)
    torch.cuda.synchronize()
    err_old = (old.float() - ref).abs().max().item()
    assert err <= max(2.0 * err_old, tol / 2)


def test_persistent_consumers_are_covered(eng):
    """The GEGLU cases above with >= 512 tiles (or K >= 1024 and >= 256) run idb_gemm_kernel_pl: plan variant 4."""
    import ctypes as C
    from faceposegenerator_amd import _lib as L
    x = torch.zeros(16, device=DEV)
    for (m, c, n_out, want) in [(4000, 320, 2560, True), (16384, 640, 5120, True), (512, 1280, 10240, True), (1024, 640, 5120, False)]:
        d = L.GemmDesc()
        d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = eng.dt, m, 1, 1, 1, n_out, 1
        d.src[0].ptr, d.src[0].channels, d.src[0].taps, d.src[0].in_h, d.src[0].in_w = x.data_ptr(), c, 1, 1, 1
        d.w, d.geglu, d.out, d.out_dtype, d.out_ld = x.data_ptr(), 1, x.data_ptr(), eng.dt, n_out // 2
        tile, sk, blocks = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(eng.lib.idb_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk), C.byref(blocks)), "idb_gemm_plan")
        assert (tile.value // 10 == 4) == want, (m, c, n_out, tile.value)
        assert eng.lib.idb_gemm_folds_layernorm(C.byref(d)) == (0 if want else 1)      # off by default in the persistent variant ...
        d.flags = 256
        assert eng.lib.idb_gemm_folds_layernorm(C.byref(d)) == 1                        # ... on with flags bit 8


def test_split_k_plans_decline_the_fold(eng):
    m, c = 128, 1280                                   # M = 128: the plan splits K
    h = _rand((m, c), 1).to(eng.tdt)
    w = eng._pack_mat(_rand((c, c), 2, c ** -0.5))
    out = eng.linear(h, w, c, c, row_stats=True)
    assert getattr(out, "_rs", None) is None
    dummy = torch.zeros(m * 8 * 2, device=DEV)
    vec = torch.zeros(c, device=DEV)
    assert eng.linear(h, w, c, c, ln=(dummy, 8, vec, vec, 1e-5)) is None


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_unet_forward_with_and_without_the_fold(lib, dtype):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    ucfg, vcfg = S.TINY_UNET, S.TINY_VAE
    usd, vsd, lora_raw = W.synth_unet(ucfg, 7), W.synth_vae(vcfg, 8), W.synth_lora(ucfg, 3)
    pipe = StableDiffusionPipeline(ucfg, vcfg, usd, vsd, torch_dtype=dtype).to(DEV)
    pipe.load_lora_weights(lora_raw)
    eng = pipe._engine()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 4, 32, 32, generator=g)
    ctx = torch.randn(4, 77, ucfg.cross_attention_dim, generator=g)
    ref = O.unet_forward(usd, ucfg, x, 501, ctx, O.normalize_lora_keys(lora_raw))
    assert eng._ln_fold
    got = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    n_fold = eng.last_forward_launches
    eng._ln_fold = False
    base = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    n_base = eng.last_forward_launches
    eng._ln_fold = True
    mx_tol, rel_tol = {"bf16": (6e-2, 2e-2), "f16": (1e-2, 3e-3)}[dtype]
    e_f, e_b = ((got - ref).norm() / ref.norm()).item(), ((base - ref).norm() / ref.norm()).item()
    print(f"[{dtype}] UNet forward (LoRA): folded LayerNorm rel-rms {e_f:.3e} ({n_fold} launches) vs idb_layernorm {e_b:.3e} ({n_base} launches)")
    assert (got - ref).abs().max().item() < mx_tol and e_f < rel_tol and n_fold < n_base
    # identity switch: the folded operands follow the LoRA set (in place: captured graphs keep their addresses)
    pipe.unload_lora_weights()
    ref0 = O.unet_forward(usd, ucfg, x, 501, ctx)
    got0 = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    assert ((got0 - ref0).norm() / ref0.norm()).item() < rel_tol


def test_feed_forward_in_row_chunks_matches_the_single_piece(lib):
    """engine._ff_chunk_bytes: ff.net.0 -> ff.net.2 over row chunks (default for intermediates beyond 320 MB) against the one-piece
    form on the reduced graph with a chunk limit small enough to trigger; the GEMM plans differ with M, so equality is to rounding."""
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from oracle import sd21_oracle as O
    ucfg, vcfg = S.TINY_UNET, S.TINY_VAE
    usd = W.synth_unet(ucfg, 7)
    pipe = StableDiffusionPipeline(ucfg, vcfg, usd, W.synth_vae(vcfg, 8), torch_dtype="f16").to(DEV)
    eng = pipe._engine()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 4, 32, 32, generator=g)
    ctx = torch.randn(4, 77, ucfg.cross_attention_dim, generator=g)
    ref = O.unet_forward(usd, ucfg, x, 501, ctx)
    eng._ff_chunk_bytes = 0
    whole = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    n_whole = eng.last_forward_launches
    eng._ff_chunk_bytes = 64 << 10                     # 64 KB of intermediate per chunk: every transformer block splits
    parts = pipe.unet(x.to(DEV), 501, ctx.to(DEV), return_dict=False)[0].cpu()
    assert eng.last_forward_launches > n_whole
    e_w, e_p = ((whole - ref).norm() / ref.norm()).item(), ((parts - ref).norm() / ref.norm()).item()
    assert e_p < 3e-3 and abs(e_p - e_w) < 5e-4, (e_w, e_p)
    assert ((parts - whole).norm() / whole.norm()).item() < 2e-3
