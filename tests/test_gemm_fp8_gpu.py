"""fp8 (OCP e4m3) implicit GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 (idb_gemm_fp8, BASELINE configs[4] at kernel level) against
torch fp32 ops on the DEQUANTISED operands: products of two e4m3 values are exact in fp32, so only the accumulation order and the
one rounding of the output differ — same tolerance as the 16-bit kernels.  Also: the quantiser against torch's float8_e4m3fn cast,
the weight packer's per-channel scales and zero padding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module", params=["f16", "bf16"])
def eng(request, lib):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    return HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, request.param)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _deq_w(w8, scales, cout, cin, taps):
    cpad = (cin + 127) // 128 * 128
    w = w8.view(F8).float().view(cout, taps, cpad)
    assert (w[:, :, cin:] == 0).all()
    return w[:, :, :cin] * scales[:, None, None]                       # [cout][tap][cin]


def test_quantiser_matches_torch_e4m3(eng):
    x = (_rand((64, 320), 1, 3.0)).to(eng.tdt)
    x[0, :8] = torch.tensor([0.0, 1e-4, -1e-4, 500.0, -500.0, 448.0, 0.0175, 1.0]).to(eng.tdt)
    scale = 0.05
    q = eng.quantize_fp8(x, scale)
    ref = (x.float() / scale).clamp(-448, 448).to(F8)
    assert torch.equal(q.view(F8).float(), ref.float())


@pytest.mark.parametrize("b,h,w_,cin,cout,taps,stride,up", [
    (2, 16, 16, 128, 160, 9, 1, 0), (1, 32, 32, 320, 320, 9, 1, 0), (2, 8, 8, 64, 128, 9, 1, 0), (1, 16, 16, 192, 256, 9, 2, 0),
    (1, 8, 8, 128, 160, 9, 1, 1), (3, 24, 24, 320, 640, 9, 1, 0), (1, 96, 96, 320, 320, 9, 1, 0), (4, 1, 77, 1024, 640, 1, 1, 0),
    (2, 32, 32, 320, 960, 1, 1, 0), (1, 13, 11, 64, 72, 9, 1, 0)])
def test_gemm_fp8_conv_and_linear(eng, b, h, w_, cin, cout, taps, stride, up):
    x = _rand((b, cin, h, w_), 10, 2.0).to(eng.tdt)
    wt = _rand((cout, cin, 3, 3) if taps == 9 else (cout, cin, 1, 1), 11, (taps * cin) ** -0.5)
    bias, sb = _rand((cout,), 12), _rand((b, cout), 13)
    x_scale = float(x.float().abs().max()) / 448.0
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    x8 = eng.quantize_fp8(x_nhwc, x_scale)
    w8, ws = eng.pack_weight_fp8(wt)
    # per-channel scales = absmax / 448
    assert torch.allclose(ws, wt.flatten(1).abs().amax(1) / 448.0, rtol=1e-6)
    xd = (x8.view(F8).float() * x_scale).permute(0, 3, 1, 2)           # dequantised NCHW
    wd = _deq_w(w8, ws, cout, cin, taps).permute(0, 2, 1).reshape(cout, cin, 3, 3) if taps == 9 else _deq_w(w8, ws, cout, cin, 1).permute(0, 2, 1).reshape(cout, cin, 1, 1)
    xin = F.interpolate(xd, scale_factor=2.0, mode="nearest") if up else xd
    ref = F.conv2d(xin, wd, bias, stride=stride, padding=1 if taps == 9 else 0) + sb[:, :, None, None]
    oh, ow = ref.shape[2], ref.shape[3]
    res = _rand((b, cout, oh, ow), 14).to(eng.tdt)
    ref = ref + res.float()
    out = eng.gemm_fp8(x8, x_scale, cin, taps, h, w_, w8, ws, cout, b, oh, ow, bias=bias, sbias=(sb, 0, cout),
                       residual=res.permute(0, 2, 3, 1).contiguous(), stride=stride, upsample=up)
    torch.cuda.synchronize()
    got = out.view(b, oh, ow, cout).permute(0, 3, 1, 2).float()
    tol = (2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9) * max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= tol, f"fp8 gemm {b}x{h}x{w_} {cin}->{cout} taps {taps}: max err {err:.4e} vs tol {tol:.4e}"
    # and against the UNQUANTISED fp32 conv: the quantisation error of the 8-bit operands (informative bound: a few percent)
    full = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float(), wt, bias, stride=stride,
                    padding=1 if taps == 9 else 0) + sb[:, :, None, None] + res.float()
    rel = ((got - full).norm() / full.norm()).item()
    assert rel < 6e-2, rel


@pytest.mark.parametrize("b,side,cin,cout", [(2, 32, 320, 320), (3, 16, 128, 640), (1, 8, 256, 1280)])
def test_gemm_fp8_emits_groupnorm_partials(eng, b, side, cin, cout):
    """idb_gemm_fp8_desc.gn_partials from the kernel's own LDS-staged epilogue (160-wide tiles, groups of 10 / 20 / 40 channels):
    {sum, sum of squares} of the ROUNDED output per (sample, 64-row chunk, group), and the same output tensor as without them."""
    x = _rand((b, side, side, cin), 20, 2.0).to(eng.tdt)
    wt = _rand((cout, cin, 3, 3), 21, (9 * cin) ** -0.5)
    bias = _rand((cout,), 22)
    x_scale = float(x.float().abs().max()) / 448.0
    x8 = eng.quantize_fp8(x, x_scale)
    w8, ws = eng.pack_weight_fp8(wt)
    plain = eng.gemm_fp8(x8, x_scale, cin, 9, side, side, w8, ws, cout, b, side, side, bias=bias)
    out = eng.gemm_fp8(x8, x_scale, cin, 9, side, side, w8, ws, cout, b, side, side, bias=bias, gn_stats=32)
    st = getattr(out, "_gn", None)
    assert st is not None and st[1] == side * side // 64
    torch.cuda.synchronize()
    assert torch.equal(out, plain)
    cpg, hw = cout // 32, side * side
    yr = out.double().view(b, hw // 64, 64, 32, cpg)
    part = st[0].view(b, hw // 64, 32, 2).double()
    assert torch.allclose(part[..., 0], yr.sum(dim=(2, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1], (yr * yr).sum(dim=(2, 4)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("b,h,cin,cout,taps", [(64, 32, 320, 320, 9), (128, 16, 1280, 640, 9), (64, 32, 1280, 320, 1), (32, 48, 192, 320, 9)])
def test_gemm_fp8_256_row_loader_wave_tiles_bit_identical(eng, b, h, cin, cout, taps):
    """Large grids with K >= 1280 run the 256-row loader-wave form (idb_gemm8_kernel_lw: 8 MFMA + 4 DMA-only waves, one workgroup
    per CU); IDB_GEMM8_BIG_TILES=0 forces the 128-row kernel: same fragment mapping and accumulation order, so bit-identical —
    residual, per-sample bias and the GroupNorm-statistics epilogue included."""
    import os
    x = _rand((b, h, h, cin), 20, 2.0).to(eng.tdt)
    wt = _rand((cout, cin, 3, 3) if taps == 9 else (cout, cin, 1, 1), 21, (taps * cin) ** -0.5)
    bias, sb = _rand((cout,), 22), _rand((b, cout), 23)
    res = _rand((b * h * h, cout), 24).to(eng.tdt)
    x_scale = 6.0 / 448.0
    x8 = eng.quantize_fp8(x, x_scale)
    w8, ws = eng.pack_weight_fp8(wt)
    outs = []
    for big in ("512", "0"):
        os.environ["IDB_GEMM8_BIG_TILES"] = big
        try:
            o = eng.gemm_fp8(x8, x_scale, cin, taps, h, h, w8, ws, cout, b, h, h, bias=bias, sbias=(sb, 0, cout), residual=res, gn_stats=32)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("IDB_GEMM8_BIG_TILES", None)
        outs.append((o.clone(), None if getattr(o, "_gn", None) is None else o._gn[0].clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None and outs[1][1] is not None:
        assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-2)
