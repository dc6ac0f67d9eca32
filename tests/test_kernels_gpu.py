"""GPU parity tests of each C-ABI kernel against plain torch fp32 ops on the SAME (operand-dtype-rounded)
inputs.  Tolerances: the kernels accumulate in fp32 and round the output once to bf16 (2^-9 relative) or
f16 (2^-11), so errors are bounded by a few output ulps plus fp32 accumulation-order noise."""
import math

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


def _rt(eng, x):
    """round-trip through the operand dtype (what the kernel sees), back to fp32"""
    return x.to(eng.tdt).float()


def _tol(eng, scale=1.0):
    return (2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9) * scale


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _check(out, ref, tol, what=""):
    err = (out.float() - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err <= tol * max(1.0, mag), f"{what}: max err {err:.4e} vs tol {tol * max(1.0, mag):.4e} (|ref| max {mag:.3f})"


# ---------------------------------------------------------------------------------------------------
# implicit GEMM
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k,tile,split_k", [
    (256, 320, 320, 1, 1), (300, 128, 64, 2, 1), (128, 160, 1280, 3, 4), (77, 256, 1024, 4, 3),
    (1000, 4, 576, 5, 1), (64, 3, 128, 5, 2), (512, 640, 2560, 0, 0), (4096, 960, 320, 0, 0), (130, 1280, 1280, 0, 0),
    (256, 320, 320, 11, 1), (300, 128, 64, 12, 1), (128, 160, 1280, 13, 4), (77, 256, 1024, 14, 3), (700, 640, 128, 11, 1),
    (512, 384, 192, 12, 2), (256, 320, 320, 31, 1), (300, 128, 64, 32, 1), (128, 160, 1280, 33, 4),
    (1000, 4, 576, 35, 1), (700, 640, 128, 31, 2), (130, 256, 64, 14, 1), (200, 320, 704, 21, 1),
    (256, 320, 320, 6, 1), (300, 128, 64, 7, 1), (1000, 480, 192, 6, 1), (700, 640, 1280, 6, 3), (513, 256, 128, 7, 2),
    (256, 320, 320, 8, 1), (300, 128, 64, 9, 1), (300, 128, 64, 19, 1), (700, 384, 640, 17, 1), (700, 320, 640, 16, 1), (300, 480, 1280, 18, 2), (130, 160, 192, 18, 1), (200, 256, 1280, 17, 3), (513, 256, 704, 19, 2), (1000, 480, 192, 8, 2), (70, 640, 1280, 9, 5), (200, 384, 640, 7, 3),
    (256, 320, 320, 41, 1), (300, 128, 64, 42, 1), (5000, 320, 192, 41, 1), (70000, 256, 128, 42, 1), (66000, 960, 320, 41, 1)])
def test_gemm_linear(eng, m, n, k, tile, split_k):
    a = _rand((m, k), 1).to(eng.tdt)
    w = _rand((n, k), 2, k ** -0.5).to(eng.tdt)
    bias = _rand((n,), 3)
    res = _rand((m, n), 4).to(eng.tdt) if n % 4 == 0 else None
    ref = a.float() @ w.float().t() + bias + (res.float() if res is not None else 0)
    out = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=tile, split_k=split_k)
    torch.cuda.synchronize()
    _check(out, ref, _tol(eng), f"linear {m}x{n}x{k}")
    if tile // 10 == 4:
        return                                   # the persistent variant writes operand-dtype outputs only
    out32 = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, out_f32=True, out_scale=0.5, tile=tile, split_k=split_k)
    torch.cuda.synchronize()
    ref32 = 0.5 * (a.float() @ w.float().t()) + bias
    _check(out32, ref32, 2e-5 * math.sqrt(k), "linear f32 out")


def test_gemm_identity_asymmetric(eng):
    """A = I with an asymmetric W catches transposed / permuted fragment layouts exactly."""
    k = n = 128
    a = torch.eye(k, device=DEV).to(eng.tdt)
    w = (torch.arange(n * k, device=DEV).reshape(n, k) % 251).float().to(eng.tdt)
    out = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, k, 1, 1, out_f32=True)
    torch.cuda.synchronize()
    assert torch.equal(out, w.float().t().contiguous())


def test_gemm_geglu(eng):
    m, c = 200, 128
    a = _rand((m, c), 5).to(eng.tdt)
    w = _rand((8 * c, c), 6, c ** -0.5)
    bias = _rand((8 * c,), 7)
    perm = eng._geglu_perm(8 * c).to(DEV)
    wp = eng._pack_mat(w, geglu=True)
    assert torch.equal(wp, w.to(eng.tdt)[perm])
    proj = a.float() @ w.to(eng.tdt).float().t() + bias
    val, gate = proj.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    out = eng.gemm([(a, c, 1, 1, 1, 0)], wp, 8 * c, m, 1, 1, bias=bias[perm].contiguous(), geglu=True)
    torch.cuda.synchronize()
    assert out.shape == (m, 4 * c)
    _check(out, ref, _tol(eng), "geglu")
    out_p = eng.gemm([(a, c, 1, 1, 1, 0)], wp, 8 * c, m, 1, 1, bias=bias[perm].contiguous(), geglu=True, tile=42)   # persistent variant
    torch.cuda.synchronize()
    assert torch.equal(out_p, out)


@pytest.mark.parametrize("b,h,w_,cin,cout,stride,up,tile", [
    (2, 16, 16, 64, 128, 1, 0, 0), (1, 8, 8, 128, 64, 1, 0, 4), (2, 16, 16, 64, 64, 2, 0, 0),
    (1, 8, 8, 128, 128, 1, 1, 0), (2, 13, 11, 64, 320, 1, 0, 1), (1, 32, 32, 320, 320, 1, 0, 0), (3, 8, 8, 192, 4, 1, 0, 0),
    (2, 16, 16, 64, 128, 1, 0, 12), (2, 16, 16, 64, 64, 2, 0, 14), (1, 8, 8, 128, 128, 1, 1, 13), (2, 13, 11, 64, 320, 1, 0, 11),
    (1, 6, 10, 128, 160, 1, 1, 11), (2, 9, 7, 64, 128, 2, 0, 12), (2, 16, 16, 64, 128, 1, 0, 32), (2, 16, 16, 64, 64, 2, 0, 14),
    (1, 8, 8, 128, 128, 1, 1, 33), (2, 13, 11, 64, 320, 1, 0, 31), (3, 8, 8, 192, 4, 1, 0, 35), (1, 32, 32, 320, 320, 1, 0, 31),
    (2, 16, 16, 64, 128, 1, 0, 7), (2, 13, 11, 64, 320, 1, 0, 6), (1, 32, 32, 320, 320, 1, 0, 6), (2, 16, 16, 64, 64, 2, 0, 7),
    (2, 13, 11, 64, 320, 1, 0, 8), (1, 32, 32, 320, 320, 1, 0, 8), (2, 13, 11, 64, 320, 1, 0, 18), (1, 32, 32, 320, 320, 1, 0, 16), (1, 8, 8, 128, 160, 1, 1, 16), (2, 16, 16, 64, 128, 2, 0, 9), (1, 8, 8, 128, 128, 1, 1, 9),
    (1, 8, 8, 128, 160, 1, 1, 6)])
def test_gemm_conv3x3(eng, b, h, w_, cin, cout, stride, up, tile):
    x = _rand((b, cin, h, w_), 10).to(eng.tdt)
    w = _rand((cout, cin, 3, 3), 11, (9 * cin) ** -0.5)
    bias = _rand((cout,), 12)
    sb = _rand((b, cout), 13)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, w.to(eng.tdt).float(), bias, stride=stride, padding=1) + sb[:, :, None, None]
    oh, ow = ref.shape[2], ref.shape[3]
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    wp = eng._pack_conv(w)
    out = eng.gemm([(x_nhwc, cin, 9, h, w_, up)], wp, cout, b, oh, ow, bias=bias, sbias=(sb, 0, cout), stride=stride, tile=tile)
    torch.cuda.synchronize()
    _check(out.view(b, oh, ow, cout).permute(0, 3, 1, 2), ref, _tol(eng), "conv3x3")


def test_gemm_concat_shortcut(eng):
    """conv2(3x3 over n2) + 1x1 shortcut over cat([xa, xb]) fused as K segments."""
    b, h, w_, ca, cb, cout = 2, 8, 8, 128, 64, 128
    n2 = _rand((b, cout, h, w_), 20).to(eng.tdt)
    xa = _rand((b, ca, h, w_), 21).to(eng.tdt)
    xb = _rand((b, cb, h, w_), 22).to(eng.tdt)
    w2 = _rand((cout, cout, 3, 3), 23, (9 * cout) ** -0.5)
    ws = _rand((cout, ca + cb, 1, 1), 24, (ca + cb) ** -0.5)
    bias = _rand((cout,), 25)
    ref = F.conv2d(n2.float(), w2.to(eng.tdt).float(), None, padding=1) + \
        F.conv2d(torch.cat([xa, xb], 1).float(), ws.to(eng.tdt).float(), bias)
    wf = torch.cat([eng._pack_conv(w2), eng._pack_mat(ws.reshape(cout, ca + cb))], dim=1).contiguous()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    out = eng.gemm([(nhwc(n2), cout, 9, h, w_, 0), (nhwc(xa), ca, 1, h, w_, 0), (nhwc(xb), cb, 1, h, w_, 0)], wf, cout, b, h, w_,
                   bias=bias)
    torch.cuda.synchronize()
    _check(out.view(b, h, w_, cout).permute(0, 3, 1, 2), ref, _tol(eng), "concat+shortcut")
    for sk, tile in ((2, 0), (5, 0), (1, 12), (3, 14), (7, 12), (1, 32), (4, 4), (2, 22), (1, 7), (3, 7), (2, 9), (1, 9)):
        out = eng.gemm([(nhwc(n2), cout, 9, h, w_, 0), (nhwc(xa), ca, 1, h, w_, 0), (nhwc(xb), cb, 1, h, w_, 0)], wf, cout, b,
                       h, w_, bias=bias, split_k=sk, tile=tile)
        torch.cuda.synchronize()
        _check(out.view(b, h, w_, cout).permute(0, 3, 1, 2), ref, _tol(eng), f"concat+shortcut splitk={sk} tile={tile}")


def test_gemm_lds_epilogue_equals_direct_epilogue(eng):
    """The LDS-staged coalesced epilogue and the direct one round once from the same fp32 value: bit-identical."""
    for (m, n, k, tile) in ((300, 320, 128, 1), (257, 256, 192, 2), (130, 160, 64, 3), (64, 384, 64, 4), (1000, 32, 128, 5),
                            (600, 320, 128, 6), (300, 256, 64, 7), (300, 320, 128, 8), (257, 256, 192, 9)):
        a = _rand((m, k), 1).to(eng.tdt)
        w = _rand((n, k), 2, k ** -0.5).to(eng.tdt)
        bias, res = _rand((n,), 3), _rand((m, n), 4).to(eng.tdt)
        o1 = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=tile).clone()
        o2 = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=tile, flags=4).clone()
        torch.cuda.synchronize()
        assert torch.equal(o1, o2), (m, n, k, tile)
    c = 128
    a = _rand((200, c), 5).to(eng.tdt)
    wp = eng._pack_mat(_rand((8 * c, c), 6, c ** -0.5), geglu=True)
    bias = _rand((8 * c,), 7)
    o1 = eng.gemm([(a, c, 1, 1, 1, 0)], wp, 8 * c, 200, 1, 1, bias=bias, geglu=True).clone()
    o2 = eng.gemm([(a, c, 1, 1, 1, 0)], wp, 8 * c, 200, 1, 1, bias=bias, geglu=True, flags=4).clone()
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


def test_gemm_fused_splitk_reduce_is_bit_identical_and_stable(eng):
    """In-kernel split-K reduction (agent-scope release + ticket + acquire, last arriver re-sums every slab in fixed
    order) vs the two-launch path (flags bit 3): bit-identical, on every repetition, for bf16/fp32 outputs, residual,
    partial tiles, many splits — with other work in flight so that arrival order varies, and with the reducer's L1/L2
    pre-warmed on the slab addresses (the workspace is re-used across launches)."""
    torch.manual_seed(0)
    noise_a = torch.randn(4096, 1024, device=DEV).to(eng.tdt)
    noise_w = torch.randn(1024, 1024, device=DEV).to(eng.tdt)
    cases = [(512, 1280, 1280, 4, 4), (128, 1280, 5120, 1, 16), (130, 320, 2560, 3, 7), (2048, 640, 2560, 4, 3),
             (300, 4, 1152, 5, 6), (77, 3, 640, 5, 5), (512, 1280, 11520, 1, 30), (200, 264, 1024, 2, 8)]
    for (m, n, k, tile, sk) in cases:
        a = _rand((m, k), 1).to(eng.tdt)
        w = _rand((n, k), 2, k ** -0.5).to(eng.tdt)
        bias = _rand((n,), 3)
        res = _rand((m, n), 4).to(eng.tdt) if n % 8 == 0 else None
        for out_f32 in ((False, True) if res is None else (False,)):
            kw = dict(bias=bias, residual=res, tile=tile, split_k=sk, out_f32=out_f32)
            eng.arena.reset()
            ref = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, flags=8, **kw).clone()
            for rep in range(12):
                eng.arena.reset()
                if rep % 3 == 0:      # unrelated traffic before/around the launch: uneven load, different arrival orders
                    eng.gemm([(noise_a, 1024, 1, 1, 1, 0)], noise_w, 1024, 4096, 1, 1)
                got = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, flags=16, **kw)      # bit 4: fused reduce at any split
                torch.cuda.synchronize()
                assert torch.equal(got, ref), (m, n, k, tile, sk, out_f32, rep)
    assert int(eng._counters.abs().sum().item()) == 0          # tickets are left at zero


def test_gemm_rejects_bad_args(eng):
    from faceposegenerator_amd._lib import IdbError
    a = _rand((64, 100), 1).to(eng.tdt)
    w = _rand((64, 100), 2).to(eng.tdt)
    with pytest.raises(IdbError):
        eng.gemm([(a, 100, 1, 1, 1, 0)], w, 64, 64, 1, 1)


# ---------------------------------------------------------------------------------------------------
# norms / softmax
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,hw,c0,c1,silu,eps", [(2, 256, 64, 0, True, 1e-5), (3, 64, 320, 0, False, 1e-6), (2, 100, 1280, 640, True, 1e-5),
                                                 (1, 4096, 128, 0, True, 1e-6), (2, 64, 1280, 1280, True, 1e-5), (4, 1024, 640, 320, True, 1e-5),
                                                 (2, 1024, 1280, 640, True, 1e-5), (2, 1024, 640, 0, False, 1e-5), (2, 256, 1280, 0, True, 1e-5),
                                                 (3, 16, 64, 0, True, 1e-5), (2, 4, 128, 128, False, 1e-6), (2, 1600, 320, 0, True, 1e-5)])
def test_groupnorm(eng, b, hw, c0, c1, silu, eps):
    x0 = (_rand((b, hw, c0), 30) * 2 + 0.5).to(eng.tdt)
    x1 = (_rand((b, hw, c1), 31) - 1.0).to(eng.tdt) if c1 else None
    c = c0 + c1
    gamma, beta = _rand((c,), 32) * 0.2 + 1, _rand((c,), 33) * 0.1
    xcat = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(xcat.float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    out = eng.groupnorm(x0, c0, x1, c1, b, hw, gamma, beta, eps, silu, groups=32)
    torch.cuda.synchronize()
    _check(out.view(b, hw, c).permute(0, 2, 1), ref, _tol(eng), "groupnorm")


@pytest.mark.parametrize("b,side,cin,cout,split_k,res", [(2, 16, 128, 320, 4, True), (2, 32, 64, 640, 8, False), (1, 8, 256, 1280, 16, True),
                                                         (2, 16, 128, 320, 1, True), (3, 8, 64, 128, 3, False), (2, 64, 64, 320, 2, False),
                                                         # no split, 160-wide tiles: the GEMM's own epilogue emits them (cpg 10 / 20 / 40; 128- and 64-row tiles)
                                                         (8, 64, 64, 320, 1, True), (8, 32, 64, 640, 1, False), (16, 16, 128, 1280, 1, True),
                                                         (1, 64, 64, 320, 1, False), (5, 8, 64, 320, 1, True)])
def test_groupnorm_statistics_from_the_gemm(eng, b, side, cin, cout, split_k, res):
    """idb_gemm_desc.gn_partials: the conv's split-K reduce launch (or, without a split, an extra statistics launch) emits the
    per-(sample, 64-row block, group) sums of its rounded output; idb_groupnorm(partials_in) then only normalises.  The result
    must agree with the ordinary two-pass GroupNorm of the same tensor (same rounded inputs; summation order differs) and with
    torch."""
    g = torch.Generator().manual_seed(17)
    hw = side * side
    x = torch.randn(b * hw, cin, generator=g).to(DEV).to(eng.tdt)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g).to(DEV)
    resid = torch.randn(b * hw, cout, generator=g).to(DEV).to(eng.tdt) if res else None
    gamma, beta = _rand((cout,), 32) * 0.2 + 1, _rand((cout,), 33) * 0.1
    wp = eng._pack_conv(w)
    eng.arena.reset()
    y = eng.gemm([(x, cin, 9, side, side, 0)], wp, cout, b, side, side, bias=bias, residual=resid, split_k=split_k, gn_stats=32)
    if split_k == 1 and getattr(y, "_gn", None) is None:   # neither a reduce launch nor a 160-wide LDS-staged epilogue: the engine does not ask
        eng.arena.free(y)
        y = eng.gemm([(x, cin, 9, side, side, 0)], wp, cout, b, side, side, bias=bias, residual=resid, split_k=1, gn_stats=32,
                     gn_stats_always=True)
    assert getattr(y, "_gn", None) is not None and y._gn[1] == hw // 64
    torch.cuda.synchronize()
    cpg = cout // 32                     # the partials themselves: {sum, sum of squares} of the ROUNDED output per (sample, 64 rows, group)
    yr = y.double().view(b, hw // 64, 64, 32, cpg)
    part = y._gn[0].view(b, hw // 64, 32, 2).double()
    assert torch.allclose(part[..., 0], yr.sum(dim=(2, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1], (yr * yr).sum(dim=(2, 4)), rtol=1e-5, atol=1e-3)
    y_plain = eng.gemm([(x, cin, 9, side, side, 0)], wp, cout, b, side, side, bias=bias, residual=resid, split_k=split_k)
    fused = eng.groupnorm(y, cout, None, 0, b, hw, gamma, beta, 1e-5, True, groups=32)
    assert y._gn is None
    plain = eng.groupnorm(y_plain, cout, None, 0, b, hw, gamma, beta, 1e-5, True, groups=32)
    torch.cuda.synchronize()
    assert torch.equal(y, y_plain)                                   # the reduce+statistics kernel writes the same tensor
    ref = F.silu(F.group_norm(y_plain.float().view(b, hw, cout).permute(0, 2, 1), 32, gamma, beta, 1e-5))
    _check(fused.view(b, hw, cout).permute(0, 2, 1), ref, _tol(eng), "groupnorm(statistics from the GEMM)")
    assert (fused.float() - plain.float()).abs().max().item() <= _tol(eng) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("b,hw,c0,c1", [(2, 4096, 320, 0), (2, 4096, 640, 320), (2, 1024, 1280, 640), (2, 256, 1280, 1280), (1, 4096, 512, 0)])
def test_groupnorm_single_launch_handoff(eng, b, hw, c0, c1):
    """Opt-in single-launch form (workgroups hand partial sums over through self-resetting counters): many launches back to
    back through ONE workspace with different data each time (a stale partial from the previous launch would show as a wrong
    mean/variance), the counters end at zero, and the result agrees with the two-launch form."""
    from faceposegenerator_amd import _lib as L
    c = c0 + c1
    gamma, beta = _rand((c,), 32) * 0.2 + 1, _rand((c,), 33) * 0.1
    sync = torch.zeros(4096, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def gn(x0, x1, counters):
        out = torch.empty((b * hw, c), dtype=eng.tdt, device=DEV)
        L.check(eng.lib.idb_groupnorm(x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, b, hw, 32, 1e-5, gamma.data_ptr(),
                                      beta.data_ptr(), 1, out.data_ptr(), eng.dt, eng._gn_ws.data_ptr(), eng._gn_ws.numel(),
                                      None if counters is None else counters.data_ptr(), 0 if counters is None else counters.numel(),
                                      None, 0, st))
        return out

    runs = []
    for i in range(12):
        x0 = (_rand((b, hw, c0), 300 + i) * (1 + i) + 0.5 * i).to(eng.tdt)
        x1 = (_rand((b, hw, c1), 400 + i) - 1.0 * i).to(eng.tdt) if c1 else None
        runs.append((x0, x1, gn(x0, x1, sync)))
    torch.cuda.synchronize()
    assert int(sync.abs().sum().item()) == 0
    for x0, x1, out in runs:
        xcat = torch.cat([x0, x1], -1) if c1 else x0
        ref = F.silu(F.group_norm(xcat.float().permute(0, 2, 1), 32, gamma, beta, 1e-5))
        _check(out.view(b, hw, c).permute(0, 2, 1), ref, _tol(eng), "groupnorm(single launch)")
    x0, x1, out = runs[-1]
    out2 = gn(x0, x1, None)
    torch.cuda.synchronize()
    assert (out.float() - out2.float()).abs().max().item() <= _tol(eng) * max(1.0, out2.float().abs().max().item())


@pytest.mark.parametrize("rows,c", [(1000, 320), (77, 640), (513, 1280), (64, 64)])
def test_layernorm(eng, rows, c):
    x = (_rand((rows, c), 40) * 3 + 1).to(eng.tdt)
    gamma, beta = _rand((c,), 41) * 0.2 + 1, _rand((c,), 42) * 0.1
    ref = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5)
    out = eng.layernorm(x, rows, c, gamma, beta)
    torch.cuda.synchronize()
    _check(out, ref, _tol(eng), "layernorm")


def test_softmax_rows(eng):
    x = (_rand((300, 4096), 50) * 4).to(eng.tdt)
    ref = torch.softmax(x.float(), -1)
    from faceposegenerator_amd import _lib as L
    L.check(eng.lib.idb_softmax_rows(x.data_ptr(), 300, 4096, eng.dt, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    _check(x, ref, _tol(eng, 0.5), "softmax")      # output values up to ~1 rounded once to the operand dtype


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------
# (2,10,1024), (3,7,1000): grids of 128-256 workgroups -> the 8-wave key-split form (wave pairs merge through LDS);
# (2,5,4096), (2,11,1600): 257-511 workgroups of 128 rows -> the 12-wave 192-row form
@pytest.mark.parametrize("b,heads,n", [(2, 5, 1024), (1, 2, 4096), (3, 4, 64), (2, 20, 256), (1, 3, 144), (2, 5, 4096), (2, 10, 1024), (3, 7, 1000), (2, 11, 1600)])
def test_self_attention(eng, b, heads, n):
    c = heads * 64
    qkv = _rand((b * n, 3 * c), 60).to(eng.tdt)
    q, k, v = [t.float().view(b, n, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b * n, c)
    p = qkv.data_ptr()
    out = eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, b, heads, n, n, n)
    torch.cuda.synchronize()
    _check(out, ref, _tol(eng), "self-attention")


def test_attention_peaked_softmax(eng):
    """One dominant key per query (forces the online-softmax running max to jump between tiles)."""
    _peaked(eng, 1, 2, 512)
    _peaked(eng, 2, 10, 1024)            # key-split form: the dominant key of a query sits in either wave's half of a tile


def _peaked(eng, b, heads, n):
    c = heads * 64
    g = torch.Generator().manual_seed(61)
    q = torch.randn(b * n, c, generator=g)
    k = torch.randn(b * n, c, generator=g)
    idx = torch.cat([torch.randperm(n, generator=g) + j * n for j in range(b)])
    k[idx] += 3.0 * q                      # key idx[i] aligned with query i -> large logit late/early in the sweep
    v = torch.randn(b * n, c, generator=g)
    qkv = torch.cat([q, k, v], -1).to(DEV).to(eng.tdt)
    qq, kk, vv = [t.float().view(b, n, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(b * n, c)
    p = qkv.data_ptr()
    out = eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, b, heads, n, n, n)
    torch.cuda.synchronize()
    _check(out, ref, _tol(eng), "peaked attention")


@pytest.mark.parametrize("b,heads,n,n_ctx", [(2, 5, 1024, 77), (2, 20, 64, 77), (1, 10, 256, 64), (1, 1, 128, 130), (2, 10, 1024, 600), (2, 10, 1024, 545)])
def test_cross_attention(eng, b, heads, n, n_ctx):
    c = heads * 64
    qm = _rand((b * n, c), 70).to(eng.tdt)
    kv = _rand((b * n_ctx, 2 * c), 71).to(eng.tdt)
    q = qm.float().view(b, n, heads, 64).transpose(1, 2)
    k, v = [t.float().view(b, n_ctx, heads, 64).transpose(1, 2) for t in kv.chunk(2, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b * n, c)
    p = kv.data_ptr()
    out = eng.attention(qm, c, p, p + 2 * c, 2 * c, b, heads, n, n_ctx, n_ctx)
    torch.cuda.synchronize()
    _check(out, ref, _tol(eng), "cross-attention")


# ---------------------------------------------------------------------------------------------------
# small fp32 kernels
# ---------------------------------------------------------------------------------------------------
def test_time_embedding_path(eng):
    from faceposegenerator_amd import _lib as L
    from oracle import sd21_oracle as O
    st = torch.cuda.current_stream().cuda_stream
    ts = torch.tensor([958.0, 925.0, 34.0, 1.0, 0.0], device=DEV)
    out = torch.empty((5, 320), device=DEV)
    L.check(eng.lib.idb_timestep_sinusoid(ts.data_ptr(), out.data_ptr(), 5, 320, st))
    ref = O.timestep_embedding(ts.cpu(), 320)
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 2e-4           # fp32 sin/cos of arguments up to ~1e3
    x, w, b = _rand((30, 320), 80), _rand((1280, 320), 81, 320 ** -0.5), _rand((1280,), 82)
    y = torch.empty((30, 1280), device=DEV)
    for silu in (0, 1):
        L.check(eng.lib.idb_linear_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 30, 1280, 320, silu, st))
        torch.cuda.synchronize()
        ref = F.linear(F.silu(x) if silu else x, w, b)
        assert (y - ref).abs().max().item() < 1e-4


def test_conv_in(eng):
    from faceposegenerator_amd import _lib as L
    b, h, w_, cout = 2, 16, 16, 64
    x = _rand((b, 4, h, w_), 90)
    w, bias = _rand((cout, 4, 3, 3), 91, 1 / 6.0), _rand((cout,), 92)
    pw, pb = _rand((4, 4), 93, 0.5), _rand((4,), 94)
    out = torch.empty((2 * b * h * w_, cout), dtype=eng.tdt, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.check(eng.lib.idb_conv_in(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), b, 2, 4, h, w_, cout, 1.0, None, None,
                                eng.dt, st))
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, bias, padding=1)
    got = out.view(2, b, h, w_, cout).permute(0, 1, 4, 2, 3)
    _check(got[0], ref, _tol(eng), "conv_in")
    assert torch.equal(got[0], got[1])
    L.check(eng.lib.idb_conv_in(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), b, 1, 4, h, w_, cout, 1 / 0.18215,
                                pw.data_ptr(), pb.data_ptr(), eng.dt, st))
    torch.cuda.synchronize()
    ref = F.conv2d(F.conv2d(x / 0.18215, pw.view(4, 4, 1, 1), pb), w, bias, padding=1)
    _check(out[: b * h * w_].view(b, h, w_, cout).permute(0, 3, 1, 2), ref, _tol(eng, 4.0), "conv_in + post_quant")


@pytest.mark.parametrize("b,h,w_,cout,rep,scale", [(4, 64, 64, 320, 2, 1.0), (2, 128, 64, 320, 1, 0.5), (3, 96, 96, 128, 2, 1.0), (4, 64, 65, 64, 1, 1.0)])
def test_conv_in_large(eng, b, h, w_, cout, rep, scale):
    """conv_in at the sizes of the large-batch runs (full 320 channels, CFG duplication, input scale, odd width)."""
    from faceposegenerator_amd import _lib as L
    x = _rand((b, 4, h, w_), 95, 3.0)
    w, bias = _rand((cout, 4, 3, 3), 96, 1 / 6.0), _rand((cout,), 97)
    out = torch.full((rep * b * h * w_, cout), float("nan"), dtype=eng.tdt, device=DEV)
    L.check(eng.lib.idb_conv_in(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), b, rep, 4, h, w_, cout, scale, None, None,
                                eng.dt, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = F.conv2d(x * scale, w, bias, padding=1)
    got = out.view(rep, b, h, w_, cout).permute(0, 1, 4, 2, 3)
    _check(got[0], ref, _tol(eng), "conv_in (large)")
    for r in range(1, rep):
        assert torch.equal(got[0], got[r])


def test_cfg_ddpm_step_and_scheduler(eng):
    from faceposegenerator_amd.scheduler import DDPMScheduler
    from oracle import sd21_oracle as O
    sch = DDPMScheduler()
    sch.set_timesteps(30)
    ac, ts = O.ddpm_tables(), O.ddpm_timesteps(30)
    assert sch.timesteps.tolist() == ts
    b, hw = 2, 64
    for t in (958, 496, 1):
        eps_u, eps_c = _rand((b, 4, 8, 8), 100 + t), _rand((b, 4, 8, 8), 101 + t)
        x, nz = _rand((b, 4, 8, 8), 102 + t), _rand((b, 4, 8, 8), 103 + t)
        ref_prev, ref_x0 = O.ddpm_step(ac, ts, t, (eps_u + 5.0 * (eps_c - eps_u)).cpu(), x.cpu(), nz.cpu())
        # scheduler API (no CFG)
        out = sch.step((eps_u + 5.0 * (eps_c - eps_u)), t, x, variance_noise=nz)
        assert (out.prev_sample.cpu() - ref_prev).abs().max().item() < 2e-5 * max(1.0, ref_prev.abs().max().item())
        assert (out.pred_original_sample.cpu() - ref_x0).abs().max().item() < 2e-5 * max(1.0, ref_x0.abs().max().item())
        # fused CFG form used by the sampling loop
        from faceposegenerator_amd import _lib as L
        eps = torch.cat([eps_u, eps_c]).permute(0, 2, 3, 1).contiguous()
        lat = x.clone()
        coef = torch.tensor(list(sch.step_coefficients(t)) + [5.0], device=DEV)
        L.check(eng.lib.idb_cfg_ddpm_step(eps.data_ptr(), lat.data_ptr(), nz.data_ptr(), coef.data_ptr(), None, b, 4, hw, 1, 0,
                                          torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert (lat.cpu() - ref_prev).abs().max().item() < 2e-5 * max(1.0, ref_prev.abs().max().item())


def test_postprocess(eng):
    from faceposegenerator_amd import _lib as L
    from oracle import sd21_oracle as O
    x = _rand((2, 16, 16, 3), 110) * 1.5
    img = torch.empty_like(x)
    u8 = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    L.check(eng.lib.idb_postprocess(x.data_ptr(), img.data_ptr(), u8.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = O.postprocess_np(x.cpu().permute(0, 3, 1, 2))
    assert torch.equal(img.cpu(), ref)
    assert torch.equal(u8.cpu(), O.to_uint8(ref.clone()))


def test_lora_merge_and_pack(eng):
    from faceposegenerator_amd import _lib as L
    rows, cols, r = 128, 192, 4
    w, a, b = _rand((rows, cols), 120), _rand((r, cols), 121), _rand((rows, r), 122)
    dst = torch.empty((rows, cols), dtype=eng.tdt, device=DEV)
    L.check(eng.lib.idb_lora_merge(w.data_ptr(), a.data_ptr(), b.data_ptr(), dst.data_ptr(), rows, cols, r, 0.5, eng.dt,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = (w + 0.5 * b @ a)
    assert (dst.float() - ref).abs().max().item() <= _tol(eng) * ref.abs().max().item()
    cw = _rand((8, 64, 3, 3), 123)
    packed = eng._pack_conv(cw)
    torch.cuda.synchronize()
    assert torch.equal(packed, cw.permute(0, 2, 3, 1).reshape(8, -1).to(eng.tdt))


def test_gemm_epilogue_emits_groupnorm_partials_only_for_whole_groups(eng):
    """idb_gemm_emits_gn_partials: 2 = from the LDS-staged epilogue (no split, 160-wide tile, whole groups per column tile),
    1 = from the split-K reduce launch, 0 = it would take an extra launch (128-wide tiles cut groups of 10 / 20 / 40 channels)."""
    import ctypes as C
    from faceposegenerator_amd import _lib as L
    x = torch.zeros(64, device=DEV)

    def mode(m_hw, batch, cin, taps, n, tile=0):
        d = L.GemmDesc()
        side = int(m_hw ** 0.5)
        d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = eng.dt, batch, side, side, 1, n, 1
        d.src[0].ptr, d.src[0].channels, d.src[0].taps, d.src[0].in_h, d.src[0].in_w = x.data_ptr(), cin, taps, side, side
        d.w, d.out, d.out_dtype, d.out_ld, d.tile = x.data_ptr(), x.data_ptr(), eng.dt, n, tile
        return eng.lib.idb_gemm_emits_gn_partials(C.byref(d), 32)

    assert mode(4096, 128, 320, 9, 320) == 2          # batch-64 conv: 128x160, no split
    assert mode(4096, 128, 320, 1, 320) == 2          # proj_out
    assert mode(4096, 2, 320, 9, 320) == 2            # the same conv at batch 1: 64x160 tiles, still no split
    assert mode(256, 2, 1280, 9, 1280) == 1           # batch-1 conv at 16x16: split-K, statistics from the reduce launch
    assert mode(4096, 128, 320, 9, 320, tile=9) == 0  # 128x128 tiles cut the groups
    assert mode(4096, 128, 320, 9, 336) == 0          # 336 / 32 is not an integer group width


# ---------------------------------------------------------------------------------------------------
# K-tiled weight layout (idb_tile_weight, idb_gemm_desc.w_layout = 1) and the folded-LayerNorm vectors
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k,tile,split_k", [
    (256, 320, 320, 1, 1), (128, 160, 1280, 3, 4), (77, 256, 1024, 4, 3), (1000, 4, 576, 5, 1), (64, 3, 128, 5, 2),
    (512, 640, 2560, 0, 0), (4096, 960, 320, 0, 0), (130, 1280, 1280, 0, 0), (128, 1280, 11520, 0, 0), (700, 640, 128, 11, 1),
    (513, 250, 704, 19, 2), (300, 480, 1280, 18, 2), (200, 384, 640, 7, 3), (700, 320, 640, 16, 1), (256, 320, 320, 31, 1),
    (5000, 320, 192, 41, 1), (66000, 960, 320, 41, 1), (300, 136, 64, 42, 1)])
def test_gemm_tiled_weight_layout(eng, m, n, k, tile, split_k):
    """The same GEMM from [n][K] rows and from the 16-row K-tiled blocks: bit-identical (only the addresses differ)."""
    a = _rand((m, k), 21).to(eng.tdt)
    w = _rand((n, k), 22, k ** -0.5).to(eng.tdt)
    bias = _rand((n,), 23)
    wt = eng.tile_weight(w)
    assert wt.numel() == (n + 15) // 16 * 16 * k
    blocks = wt.view((n + 15) // 16, k // 64, 16, 64)
    wpad = torch.zeros(((n + 15) // 16 * 16, k), dtype=eng.tdt, device=DEV)
    wpad[:n] = w
    assert torch.equal(blocks.permute(0, 2, 1, 3).reshape(-1, k), wpad)          # the layout itself, incl. zero rows past n
    ref = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, tile=tile, split_k=split_k, out_f32=(tile // 10 != 4))
    out = eng.gemm([(a, k, 1, 1, 1, 0)], wt, n, m, 1, 1, bias=bias, tile=tile, split_k=split_k, out_f32=(tile // 10 != 4))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_conv_tiled_weight_layout(eng):
    b, h, cin, cout = 2, 16, 128, 320
    x = _rand((b, h, h, cin), 31).to(eng.tdt)
    wc = _rand((cout, cin, 3, 3), 32, (9 * cin) ** -0.5)
    w = eng._pack_conv(wc)
    for tile in (0, 8, 6, 13):
        ref = eng.gemm([(x, cin, 9, h, h, 0)], w, cout, b, h, h, tile=tile)
        out = eng.gemm([(x, cin, 9, h, h, 0)], eng.tile_weight(w), cout, b, h, h, tile=tile)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("rows,cols,rank,geglu", [(320, 320, 4, 0), (960, 320, 0, 0), (2560, 320, 0, 1), (1280, 1280, 8, 0)])
def test_ln_fold_vectors(eng, rows, cols, rank, geglu):
    w = _rand((rows, cols), 41, cols ** -0.5)
    gamma, beta, bias = 1.0 + 0.2 * _rand((cols,), 42), _rand((cols,), 43), _rand((rows,), 44)
    la = _rand((rank, cols), 45, 0.25) if rank else None
    lb = _rand((rows, rank), 46, 0.02) if rank else None
    sc = 0.75
    perm = eng._geglu_perm(rows).to(DEV) if geglu else torch.arange(rows, device=DEV)
    merged = w.double() + (sc * (lb.double() @ la.double()) if rank else 0.0)
    wq = torch.empty((rows, cols), dtype=eng.tdt, device=DEV)
    if geglu:
        from faceposegenerator_amd import _lib as L
        L.check(eng.lib.idb_pack_matrix_scaled(w.data_ptr(), wq.data_ptr(), rows, cols, 1, gamma.data_ptr(), eng.dt, 0), "pack_scaled")
        # one rounding of the exact product (v_fma_mix) vs torch's fp32 product rounded again: equal up to one output ulp
        ref_q = (w.double() * gamma.double()[None, :])[perm]
        assert ((wq.double() - ref_q).abs() <= (2.0 ** -8 if eng.dtype_name == "bf16" else 2.0 ** -11) * ref_q.abs() + 6e-8).all()      # + the f16 sub-normal step 2^-24
    else:
        wq.copy_((merged * gamma.double()[None, :]).float().to(eng.tdt))
    u = torch.empty((rows,), dtype=torch.float32, device=DEV)
    v = torch.empty((rows,), dtype=torch.float32, device=DEV)
    from faceposegenerator_amd import _lib as L
    L.check(eng.lib.idb_ln_fold_vectors(w.data_ptr(), None if la is None else la.data_ptr(), None if lb is None else lb.data_ptr(), rank, sc,
                                        wq.data_ptr(), beta.data_ptr(), bias.data_ptr(), u.data_ptr(), v.data_ptr(), rows, cols, geglu, eng.dt, 0),
            "idb_ln_fold_vectors")
    torch.cuda.synchronize()
    u_ref = wq.double().sum(dim=1)
    v_ref = (merged @ beta.double())[perm] + bias.double()
    assert (u.double() - u_ref).abs().max().item() < 2e-5 * max(1.0, u_ref.abs().max().item())
    assert (v.double() - v_ref).abs().max().item() < 2e-5 * max(1.0, v_ref.abs().max().item())


# ---------------------------------------------------------------------------------------------------
# loader-wave variant (idb_gemm_kernel_lw: tile ids 5x / 6x / 7x): same fragment mapping and accumulation order as the ring-3
# kernels of the same tile, so outputs must be BIT-IDENTICAL — over ragged M / N, split-K, conv taps, strides, upsampling,
# K-segment concatenation, folded LayerNorm and the GroupNorm-statistics epilogue
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [5, 6, 7])
@pytest.mark.parametrize("m,n,k,shape,split_k", [
    (256, 320, 320, 6, 1), (300, 128, 64, 7, 1), (700, 640, 1280, 6, 3), (513, 250, 704, 9, 2), (1000, 480, 192, 8, 2),
    (70, 640, 1280, 9, 5), (130, 160, 192, 8, 1), (77, 256, 1024, 4, 3), (8192, 320, 2880, 6, 1), (128, 1280, 11520, 7, 16),
    (2048, 640, 5760, 8, 4), (64, 64, 64, 4, 1)])
def test_gemm_loader_waves_bit_identical(eng, variant, m, n, k, shape, split_k):
    a = _rand((m, k), 51).to(eng.tdt)
    w = _rand((n, k), 52, k ** -0.5).to(eng.tdt)
    bias = _rand((n,), 53)
    res = _rand((m, n), 54).to(eng.tdt) if n % 8 == 0 else None
    ref = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=10 + shape, split_k=split_k)
    out = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=10 * variant + shape, split_k=split_k)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    _check(out, a.float() @ w.float().t() + bias + (res.float() if res is not None else 0), _tol(eng), "lw linear")
    wt = eng.tile_weight(w)
    out_t = eng.gemm([(a, k, 1, 1, 1, 0)], wt, n, m, 1, 1, bias=bias, residual=res, tile=10 * variant + shape, split_k=split_k)
    torch.cuda.synchronize()
    assert torch.equal(out_t, ref)


@pytest.mark.parametrize("variant", [5, 6, 7])
@pytest.mark.parametrize("b,h,w_,cin,cout,stride,up,shape", [
    (2, 16, 16, 64, 128, 1, 0, 7), (2, 13, 11, 64, 320, 1, 0, 6), (1, 32, 32, 320, 320, 1, 0, 8), (2, 16, 16, 64, 64, 2, 0, 7),
    (1, 8, 8, 128, 160, 1, 1, 6), (2, 9, 7, 64, 128, 2, 0, 9), (3, 8, 8, 192, 64, 1, 0, 4)])
def test_conv_loader_waves_bit_identical(eng, variant, b, h, w_, cin, cout, stride, up, shape):
    x = _rand((b, h, w_, cin), 61).to(eng.tdt)
    wc = _rand((cout, cin, 3, 3), 62, (9 * cin) ** -0.5)
    w = eng._pack_conv(wc)
    bias = _rand((cout,), 63)
    oh, ow = ((h << up) + stride - 1) // stride, ((w_ << up) + stride - 1) // stride
    ref = eng.gemm([(x, cin, 9, h, w_, up)], w, cout, b, oh, ow, bias=bias, stride=stride, tile=10 + shape)
    out = eng.gemm([(x, cin, 9, h, w_, up)], w, cout, b, oh, ow, bias=bias, stride=stride, tile=10 * variant + shape)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    xin = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest") if up else x.float().permute(0, 3, 1, 2)
    want = F.conv2d(xin, wc.to(eng.tdt).float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    _check(out, want, _tol(eng), "lw conv")


def test_resnet_conv2_with_shortcut_loader_waves(eng):
    """conv2 + 1x1 shortcut over a skip concatenation as ONE GEMM with three K segments, plus the GroupNorm-statistics epilogue."""
    b, h, c1, c2, cout = 2, 16, 128, 64, 320
    n2 = _rand((b, h, h, cout), 71).to(eng.tdt)
    xa, xb = _rand((b, h, h, c1), 72).to(eng.tdt), _rand((b, h, h, c2), 73).to(eng.tdt)
    w = _rand((cout, 9 * cout + c1 + c2), 74, (9 * cout) ** -0.5).to(eng.tdt)
    bias = _rand((cout,), 75)
    srcs = [(n2, cout, 9, h, h, 0), (xa, c1, 1, h, h, 0), (xb, c2, 1, h, h, 0)]
    ref = eng.gemm(srcs, w, cout, b, h, h, bias=bias, tile=16, gn_stats=32)
    out = eng.gemm(srcs, w, cout, b, h, h, bias=bias, tile=56, gn_stats=32)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert (getattr(ref, "_gn", None) is None) == (getattr(out, "_gn", None) is None)
    if getattr(ref, "_gn", None) is not None:
        assert torch.equal(ref._gn[0], out._gn[0])


@pytest.mark.parametrize("m,n,k,shape,split_k", [(4096, 320, 320, 8, 1), (1000, 480, 704, 8, 1), (700, 256, 1280, 9, 2), (513, 250, 192, 9, 1),
                                                 (256, 160, 64, 8, 1), (5000, 640, 2560, 8, 3)])
def test_gemm_256_row_loader_wave_tiles_bit_identical(eng, m, n, k, shape, split_k):
    """tile ids 88 / 89: 256x160 / 256x128 with 8 MFMA waves + 4 loader waves, one workgroup per CU (all 160 KB of LDS)."""
    a = _rand((m, k), 81).to(eng.tdt)
    w = eng.tile_weight(_rand((n, k), 82, k ** -0.5).to(eng.tdt))
    bias = _rand((n,), 83)
    res = _rand((m, n), 84).to(eng.tdt) if n % 8 == 0 else None
    ref = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=shape, split_k=split_k)
    out = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, residual=res, tile=80 + shape, split_k=split_k)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_conv_256_row_loader_wave_tile_bit_identical_with_gn_partials(eng):
    b, h, cin, cout = 4, 32, 128, 320
    x = _rand((b, h, h, cin), 91).to(eng.tdt)
    w = eng.tile_weight(eng._pack_conv(_rand((cout, cin, 3, 3), 92, (9 * cin) ** -0.5)))
    bias = _rand((cout,), 93)
    ref = eng.gemm([(x, cin, 9, h, h, 0)], w, cout, b, h, h, bias=bias, tile=8, gn_stats=32)
    out = eng.gemm([(x, cin, 9, h, h, 0)], w, cout, b, h, h, bias=bias, tile=88, gn_stats=32)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert getattr(out, "_gn", None) is not None and torch.allclose(ref._gn[0], out._gn[0], rtol=1e-5, atol=1e-3)


# ---------------------------------------------------------------------------------------------------
# Patch-resident 3x3 conv (tile ids 98 / 99, idb_conv_patch_kernel): the halo patch of a 256-pixel tile loaded once per 64-channel
# chunk, K walked chunk-major.  Different accumulation ORDER than the tap-major kernels, so: equal to the fp32 reference within the
# operand tolerance, and within a few ulp of the tap-major 256-row tile (tile ids 88 / 89) — not bit-identical
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,h,w_,cin,cout,shape", [
    (2, 64, 64, 64, 320, 8),      # 4 image rows per tile, 2 column tiles
    (1, 64, 64, 320, 160, 8),     # 5 chunks x 9 taps: the double-buffered patch wraps
    (4, 32, 32, 128, 256, 9),     # 8 image rows per tile, 256x128 tile
    (3, 16, 16, 192, 320, 8),     # one whole image per tile
    (8, 8, 8, 128, 160, 8),       # FOUR whole images per tile (halo rows between them)
    (4, 8, 8, 64, 100, 9),        # ragged column tile (n = 100)
    (2, 32, 16, 64, 128, 9)])     # non-square image: 16 rows of width 16 per tile, half an image
def test_conv_patch_resident_tiles(eng, b, h, w_, cin, cout, shape):
    x = _rand((b, h, w_, cin), 131).to(eng.tdt)
    wc = _rand((cout, cin, 3, 3), 132, (9 * cin) ** -0.5)
    w = eng.tile_weight(eng._pack_conv(wc))
    bias = _rand((cout,), 133)
    res = _rand((b * h * w_, cout), 134).to(eng.tdt) if cout % 8 == 0 else None
    ref = eng.gemm([(x, cin, 9, h, w_, 0)], w, cout, b, h, w_, bias=bias, residual=res, tile=80 + shape)
    out = eng.gemm([(x, cin, 9, h, w_, 0)], w, cout, b, h, w_, bias=bias, residual=res, tile=90 + shape)
    torch.cuda.synchronize()
    want = F.conv2d(x.float().permute(0, 3, 1, 2), wc.to(eng.tdt).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    if res is not None:
        want = want + res.float()
    _check(out, want, _tol(eng), "patch conv")
    ulp = 2.0 ** (-10 if eng.tdt == torch.float16 else -7)
    assert (out.float() - ref.float()).abs().max().item() <= 2 * ulp * max(1.0, want.abs().max().item())
    # the same through rows-layout weights
    out_r = eng.gemm([(x, cin, 9, h, w_, 0)], eng._pack_conv(wc), cout, b, h, w_, bias=bias, residual=res, tile=90 + shape)
    torch.cuda.synchronize()
    assert torch.equal(out_r, out)


@pytest.mark.parametrize("b,h,w_,cin,cout,shape,split_k", [
    (2, 64, 64, 64, 320, 6, 1),      # 64x160 tile = ONE image row, 3-row patch
    (2, 64, 64, 192, 160, 8, 1),     # 128x160 tile = two image rows
    (2, 32, 32, 128, 256, 7, 2),     # split-K: a split may start inside a chunk
    (2, 16, 16, 320, 320, 8, 4),     # 45 K-steps over 4 splits: splits start mid-chunk
    (2, 8, 8, 128, 160, 6, 2),       # 64 rows = one whole 8x8 image
    (4, 8, 8, 192, 320, 8, 3),       # 128 rows = two whole images
    (1, 16, 16, 192, 64, 4, 3),      # 64x64 tile
    (2, 32, 32, 192, 256, 9, 1)])    # 128x128 tile
def test_conv_patch_resident_small_tiles_and_split_k(eng, b, h, w_, cin, cout, shape, split_k):
    """tile ids 10x: the patch-resident conv on the 64- / 128-row tiles of the one-workgroup-per-CU plans, split-K over K-steps of the chunk-major walk."""
    x = _rand((b, h, w_, cin), 151).to(eng.tdt)
    wc = _rand((cout, cin, 3, 3), 152, (9 * cin) ** -0.5)
    w = eng.tile_weight(eng._pack_conv(wc))
    bias = _rand((cout,), 153)
    res = _rand((b * h * w_, cout), 154).to(eng.tdt)
    ref = eng.gemm([(x, cin, 9, h, w_, 0)], w, cout, b, h, w_, bias=bias, residual=res, tile=70 + shape, split_k=split_k, gn_stats=32 if cout % 32 == 0 else 0)
    out = eng.gemm([(x, cin, 9, h, w_, 0)], w, cout, b, h, w_, bias=bias, residual=res, tile=100 + shape, split_k=split_k, gn_stats=32 if cout % 32 == 0 else 0)
    torch.cuda.synchronize()
    want = F.conv2d(x.float().permute(0, 3, 1, 2), wc.to(eng.tdt).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res.float()
    _check(out, want, _tol(eng), "patch conv, small tile")
    ulp = 2.0 ** (-10 if eng.tdt == torch.float16 else -7)
    assert (out.float() - ref.float()).abs().max().item() <= 2 * ulp * max(1.0, want.abs().max().item())
    assert (getattr(ref, "_gn", None) is None) == (getattr(out, "_gn", None) is None)
    if getattr(ref, "_gn", None) is not None:
        assert torch.allclose(ref._gn[0], out._gn[0], rtol=2e-3, atol=0.5)


def test_conv_patch_resident_with_shortcut_segments_and_gn_partials(eng):
    """conv2 + 1x1 shortcut over a skip concatenation: a 9-tap segment followed by two halo-less 1-tap segments; GroupNorm-statistics
    epilogue and a per-sample bias ride along."""
    b, h, c1, c2, cout = 4, 32, 128, 64, 320
    n2 = _rand((b, h, h, cout), 141).to(eng.tdt)
    xa, xb = _rand((b, h, h, c1), 142).to(eng.tdt), _rand((b, h, h, c2), 143).to(eng.tdt)
    wfull = _rand((cout, 9 * cout + c1 + c2), 144, (9 * cout) ** -0.5).to(eng.tdt)
    w = eng.tile_weight(wfull)
    bias = _rand((cout,), 145)
    srcs = [(n2, cout, 9, h, h, 0), (xa, c1, 1, h, h, 0), (xb, c2, 1, h, h, 0)]
    ref = eng.gemm(srcs, w, cout, b, h, h, bias=bias, tile=88, gn_stats=32)
    out = eng.gemm(srcs, w, cout, b, h, h, bias=bias, tile=98, gn_stats=32)
    torch.cuda.synchronize()
    ulp = 2.0 ** (-10 if eng.tdt == torch.float16 else -7)
    assert (out.float() - ref.float()).abs().max().item() <= 2 * ulp * max(1.0, ref.float().abs().max().item())
    assert getattr(out, "_gn", None) is not None and getattr(ref, "_gn", None) is not None
    assert torch.allclose(ref._gn[0], out._gn[0], rtol=2e-3, atol=0.5)
    # against fp32: conv over the packed [cout][tap][cin] block + the two 1x1 blocks
    wc = wfull[:, :9 * cout].float().reshape(cout, 3, 3, cout).permute(0, 3, 1, 2)
    want = F.conv2d(n2.float().permute(0, 3, 1, 2), wc, bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    want = want + xa.float().reshape(-1, c1) @ wfull[:, 9 * cout:9 * cout + c1].float().t() + xb.float().reshape(-1, c2) @ wfull[:, 9 * cout + c1:].float().t()
    _check(out, want, _tol(eng), "patch conv + shortcut segments")


def test_conv_patch_planner_and_refusals(eng):
    """auto plan: a large-grid 3x3 stride-1 conv takes the patch-resident tile (98); what it cannot run stays on the tap-major tile (auto)
    or is refused (forced)."""
    import ctypes as C
    from faceposegenerator_amd import _lib as L

    def plan(b, h, cin, cout, stride=1, up=0, taps=9, tile=0):
        x = torch.empty((b, h >> up, h >> up, cin), dtype=eng.tdt, device="cuda")
        w = torch.empty((cout, taps * cin), dtype=eng.tdt, device="cuda")
        out = torch.empty((b * (h // stride) ** 2, cout), dtype=eng.tdt, device="cuda")
        d = L.GemmDesc()
        d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = eng.dt, b, h // stride, h // stride, stride, cout, 1
        d.src[0].ptr, d.src[0].channels, d.src[0].taps, d.src[0].in_h, d.src[0].in_w, d.src[0].upsample = x.data_ptr(), cin, taps, h >> up, h >> up, up
        d.w, d.out, d.out_dtype, d.out_ld, d.tile = w.data_ptr(), out.data_ptr(), eng.dt, cout, tile
        t, sk, bl = C.c_int32(), C.c_int32(), C.c_int32()
        rc = eng.lib.idb_gemm_plan(C.byref(d), C.byref(t), C.byref(sk), C.byref(bl))
        return rc, t.value, sk.value

    assert plan(128, 64, 320, 320) == (0, 98, 1)                 # configs[2]'s conv 320->320 @64x64 (B_eff 128)
    assert plan(128, 16, 1280, 1280) == (0, 98, 1)
    assert plan(128, 64, 320, 320, stride=2)[1] == 88            # Downsample2D: tap-major
    assert plan(128, 64, 640, 640, up=1)[1] == 88                # Upsample2D conv: tap-major
    assert plan(128, 64, 1280, 320, taps=1)[1] == 88             # K = 1280 projection
    assert plan(2, 64, 320, 320)[1] // 10 != 9                   # batch 1: one workgroup per CU plans
    assert plan(128, 64, 320, 320, stride=2, tile=98)[0] == -2
    assert plan(3, 8, 320, 320, tile=98)[0] == -2   # 192 pixels: not whole tiles


# ---------------------------------------------------------------------------------------------------
# GroupNorm(+SiLU) applied by normalizer waves inside the conv (idb_gemm_desc.gn_in_*): the transform is gn_apply_kernel's
# arithmetic on the same partial sums, so the result must equal idb_groupnorm(partials_in) + idb_gemm BIT FOR BIT
# ---------------------------------------------------------------------------------------------------
def _gn_ref_and_fused(eng, srcs_raw, c_norm, nsrc_norm, w, cout, b, h, w_, tile, silu, eps, seed, extra=None, split_k=0, ref_tile=None, **kw):
    """srcs_raw: [(tensor [b,h,w,c], c, taps)]; the first nsrc_norm are normalised as one GroupNorm(32) over their concatenation."""
    G = 32
    gamma, beta = 1.0 + 0.3 * _rand((c_norm,), seed), 0.2 * _rand((c_norm,), seed + 1)
    x0, c0 = srcs_raw[0][0], srcs_raw[0][1]
    x1, c1 = (srcs_raw[1][0], srcs_raw[1][1]) if nsrc_norm == 2 else (None, 0)
    from faceposegenerator_amd import _lib as L
    xn = eng.arena.alloc((b * h * w_, c_norm), eng.tdt)
    hw = h * w_
    if x1 is None and hw % 64 == 0:
        # partials in the producer's layout (one per 64-row chunk and group), fed to BOTH paths: bit-identical results required
        xf = x0.float().reshape(b, hw // 64, 64, G, c0 // G)
        part = torch.stack([xf.sum(dim=(2, 4)), (xf * xf).sum(dim=(2, 4))], dim=-1).contiguous()
        chunks = hw // 64
        pin, pch = part.data_ptr(), chunks
    else:
        # skip concatenation: the reference recomputes its statistics (idb_groupnorm's own first pass), the fused kernel takes
        # idb_groupnorm_stats' partials — the same sums in a different order, so the caller compares with a one-ulp tolerance
        part, chunks = eng.gn_statistics(x0, c0, x1, c1, b, hw, G)
        pin, pch = None, 0
    L.check(eng.lib.idb_groupnorm(x0.data_ptr(), c0, None if x1 is None else x1.data_ptr(), c1, b, hw, G, eps, gamma.data_ptr(), beta.data_ptr(),
                                  int(silu), xn.data_ptr(), eng.dt, eng._gn_ws.data_ptr(), eng._gn_ws.numel(), None, 0, pin, pch, 0),
            "idb_groupnorm")
    taps0 = srcs_raw[0][2]
    ref_srcs = [(xn, c_norm, taps0, h, w_, 0)] + [(t, c, tp, h, w_, 0) for (t, c, tp) in srcs_raw[nsrc_norm:]]
    if nsrc_norm == 2:       # the fused form's K order is [src0 taps][src1 taps]: the reference GEMM needs the [tap][c0+c1] weight
        wref = kw.pop("w_ref")
    else:
        wref = w
    ref = eng.gemm(ref_srcs, wref, cout, b, h, w_, tile=ref_tile or (tile - (tile // 10) * 10 + 10), split_k=split_k, **(extra or {}))
    fused = eng.gemm([(t, c, tp, h, w_, 0) for (t, c, tp) in srcs_raw], w, cout, b, h, w_, tile=tile, split_k=split_k,
                     gn_in=(part, chunks, G, eps, gamma, beta, silu, nsrc_norm), **(extra or {}))
    torch.cuda.synchronize()
    assert fused is not None, "the library declined the fused GroupNorm for this plan"
    return ref, fused


@pytest.mark.parametrize("b,h,cin,cout,tile,split_k", [(2, 16, 128, 320, 76, 1), (2, 32, 64, 128, 77, 1), (1, 16, 320, 320, 56, 2), (2, 8, 256, 256, 57, 4),
                                                        (2, 8, 640, 128, 74, 1), (3, 16, 192, 160, 56, 1), (2, 8, 128, 160, 76, 1)])
def test_fused_groupnorm_silu_conv3x3_bit_identical(eng, b, h, cin, cout, tile, split_k):
    x = _rand((b, h, h, cin), 101, 1.5).to(eng.tdt)
    w = eng.tile_weight(eng._pack_conv(_rand((cout, cin, 3, 3), 102, (9 * cin) ** -0.5)))
    bias, sb = _rand((cout,), 103), _rand((b, cout), 104)
    ref, fused = _gn_ref_and_fused(eng, [(x, cin, 9)], cin, 1, w, cout, b, h, h, tile, True, 1e-5, 105, extra=dict(bias=bias, sbias=(sb, 0, cout)),
                                   split_k=split_k)
    assert torch.equal(fused, ref)


def test_fused_groupnorm_over_skip_concat_and_shortcut(eng):
    """norm1 + conv1 over cat[x, skip] as two 3x3 sources (K order [src0 taps][src1 taps]) and norm2 + conv2 + 1x1 shortcut over the raw
    inputs as three sources (only the first normalised)."""
    b, h, ca, cb, cout = 2, 16, 128, 64, 192
    xa, xb = _rand((b, h, h, ca), 111).to(eng.tdt), _rand((b, h, h, cb), 112, 2.0).to(eng.tdt)
    wc = _rand((cout, ca + cb, 3, 3), 113, (9 * (ca + cb)) ** -0.5)
    w_ref = eng.tile_weight(eng._pack_conv(wc))
    w_split = eng.tile_weight(torch.cat([eng._pack_conv(wc[:, :ca].contiguous()), eng._pack_conv(wc[:, ca:].contiguous())], dim=1).contiguous())
    bias = _rand((cout,), 114)
    ref, fused = _gn_ref_and_fused(eng, [(xa, ca, 9), (xb, cb, 9)], ca + cb, 2, w_split, cout, b, h, h, 76, True, 1e-5, 115, extra=dict(bias=bias),
                                   w_ref=w_ref)
    # statistics from two different first passes: equal up to an output ulp here and there
    assert (fused.float() - ref.float()).abs().max().item() <= _tol(eng) * max(1.0, ref.float().abs().max().item())
    assert (fused != ref).float().mean().item() < 0.02
    # conv2 + shortcut: [h1 (normalised) 3x3 | xa 1x1 | xb 1x1]
    h1 = _rand((b, h, h, cout), 116, 1.3).to(eng.tdt)
    w2 = eng.tile_weight(torch.cat([eng._pack_conv(_rand((cout, cout, 3, 3), 117, (9 * cout) ** -0.5)),
                                    eng._pack_mat(_rand((cout, ca + cb), 118, (ca + cb) ** -0.5))], dim=1).contiguous())
    ref, fused = _gn_ref_and_fused(eng, [(h1, cout, 9), (xa, ca, 1), (xb, cb, 1)], cout, 1, w2, cout, b, h, h, 76, True, 1e-5, 119,
                                   extra=dict(bias=bias, gn_stats=32))
    assert torch.equal(fused, ref)
    # (the statistics of the output are summed by 256 threads here and 512 in the unfused kernel: same values, different order)
    assert (getattr(fused, "_gn", None) is None) == (getattr(ref, "_gn", None) is None)
    if getattr(fused, "_gn", None) is not None:
        assert torch.allclose(fused._gn[0], ref._gn[0], rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("b,h,cin,cout,shape,split_k", [(2, 64, 128, 320, 6, 1), (2, 64, 64, 160, 8, 1), (2, 32, 256, 256, 9, 2), (1, 16, 320, 320, 8, 4),
                                                         (2, 8, 256, 160, 6, 2), (2, 16, 192, 64, 4, 3), (3, 32, 192, 128, 7, 1)])
def test_fused_groupnorm_silu_in_patch_resident_conv_bit_identical(eng, b, h, cin, cout, shape, split_k):
    """tile ids 10x with gn_in: the transforming patch loaders normalise each halo patch once per chunk; operands and accumulation order are
    those of idb_groupnorm + the unfused patch-resident conv, so the result is bit-identical (padding pixels stay zero)."""
    x = _rand((b, h, h, cin), 161, 1.5).to(eng.tdt)
    w = eng.tile_weight(eng._pack_conv(_rand((cout, cin, 3, 3), 162, (9 * cin) ** -0.5)))
    bias, sb = _rand((cout,), 163), _rand((b, cout), 164)
    res = _rand((b * h * h, cout), 165).to(eng.tdt)
    for silu in (True, False):
        ref, fused = _gn_ref_and_fused(eng, [(x, cin, 9)], cin, 1, w, cout, b, h, h, 100 + shape, silu, 1e-5, 166,
                                       extra=dict(bias=bias, sbias=(sb, 0, cout), residual=res, gn_stats=32 if cout % 32 == 0 else 0),
                                       split_k=split_k, ref_tile=100 + shape)
        assert torch.equal(fused, ref)
    # the auto plan takes the same kernel when a GroupNorm is handed in
    if eng.fuses_groupnorm([(cin, 9)], w, cout, b, h, h, 32, 1):
        auto = eng.gemm([(x, cin, 9, h, h, 0)], w, cout, b, h, h, bias=bias, sbias=(sb, 0, cout), residual=res,
                        gn_in=_last_gn_in(eng, x, cin, b, h, False, 166))
        torch.cuda.synchronize()
        assert (auto.float() - ref.float()).abs().max().item() <= 4 * _tol(eng) * max(1.0, ref.float().abs().max().item())


def _last_gn_in(eng, x, cin, b, h, silu, seed):
    G = 32
    gamma, beta = 1.0 + 0.3 * _rand((cin,), seed), 0.2 * _rand((cin,), seed + 1)
    xf = x.float().reshape(b, h * h // 64, 64, G, cin // G)
    part = torch.stack([xf.sum(dim=(2, 4)), (xf * xf).sum(dim=(2, 4))], dim=-1).contiguous()
    return (part, h * h // 64, G, 1e-5, gamma, beta, silu, 1)


def test_fused_groupnorm_over_skip_concat_in_patch_resident_conv(eng):
    """norm1 + conv1 over cat[x, skip] as two normalised 3x3 K segments of the patch-resident conv."""
    b, h, ca, cb, cout = 2, 16, 128, 64, 192
    xa, xb = _rand((b, h, h, ca), 171).to(eng.tdt), _rand((b, h, h, cb), 172, 2.0).to(eng.tdt)
    wc = _rand((cout, ca + cb, 3, 3), 173, (9 * (ca + cb)) ** -0.5)
    w_ref = eng.tile_weight(eng._pack_conv(wc))
    w_split = eng.tile_weight(torch.cat([eng._pack_conv(wc[:, :ca].contiguous()), eng._pack_conv(wc[:, ca:].contiguous())], dim=1).contiguous())
    bias = _rand((cout,), 174)
    ref, fused = _gn_ref_and_fused(eng, [(xa, ca, 9), (xb, cb, 9)], ca + cb, 2, w_split, cout, b, h, h, 106, True, 1e-5, 175, extra=dict(bias=bias),
                                   w_ref=w_ref, ref_tile=106)
    # statistics from two different first passes: equal up to an output ulp here and there
    assert (fused.float() - ref.float()).abs().max().item() <= _tol(eng) * max(1.0, ref.float().abs().max().item())
    assert (fused != ref).float().mean().item() < 0.02


def test_fused_groupnorm_proj_in_no_silu_with_row_stats(eng):
    """Transformer2DModel.norm (eps 1e-6, no SiLU) inside proj_in (a 1x1 source), with the row statistics of a following folded LayerNorm."""
    b, h, c = 2, 16, 320
    x = _rand((b, h, h, c), 121, 2.0).to(eng.tdt)
    w = eng.tile_weight(eng._pack_mat(_rand((c, c), 122, c ** -0.5)))
    bias = _rand((c,), 123)
    ref, fused = _gn_ref_and_fused(eng, [(x, c, 1)], c, 1, w, c, b, h, h, 56, False, 1e-6, 124, extra=dict(bias=bias, row_stats=True))
    assert torch.equal(fused, ref)
    if getattr(ref, "_rs", None) is not None:       # row statistics: summed by 4 threads per row here, 8 in the 8-wave kernel (same values, other order)
        assert getattr(fused, "_rs", None) is not None and torch.allclose(fused._rs[0], ref._rs[0], rtol=1e-5, atol=1e-2)
