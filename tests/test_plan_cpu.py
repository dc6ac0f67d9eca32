"""Host-only planner checks (no GPU call): idb_gemm_plan / idb_gemm_fuses_groupnorm / idb_gemm_workspace_bytes decide from the descriptor
alone, so the tile forms DESIGN.md section 4 describes — 256-row loader-wave tiles and the patch-resident 3x3 conv for large grids, the
small-tile patch conv with a fused GroupNorm for the batch-1 UNet, tap-major kernels for everything the patch form cannot run — are pinned
here for the layer shapes of SD-2.1-base (BASELINE configs[1] = B_eff 2, configs[2] = B_eff 128)."""
import ctypes as C

import pytest

from faceposegenerator_amd import _lib as L

PTR = 1 << 20          # any non-null 16-byte-aligned address: the planner never dereferences


def _desc(b, h, srcs, cout, stride=1, tile=0, split_k=0, gn=0, dt=None):
    """srcs: [(channels, taps, upsample)]; gn = number of leading sources normalised in-kernel (0: none)."""
    d = L.GemmDesc()
    oh = h // stride
    d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = (L.IDB_F16 if dt is None else dt), b, oh, oh, stride, cout, len(srcs)
    for i, (ch, taps, up) in enumerate(srcs):
        d.src[i].ptr, d.src[i].channels, d.src[i].taps, d.src[i].in_h, d.src[i].in_w, d.src[i].upsample = PTR, ch, taps, h >> up, h >> up, up
    d.w, d.out, d.out_dtype, d.out_ld, d.tile, d.split_k, d.w_layout = PTR, PTR, d.dtype, cout, tile, split_k, 1
    if gn:
        d.gn_in_partials, d.gn_in_chunks, d.gn_in_groups, d.gn_in_eps = PTR, min(64, max(1, oh * oh // 64)), 32, 1e-5
        d.gn_in_gamma, d.gn_in_beta, d.gn_in_silu, d.gn_in_nsrc = PTR, PTR, 1, gn
    return d


def _plan(d):
    lib = L.load()
    t, sk, bl = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.idb_gemm_plan(C.byref(d), C.byref(t), C.byref(sk), C.byref(bl))
    return rc, t.value, sk.value, bl.value


@pytest.mark.parametrize("h,cin,cout", [(64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280)])
def test_batch64_pure_3x3_convs_take_the_patch_resident_256_row_tile(h, cin, cout):
    rc, tile, sk, blocks = _plan(_desc(128, h, [(cin, 9, 0)], cout))
    assert (rc, tile, sk) == (0, 98, 1)
    assert blocks == (128 * h * h // 256) * (cout // 160)


def test_batch64_everything_else_keeps_the_tap_major_tiles():
    assert _plan(_desc(128, 64, [(320, 9, 0)], 320, stride=2))[1] == 88                 # Downsample2D
    assert _plan(_desc(128, 64, [(640, 9, 1)], 640))[1] == 88                           # Upsample2D (nearest 2x fused)
    assert _plan(_desc(128, 32, [(640, 9, 0), (320, 1, 0)], 640))[1] == 88              # conv2 + fused 1x1 shortcut
    assert _plan(_desc(128, 64, [(1280, 1, 0)], 320))[1] == 88                          # FF-out, K = 1280
    assert _plan(_desc(128, 64, [(320, 1, 0)], 960))[1] // 10 == 0                      # K = 320 projection: 128-row two-per-CU tile
    assert _plan(_desc(128, 48, [(320, 9, 0)], 320))[1] == 88                           # 96x96 latents / 2: W = 48 is no divisor of 256
    assert _plan(_desc(128, 8, [(1280, 9, 0)], 1280))[1] == 8                           # 8x8 level: 256 workgroups of 256 rows would not fill two rounds


@pytest.mark.parametrize("h,cin,cout,fuses", [(64, 320, 320, 1), (32, 640, 640, 1), (16, 1280, 1280, 1), (8, 1280, 1280, 1)])
def test_batch1_resnet_convs_fuse_their_groupnorm_through_the_small_patch_tiles(h, cin, cout, fuses):
    lib = L.load()
    plain = _plan(_desc(2, h, [(cin, 9, 0)], cout))
    assert plain[0] == 0 and 5 <= plain[1] // 10 <= 7, plain                            # a one-workgroup-per-CU loader-wave plan
    d = _desc(2, h, [(cin, 9, 0)], cout, gn=1)
    assert lib.idb_gemm_fuses_groupnorm(C.byref(d)) == fuses
    rc, tile, sk, _ = _plan(d)
    if fuses:
        assert rc == 0 and tile // 10 == 10 and tile % 10 == plain[1] % 10 and sk == plain[2]   # the same tile shape and split, the patch kernel
        need = lib.idb_gemm_workspace_bytes(C.byref(d))
        assert need == (sk * 2 * h * h * cout * 4 if sk > 1 else 0)
    else:
        assert tile // 10 != 10


def test_groupnorm_fusion_refusals():
    lib = L.load()
    # skip concatenation as two normalised 3x3 sources: fuses; conv2 + raw 1x1 shortcut segments: does not
    assert lib.idb_gemm_fuses_groupnorm(C.byref(_desc(2, 32, [(640, 9, 0), (320, 9, 0)], 640, gn=2))) == 1
    assert lib.idb_gemm_fuses_groupnorm(C.byref(_desc(2, 32, [(640, 9, 0), (320, 1, 0)], 640, gn=1))) == 0
    # large grids never fuse (256-row tiles: no transforming loaders there)
    assert lib.idb_gemm_fuses_groupnorm(C.byref(_desc(128, 32, [(640, 9, 0)], 640, gn=1))) == 0
    # Transformer2DModel norm + proj_in (1x1) keeps the tap-major normalizer-wave kernel
    d = _desc(2, 64, [(320, 1, 0)], 320, gn=1)
    assert lib.idb_gemm_fuses_groupnorm(C.byref(d)) == 1 and 5 <= _plan(d)[1] // 10 <= 7
    # forced patch tiles on shapes they cannot run are refused, not mis-run
    assert _plan(_desc(128, 64, [(320, 9, 0)], 320, stride=2, tile=98))[0] == -2
    assert _plan(_desc(3, 8, [(320, 9, 0)], 320, tile=98))[0] == -2
    assert _plan(_desc(2, 64, [(320, 9, 0)], 320, tile=108, split_k=2))[0:3] == (0, 108, 2)
