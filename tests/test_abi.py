"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol that
include/idb_kernels.h declares (no compute calls: there is no GPU here)."""
import os
import re

from faceposegenerator_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "idb_kernels.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(idb_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.idb_version() >= 100


def test_last_error_is_a_string(lib):
    assert isinstance(lib.idb_last_error(), bytes)


def test_invalid_descriptor_is_rejected_without_gpu(lib):
    # argument validation happens before any HIP call, so it is testable on CPU
    import ctypes as C
    d = _lib.GemmDesc()
    d.dtype = 7
    assert lib.idb_gemm(C.byref(d), None, 0, None) == -1
    assert b"dtype" in lib.idb_last_error()
    d.dtype, d.batch, d.out_h, d.out_w, d.n, d.stride, d.nsrc = 0, 1, 1, 1, 64, 1, 1
    d.src[0].ptr, d.src[0].channels, d.src[0].taps, d.src[0].in_h, d.src[0].in_w = 0x1000, 100, 1, 1, 1
    assert lib.idb_gemm(C.byref(d), None, 0, None) == -1
    assert b"multiple of 64" in lib.idb_last_error()


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    with pytest.raises(_lib.IdbError):
        HipEngine(S.TINY_UNET, S.TINY_VAE, None, None)
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    pipe = StableDiffusionPipeline(S.TINY_UNET, S.TINY_VAE, {}, {})
    with pytest.raises(ValueError):
        pipe.to("cpu")
