"""SURVEY.md §8f-3, align-and-crop half (utils/detect_align_crop_data.py:135-197): the oracle's restatements of
skimage's Umeyama estimator and of cv2.warpAffine's 8-bit bilinear path, pinned by analytic properties (neither library is
importable here, the reference holds no fixtures for this stage: parity with them is unpinned), and the host module's own
estimator against the oracle's."""
import numpy as np
import pytest

from faceposegenerator_amd import face_align as FA
from oracle import face_align_oracle as FO


def _sim(scale, ang, tx, ty):
    c, s = np.cos(ang) * scale, np.sin(ang) * scale
    return np.array([[c, -s, tx], [s, c, ty]])


def test_template_is_the_shifted_arcface_one():
    assert np.allclose(FA.ARCFACE_TEMPLATE, FO.ARCFACE_TEMPLATE.astype(np.float64), atol=1e-5)
    assert abs(FA.ARCFACE_TEMPLATE[0, 0] - 38.2946) < 1e-5 and abs(FA.ARCFACE_TEMPLATE[2, 1] - 71.7366) < 1e-5   # float32 constants


@pytest.mark.parametrize("scale,ang,tx,ty", [(1.0, 0.0, 0.0, 0.0), (1.7, 0.4, 5.0, -3.0), (0.31, -2.2, 300.0, 120.5), (2.5, 3.1, -40.0, 7.0)])
def test_umeyama_recovers_a_known_similarity(scale, ang, tx, ty):
    m = _sim(scale, ang, tx, ty)
    src = np.random.default_rng(0).normal(size=(5, 2)) * 30 + 100
    dst = src @ m[:, :2].T + m[:, 2]
    assert np.allclose(FO.umeyama(src, dst)[:2], m, atol=1e-9)
    assert np.allclose(FA.similarity_from_points(src, dst), m, atol=1e-9)


def test_host_estimator_equals_oracle_on_noisy_landmarks():
    rng = np.random.default_rng(1)
    for _ in range(20):
        m = _sim(rng.uniform(0.2, 4), rng.uniform(-3.1, 3.1), rng.uniform(-50, 300), rng.uniform(-50, 300))
        lm = (FO.ARCFACE_TEMPLATE.astype(np.float64) + rng.normal(size=(5, 2)) * 1.5 - m[:, 2]) @ np.linalg.inv(m[:, :2]).T
        a, b = FA.estimate_norm(lm), FO.estimate_norm(lm)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9)
        mapped = np.float32(lm) @ a[:, :2].T + a[:, 2]
        assert np.abs(mapped - FO.ARCFACE_TEMPLATE).max() < 8.0          # noisy landmarks land near the template
    with pytest.raises(ValueError):
        FA.estimate_norm(np.zeros((4, 2)))
    with pytest.raises(ValueError):
        FA.estimate_norm(np.zeros((5, 2)) + 3.0)                         # all landmarks identical


def test_warp_identity_translation_halfpixel_and_border():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(150, 140, 3), dtype=np.uint8)
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(FO.warp_affine_u8(img, eye), img[:112, :112])
    shift = np.array([[1.0, 0, -7.0], [0, 1.0, -20.0]])                  # dst(x, y) = src(x + 7, y + 20)
    assert np.array_equal(FO.warp_affine_u8(img, shift), img[20:132, 7:119])
    half = np.array([[1.0, 0, -0.5], [0, 1.0, 0.0]])                     # dst(x) = (src(x) + src(x + 1) + 1) >> 1
    want = ((img[:112, :112].astype(int) + img[:112, 1:113].astype(int) + 1) >> 1).astype(np.uint8)
    assert np.array_equal(FO.warp_affine_u8(img, half), want)
    far = np.array([[1.0, 0, 500.0], [0, 1.0, 0.0]])                     # everything maps from outside the image
    assert np.array_equal(FO.warp_affine_u8(img, far, border=9), np.full((112, 112, 3), 9, np.uint8))
    edge = np.array([[1.0, 0, 1.0], [0, 1.0, 0.0]])                      # column 0 comes from x = -1: border
    out = FO.warp_affine_u8(img, edge, border=0)
    assert np.array_equal(out[:, 0], np.zeros((112, 3), np.uint8)) and np.array_equal(out[:, 1:], img[:112, :111])
    up2 = FO.warp_affine_u8(img, np.array([[2.0, 0, 0], [0, 2.0, 0]]), (64, 64))
    assert np.array_equal(up2[::2, ::2], img[:32, :32])                  # even destination pixels hit source pixels exactly


def test_pad_for_detection():
    img = np.ones((10, 7, 3), np.uint8)
    p = FA.pad_for_detection(img)
    assert p.shape == (20, 13, 3) and p[:5].sum() == 0 and p[5:15, 3:10].min() == 1
