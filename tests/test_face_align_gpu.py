"""GPU parity of the align-and-crop warp (idb_warp_affine_u8, faceposegenerator_amd/face_align.py) against the oracle's
restatement of cv2.warpAffine: integer arithmetic on both sides, so the bar is bit-exactness."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sim(scale, ang, tx, ty):
    c, s = np.cos(ang) * scale, np.sin(ang) * scale
    return np.array([[c, -s, tx], [s, c, ty]])


@pytest.mark.parametrize("h,w,c,out_hw,border", [(200, 180, 3, (112, 112), 0), (64, 64, 1, (112, 112), 77), (512, 512, 3, (112, 112), 0),
                                                 (33, 47, 4, (50, 70), 255)])
def test_warp_affine_matches_oracle_bit_exactly(lib, h, w, c, out_hw, border):
    from faceposegenerator_amd import face_align as FA
    from oracle import face_align_oracle as FO
    rng = np.random.default_rng(5)
    b = 6
    imgs = rng.integers(0, 256, size=(b, h, w, c), dtype=np.uint8)
    mats = [_sim(1.0, 0.0, 0.0, 0.0), _sim(1.0, 0.0, -3.0, -5.0), _sim(0.37, 0.3, 10.0, 4.0), _sim(2.3, -1.2, 60.0, 80.0),
            _sim(0.9, 3.0, 150.0, 120.0), _sim(1.13, 0.77, -40.5, 33.25)]
    out = FA.warp_affine(torch.from_numpy(imgs).to(DEV), np.stack(mats), out_hw, border).cpu().numpy()
    for i in range(b):
        ref = FO.warp_affine_u8(imgs[i], mats[i], out_hw, border)
        assert np.array_equal(out[i], ref), f"image {i}: {np.abs(out[i].astype(int) - ref.astype(int)).max()} levels off"
    with pytest.raises(ValueError):
        FA.warp_affine(torch.from_numpy(imgs).to(DEV), np.stack(mats[:2]), out_hw, border)


def test_norm_crop_from_landmarks(lib):
    """A face whose landmarks are the template pushed through a known similarity: the crop equals the oracle's norm_crop and
    maps the landmark pixels onto the template positions."""
    from faceposegenerator_amd import face_align as FA
    from oracle import face_align_oracle as FO
    rng = np.random.default_rng(6)
    imgs = rng.integers(0, 256, size=(4, 512, 512, 3), dtype=np.uint8)
    lms = []
    for i in range(4):
        m = _sim(rng.uniform(1.5, 3.5), rng.uniform(-0.5, 0.5), rng.uniform(80, 200), rng.uniform(60, 180))     # template -> image
        lm = FO.ARCFACE_TEMPLATE.astype(np.float64) @ m[:, :2].T + m[:, 2] + rng.normal(size=(5, 2))
        lms.append(lm)
    crops = FA.norm_crop(torch.from_numpy(imgs).to(DEV), np.stack(lms)).cpu().numpy()
    assert crops.shape == (4, 112, 112, 3)
    for i in range(4):
        assert np.array_equal(crops[i], FO.norm_crop(imgs[i], lms[i]))
