"""Align-and-crop half of the reference's face preprocessing: the step right after the sampler on the way to
face-recognition training — 5 landmarks -> similarity transform onto the ArcFace 112x112 template -> warped crop
(/root/reference/utils/detect_align_crop_data.py:135-197, ``estimate_norm`` + ``norm_crop``).

The landmark DETECTOR of that script (facenet_pytorch MTCNN, :18-20, :99) is ``faceposegenerator_amd.mtcnn.MTCNN`` (P/R/O-Net
cascade on HIP kernels; its trained weights are not available offline, so tests use seeded synthetic ones):
``landmarks = mtcnn.detect(batch, landmarks=True)[2]`` then ``norm_crop(batch, [l[0] for l in landmarks])``.  The warp runs on
the GPU (``idb_warp_affine_u8``) directly on the sampler's uint8 NHWC output; the 5-point least-squares fit is host arithmetic
on 10 numbers per face, as in the reference."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

# insightface template for 112x112 crops, x shifted by +8 as detect_align_crop_data.py:183-197 does (in place, so both of its
# names see the shift)
_BASE = np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366], [33.5493, 92.3655], [62.7299, 92.2041]], dtype=np.float32)
_BASE[:, 0] += 8.0                                    # float32 arithmetic, like the reference's in-place shift
ARCFACE_TEMPLATE = _BASE.astype(np.float64)


def similarity_from_points(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """2x3 least-squares similarity (rotation, uniform scale, translation; no reflection) with dst ~ M [src; 1] (Umeyama 1991,
    the estimator behind skimage's SimilarityTransform.estimate)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    if src.shape != dst.shape or src.ndim != 2 or src.shape[1] != 2 or src.shape[0] < 2:
        raise ValueError("need two (N, 2) point sets, N >= 2")
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    u, sv, vt = np.linalg.svd(cov)
    sign = np.ones(2)
    if np.linalg.det(cov) < 0:
        sign[1] = -1.0
    var = (xs ** 2).sum() / src.shape[0]
    if var == 0.0 or np.linalg.matrix_rank(cov) == 0:
        raise ValueError("degenerate landmark set")
    if np.linalg.matrix_rank(cov) == 1:               # collinear points: keep a proper rotation
        sign[1] = 1.0 if np.linalg.det(u) * np.linalg.det(vt) > 0 else -1.0
    rot = u @ np.diag(sign) @ vt
    scale = (sv * sign).sum() / var
    m = np.empty((2, 3))
    m[:, :2] = scale * rot
    m[:, 2] = mu_d - scale * (rot @ mu_s)
    return m


def estimate_norm(landmarks: np.ndarray, image_size: int = 112) -> np.ndarray:
    """detect_align_crop_data.py:135-168 for its single template: landmarks (5, 2) in pixels -> 2x3 matrix."""
    if image_size != 112:
        raise ValueError("the reference asserts image_size == 112")
    lm = np.asarray(landmarks)
    if lm.shape != (5, 2):
        raise ValueError("landmarks must have shape (5, 2)")
    return similarity_from_points(np.float32(lm), np.float32(ARCFACE_TEMPLATE))


def _invert(m: np.ndarray) -> np.ndarray:
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([a11, a12, -a11 * m[0, 2] - a12 * m[1, 2], a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]], np.float64)


def warp_affine(images: torch.Tensor, matrices, out_hw=(112, 112), border_value: int = 0) -> torch.Tensor:
    """``cv2.warpAffine(img, M, (w, h), borderValue=...)`` for a batch of uint8 NHWC images on the GPU; matrices: (B, 2, 3)
    forward maps (source -> destination), one per image."""
    if images.dtype != torch.uint8 or images.ndim != 4 or not images.is_cuda:
        raise ValueError("images must be a CUDA uint8 tensor [B, H, W, C]")
    mats = np.asarray(matrices, np.float64).reshape(-1, 2, 3)
    b, h, w, c = images.shape
    if mats.shape[0] != b:
        raise ValueError("one 2x3 matrix per image")
    inv = torch.from_numpy(np.stack([_invert(m) for m in mats])).to(images.device)
    img = images.contiguous()
    out = torch.empty((b, out_hw[0], out_hw[1], c), dtype=torch.uint8, device=images.device)
    L.check(L.load().idb_warp_affine_u8(img.data_ptr(), b, h, w, c, inv.data_ptr(), out.data_ptr(), out_hw[0], out_hw[1],
                                        int(border_value), torch.cuda.current_stream().cuda_stream), "idb_warp_affine_u8")
    return out


def norm_crop(images: torch.Tensor, landmarks, image_size: int = 112) -> torch.Tensor:
    """Batched ``norm_crop`` (detect_align_crop_data.py:172-181): uint8 NHWC images + (B, 5, 2) landmarks -> (B, 112, 112, C)."""
    lms = np.asarray(landmarks).reshape(-1, 5, 2)
    return warp_affine(images, np.stack([estimate_norm(lm, image_size) for lm in lms]), (image_size, image_size), 0)


def pad_for_detection(image: np.ndarray) -> np.ndarray:
    """InferenceDataset.__getitem__ (:66-72): zero borders of half the height / width on every side before detection."""
    h, w, _ = image.shape
    return np.pad(image, ((h // 2, h // 2), (w // 2, w // 2), (0, 0)), mode="constant", constant_values=0)
