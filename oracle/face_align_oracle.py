"""CPU restatement of the align-and-crop half of the reference's face preprocessing — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product path (faceposegenerator_amd/face_align.py + idb_warp_affine_u8) never does.

What it restates (/root/reference/utils/detect_align_crop_data.py):
  * estimate_norm (:135-168): skimage.transform.SimilarityTransform.estimate(lmk, template) = Umeyama's least-squares
    similarity (S. Umeyama, PAMI 1991) onto the ArcFace 112x112 template whose x coordinates the script shifts by +8
    (:183-197; the "+= 8.0" is applied in place to the array both names point to);
  * norm_crop (:172-181): cv2.warpAffine(img, M, (112, 112), borderValue=0.0) — OpenCV's default INTER_LINEAR, BORDER_CONSTANT
    path for 8-bit images: M is inverted in double precision, destination coordinates are mapped in 10-bit fixed point with
    a 5-bit sub-pixel fraction (AB_BITS 10, INTER_BITS 5, rounding offset 16), and the four taps are blended with 15-bit
    integer weights (32 - fx)(32 - fy) * 32 ... that sum to 32768 exactly, result (sum + 16384) >> 15.
  * the MTCNN detector (:18-20, :99) is NOT restated: facenet_pytorch and its weights are absent.

PARITY UNPINNED: neither cv2 nor skimage is importable in the build container and the reference holds no fixtures for this
stage; the restatement follows the published algorithms above.  Analytic pins in tests/test_face_align_cpu.py: exact
recovery of known similarities, identity / integer-translation warps are exact copies, half-pixel shifts average neighbours.
"""
import numpy as np

ARCFACE_TEMPLATE = np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366], [33.5493, 92.3655], [62.7299, 92.2041]],
                            dtype=np.float32)
ARCFACE_TEMPLATE[:, 0] += 8.0          # detect_align_crop_data.py:196-197


def umeyama(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 homogeneous similarity T minimising |T src - dst|^2 (skimage _umeyama with estimate_scale=True)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(0), dst.mean(0)
    sd, dd = src - src_mean, dst - dst_mean
    A = dd.T @ sd / num
    d = np.ones(dim)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.full((dim + 1, dim + 1), np.nan)
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def estimate_norm(lmk: np.ndarray) -> np.ndarray:
    """2x3 matrix mapping the 5 detected landmarks onto the template (estimate_norm, one template => index 0)."""
    assert lmk.shape == (5, 2)
    return umeyama(np.float32(lmk), ARCFACE_TEMPLATE)[0:2, :]


def invert_affine(m: np.ndarray) -> np.ndarray:
    """cv::invertAffineTransform in double precision."""
    m = np.asarray(m, np.float64)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = m[1, 1] * D, m[0, 0] * D
    a12, a21 = -m[0, 1] * D, -m[1, 0] * D
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]], np.float64)


def warp_affine_u8(img: np.ndarray, m: np.ndarray, out_hw=(112, 112), border: int = 0) -> np.ndarray:
    """cv2.warpAffine(img, m, (w, h), borderValue=border) for uint8 HWC images (INTER_LINEAR, BORDER_CONSTANT)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    oh, ow = out_hw
    mi = invert_affine(m)
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    rd = AB_SCALE // (1 << INTER_BITS) // 2
    xs = np.arange(ow, dtype=np.float64)
    adelta = np.rint(mi[0, 0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(mi[1, 0] * xs * AB_SCALE).astype(np.int64)
    out = np.empty((oh, ow, C), np.uint8)
    src = img.astype(np.int64)
    for y in range(oh):
        X0 = int(np.rint((mi[0, 1] * y + mi[0, 2]) * AB_SCALE)) + rd
        Y0 = int(np.rint((mi[1, 1] * y + mi[1, 2]) * AB_SCALE)) + rd
        X = (X0 + adelta) >> (AB_BITS - INTER_BITS)
        Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS)
        sx, sy = X >> INTER_BITS, Y >> INTER_BITS
        fx, fy = X & 31, Y & 31
        acc = np.zeros((ow, C), np.int64)
        for dy, dx, wgt in ((0, 0, (32 - fx) * (32 - fy) * 32), (0, 1, fx * (32 - fy) * 32), (1, 0, (32 - fx) * fy * 32), (1, 1, fx * fy * 32)):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            tap = np.full((ow, C), border, np.int64)
            tap[ok] = src[yy[ok], xx[ok]]
            acc += wgt[:, None] * tap
        out[y] = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out


def norm_crop(img: np.ndarray, landmark: np.ndarray) -> np.ndarray:
    return warp_affine_u8(img, estimate_norm(landmark), (112, 112), 0)
