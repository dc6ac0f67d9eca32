"""MTCNN cascade on the HIP kernels (faceposegenerator_amd.mtcnn) against the CPU oracle (oracle/mtcnn_oracle.py), seeded synthetic
weights of the published P/R/O-Net shapes: every network on the same inputs (fp32: 1e-4), the whole cascade on a two-image batch
(same boxes, landmarks within 1e-2 px), and the reference's call pattern detect(...)[2][k][0] -> norm_crop."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _images(seed=3, b=2, h=160, w=192):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(b, 3, h // 8, w // 8, generator=g)
    return (F.interpolate(base, size=(h, w), mode="bilinear") * 255).permute(0, 2, 3, 1).to(torch.uint8).contiguous()


@pytest.fixture(scope="module")
def det(lib):
    from faceposegenerator_amd import mtcnn as M
    w = M.synth_weights(5)
    return M.MTCNN(select_largest=True, post_process=False, device=DEV, weights=w), w


def test_networks_match_the_oracle(det):
    from oracle import mtcnn_oracle as O
    m, w = det
    img = _images()
    imgs_f = img.permute(0, 3, 1, 2).float()
    # pyramid level: area resize + normalisation + P-Net
    for (oh, ow) in ((97, 116), (49, 58), (13, 15)):
        full = np.array([[b, 0, 160, 0, 192] for b in range(2)], dtype=np.int32)
        x = m._resample(img.to(DEV), full, oh, ow)
        ref = (O.imresample(imgs_f, (oh, ow)) - 127.5) * 0.0078125
        assert (x.cpu() - ref).abs().max().item() < 1e-5
        reg, prob = m.pnet(x)
        r_ref, p_ref = O.pnet(w["pnet"], ref)
        assert (reg.cpu() - r_ref).abs().max().item() < 1e-4 and (prob.cpu() - p_ref[:, 1]).abs().max().item() < 1e-5
    # crops: odd windows, 24x24 and 48x48
    spec = np.array([[0, 3, 70, 10, 61], [1, 0, 160, 0, 192], [1, 100, 113, 5, 40], [0, 20, 21, 30, 31]], dtype=np.int32)
    for size, net in ((24, "rnet"), (48, "onet")):
        x = m._resample(img.to(DEV), spec, size, size)
        ref = torch.cat([(O.imresample(imgs_f[i:i + 1, :, y0:y1, x0:x1], (size, size)) - 127.5) * 0.0078125 for i, y0, y1, x0, x1 in spec.tolist()])
        assert (x.cpu() - ref).abs().max().item() < 1e-5
        g = torch.Generator().manual_seed(1)
        z = torch.randn(9, 3, size, size, generator=g) * 0.5
        if net == "rnet":
            reg, prob = m.rnet(z.to(DEV))
            r_ref, p_ref = O.rnet(w[net], z)
        else:
            reg, pts, prob = m.onet(z.to(DEV))
            r_ref, l_ref, p_ref = O.onet(w[net], z)
            assert (pts.cpu() - l_ref).abs().max().item() < 1e-4
        assert (reg.cpu() - r_ref).abs().max().item() < 1e-4 and (prob.cpu() - p_ref[:, 1]).abs().max().item() < 1e-5


def test_cascade_matches_the_oracle(det):
    from oracle import mtcnn_oracle as O
    m, w = det
    img = _images()
    boxes, inds, pts = m.detect_faces(img.to(DEV))
    b_ref, i_ref, p_ref = O.detect_face(img, w)
    print(f"MTCNN cascade: {boxes.shape[0]} faces (oracle {b_ref.shape[0]}) on a 2 x 160x192 batch")
    assert boxes.shape[0] == b_ref.shape[0] > 0 and (inds == i_ref.numpy()).all()
    assert np.abs(boxes[:, :4] - b_ref[:, :4].numpy()).max() < 1e-2 and np.abs(boxes[:, 4] - b_ref[:, 4].numpy()).max() < 1e-5
    assert np.abs(pts - p_ref.numpy()).max() < 1e-2


def test_detect_api_and_align_crop(det):
    """utils/detect_align_crop_data.py:99-115: boxes, probs, landmarks = mtcnn.detect(img_batch, landmarks=True);
    facial5points = landmark[0]; norm_crop(img, facial5points, 112)."""
    from faceposegenerator_amd import face_align as FA
    from oracle import mtcnn_oracle as O
    m, w = det
    img = _images(seed=4)
    boxes, probs, landmarks = m.detect(img, landmarks=True)
    # upstream detect() returns EVERY face of an image, largest first with select_largest (keep_all only acts in forward());
    # the reference takes landmark[0]
    assert boxes.shape == (2,) and landmarks[0].shape[1:] == (5, 2) and boxes[0].shape[1:] == (4,) and probs[0].shape == (boxes[0].shape[0],)
    b_ref, i_ref, p_ref = O.detect_face(img, w)
    for k in range(2):
        bb, pp, ll = O.select_largest_first(b_ref, i_ref, p_ref, k)
        assert boxes[k].shape[0] == int((i_ref == k).sum())
        area = (boxes[k][:, 2] - boxes[k][:, 0]) * (boxes[k][:, 3] - boxes[k][:, 1])
        assert (np.diff(area) <= 0).all()
        assert np.abs(boxes[k][:1] - bb.numpy()).max() < 1e-2 and np.abs(landmarks[k][:1] - ll.numpy()).max() < 1e-2
    blank = torch.zeros(1, 64, 64, 3, dtype=torch.uint8)
    b0, p0, l0 = m.detect(blank, landmarks=True)
    # (a blank image may or may not yield candidates with synthetic weights; the call must return one entry per image)
    assert b0.shape == (1,) and l0.shape == (1,)
    lm = np.array([[70.0, 60.0], [120.0, 62.0], [95.0, 90.0], [75.0, 120.0], [118.0, 121.0]], dtype=np.float32)
    crop = FA.norm_crop(img[:1].to(DEV), lm[None])                  # landmarks in, 112x112 aligned crop out
    assert tuple(crop.shape) == (1, 112, 112, 3) and crop.dtype == torch.uint8


@pytest.mark.parametrize("n,images,method,plus_one,thr", [(3000, 4, "Union", False, 0.5), (5000, 16, "Union", False, 0.7), (2500, 3, "Min", True, 0.7),
                                                          (700, 1, "Union", False, 0.7), (64 * 9 + 1, 2, "Min", True, 0.5)])
def test_device_nms_mask_keeps_exactly_the_host_routines_boxes(det, n, images, method, plus_one, thr):
    """idb_nms_mask + the host scan (MTCNN._bnms) against the pure-numpy _batched_nms on crowded random boxes: identical indices in
    identical order — score ties, duplicate boxes, zero-area boxes (NaN overlap under "Union" without +1) and boxes of different
    images included."""
    from faceposegenerator_amd import mtcnn as M
    m, _ = det
    rng = np.random.default_rng(n)
    ctr = rng.uniform(0, 400, size=(n, 2))
    wh = rng.uniform(4, 120, size=(n, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    boxes[::97] = boxes[1::97][: boxes[::97].shape[0]]                 # exact duplicates
    boxes[5::211, 2:] = boxes[5::211, :2]                               # zero-area boxes
    scores = rng.uniform(0.5, 1.0, size=n).astype(np.float32)
    scores[::13] = np.float32(0.75)                                     # ties
    idxs = rng.integers(0, images, size=n)
    want = M._batched_nms(boxes, scores, idxs, thr, method, plus_one)
    assert n >= m.NMS_DEVICE_MIN
    got = m._bnms(boxes, scores, idxs, thr, method, plus_one)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert 0 < want.size < n
