"""MTCNN face detector — the landmark detector in front of the align-and-crop step.

API mirror of the calls the reference makes (/root/reference/utils/detect_align_crop_data.py):
  * ``mtcnn = MTCNN(select_largest=True, post_process=False, device="cuda:0")``                      :18-20
  * ``boxes, probs, landmarks = mtcnn.detect(img_batch, landmarks=True)`` on a uint8 NHWC batch        :99
    -> per image ``box`` [n, 4] / ``prob`` [n] / ``landmark`` [n, 5, 2] (or None when no face), faces ordered largest first;
       the reference takes ``landmark[0]`` (:112) and hands it to ``norm_crop`` (faceposegenerator_amd.face_align).

Upstream is ``facenet_pytorch`` (not installed, its weights not available offline): the cascade below restates its published
algorithm — image pyramid (factor 0.709, min face 20), P-Net on every scale, box generation (stride 2, cell 12), NMS 0.5 per
scale and 0.7 across scales, regression + squaring, R-Net on 24x24 and O-Net on 48x48 area-resized crops with thresholds
0.6 / 0.7 / 0.7, final "Min"-overlap NMS — PARITY UNPINNED against upstream; the HIP path is checked against
oracle/mtcnn_oracle.py (plain torch CPU ops) with seeded synthetic weights.

Every network layer is a HIP kernel of libidb_kernels.so (idb_crop_resize_area_u8, idb_conv2d_f32, idb_maxpool2d_f32,
idb_softmax_pairs_f32; fp32, csrc/idb_mtcnn.hip); the data-dependent control logic between the stages works on a few hundred
boxes and runs on the host, as in upstream."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

SD = Dict[str, torch.Tensor]

# layer lists: (kind, name, args)
PNET = [("conv", "conv1", (3, 10, 3), "prelu1"), ("pool", 2, 2), ("conv", "conv2", (10, 16, 3), "prelu2"), ("conv", "conv3", (16, 32, 3), "prelu3")]
RNET = [("conv", "conv1", (3, 28, 3), "prelu1"), ("pool", 3, 2), ("conv", "conv2", (28, 48, 3), "prelu2"), ("pool", 3, 2),
        ("conv", "conv3", (48, 64, 2), "prelu3")]
ONET = [("conv", "conv1", (3, 32, 3), "prelu1"), ("pool", 3, 2), ("conv", "conv2", (32, 64, 3), "prelu2"), ("pool", 3, 2),
        ("conv", "conv3", (64, 64, 3), "prelu3"), ("pool", 2, 2), ("conv", "conv4", (64, 128, 2), "prelu4")]


def param_shapes() -> Dict[str, Dict[str, Tuple[int, ...]]]:
    """State-dict layout of facenet_pytorch's PNet / RNet / ONet (names as in its ``.pt`` files)."""
    def convs(layers):
        out = {}
        for l in layers:
            if l[0] == "conv":
                cin, cout, k = l[2]
                out[f"{l[1]}.weight"], out[f"{l[1]}.bias"], out[f"{l[3]}.weight"] = (cout, cin, k, k), (cout,), (cout,)
        return out
    p = convs(PNET)
    p.update({"conv4_1.weight": (2, 32, 1, 1), "conv4_1.bias": (2,), "conv4_2.weight": (4, 32, 1, 1), "conv4_2.bias": (4,)})
    r = convs(RNET)
    r.update({"dense4.weight": (128, 576), "dense4.bias": (128,), "prelu4.weight": (128,), "dense5_1.weight": (2, 128), "dense5_1.bias": (2,),
              "dense5_2.weight": (4, 128), "dense5_2.bias": (4,)})
    o = convs(ONET)
    o.update({"dense5.weight": (256, 1152), "dense5.bias": (256,), "prelu5.weight": (256,), "dense6_1.weight": (2, 256), "dense6_1.bias": (2,),
              "dense6_2.weight": (4, 256), "dense6_2.bias": (4,), "dense6_3.weight": (10, 256), "dense6_3.bias": (10,)})
    return {"pnet": p, "rnet": r, "onet": o}


def synth_weights(seed: int = 5) -> Dict[str, SD]:
    """Seeded synthetic P/R/O-Net weights (the trained ones ship inside the facenet_pytorch wheel, which is not here).  The class
    biases are calibrated for the default seed so that, as with trained weights, each stage passes a fraction of its candidates
    (P-Net ~10 % of the positions, R-Net ~60 %, O-Net ~30 %) and the whole cascade is exercised."""
    out = {}
    for ni, (net, shapes) in enumerate(param_shapes().items()):
        sd = {}
        for i, name in enumerate(sorted(shapes)):
            g = torch.Generator().manual_seed(seed * 1000 + ni * 100 + i)
            shp = shapes[name]
            if name.startswith("prelu"):
                sd[name] = 0.1 + 0.3 * torch.rand(shp, generator=g)
            elif name.endswith(".weight"):
                fan = int(np.prod(shp[1:]))
                sd[name] = torch.randn(shp, generator=g) * (2.0 / fan) ** 0.5
            else:
                sd[name] = 0.1 * torch.randn(shp, generator=g)
        cls = {"pnet": "conv4_1", "rnet": "dense5_1", "onet": "dense6_1"}[net]
        sd[f"{cls}.bias"] = torch.tensor({"pnet": [-0.45, 0.45], "rnet": [0.0, 0.0], "onet": [-1.25, 1.25]}[net])
        out[net] = sd
    return out


# ---- host control logic (numpy; upstream: facenet_pytorch/models/utils/detect_face.py) -------------------------------------
def _nms(boxes: np.ndarray, scores: np.ndarray, thr: float, method: str = "Union", plus_one: bool = False) -> np.ndarray:
    """Greedy NMS in descending score order.  Union: IoU > thr suppresses (torchvision.ops.nms: areas without +1; a NaN overlap of two
    zero-area boxes is NOT > thr, so both stay, as torchvision keeps them); Min: overlap over the smaller box, areas with +1, kept iff
    o <= thr (upstream's nms_numpy: a NaN overlap drops the box)."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    one = 1.0 if plus_one else 0.0
    area = (x2 - x1 + one) * (y2 - y1 + one)
    order = np.argsort(-scores, kind="stable")
    keep = []
    while order.size:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + one)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + one)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):        # 0 / 0 between degenerate boxes: handled by the comparisons below
            o = inter / np.minimum(area[i], area[rest]) if method == "Min" else inter / (area[i] + area[rest] - inter)
        order = rest[o <= thr] if method == "Min" else rest[~(o > thr)]
    return np.asarray(keep, dtype=np.int64)


def _batched_nms(boxes, scores, idxs, thr, method="Union", plus_one=False) -> np.ndarray:
    keep = []
    for b in np.unique(idxs):
        sel = np.nonzero(idxs == b)[0]
        keep.append(sel[_nms(boxes[sel], scores[sel], thr, method, plus_one)])
    if not keep:
        return np.zeros((0,), dtype=np.int64)
    keep = np.concatenate(keep)
    return keep[np.argsort(-scores[keep], kind="stable")]          # torchvision returns the kept boxes in descending score order


def _bbreg(bb: np.ndarray, reg: np.ndarray) -> np.ndarray:
    w = bb[:, 2] - bb[:, 0] + 1
    h = bb[:, 3] - bb[:, 1] + 1
    out = bb.copy()
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = bb[:, 0] + reg[:, 0] * w, bb[:, 1] + reg[:, 1] * h, bb[:, 2] + reg[:, 2] * w, bb[:, 3] + reg[:, 3] * h
    return out


def _rerec(bb: np.ndarray) -> np.ndarray:
    h, w = bb[:, 3] - bb[:, 1], bb[:, 2] - bb[:, 0]
    l = np.maximum(w, h)
    out = bb.copy()
    out[:, 0] = bb[:, 0] + w * 0.5 - l * 0.5
    out[:, 1] = bb[:, 1] + h * 0.5 - l * 0.5
    out[:, 2], out[:, 3] = out[:, 0] + l, out[:, 1] + l
    return out


def _pad(bb: np.ndarray, w: int, h: int):
    b = np.trunc(bb[:, :4]).astype(np.int32)
    x, y, ex, ey = b[:, 0].copy(), b[:, 1].copy(), b[:, 2].copy(), b[:, 3].copy()
    x[x < 1] = 1
    y[y < 1] = 1
    ex[ex > w] = w
    ey[ey > h] = h
    return y, ey, x, ex


def pyramid_scales(h: int, w: int, min_face_size: int = 20, factor: float = 0.709) -> List[float]:
    m = 12.0 / min_face_size
    minl = min(h, w) * m
    s, out = m, []
    while minl >= 12:
        out.append(s)
        s *= factor
        minl *= factor
    return out


class MTCNN:
    def __init__(self, image_size: int = 160, margin: int = 0, min_face_size: int = 20, thresholds=(0.6, 0.7, 0.7), factor: float = 0.709,
                 post_process: bool = True, select_largest: bool = True, keep_all: bool = False, device="cuda:0",
                 weights: Optional[Dict[str, SD]] = None):
        if not torch.cuda.is_available():
            raise L.IdbError("MTCNN needs a GPU (there is no CPU fallback)")
        self.lib = L.load()
        self.device = torch.device(device)
        self.min_face_size, self.thresholds, self.factor = min_face_size, list(thresholds), factor
        self.select_largest, self.keep_all, self.post_process = select_largest, keep_all, post_process
        w = weights if weights is not None else synth_weights()
        self.w = {net: {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in sd.items()} for net, sd in w.items()}
        # dense layers as convolutions over the whole map: upstream flattens x.permute(0, 3, 2, 1), i.e. index (w * H + h) * C + c
        r, o = self.w["rnet"], self.w["onet"]
        r["dense4.conv"] = r["dense4.weight"].view(128, 3, 3, 64).permute(0, 3, 2, 1).contiguous()       # [out][c][h][w]
        o["dense5.conv"] = o["dense5.weight"].view(256, 3, 3, 128).permute(0, 3, 2, 1).contiguous()

    # ---- kernels --------------------------------------------------------------------------------------------------------
    def _st(self) -> int:
        return torch.cuda.current_stream().cuda_stream

    def _conv(self, x, wt, bias, slope=None):
        b, cin, h, w_ = x.shape
        cout, _, kh, kw = wt.shape
        y = torch.empty((b, cout, h - kh + 1, w_ - kw + 1), dtype=torch.float32, device=self.device)
        L.check(self.lib.idb_conv2d_f32(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), None if slope is None else slope.data_ptr(), y.data_ptr(),
                                        b, cin, h, w_, cout, kh, kw, self._st()), "idb_conv2d_f32")
        return y

    def _pool(self, x, k, s):
        b, c, h, w_ = x.shape

        def o(n):
            v = (n - k + s - 1) // s + 1
            return max(1, v - 1 if (v - 1) * s >= n else v)
        y = torch.empty((b, c, o(h), o(w_)), dtype=torch.float32, device=self.device)
        L.check(self.lib.idb_maxpool2d_f32(x.data_ptr(), y.data_ptr(), b * c, h, w_, k, s, self._st()), "idb_maxpool2d_f32")
        return y

    def _trunk(self, net: str, layers, x):
        sd = self.w[net]
        for l in layers:
            x = self._conv(x, sd[f"{l[1]}.weight"], sd[f"{l[1]}.bias"], sd[f"{l[3]}.weight"]) if l[0] == "conv" else self._pool(x, l[1], l[2])
        return x

    def _prob1(self, logits):
        b = logits.shape[0]
        hw = logits.numel() // (2 * b)
        p = torch.empty((b, hw), dtype=torch.float32, device=self.device)
        L.check(self.lib.idb_softmax_pairs_f32(logits.data_ptr(), p.data_ptr(), b, hw, self._st()), "idb_softmax_pairs_f32")
        return p

    def _resample(self, imgs_u8, boxes_np, oh, ow):
        """boxes_np int32 [n][5] = image, y0, y1, x0, x1 -> normalised fp32 [n][3][oh][ow]"""
        n = boxes_np.shape[0]
        bx = torch.from_numpy(np.ascontiguousarray(boxes_np, dtype=np.int32)).to(self.device)
        out = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=self.device)
        b, h, w_, c = imgs_u8.shape
        L.check(self.lib.idb_crop_resize_area_u8(imgs_u8.data_ptr(), b, h, w_, c, bx.data_ptr(), n, out.data_ptr(), oh, ow, 127.5, 0.0078125,
                                                 self._st()), "idb_crop_resize_area_u8")
        return out

    def pnet(self, x):
        sd = self.w["pnet"]
        f = self._trunk("pnet", PNET, x)
        prob = self._prob1(self._conv(f, sd["conv4_1.weight"], sd["conv4_1.bias"])).view(f.shape[0], f.shape[2], f.shape[3])
        return self._conv(f, sd["conv4_2.weight"], sd["conv4_2.bias"]), prob

    def rnet(self, x):
        sd = self.w["rnet"]
        f = self._conv(self._trunk("rnet", RNET, x), sd["dense4.conv"], sd["dense4.bias"], sd["prelu4.weight"])        # [n,128,1,1]
        prob = self._prob1(self._conv(f, sd["dense5_1.weight"].view(2, 128, 1, 1), sd["dense5_1.bias"])).view(-1)
        return self._conv(f, sd["dense5_2.weight"].view(4, 128, 1, 1), sd["dense5_2.bias"]).view(-1, 4), prob

    def onet(self, x):
        sd = self.w["onet"]
        f = self._conv(self._trunk("onet", ONET, x), sd["dense5.conv"], sd["dense5.bias"], sd["prelu5.weight"])        # [n,256,1,1]
        prob = self._prob1(self._conv(f, sd["dense6_1.weight"].view(2, 256, 1, 1), sd["dense6_1.bias"])).view(-1)
        reg = self._conv(f, sd["dense6_2.weight"].view(4, 256, 1, 1), sd["dense6_2.bias"]).view(-1, 4)
        pts = self._conv(f, sd["dense6_3.weight"].view(10, 256, 1, 1), sd["dense6_3.bias"]).view(-1, 10)
        return reg, pts, prob

    # ---- cascade --------------------------------------------------------------------------------------------------------
    def _crops(self, imgs, boxes, inds, size, w_, h):
        y, ey, x, ex = _pad(boxes, w_, h)
        ok = (ey > y - 1) & (ex > x - 1)
        spec = np.stack([inds, y - 1, ey, x - 1, ex], axis=1).astype(np.int32)[ok]
        return self._resample(imgs, spec, size, size), ok

    NMS_DEVICE_MIN = 256      # below this many candidates the host loop of _nms is cheaper than a launch + two copies

    def _bnms(self, boxes, scores, idxs, thr, method="Union", plus_one=False) -> np.ndarray:
        """_batched_nms with the O(n^2) overlap tests on the GPU (idb_nms_mask: the suppression bit matrix in the host routine's fp32
        operation order) and the greedy scan over its rows here: the same kept indices in the same order.  With thousands of
        candidates (synthetic weights; a crowded scene) the numpy loop was 0.84 s of a 1.02 s detect() on 16 images."""
        n = boxes.shape[0]
        if n < self.NMS_DEVICE_MIN:
            return _batched_nms(boxes, scores, idxs, thr, method, plus_one)
        order = np.argsort(-scores, kind="stable")
        bs = torch.from_numpy(np.ascontiguousarray(boxes[order, :4], dtype=np.float32)).to(self.device)
        im = torch.from_numpy(np.ascontiguousarray(idxs[order]).astype(np.int32)).to(self.device)
        words = (n + 63) // 64
        mask = torch.empty((n, words), dtype=torch.int64, device=self.device)
        L.check(self.lib.idb_nms_mask(bs.data_ptr(), im.data_ptr(), n, float(np.float32(thr)), int(method == "Min"), int(plus_one),
                                      mask.data_ptr(), self._st()), "idb_nms_mask")
        m = mask.cpu().numpy().view(np.uint64)
        removed = np.zeros(words, dtype=np.uint64)
        keep = []
        for i in range(n):
            if (int(removed[i >> 6]) >> (i & 63)) & 1:
                continue
            keep.append(i)
            removed |= m[i]
        kept = order[np.asarray(keep, dtype=np.int64)]
        kept = kept[np.argsort(idxs[kept], kind="stable")]           # per-image groups in score order, as _batched_nms concatenates them
        return kept[np.argsort(-scores[kept], kind="stable")]

    def detect_faces(self, imgs: torch.Tensor):
        """imgs uint8 [B,H,W,3] on the device -> (boxes [n,5] float32 incl. score, image index [n], points [n,5,2]) as numpy."""
        B, h, w_, _ = imgs.shape
        thr = self.thresholds
        full = np.array([[b, 0, h, 0, w_] for b in range(B)], dtype=np.int32)
        all_boxes, all_inds = [], []
        for scale in pyramid_scales(h, w_, self.min_face_size, self.factor):
            x = self._resample(imgs, full, int(h * scale + 1), int(w_ * scale + 1))
            reg, prob = self.pnet(x)
            prob_np = prob.cpu().numpy()
            mask = prob_np >= thr[0]
            if not mask.any():
                continue
            bi, yy, xx = np.nonzero(mask)
            r = reg.cpu().numpy()[bi, :, yy, xx]                                   # [n,4]
            bbx = np.stack([xx, yy], axis=1).astype(np.float32)
            q1 = np.floor((2 * bbx + 1) / np.float32(scale))
            q2 = np.floor((2 * bbx + 12 - 1 + 1) / np.float32(scale))
            bs = np.concatenate([q1, q2, prob_np[mask][:, None], r], axis=1).astype(np.float32)
            pick = self._bnms(bs[:, :4], bs[:, 4], bi, 0.5)
            all_boxes.append(bs[pick])
            all_inds.append(bi[pick])
        pts = np.zeros((0, 5, 2), dtype=np.float32)
        if not all_boxes:
            return np.zeros((0, 5), dtype=np.float32), np.zeros((0,), dtype=np.int64), pts
        boxes, inds = np.concatenate(all_boxes), np.concatenate(all_inds)
        pick = self._bnms(boxes[:, :4], boxes[:, 4], inds, 0.7)
        boxes, inds = boxes[pick], inds[pick]
        regw, regh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        boxes = np.stack([boxes[:, 0] + boxes[:, 5] * regw, boxes[:, 1] + boxes[:, 6] * regh, boxes[:, 2] + boxes[:, 7] * regw,
                          boxes[:, 3] + boxes[:, 8] * regh, boxes[:, 4]], axis=1)
        boxes = _rerec(boxes)
        # second stage
        if boxes.shape[0]:
            x, ok = self._crops(imgs, boxes, inds, 24, w_, h)
            boxes, inds = boxes[ok], inds[ok]
            reg, prob = self.rnet(x)
            score, mv = prob.cpu().numpy(), reg.cpu().numpy()
            ip = score > thr[1]
            boxes = np.concatenate([boxes[ip, :4], score[ip][:, None]], axis=1)
            inds, mv = inds[ip], mv[ip]
            pick = self._bnms(boxes[:, :4], boxes[:, 4], inds, 0.7)
            boxes, inds, mv = boxes[pick], inds[pick], mv[pick]
            boxes = _rerec(_bbreg(boxes, mv))
        # third stage
        if boxes.shape[0]:
            x, ok = self._crops(imgs, boxes, inds, 48, w_, h)
            boxes, inds = boxes[ok], inds[ok]
            reg, lm, prob = self.onet(x)
            score, mv, lm = prob.cpu().numpy(), reg.cpu().numpy(), lm.cpu().numpy()
            ip = score > thr[2]
            boxes = np.concatenate([boxes[ip, :4], score[ip][:, None]], axis=1)
            inds, mv, lm = inds[ip], mv[ip], lm[ip]
            wi, hi = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
            px = wi[:, None] * lm[:, :5] + boxes[:, 0:1] - 1
            py = hi[:, None] * lm[:, 5:10] + boxes[:, 1:2] - 1
            pts = np.stack([px, py], axis=2).astype(np.float32)
            boxes = _bbreg(boxes, mv)
            pick = self._bnms(boxes[:, :4], boxes[:, 4], inds, 0.7, "Min", plus_one=True)
            boxes, inds, pts = boxes[pick], inds[pick], pts[pick]
        return boxes.astype(np.float32), inds, pts

    def detect(self, img, landmarks: bool = False):
        """``mtcnn.detect(img_batch, landmarks=True)`` (detect_align_crop_data.py:99): uint8 NHWC batch (tensor or array) ->
        per-image object arrays of boxes [n,4], probabilities [n] and landmarks [n,5,2]; None where no face was found."""
        t = torch.as_tensor(np.asarray(img) if not torch.is_tensor(img) else img)
        single = t.ndim == 3
        if single:
            t = t[None]
        if t.dtype != torch.uint8 or t.ndim != 4 or t.shape[-1] != 3:
            raise ValueError("expected a uint8 image batch [B, H, W, 3]")
        t = t.to(self.device).contiguous()
        boxes, inds, pts = self.detect_faces(t)
        out_b, out_p, out_l = [], [], []
        for b in range(t.shape[0]):
            sel = np.nonzero(inds == b)[0]
            if sel.size == 0:
                out_b.append(None), out_p.append([None]), out_l.append(None)
                continue
            bb, pp, ll = boxes[sel, :4], boxes[sel, 4], pts[sel]
            if self.select_largest:
                order = np.argsort((bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1]), kind="stable")[::-1]
                bb, pp, ll = bb[order], pp[order], ll[order]
            # every face is returned, as upstream's detect() does: keep_all only acts in forward() / select_boxes there
            out_b.append(bb), out_p.append(pp), out_l.append(ll)
        ob, op, ol = np.empty(len(out_b), dtype=object), np.empty(len(out_b), dtype=object), np.empty(len(out_b), dtype=object)
        for i in range(len(out_b)):
            ob[i], op[i], ol[i] = out_b[i], out_p[i], out_l[i]
        if single:
            ob, op, ol = ob[0], op[0], ol[0]
        return (ob, op, ol) if landmarks else (ob, op)
