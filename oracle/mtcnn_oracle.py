"""ORACLE — test infrastructure only.  CPU fp32 restatement of facenet_pytorch's MTCNN detector as the reference uses it
(/root/reference/utils/detect_align_crop_data.py:18-20 ``MTCNN(select_largest=True, post_process=False)``, :99
``mtcnn.detect(img_batch, landmarks=True)``).

PARITY UNPINNED: ``facenet_pytorch`` (unpinned in the reference's imports, not in requirements.txt) is not installed, its
trained P/R/O-Net weights ship inside that wheel, and the reference holds no fixtures for this stage.  This file restates the
published algorithm (Zhang et al. 2016; facenet_pytorch ``models/mtcnn.py`` PNet/RNet/ONet and ``models/utils/detect_face.py``
``detect_face``) with plain torch ops: F.interpolate(mode="area"), F.conv2d, F.prelu, F.max_pool2d(ceil_mode=True), softmax,
torchvision-style NMS written out as an IoU-matrix sweep.  Only tests/ may import it."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _trunk(sd: SD, x, plan):
    for op in plan:
        if op[0] == "c":
            x = F.prelu(F.conv2d(x, sd[f"conv{op[1]}.weight"], sd[f"conv{op[1]}.bias"]), sd[f"prelu{op[1]}.weight"])
        else:
            x = F.max_pool2d(x, op[1], op[2], ceil_mode=True)
    return x


def pnet(sd: SD, x):
    x = _trunk(sd, x, [("c", 1), ("p", 2, 2), ("c", 2), ("c", 3)])
    a = F.softmax(F.conv2d(x, sd["conv4_1.weight"], sd["conv4_1.bias"]), dim=1)
    return F.conv2d(x, sd["conv4_2.weight"], sd["conv4_2.bias"]), a


def rnet(sd: SD, x):
    x = _trunk(sd, x, [("c", 1), ("p", 3, 2), ("c", 2), ("p", 3, 2), ("c", 3)])
    x = x.permute(0, 3, 2, 1).contiguous()
    x = F.prelu(F.linear(x.view(x.shape[0], -1), sd["dense4.weight"], sd["dense4.bias"]), sd["prelu4.weight"])
    a = F.softmax(F.linear(x, sd["dense5_1.weight"], sd["dense5_1.bias"]), dim=1)
    return F.linear(x, sd["dense5_2.weight"], sd["dense5_2.bias"]), a


def onet(sd: SD, x):
    x = _trunk(sd, x, [("c", 1), ("p", 3, 2), ("c", 2), ("p", 3, 2), ("c", 3), ("p", 2, 2), ("c", 4)])
    x = x.permute(0, 3, 2, 1).contiguous()
    x = F.prelu(F.linear(x.view(x.shape[0], -1), sd["dense5.weight"], sd["dense5.bias"]), sd["prelu5.weight"])
    a = F.softmax(F.linear(x, sd["dense6_1.weight"], sd["dense6_1.bias"]), dim=1)
    return F.linear(x, sd["dense6_2.weight"], sd["dense6_2.bias"]), F.linear(x, sd["dense6_3.weight"], sd["dense6_3.bias"]), a


def imresample(img, sz):
    return F.interpolate(img, size=sz, mode="area")


def nms(boxes, scores, thr, method="Union", plus_one=False):
    """Greedy NMS through a pairwise-overlap matrix: box j is dropped if a KEPT box with a higher score overlaps it by more than thr."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].double()
    one = 1.0 if plus_one else 0.0
    area = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    w = (torch.minimum(b[:, None, 2], b[None, :, 2]) - torch.maximum(b[:, None, 0], b[None, :, 0]) + one).clamp(min=0)
    h = (torch.minimum(b[:, None, 3], b[None, :, 3]) - torch.maximum(b[:, None, 1], b[None, :, 1]) + one).clamp(min=0)
    inter = w * h
    ov = inter / torch.minimum(area[:, None], area[None, :]) if method == "Min" else inter / (area[:, None] + area[None, :] - inter)
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if alive[i]:
            keep.append(i)
            alive &= (ov[i] <= thr) if method == "Min" else ~(ov[i] > thr)    # NaN: nms_numpy ("Min") drops, torchvision ("Union") keeps
            alive[i] = False
    return order[torch.tensor(keep, dtype=torch.long)]


def batched_nms(boxes, scores, idxs, thr, method="Union", plus_one=False):
    keep = [torch.nonzero(idxs == b).flatten()[nms(boxes[idxs == b].float(), scores[idxs == b], thr, method, plus_one)] for b in idxs.unique()]
    if not keep:
        return torch.zeros(0, dtype=torch.long)
    keep = torch.cat(keep)
    return keep[torch.argsort(scores[keep], descending=True, stable=True)]


def bbreg(bb, reg):
    w = bb[:, 2] - bb[:, 0] + 1
    h = bb[:, 3] - bb[:, 1] + 1
    out = bb.clone()
    out[:, :4] = torch.stack([bb[:, 0] + reg[:, 0] * w, bb[:, 1] + reg[:, 1] * h, bb[:, 2] + reg[:, 2] * w, bb[:, 3] + reg[:, 3] * h], dim=1)
    return out


def rerec(bb):
    h, w = bb[:, 3] - bb[:, 1], bb[:, 2] - bb[:, 0]
    l = torch.max(w, h)
    out = bb.clone()
    out[:, 0] = bb[:, 0] + w * 0.5 - l * 0.5
    out[:, 1] = bb[:, 1] + h * 0.5 - l * 0.5
    out[:, 2:4] = out[:, :2] + l[:, None]
    return out


def pad(bb, w, h):
    b = bb[:, :4].trunc().int()
    x, y, ex, ey = b[:, 0].clamp(min=1), b[:, 1].clamp(min=1), b[:, 2].clamp(max=w), b[:, 3].clamp(max=h)
    return y, ey, x, ex


def _crops(imgs, boxes, inds, size, w, h):
    y, ey, x, ex = pad(boxes, w, h)
    out, ok = [], []
    for k in range(boxes.shape[0]):
        good = bool(ey[k] > y[k] - 1) and bool(ex[k] > x[k] - 1)
        ok.append(good)
        if good:
            out.append(imresample(imgs[inds[k], :, (y[k] - 1):ey[k], (x[k] - 1):ex[k]].unsqueeze(0), (size, size)))
    ok = torch.tensor(ok, dtype=torch.bool)
    return ((torch.cat(out) - 127.5) * 0.0078125 if out else torch.zeros(0, 3, size, size)), ok


@torch.no_grad()
def detect_face(imgs_u8: torch.Tensor, w: Dict[str, SD], minsize: int = 20, threshold=(0.6, 0.7, 0.7), factor: float = 0.709):
    """imgs uint8 [B,H,W,3] -> (boxes [n,5], image index [n], points [n,5,2])."""
    imgs = imgs_u8.permute(0, 3, 1, 2).float()
    h, wd = imgs.shape[2:4]
    m = 12.0 / minsize
    minl = min(h, wd) * m
    scale_i, scales = m, []
    while minl >= 12:
        scales.append(scale_i)
        scale_i *= factor
        minl *= factor
    boxes, image_inds = [], []
    for scale in scales:
        im = (imresample(imgs, (int(h * scale + 1), int(wd * scale + 1))) - 127.5) * 0.0078125
        reg, probs = pnet(w["pnet"], im)
        p1 = probs[:, 1]
        mask = p1 >= threshold[0]
        mi = mask.nonzero()
        if mi.shape[0] == 0:
            continue
        score = p1[mask]
        r = reg.permute(1, 0, 2, 3)[:, mask].permute(1, 0)
        bb = mi[:, 1:].float().flip(1)
        q1 = ((2 * bb + 1) / scale).floor()
        q2 = ((2 * bb + 12 - 1 + 1) / scale).floor()
        bs = torch.cat([q1, q2, score[:, None], r], dim=1)
        pick = batched_nms(bs[:, :4], bs[:, 4], mi[:, 0], 0.5)
        boxes.append(bs[pick])
        image_inds.append(mi[:, 0][pick])
    points = torch.zeros(0, 5, 2)
    if not boxes:
        return torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), points
    boxes, inds = torch.cat(boxes), torch.cat(image_inds)
    pick = batched_nms(boxes[:, :4], boxes[:, 4], inds, 0.7)
    boxes, inds = boxes[pick], inds[pick]
    regw, regh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    boxes = torch.stack([boxes[:, 0] + boxes[:, 5] * regw, boxes[:, 1] + boxes[:, 6] * regh, boxes[:, 2] + boxes[:, 7] * regw,
                         boxes[:, 3] + boxes[:, 8] * regh, boxes[:, 4]], dim=1)
    boxes = rerec(boxes)
    if boxes.shape[0]:
        im, ok = _crops(imgs, boxes, inds, 24, wd, h)
        boxes, inds = boxes[ok], inds[ok]
        out0, out1 = rnet(w["rnet"], im)
        score = out1[:, 1]
        ip = score > threshold[1]
        boxes = torch.cat([boxes[ip, :4], score[ip][:, None]], dim=1)
        inds, mv = inds[ip], out0[ip]
        pick = batched_nms(boxes[:, :4], boxes[:, 4], inds, 0.7)
        boxes, inds, mv = boxes[pick], inds[pick], mv[pick]
        boxes = rerec(bbreg(boxes, mv))
    if boxes.shape[0]:
        im, ok = _crops(imgs, boxes, inds, 48, wd, h)
        boxes, inds = boxes[ok], inds[ok]
        out0, out1, out2 = onet(w["onet"], im)
        score = out2[:, 1]
        ip = score > threshold[2]
        boxes = torch.cat([boxes[ip, :4], score[ip][:, None]], dim=1)
        inds, mv, lm = inds[ip], out0[ip], out1[ip]
        wi, hi = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
        px = wi[:, None] * lm[:, :5] + boxes[:, 0:1] - 1
        py = hi[:, None] * lm[:, 5:10] + boxes[:, 1:2] - 1
        points = torch.stack([px, py], dim=2)
        boxes = bbreg(boxes, mv)
        pick = batched_nms(boxes[:, :4], boxes[:, 4], inds, 0.7, "Min", plus_one=True)
        boxes, inds, points = boxes[pick], inds[pick], points[pick]
    return boxes, inds, points


def select_largest_first(boxes, inds, points, image: int):
    sel = torch.nonzero(inds == image).flatten()
    if sel.numel() == 0:
        return None, None, None
    bb, pp, ll = boxes[sel, :4], boxes[sel, 4], points[sel]
    order = torch.argsort((bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1]), stable=True).flip(0)
    return bb[order][:1], pp[order][:1], ll[order][:1]
