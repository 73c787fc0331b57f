"""Oracle (test infrastructure): heat-map -> text boxes, box grouping.

Restates (numpy, float32 where upstream is float32):
  ``easyocr/craft_utils.py::{getDetBoxes_core,adjustResultCoordinates}``,
  ``easyocr/detection.py::get_textbox`` (int32 cast),
  ``easyocr/utils.py::group_text_box``, ``easyocr/easyocr.py::Reader.detect`` (min_size),
  and the OpenCV 4.10 pieces they call: ``threshold`` (THRESH_BINARY),
  ``connectedComponentsWithStats`` (4-connectivity, raster-order labels), ``dilate``
  (rect kernel, centre anchor, isolated ROI), ``convexHull`` (Sklansky, clockwise=false),
  ``rotatingCalipers`` / ``minAreaRect`` / ``boxPoints``.
The reference reaches all of it through ``reader.readtext(...)`` at
``pipeline_demo/extractor/enhanced_extractor.py:520``.  PARITY UNPINNED (no cv2 here).
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage

f32 = np.float32


# ----------------------------------------------------------------------------- CCL
def connected_components_4(mask: np.ndarray):
    """cv2.connectedComponentsWithStats(mask, connectivity=4) -> (n, labels i32, stats[n,5]).

    Labels are numbered in raster order of each component's first pixel (what the
    SAUF labelling + flatten of OpenCV yields).  stats = (left, top, width, height, area).
    """
    structure = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
    labels, n = ndimage.label(mask != 0, structure=structure)
    labels = labels.astype(np.int32)
    stats = np.zeros((n + 1, 5), dtype=np.int32)
    objs = ndimage.find_objects(labels)
    areas = np.bincount(labels.ravel(), minlength=n + 1)
    for k, sl in enumerate(objs, start=1):
        ys, xs = sl
        stats[k] = (xs.start, ys.start, xs.stop - xs.start, ys.stop - ys.start, areas[k])
    stats[0, 4] = areas[0]
    return n + 1, labels, stats


# ------------------------------------------------------------------- convex hull
def convex_hull_ccw(points: np.ndarray) -> np.ndarray:
    """cv2.convexHull(points, clockwise=False, returnPoints=True) on integer points.

    Strictly convex vertices, starting at the right-most point (max x, then max y) and
    visiting max-y, min-x, min-y in that cyclic order (OpenCV's output for clockwise=false).
    """
    pts = np.unique(np.asarray(points, dtype=np.int64).reshape(-1, 2), axis=0)  # sorted by x then y
    n = len(pts)
    if n <= 1:
        return pts.astype(np.float32)
    if n == 2:
        return pts[::-1].astype(np.float32)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []  # min-y side, left -> right
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []  # max-y side, right -> left
    for p in pts[::-1]:
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    hull = upper[:-1] + lower[:-1]  # starts at pts[-1] = (max x, max y among those)
    return np.array(hull, dtype=np.float32)


# ------------------------------------------------------------- rotating calipers
def _rotating_calipers_minarea(points: np.ndarray):
    """imgproc/src/rotcalipers.cpp::rotatingCalipers(CALIPERS_MINAREARECT), float32 step by step."""
    n = len(points)
    px = points[:, 0].astype(f32)
    py = points[:, 1].astype(f32)
    vx = np.zeros(n, f32)
    vy = np.zeros(n, f32)
    inv = np.zeros(n, f32)
    left = bottom = right = top = 0
    left_x = right_x = px[0]
    top_y = bottom_y = py[0]
    for i in range(n):
        if px[i] < left_x:
            left_x, left = px[i], i
        if px[i] > right_x:
            right_x, right = px[i], i
        if py[i] > top_y:
            top_y, top = py[i], i
        if py[i] < bottom_y:
            bottom_y, bottom = py[i], i
        j = i + 1 if i + 1 < n else 0
        dx = float(px[j]) - float(px[i])
        dy = float(py[j]) - float(py[i])
        vx[i] = f32(dx)
        vy[i] = f32(dy)
        inv[i] = f32(1.0 / math.sqrt(dx * dx + dy * dy))
    orientation = f32(0)
    ax, ay = float(vx[n - 1]), float(vy[n - 1])
    for i in range(n):
        bx, by = float(vx[i]), float(vy[i])
        convexity = ax * by - ay * bx
        if convexity != 0:
            orientation = f32(1.0) if convexity > 0 else f32(-1.0)
            break
        ax, ay = bx, by
    if orientation == 0:
        raise ValueError("degenerate hull")
    base_a, base_b = orientation, f32(0)
    seq = [bottom, right, top, left]
    minarea = f32(np.finfo(np.float32).max)
    best = None
    for _ in range(n):
        dp = [
            f32(f32(base_a * vx[seq[0]]) + f32(base_b * vy[seq[0]])),
            f32(f32(-base_b * vx[seq[1]]) + f32(base_a * vy[seq[1]])),
            f32(f32(-base_a * vx[seq[2]]) - f32(base_b * vy[seq[2]])),
            f32(f32(base_b * vx[seq[3]]) - f32(base_a * vy[seq[3]])),
        ]
        maxcos = f32(dp[0] * inv[seq[0]])
        main = 0
        for i in range(1, 4):
            c = f32(dp[i] * inv[seq[i]])
            if c > maxcos:
                main, maxcos = i, c
        p = seq[main]
        lead_x = f32(vx[p] * inv[p])
        lead_y = f32(vy[p] * inv[p])
        if main == 0:
            base_a, base_b = lead_x, lead_y
        elif main == 1:
            base_a, base_b = lead_y, f32(-lead_x)
        elif main == 2:
            base_a, base_b = f32(-lead_x), f32(-lead_y)
        else:
            base_a, base_b = f32(-lead_y), lead_x
        seq[main] = 0 if seq[main] + 1 == n else seq[main] + 1
        dx = f32(px[seq[1]] - px[seq[3]])
        dy = f32(py[seq[1]] - py[seq[3]])
        width = f32(f32(dx * base_a) + f32(dy * base_b))
        dx = f32(px[seq[2]] - px[seq[0]])
        dy = f32(py[seq[2]] - py[seq[0]])
        height = f32(f32(-dx * base_b) + f32(dy * base_a))
        area = f32(width * height)
        if area <= minarea:
            minarea = area
            best = (seq[3], base_a, width, base_b, height, seq[0])
    li, A1, w, B1, h, bi = best
    A2, B2 = f32(-B1), A1
    C1 = f32(f32(A1 * px[li]) + f32(py[li] * B1))
    C2 = f32(f32(A2 * px[bi]) + f32(py[bi] * B2))
    idet = f32(f32(1.0) / f32(f32(A1 * B2) - f32(A2 * B1)))
    ox = f32(f32(f32(C1 * B2) - f32(C2 * B1)) * idet)
    oy = f32(f32(f32(A1 * C2) - f32(A2 * C1)) * idet)
    return (ox, oy), (f32(A1 * w), f32(B1 * w)), (f32(A2 * h), f32(B2 * h))


def min_area_rect(points: np.ndarray):
    """cv2.minAreaRect -> ((cx, cy), (w, h), angle_deg) with float32 members."""
    hull = convex_hull_ccw(points)
    n = len(hull)
    cx = cy = w = h = f32(0)
    ang = f32(0)
    if n > 2:
        o, v1, v2 = _rotating_calipers_minarea(hull)
        cx = f32(o[0] + f32(f32(v1[0] + v2[0]) * f32(0.5)))
        cy = f32(o[1] + f32(f32(v1[1] + v2[1]) * f32(0.5)))
        w = f32(math.sqrt(float(v1[0]) * float(v1[0]) + float(v1[1]) * float(v1[1])))
        h = f32(math.sqrt(float(v2[0]) * float(v2[0]) + float(v2[1]) * float(v2[1])))
        ang = f32(math.atan2(float(v1[1]), float(v1[0])))
    elif n == 2:
        cx = f32(f32(hull[0, 0] + hull[1, 0]) * f32(0.5))
        cy = f32(f32(hull[0, 1] + hull[1, 1]) * f32(0.5))
        dx = float(hull[1, 0]) - float(hull[0, 0])
        dy = float(hull[1, 1]) - float(hull[0, 1])
        w = f32(math.sqrt(dx * dx + dy * dy))
        ang = f32(math.atan2(dy, dx))
    elif n == 1:
        cx, cy = f32(hull[0, 0]), f32(hull[0, 1])
    ang = f32(float(ang) * 180.0 / math.pi)
    return (cx, cy), (w, h), ang


def box_points(rect) -> np.ndarray:
    """cv2.boxPoints / RotatedRect::points -> [4,2] float32."""
    (cx, cy), (w, h), angle = rect
    _angle = float(angle) * math.pi / 180.0
    b = f32(f32(math.cos(_angle)) * f32(0.5))
    a = f32(f32(math.sin(_angle)) * f32(0.5))
    pt = np.zeros((4, 2), dtype=f32)
    pt[0, 0] = f32(f32(cx - f32(a * h)) - f32(b * w))
    pt[0, 1] = f32(f32(cy + f32(b * h)) - f32(a * w))
    pt[1, 0] = f32(f32(cx + f32(a * h)) - f32(b * w))
    pt[1, 1] = f32(f32(cy - f32(b * h)) - f32(a * w))
    pt[2, 0] = f32(f32(f32(2) * cx) - pt[0, 0])
    pt[2, 1] = f32(f32(f32(2) * cy) - pt[0, 1])
    pt[3, 0] = f32(f32(f32(2) * cx) - pt[1, 0])
    pt[3, 1] = f32(f32(f32(2) * cy) - pt[1, 1])
    return pt


# ------------------------------------------------------------------ getDetBoxes
def dilate_extents(niter: int):
    """cv2.dilate with a (1+niter)^2 rect, centre anchor: reach (left/up, right/down)."""
    return niter // 2, (niter + 1) // 2


def component_box(seg_points_xy: np.ndarray) -> np.ndarray:
    """minAreaRect + boxPoints + diamond fix + clockwise start of getDetBoxes_core."""
    rect = min_area_rect(seg_points_xy)
    box = box_points(rect)
    w = np.linalg.norm(box[0] - box[1])
    h = np.linalg.norm(box[1] - box[2])
    w, h = float(w), float(h)   # numpy 1.26 (pinned upstream) promotes f32 scalar + python float to f64
    box_ratio = max(w, h) / (min(w, h) + 1e-5)
    if abs(1 - box_ratio) <= 0.1:
        l, r = seg_points_xy[:, 0].min(), seg_points_xy[:, 0].max()
        t, b = seg_points_xy[:, 1].min(), seg_points_xy[:, 1].max()
        box = np.array([[l, t], [r, t], [r, b], [l, b]], dtype=np.float32)
    startidx = box.sum(axis=1).argmin()
    box = np.roll(box, 4 - startidx, 0)
    return np.array(box)


def get_det_boxes_core(textmap, linkmap, text_threshold=0.7, link_threshold=0.4, low_text=0.4):
    """craft_utils.py::getDetBoxes_core -> (det [list of [4,2] f32], labels, mapper)."""
    textmap = np.asarray(textmap, dtype=np.float32)
    linkmap = np.asarray(linkmap, dtype=np.float32)
    img_h, img_w = textmap.shape
    text_score = (textmap > f32(low_text)).astype(np.float32)   # cv2.threshold(..., 1, THRESH_BINARY)
    link_score = (linkmap > f32(link_threshold)).astype(np.float32)
    comb = np.clip(text_score + link_score, 0, 1).astype(np.uint8)
    n_labels, labels, stats = connected_components_4(comb)
    det, mapper = [], []
    for k in range(1, n_labels):
        x, y, w, h, size = (int(v) for v in stats[k])
        if size < 10:
            continue
        comp = labels[y:y + h, x:x + w] == k
        if np.max(textmap[y:y + h, x:x + w][comp]) < text_threshold:
            continue
        seg = comp & (text_score[y:y + h, x:x + w] == 1)     # link-only pixels removed
        niter = int(math.sqrt(size * min(w, h) / (w * h)) * 2)
        sx, ex, sy, ey = x - niter, x + w + niter + 1, y - niter, y + h + niter + 1
        sx = max(sx, 0)
        sy = max(sy, 0)
        ex = min(ex, img_w)
        ey = min(ey, img_h)
        roi = np.zeros((ey - sy, ex - sx), dtype=bool)
        roi[y - sy:y - sy + h, x - sx:x - sx + w] = seg
        lo, hi = dilate_extents(niter)
        ys, xs = np.nonzero(roi)
        out = np.zeros_like(roi)
        # rectangle dilation = union of shifted copies; done via cumulative reach per axis
        if len(ys):
            tmp = np.zeros_like(roi)
            for d in range(-lo, hi + 1):
                xx = xs + d
                ok = (xx >= 0) & (xx < roi.shape[1])
                tmp[ys[ok], xx[ok]] = True
            ys2, xs2 = np.nonzero(tmp)
            for d in range(-lo, hi + 1):
                yy = ys2 + d
                ok = (yy >= 0) & (yy < roi.shape[0])
                out[yy[ok], xs2[ok]] = True
        ys, xs = np.nonzero(out)
        pts = np.stack([xs + sx, ys + sy], axis=1)             # (x, y), raster order
        det.append(component_box(pts))
        mapper.append(k)
    return det, labels, mapper


def adjust_result_coordinates(polys, ratio_w, ratio_h, ratio_net=2):
    """craft_utils.py::adjustResultCoordinates (float32 array *= float64 pair)."""
    out = []
    for p in polys:
        q = (np.asarray(p, dtype=np.float32).astype(np.float64) * np.array([ratio_w * ratio_net, ratio_h * ratio_net])).astype(np.float32)
        out.append(q)
    return out


def boxes_to_int_polys(boxes) -> list:
    """detection.py::get_textbox: np.array(box).astype(np.int32).reshape(-1) (truncation)."""
    return [np.array(b).astype(np.int32).reshape(-1) for b in boxes]


# ----------------------------------------------------------------- group_text_box
def group_text_box(polys, slope_ths=0.1, ycenter_ths=0.5, height_ths=0.5, width_ths=1.0, add_margin=0.05, sort_output=True):
    """easyocr/utils.py::group_text_box -> (merged_list [[xmin,xmax,ymin,ymax]], free_list)."""
    horizontal_list, free_list, combined_list, merged_list = [], [], [], []
    for poly in polys:
        poly = [int(v) for v in poly]
        slope_up = (poly[3] - poly[1]) / np.maximum(10, (poly[2] - poly[0]))
        slope_down = (poly[5] - poly[7]) / np.maximum(10, (poly[4] - poly[6]))
        if max(abs(slope_up), abs(slope_down)) < slope_ths:
            x_max = max([poly[0], poly[2], poly[4], poly[6]])
            x_min = min([poly[0], poly[2], poly[4], poly[6]])
            y_max = max([poly[1], poly[3], poly[5], poly[7]])
            y_min = min([poly[1], poly[3], poly[5], poly[7]])
            horizontal_list.append([x_min, x_max, y_min, y_max, 0.5 * (y_min + y_max), y_max - y_min])
        else:
            height = np.linalg.norm([poly[6] - poly[0], poly[7] - poly[1]])
            width = np.linalg.norm([poly[2] - poly[0], poly[3] - poly[1]])
            margin = int(1.44 * add_margin * min(width, height))
            theta13 = abs(np.arctan((poly[1] - poly[5]) / np.maximum(10, (poly[0] - poly[4]))))
            theta24 = abs(np.arctan((poly[3] - poly[7]) / np.maximum(10, (poly[2] - poly[6]))))
            x1 = poly[0] - np.cos(theta13) * margin
            y1 = poly[1] - np.sin(theta13) * margin
            x2 = poly[2] + np.cos(theta24) * margin
            y2 = poly[3] - np.sin(theta24) * margin
            x3 = poly[4] + np.cos(theta13) * margin
            y3 = poly[5] + np.sin(theta13) * margin
            x4 = poly[6] - np.cos(theta24) * margin
            y4 = poly[7] + np.sin(theta24) * margin
            free_list.append([[x1, y1], [x2, y2], [x3, y3], [x4, y4]])
    if sort_output:
        horizontal_list = sorted(horizontal_list, key=lambda item: item[4])
    new_box = []
    for poly in horizontal_list:
        if len(new_box) == 0:
            b_height = [poly[5]]
            b_ycenter = [poly[4]]
            new_box.append(poly)
        else:
            if abs(np.mean(b_ycenter) - poly[4]) < ycenter_ths * np.mean(b_height):
                b_height.append(poly[5])
                b_ycenter.append(poly[4])
                new_box.append(poly)
            else:
                b_height = [poly[5]]
                b_ycenter = [poly[4]]
                combined_list.append(new_box)
                new_box = [poly]
    combined_list.append(new_box)
    for boxes in combined_list:
        if len(boxes) == 1:
            box = boxes[0]
            margin = int(add_margin * min(box[1] - box[0], box[5]))
            merged_list.append([box[0] - margin, box[1] + margin, box[2] - margin, box[3] + margin])
        else:
            boxes = sorted(boxes, key=lambda item: item[0])
            merged_box, new_box = [], []
            for box in boxes:
                if len(new_box) == 0:
                    b_height = [box[5]]
                    x_max = box[1]
                    new_box.append(box)
                else:
                    if (abs(np.mean(b_height) - box[5]) < height_ths * np.mean(b_height)) and ((box[0] - x_max) < width_ths * (box[3] - box[2])):
                        b_height.append(box[5])
                        x_max = box[1]
                        new_box.append(box)
                    else:
                        b_height = [box[5]]
                        x_max = box[1]
                        merged_box.append(new_box)
                        new_box = [box]
            if len(new_box) > 0:
                merged_box.append(new_box)
            for mbox in merged_box:
                if len(mbox) != 1:
                    x_min = min(mbox, key=lambda x: x[0])[0]
                    x_max = max(mbox, key=lambda x: x[1])[1]
                    y_min = min(mbox, key=lambda x: x[2])[2]
                    y_max = max(mbox, key=lambda x: x[3])[3]
                    box_width = x_max - x_min
                    box_height = y_max - y_min
                    margin = int(add_margin * (min(box_width, box_height)))
                    merged_list.append([x_min - margin, x_max + margin, y_min - margin, y_max + margin])
                else:
                    box = mbox[0]
                    box_width = box[1] - box[0]
                    box_height = box[3] - box[2]
                    margin = int(add_margin * (min(box_width, box_height)))
                    merged_list.append([box[0] - margin, box[1] + margin, box[2] - margin, box[3] + margin])
    return merged_list, free_list


def _diff(values):
    return max(values) - min(values)


def detect_from_heatmap(score_text, score_link, ratio, *, min_size=20, text_threshold=0.7, low_text=0.4,
                        link_threshold=0.4, slope_ths=0.1, ycenter_ths=0.5, height_ths=0.5, width_ths=0.5,
                        add_margin=0.1):
    """detection.test_net tail + get_textbox + Reader.detect for one image -> (horizontal_list, free_list, polys)."""
    boxes, _, _ = get_det_boxes_core(score_text, score_link, text_threshold, link_threshold, low_text)
    ratio_w = ratio_h = 1 / ratio
    boxes = adjust_result_coordinates(boxes, ratio_w, ratio_h)
    polys = boxes_to_int_polys(boxes)
    horizontal_list, free_list = group_text_box(polys, slope_ths, ycenter_ths, height_ths, width_ths, add_margin, True)
    if min_size:
        horizontal_list = [i for i in horizontal_list if max(i[1] - i[0], i[3] - i[2]) > min_size]
        free_list = [i for i in free_list if max(_diff([c[0] for c in i]), _diff([c[1] for c in i])) > min_size]
    return horizontal_list, free_list, polys
