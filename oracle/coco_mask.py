"""Oracle: COCO polygon -> binary mask, the `dataset.annToMask(instance)` the reference calls inside get_pts_in_mask
(see/surface_completion/datasets/shared_utils.py:66).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  annToMask is pycocotools (a third-party dependency of the reference's requirements, not vendored under /root/reference and not
installed in this image; the reference pins no version -- cocoapi PythonAPI 2.0.x): COCO.annToMask -> annToRLE -> mask.frPyObjects -> rleFrPoly /
rleMerge / rleDecode in common/maskApi.c.  This file restates those published routines step by step (same integer / double arithmetic, same order):
  rle_from_polygon   maskApi.c rleFrPoly: vertices scaled by 5 and rounded, every edge walked one step per unit of its longer axis, a boundary point
                     wherever the walk's x changes, down-sampled to a (column, row) run boundary, keys x * h + y sorted, runs = differences with
                     zero-length runs merged away
  merge_union        maskApi.c rleMerge(intersect = 0) of an annotation's polygons, here on the decoded masks (union of the parts)
  decode             maskApi.c rleDecode: column-major runs, starting with zeros
  ann_to_mask        pycocotools/coco.py annToRLE + annToMask for polygon lists and for uncompressed / compressed RLE dicts
The only anchors available here are hand-derived known answers (tests/test_isolation.py): axis-aligned rectangles, a triangle, runs that end on the
image border.

Shrinking (shrink_distance, ann_to_mask_shrunk): the reference moves every polygon's boundary inwards with shapely before annToMask
(shared_utils.py:295-330: d = half diagonal of the bounding box x percentage / 100, Polygon.buffer(-d), exterior vertices truncated to int).  shapely /
GEOS are not available here, and their vertex list (arcs as 16 chords per quadrant, noding, int truncation) is not restated: this file states the REGION
a negative buffer describes -- the points of the polygon at least d from its boundary -- sampled at the pixel centres of the polygon's own mask.  Equal
to the reference's mask except in a band of about one pixel along the shrunken boundary.  PARITY UNPINNED, and known to deviate in that band.
Since round 6 the product's get_pts_in_mask follows the vertex-list form instead (oracle/polygon_buffer.py); this region form checks
sv_polygons_to_masks_shrunk, which stays exported."""
import math

import numpy as np


def rle_from_polygon(xy, h, w):
    """xy: flat [x0, y0, x1, y1, ...] (floats) -> list of run lengths (column-major, first run = zeros)"""
    k = len(xy) // 2
    scale = 5.0
    x = [int(scale * xy[2 * j] + .5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        s = (float(ye - ys) / dx if dx else 0.0) if dx >= dy else float(xe - xs) / dy       # dx == dy == 0: one point, s unused (C divides 0.0 / 0 -> nan * 0)
        if dx >= dy:
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + .5) if dx else ys)
        else:
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + .5))
    # points along the y-boundary, down-sampled
    keys = []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + .5) / scale - .5
            if math.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + .5) / scale - .5
            yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
            yd = math.ceil(yd)
            keys.append(int(xd) * h + int(yd))
    keys.append(h * w)
    keys.sort()
    a, p = [], 0
    for t in keys:
        a.append(t - p)
        p = t
    b = [a[0]]
    j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return b


def decode(counts, h, w):
    """run lengths (column-major, zeros first) -> (h, w) uint8"""
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T.copy()


def rle_string_to_counts(s):
    """maskApi.c rleFrString: the compressed 'counts' string of an RLE dict -> run lengths"""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def ann_to_mask(ann, h, w):
    """pycocotools COCO.annToMask: polygons (list of flat lists) -> union of the parts; RLE dict -> decoded"""
    seg = ann["segmentation"]
    if isinstance(seg, list):
        m = np.zeros((h, w), np.uint8)
        for poly in seg:
            m |= decode(rle_from_polygon([float(t) for t in poly], h, w), h, w)
        return m
    counts = seg["counts"]
    hh, ww = seg["size"]
    return decode(counts if isinstance(counts, list) else rle_string_to_counts(counts), hh, ww)


def shrink_distance(xy, percentage):
    """shared_utils.py:295-306: distance from the bounding box's centre to its min corner, times percentage / 100"""
    xs, ys = xy[0::2], xy[1::2]
    return 0.5 * math.hypot(max(xs) - min(xs), max(ys) - min(ys)) * (percentage / 100.0)


def _edge_distance2(px, py, xy):
    k = len(xy) // 2
    best = np.full(px.shape, np.inf)
    for j in range(k):
        ax, ay = xy[2 * j], xy[2 * j + 1]
        bx, by = xy[2 * ((j + 1) % k)], xy[2 * ((j + 1) % k) + 1]
        ex, ey = bx - ax, by - ay
        len2 = ex * ex + ey * ey
        t = ((px - ax) * ex + (py - ay) * ey) / len2 if len2 > 0 else np.zeros(px.shape)
        t = np.clip(t, 0.0, 1.0)
        qx, qy = ax + t * ex - px, ay + t * ey - py
        best = np.minimum(best, qx * qx + qy * qy)
    return best


def ann_to_mask_shrunk(ann, h, w, percentage):
    """Union over the parts of (part's mask AND at least d from the part's boundary); an instance with a part that shrinks to nothing keeps its
    unshrunken mask (the reference returns the original polygons then: shared_utils.py:325-326)."""
    seg = ann["segmentation"]
    if percentage == 0 or not isinstance(seg, list):
        return ann_to_mask(ann, h, w)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w), np.uint8)
    for poly in seg:
        xy = [float(t) for t in poly[:2 * (len(poly) // 2)]]
        if len(xy) < 4:
            continue
        m = decode(rle_from_polygon(xy, h, w), h, w).astype(bool)
        d = shrink_distance(xy, percentage)
        part = m & (_edge_distance2(xx, yy, xy) >= d * d)
        if not part.any():
            return ann_to_mask(ann, h, w)
        out |= part.astype(np.uint8)
    return out
