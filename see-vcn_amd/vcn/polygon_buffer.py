"""Negative buffer of a simple polygon as a VERTEX LIST -- the `Polygon.buffer(-d)` + `exterior.coords` + `int()` of the reference's
shrink_instance_masks (see/surface_completion/datasets/shared_utils.py:295-330; every IMG_DET cfg sets SHRINK_MASK_PERCENTAGE: 3).

The reference takes the buffer from shapely / GEOS (third party, not under /root/reference, not installed here: PARITY UNPINNED -- nothing in this image
can produce GEOS's own output to compare with).  This module restates the published construction (GEOS operation/buffer: OffsetSegmentGenerator,
BufferBuilder) for the one case the reference uses -- a single ring, distance < 0, round joins, 16 segments per quadrant (shapely's default
`quad_segs` / `resolution`), mitre irrelevant:

  1. the ring is oriented counter-clockwise; every edge is moved to its LEFT (inwards) by d;
  2. at a vertex where the ring turns right (a reflex corner of the polygon) the two offset edges are joined by an arc around the vertex, radius d,
     clockwise, cut into chords the way OffsetSegmentGenerator::addDirectedFillet cuts it: n = int(angle / (pi / 32) + 0.5) equal steps from the end
     of the first offset edge (n < 1: no arc points); at a vertex where it turns left (a convex corner) the two offset edges overrun each other: where
     they cross, the crossing is the join (addInsideTurn adds that one point); where they do not (an edge shorter than the overrun) they are joined end
     to start -- a loop of winding number <= 0 that step 3 drops (GEOS routes that join through two points 1/81 of the way towards the vertex, inside the
     same dropped loop);
  3. the raw curve is noded (every segment cut at every crossing) and the result is the region of winding number >= 1: the pieces of the curve with
     that region on their left and not on their right, chained into rings (GEOS: depth labelling of the noded edges in BufferBuilder /
     PolygonBuilder).  Rings of positive area are the exteriors of the result's polygons -- one: a Polygon, several: a MultiPolygon, none: empty;
     rings of negative area are holes, which the reference never reads (`r_poly.exterior`);
  4. shrink_instance_masks: every exterior's vertices, closed (first vertex repeated), truncated with int() and flattened to [x0, y0, x1, y1, ...]
     -- the polygon annToMask then rasterises (here: sv_polygons_to_masks).

Known differences from GEOS that survive int(): none intended.  Not restated: BufferInputLineSimplifier (GEOS drops input vertices that deviate less
than 0.01 d from the line through their neighbours before it offsets; with pixel coordinates and d of a few pixels that is a vertex 0.03 pixels off
a straight run), the starting vertex and the direction of a ring (the rasterisation does not depend on either), repeated-point removal below
1e-6 d.  Input rings are taken to be simple (a self-intersecting annotation has no defined interior in shapely either: it raises or returns what
GEOS's noder makes of it)."""
import math
import warnings

import numpy as np

QUAD_SEGS = 16                      # shapely's default quad_segs (resolution) of buffer()


def shrink_distance(xs, ys, percentage):
    """shrink_shapely_polygon, shared_utils.py:295-306: distance from the bounding box's centre to its min corner x percentage / 100"""
    x_center, y_center = 0.5 * min(xs) + 0.5 * max(xs), 0.5 * min(ys) + 0.5 * max(ys)
    return math.hypot(x_center - min(xs), y_center - min(ys)) * (percentage / 100)


def _clean_ring(pts):
    """(n, 2) float64 without the closing duplicate and without repeated consecutive vertices, counter-clockwise; None: fewer than 3 vertices / no area"""
    p = np.asarray(pts, np.float64).reshape(-1, 2)
    if len(p) and np.array_equal(p[0], p[-1]):
        p = p[:-1]
    if len(p) < 3:
        return None
    keep = np.any(p != np.roll(p, 1, axis=0), axis=1)
    p = p[keep]
    if len(p) < 3:
        return None
    x, y = p[:, 0], p[:, 1]
    area2 = float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))
    if area2 == 0.0:
        return None
    return p if area2 > 0 else p[::-1].copy()


def raw_offset_curve(p, d, quad_segs=QUAD_SEGS):
    """p: (n, 2) counter-clockwise ring -> (m, 2) vertices of the closed raw offset curve at distance d to the left (steps 1 and 2 of the header)"""
    n = len(p)
    nxt = np.roll(p, -1, axis=0)
    e = nxt - p
    ln = np.hypot(e[:, 0], e[:, 1])
    nl = np.stack([-e[:, 1], e[:, 0]], axis=1) / ln[:, None]                 # unit left normals
    a, b = p + d * nl, nxt + d * nl                                           # offset edge i: a[i] -> b[i]
    quantum = (math.pi / 2.0) / quad_segs
    # the join behind every edge: at a LEFT turn whose two offset edges cross, the crossing replaces the end of the one and the start of the other
    # (OffsetSegmentGenerator::addInsideTurn adds exactly that point); a nearly straight left turn would otherwise leave a back-and-forth of length
    # d x angle beside the curve, thinner than any probe
    start_pt, end_pt = a.copy(), b.copy()
    left = np.zeros(n, bool)
    for i in range(n):
        j = (i + 1) % n
        cross = e[i, 0] * e[j, 1] - e[i, 1] * e[j, 0]
        if cross > 0:
            left[i] = True
            # a[i] + t e[i] = a[j] + u e[j]
            wx, wy = a[j, 0] - a[i, 0], a[j, 1] - a[i, 1]
            t = (wx * e[j, 1] - wy * e[j, 0]) / cross
            u = (wx * e[i, 1] - wy * e[i, 0]) / cross
            if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
                x = a[i] + t * e[i]
                end_pt[i] = x
                start_pt[j] = x
    out = []
    for i in range(n):
        # edge i, then the join at vertex i + 1 towards edge i + 1
        out.append(start_pt[i])
        out.append(end_pt[i])
        j = (i + 1) % n
        cross = e[i, 0] * e[j, 1] - e[i, 1] * e[j, 0]
        dot = e[i, 0] * e[j, 0] + e[i, 1] * e[j, 1]
        if cross < 0 or (cross == 0 and dot < 0):                             # right turn (or a reversal): round join, clockwise around the vertex
            c = nxt[i]
            start = math.atan2(b[i, 1] - c[1], b[i, 0] - c[0])
            end = math.atan2(a[j, 1] - c[1], a[j, 0] - c[0])
            if start <= end:
                start += 2.0 * math.pi                                       # OffsetSegmentGenerator::addCornerFillet, CLOCKWISE
            total = abs(start - end)
            nseg = int(total / quantum + 0.5)
            if nseg >= 1:
                inc = total / nseg
                for s in range(1, nseg):                                     # s = 0 is b[i] itself, s = nseg is a[j]
                    ang = start - s * inc
                    out.append(np.array([c[0] + d * math.cos(ang), c[1] + d * math.sin(ang)]))
    q = np.array(out)
    keep = np.any(q != np.roll(q, 1, axis=0), axis=1)                         # b[i] == a[j] on straight runs
    return q[keep]


def _winding(px, py, s0, s1):
    """winding number of the closed curve with segments s0 -> s1 around each point (px, py): signed crossings of the ray towards +x"""
    x0, y0, x1, y1 = s0[:, 0][None], s0[:, 1][None], s1[:, 0][None], s1[:, 1][None]
    px, py = px[:, None], py[:, None]
    up = (y0 <= py) & (y1 > py)
    dn = (y0 > py) & (y1 <= py)
    side = (x1 - x0) * (py - y0) - (px - x0) * (y1 - y0)                      # > 0: the point is left of the segment
    return np.sum(up & (side > 0), axis=1).astype(np.int64) - np.sum(dn & (side < 0), axis=1).astype(np.int64)


def _node(q, decimals):
    """every segment of the closed curve q cut at every crossing with a non-adjacent segment -> (starts, ends) of the sub-segments, in curve order.
    A crossing is computed once for both segments and, like the curve's own vertices, rounded to `decimals` places: a crossing that falls on a
    vertex (or two crossings that are one point) become the SAME coordinates, so that no piece of length 1e-15 is left for the side probes and the
    chaining below can match ends exactly"""
    m = len(q)
    s0, s1 = q, np.roll(q, -1, axis=0)
    dx, dy = (s1 - s0)[:, 0], (s1 - s0)[:, 1]
    idx = np.arange(m)
    cuts = [[] for _ in range(m)]                                             # per segment: (parameter, point)
    for r0 in range(0, m, 512):                                               # rows of the pair matrix in blocks: m reaches a few thousand on a finely digitised outline
        r = idx[r0:r0 + 512]
        # pairwise: s0[i] + t d[i] = s0[j] + u d[j]
        den = dx[r, None] * dy[None, :] - dy[r, None] * dx[None, :]
        wx, wy = s0[None, :, 0] - s0[r, None, 0], s0[None, :, 1] - s0[r, None, 1]
        with np.errstate(divide='ignore', invalid='ignore'):
            t = (wx * dy[None, :] - wy * dx[None, :]) / den
            u = (wx * dy[r, None] - wy * dx[r, None]) / den
        gap = np.abs(r[:, None] - idx[None, :])
        hit = (den != 0) & (gap > 1) & (gap != m - 1) & (t >= 0) & (t <= 1) & (u >= 0) & (u <= 1) & (r[:, None] < idx[None, :])
        for ii, j in zip(*np.nonzero(hit)):
            i = r0 + ii
            pt = np.round(s0[i] + t[ii, j] * (s1[i] - s0[i]), decimals)
            cuts[i].append((float(t[ii, j]), pt))
            cuts[j].append((float(u[ii, j]), pt))
    starts, ends = [], []
    for i in range(m):
        pts = [s0[i]] + [c[1] for c in sorted(cuts[i], key=lambda c: c[0])] + [s1[i]]
        for k in range(len(pts) - 1):
            if pts[k][0] != pts[k + 1][0] or pts[k][1] != pts[k + 1][1]:
                starts.append(pts[k])
                ends.append(pts[k + 1])
    return np.array(starts).reshape(-1, 2), np.array(ends).reshape(-1, 2)


def _buffer_rings(p, d, quad_segs, decimals):
    """steps 2 and 3 of the header for the counter-clockwise ring p; None: a chain of boundary pieces did not close at this rounding"""
    q = np.round(raw_offset_curve(p, d, quad_segs), decimals)
    q = q[np.any(q != np.roll(q, 1, axis=0), axis=1)]
    if len(q) < 3:
        return []
    qs0, qs1 = q, np.roll(q, -1, axis=0)
    a, b = _node(q, decimals)
    if len(a) == 0:
        return []
    mid = 0.5 * (a + b)
    dirv = b - a
    ln = np.hypot(dirv[:, 0], dirv[:, 1])
    nl = np.stack([-dirv[:, 1], dirv[:, 0]], axis=1) / ln[:, None]
    # a probe on either side of every piece, close to it compared with anything the construction makes (chords are ~0.1 d long, crossings of
    # chords leave pieces down to ~1e-3 d) and far compared with the rounding: 1e-8 of the ring's size, never more than a thousandth of the piece
    scale = float(np.max(np.abs(p))) + 1.0
    eps = np.minimum(1e-8 * scale, 1e-3 * ln)
    wl = _winding(mid[:, 0] + eps * nl[:, 0], mid[:, 1] + eps * nl[:, 1], qs0, qs1)
    wr = _winding(mid[:, 0] - eps * nl[:, 0], mid[:, 1] - eps * nl[:, 1], qs0, qs1)
    fwd = (wl >= 1) & (wr < 1)
    bwd = (wr >= 1) & (wl < 1)
    es = np.concatenate([a[fwd], b[bwd]])                                     # boundary pieces, the region on their left
    ee = np.concatenate([b[fwd], a[bwd]])
    if len(es) == 0:
        return []
    # chain: at a vertex with several continuations (rings touching in a point) take the one that turns left most, keeping the region on the left
    by_start = {}
    for k in range(len(es)):
        by_start.setdefault((es[k, 0], es[k, 1]), []).append(k)
    used = np.zeros(len(es), bool)
    rings = []
    for k0 in range(len(es)):
        if used[k0]:
            continue
        ring_pts, k = [es[k0]], k0
        while True:
            used[k] = True
            ring_pts.append(ee[k])
            if ee[k, 0] == es[k0, 0] and ee[k, 1] == es[k0, 1]:
                break
            cand = [c for c in by_start.get((ee[k, 0], ee[k, 1]), []) if not used[c]]
            if not cand:
                return None                                                  # an open chain: two crossings that should be one point; the caller rounds coarser
            if len(cand) > 1:
                din = ee[k] - es[k]
                ang_in = math.atan2(din[1], din[0])

                def turn(c):
                    dout = ee[c] - es[c]
                    return (math.atan2(dout[1], dout[0]) - ang_in + math.pi) % (2.0 * math.pi)
                cand.sort(key=turn, reverse=True)
            k = cand[0]
        if len(ring_pts) >= 4:
            r = np.array(ring_pts)
            x, y = r[:-1, 0], r[:-1, 1]
            if float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)) > 0:   # exteriors only (holes are never read by the reference)
                rings.append(r)

    return rings


def buffer_inward(ring, d, quad_segs=QUAD_SEGS):
    """The exterior rings of Polygon(ring).buffer(-d): a list of (k, 2) float64 arrays, each CLOSED (first vertex repeated at the end) and
    counter-clockwise, sorted by their lowest-then-leftmost vertex; [] = the empty geometry.  d <= 0 returns the ring itself."""
    p = _clean_ring(ring)
    if p is None:
        return []
    if d <= 0:
        return [np.vstack([p, p[:1]])]
    # coordinates on a 1e-9 grid (then 1e-7, 1e-5 if a chain does not close: two crossings that should be one point may straddle a rounding step)
    for decimals in (9, 7, 5):
        rings = _buffer_rings(p, float(d), quad_segs, decimals)
        if rings is not None:
            break
    if rings is None:
        raise ArithmeticError("polygon_buffer: the offset curve's pieces do not close into rings (self-intersecting input ring?)")

    def key(r):
        i = np.lexsort((r[:, 0], r[:, 1]))[0]
        return (r[i, 1], r[i, 0])
    rings.sort(key=key)
    return rings


def shrink_instance_masks(seg_masks, shrink_percentage, quad_segs=QUAD_SEGS):
    """shared_utils.py:310-330, statement for statement: every part of an instance's polygon list is replaced by the int()-truncated exterior(s) of
    its negative buffer; a part whose buffer is a single EMPTY polygon makes the function return the ORIGINAL list (whatever was collected before
    is dropped, like the reference's early `return seg_masks`); the parts of a MultiPolygon are appended one by one."""
    seg_list = []
    for seg in seg_masks:
        u, v = list(seg[::2]), list(seg[1::2])
        k = min(len(u), len(v))
        try:
            rings = buffer_inward(np.array([u[:k], v[:k]], np.float64).T, shrink_distance(u[:k], v[:k], shrink_percentage) if k else 0.0, quad_segs)
        except ArithmeticError:
            # a ring that crosses itself has no interior to shrink (shapely raises on such input or returns whatever GEOS's noder makes of it): the
            # instance keeps its polygons, like one whose part shrinks to nothing
            warnings.warn("polygon_buffer: a segmentation ring crosses itself; the instance's mask is left unshrunken", RuntimeWarning, stacklevel=2)
            return seg_masks
        if not rings:
            return seg_masks
        for r in rings:
            seg_list.append([int(val) for pair in zip(r[:, 0].tolist(), r[:, 1].tolist()) for val in pair])
    return seg_list
