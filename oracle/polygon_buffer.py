"""Oracle: shapely's `Polygon.buffer(-d)` as the reference's shrink_instance_masks uses it
(see/surface_completion/datasets/shared_utils.py:295-330).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  The buffer is GEOS's (shapely is a third-party dependency of the reference, not under /root/reference and not installed in this
image; the reference pins no version).  Nothing here can be checked against GEOS's own output; the anchors are hand-derived known answers
(tests/test_isolation.py: a rectangle, an L with its 16-chord arc, a sliver that vanishes, a dumb-bell that splits) and the region the result must
cover (every point at least d from the ring and inside it, up to the arcs' chord sag of 0.0012 d).

Restated from the published construction (GEOS operation/buffer), scalar code, one statement per step:
  offset_curve    OffsetSegmentGenerator for one ring side: every edge moved left by d; round join (fillet of n = int(angle / (pi/32) + 0.5)
                  equal chords, addDirectedFillet) where the ring turns right; where it turns left the crossing of the two offset edges
                  (addInsideTurn) or, if they do not cross, the plain end-to-start join
  crossings       all crossings of non-adjacent segments (the noder)
  winding         winding number of the raw curve around a point, here as a sum of signed angles (the product code counts ray crossings)
  buffer_inward   the boundary of {winding >= 1}: pieces with the region on the left only, chained into rings; exteriors (positive area) only --
                  BufferBuilder's depth labelling + PolygonBuilder, and the `.exterior` the reference reads
  shrink_instance_masks   shared_utils.py:310-330 statement for statement (int() truncation, MultiPolygon parts, the early return of the original
                  list when a part's buffer is one empty polygon)
Not restated (as in the product's header): BufferInputLineSimplifier, ring start / direction, repeated-point removal below 1e-6 d."""
import math

QUAD_SEGS = 16


def shrink_distance(xs, ys, percentage):
    """shared_utils.py:298-305"""
    cx, cy = 0.5 * min(xs) + 0.5 * max(xs), 0.5 * min(ys) + 0.5 * max(ys)
    return math.sqrt((cx - min(xs)) ** 2 + (cy - min(ys)) ** 2) * (percentage / 100)


def _area2(ring):
    return sum(ring[i][0] * ring[(i + 1) % len(ring)][1] - ring[(i + 1) % len(ring)][0] * ring[i][1] for i in range(len(ring)))


def _ccw_ring(points):
    ring = [(float(x), float(y)) for x, y in points]
    if len(ring) > 1 and ring[0] == ring[-1]:
        ring.pop()
    ring = [pt for i, pt in enumerate(ring) if pt != ring[i - 1]]
    if len(ring) < 3 or _area2(ring) == 0:
        return None
    return ring if _area2(ring) > 0 else ring[::-1]


def offset_curve(ring, d, quad_segs=QUAD_SEGS):
    n = len(ring)
    quantum = math.pi / 2 / quad_segs
    edges = []
    for i in range(n):
        (x0, y0), (x1, y1) = ring[i], ring[(i + 1) % n]
        ln = math.hypot(x1 - x0, y1 - y0)
        nx, ny = -(y1 - y0) / ln, (x1 - x0) / ln
        edges.append([(x0 + d * nx, y0 + d * ny), (x1 + d * nx, y1 + d * ny), (x1 - x0, y1 - y0)])
    joins = []
    for i in range(n):
        j = (i + 1) % n
        (ex, ey), (fx, fy) = edges[i][2], edges[j][2]
        cross = ex * fy - ey * fx
        pts = None
        if cross > 0:
            (ax, ay), (cx, cy) = edges[i][0], edges[j][0]
            t = ((cx - ax) * fy - (cy - ay) * fx) / cross
            u = ((cx - ax) * ey - (cy - ay) * ex) / cross
            if 0 <= t <= 1 and 0 <= u <= 1:
                pts = 'cross', (ax + t * ex, ay + t * ey)
        elif cross < 0 or ex * fx + ey * fy < 0:
            vx, vy = ring[j]
            start = math.atan2(edges[i][1][1] - vy, edges[i][1][0] - vx)
            end = math.atan2(edges[j][0][1] - vy, edges[j][0][0] - vx)
            if start <= end:
                start += 2 * math.pi
            total = abs(start - end)
            nseg = int(total / quantum + 0.5)
            arc = []
            if nseg >= 1:
                for s in range(1, nseg):
                    ang = start - s * (total / nseg)
                    arc.append((vx + d * math.cos(ang), vy + d * math.sin(ang)))
            pts = 'arc', arc
        joins.append(pts)
    curve = []
    for i in range(n):
        a, b = edges[i][0], edges[i][1]
        before, after = joins[i - 1], joins[i]
        if before and before[0] == 'cross':
            a = before[1]
        if after and after[0] == 'cross':
            b = after[1]
        curve += [a, b]
        if after and after[0] == 'arc':
            curve += after[1]
    return curve


def _round(pt, decimals):
    return (round(pt[0], decimals), round(pt[1], decimals))


def _pieces(curve, decimals):
    m = len(curve)
    cuts = [[] for _ in range(m)]
    for i in range(m):
        (ax, ay), (bx, by) = curve[i], curve[(i + 1) % m]
        for j in range(i + 2, m):
            if i == 0 and j == m - 1:
                continue
            (cx, cy), (dx, dy) = curve[j], curve[(j + 1) % m]
            den = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx)
            if den == 0:
                continue
            t = ((cx - ax) * (dy - cy) - (cy - ay) * (dx - cx)) / den
            u = ((cx - ax) * (by - ay) - (cy - ay) * (bx - ax)) / den
            if 0 <= t <= 1 and 0 <= u <= 1:
                pt = _round((ax + t * (bx - ax), ay + t * (by - ay)), decimals)
                cuts[i].append((t, pt))
                cuts[j].append((u, pt))
    out = []
    for i in range(m):
        pts = [curve[i]] + [c[1] for c in sorted(cuts[i])] + [curve[(i + 1) % m]]
        out += [(pts[k], pts[k + 1]) for k in range(len(pts) - 1) if pts[k] != pts[k + 1]]
    return out


def winding(pt, curve):
    """sum of the signed angles the curve's segments subtend at pt, / 2 pi, rounded"""
    total = 0.0
    m = len(curve)
    for i in range(m):
        ax, ay = curve[i][0] - pt[0], curve[i][1] - pt[1]
        bx, by = curve[(i + 1) % m][0] - pt[0], curve[(i + 1) % m][1] - pt[1]
        total += math.atan2(ax * by - ay * bx, ax * bx + ay * by)
    return int(round(total / (2 * math.pi)))


def _rings_at(ring, d, quad_segs, decimals):
    curve = [_round(pt, decimals) for pt in offset_curve(ring, d, quad_segs)]
    curve = [pt for i, pt in enumerate(curve) if pt != curve[i - 1]]
    if len(curve) < 3:
        return []
    scale = max(max(abs(x), abs(y)) for x, y in ring) + 1.0
    boundary = {}
    for a, b in _pieces(curve, decimals):
        ln = math.hypot(b[0] - a[0], b[1] - a[1])
        eps = min(1e-8 * scale, 1e-3 * ln)
        nx, ny = -(b[1] - a[1]) / ln, (b[0] - a[0]) / ln
        mx, my = 0.5 * (a[0] + b[0]), 0.5 * (a[1] + b[1])
        wl, wr = winding((mx + eps * nx, my + eps * ny), curve), winding((mx - eps * nx, my - eps * ny), curve)
        if wl >= 1 and wr < 1:
            boundary.setdefault(a, []).append(b)
        elif wr >= 1 and wl < 1:
            boundary.setdefault(b, []).append(a)
    rings = []
    while boundary:
        start = next(iter(boundary))
        ring_pts, cur, prev = [start], start, None
        while True:
            nxts = boundary.get(cur)
            if not nxts:
                return None
            if len(nxts) > 1 and prev is not None:                   # rings touching in a point: the sharpest left turn keeps the region on the left
                ang_in = math.atan2(cur[1] - prev[1], cur[0] - prev[0])
                nxts.sort(key=lambda q: (math.atan2(q[1] - cur[1], q[0] - cur[0]) - ang_in + math.pi) % (2 * math.pi), reverse=True)
            nxt = nxts.pop(0)
            if not nxts:
                del boundary[cur]
            ring_pts.append(nxt)
            prev, cur = cur, nxt
            if cur == start:
                break
        if len(ring_pts) >= 4 and _area2(ring_pts[:-1]) > 0:
            rings.append(ring_pts)
    rings.sort(key=lambda r: min((y, x) for x, y in r))
    return rings


def buffer_inward(points, d, quad_segs=QUAD_SEGS):
    """exterior rings (closed lists of (x, y), counter-clockwise) of Polygon(points).buffer(-d); [] = empty"""
    ring = _ccw_ring(points)
    if ring is None:
        return []
    if d <= 0:
        return [ring + ring[:1]]
    for decimals in (9, 7, 5):
        rings = _rings_at(ring, float(d), quad_segs, decimals)
        if rings is not None:
            return rings
    raise ArithmeticError("oracle.polygon_buffer: boundary pieces do not close into rings")


def shrink_instance_masks(seg_masks, shrink_percentage, quad_segs=QUAD_SEGS):
    seg_list = []
    for seg in seg_masks:
        u, v = seg[::2], seg[1::2]
        pts = [[x, y] for x, y in zip(u, v)]
        d = shrink_distance([p[0] for p in pts], [p[1] for p in pts], shrink_percentage) if pts else 0.0
        resized = buffer_inward(pts, d, quad_segs)
        if len(resized) > 1:                                          # a MultiPolygon
            for ring in resized:
                seg_list.append([int(val) for pair in ring for val in pair])
        else:
            if not resized:                                           # resized_poly.is_empty
                return seg_masks
            seg_list.append([int(val) for pair in resized[0] for val in pair])
    return seg_list
