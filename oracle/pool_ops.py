"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement (numpy float32 arithmetic, python loops: small cases only) of
  points_in_boxes_cpu  <- detector3d/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168 (MARGIN 1e-2, one 0/1 flag per (box, point))
PARITY UNPINNED at op level: the extension cannot be built here (CUDA headers) and the reference holds no test for it; restated line by line.
"""
import numpy as np

f32 = np.float32


def _local(pt, box, margin):
    x, y, z = f32(pt[0]), f32(pt[1]), f32(pt[2])
    cx, cy, cz, dx, dy, dz, rz = [f32(v) for v in box[:7]]
    if abs(f32(z - cz)) > float(dz) / 2.0:
        return False, f32(0), f32(0)
    cosa, sina = f32(np.cos(-rz)), f32(np.sin(-rz))
    sx, sy = f32(x - cx), f32(y - cy)
    lx = f32(f32(sx * cosa) + f32(sy * f32(-sina)))
    ly = f32(f32(sx * sina) + f32(sy * cosa))
    inside = (abs(float(lx)) < float(dx) / 2.0 + float(f32(margin))) and (abs(float(ly)) < float(dy) / 2.0 + float(f32(margin)))
    return inside, lx, ly


def points_in_boxes_cpu(points, boxes):
    out = np.zeros((len(boxes), len(points)), np.int32)
    for i, b in enumerate(boxes):
        for j, p in enumerate(points):
            out[i, j] = int(_local(p, b, 1e-2)[0])
    return out
