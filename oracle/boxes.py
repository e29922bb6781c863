"""ctypes front-end of oracle/geometry.c (rotated BEV overlap / IoU, NMS, points-in-boxes).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_geometry.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
        _lib.orc_box_overlap.restype = ctypes.c_float
        _lib.orc_iou_bev.restype = ctypes.c_float
        _lib.orc_nms.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def boxes_overlap_bev(a, b, iou=False):
    a, pa = _f(a)
    b, pb = _f(b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().orc_boxes_overlap_bev(pa, len(a), pb, len(b), out.ctypes.data_as(ctypes.c_void_p), int(iou))
    return out


def boxes_iou_bev(a, b):
    return boxes_overlap_bev(a, b, iou=True)


def boxes_iou3d(a, b):
    """boxes_iou3d_gpu, ops/iou3d_nms/iou3d_nms_utils.py:48-81."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ov = boxes_overlap_bev(a, b)
    amax, amin = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    bmax, bmin = (b[:, 2] + b[:, 5] / 2)[None], (b[:, 2] - b[:, 5] / 2)[None]
    oh = np.clip(np.minimum(amax, bmax) - np.maximum(amin, bmin), 0, None)
    o3 = ov * oh
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None]
    return o3 / np.clip(va + vb - o3, 1e-6, None)


def nms(boxes_sorted, thresh, normal=False):
    b, pb = _f(boxes_sorted)
    keep = np.zeros(len(b), np.int64)
    n = lib().orc_nms(pb, len(b), ctypes.c_float(thresh), keep.ctypes.data_as(ctypes.c_void_p), int(normal))
    return keep[:n]


def points_in_boxes(points, boxes):
    p, pp = _f(points)
    b, pb = _f(boxes)
    out = np.zeros(p.shape[:2], np.int32)
    lib().orc_points_in_boxes(pb, pp, p.shape[0], b.shape[1], p.shape[1], out.ctypes.data_as(ctypes.c_void_p))
    return out
