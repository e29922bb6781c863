"""Entry points of the reference's compiled extension `iou3d_nms_cuda` (detector3d/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:12-17) with
the same names, argument order, caller-allocated outputs and return values, bound to libseevcn_hip.so -- the module the reference's
unchanged iou3d_nms_utils.py imports (`from . import iou3d_nms_cuda`, iou3d_nms_utils.py:9).

  boxes_overlap_bev_gpu(boxes_a (N,7) cuda, boxes_b (M,7) cuda, ans_overlap (N,M) cuda) -> 1      iou3d_nms.cpp:49-67
  boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou) -> 1                                                iou3d_nms.cpp:70-87
  nms_gpu(boxes (N,7) cuda, keep (N) CPU int64, thresh) -> num_out                                 iou3d_nms.cpp:90-136
  nms_normal_gpu(boxes, keep, thresh) -> num_out                                                   iou3d_nms.cpp:139-186
  boxes_iou_bev_cpu(boxes_a (N,7) cpu, boxes_b (M,7) cpu, ans_iou (N,M) cpu) -> 1                  iou3d_cpu.cpp:236-end

The reference copies the suppression masks to the host and sweeps them there (iou3d_nms.cpp:111-131); here the sweep runs on the
GPU and only the kept indices come back.  boxes_iou_bev_cpu keeps its CPU-tensor interface, the pairs are evaluated on the GPU
(this library has no CPU code path)."""
import torch

from .... import _lib


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _overlap(boxes_a, boxes_b, out, iou):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (out, "ans")):
        _check_input(t, n)
    assert boxes_a.dtype == boxes_b.dtype == out.dtype == torch.float32 and boxes_a.shape[1] == boxes_b.shape[1] == 7
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    assert tuple(out.shape) == (na, nb)
    lib = _lib.load()
    _lib.check(lib.sv_boxes_overlap_bev(_lib.ptr(boxes_a) if na else None, na, _lib.ptr(boxes_b) if nb else None, nb,
                                        _lib.ptr(out) if out.numel() else None, int(iou), _lib.stream()), "sv_boxes_overlap_bev")
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    return _overlap(boxes_a, boxes_b, ans_overlap, False)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    return _overlap(boxes_a, boxes_b, ans_iou, True)


def _nms(boxes, keep, thresh, normal):
    _check_input(boxes, "boxes")
    if not keep.is_contiguous():
        raise RuntimeError("keep must be contiguous")
    assert boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.shape[1] == 7
    assert keep.dtype == torch.int64 and keep.numel() >= boxes.shape[0]
    lib = _lib.load()
    n, dev = boxes.shape[0], boxes.device
    keep_dev = keep if keep.is_cuda else torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    num_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    scratch = _lib.workspace.scratch("nms", lib.sv_nms_scratch_bytes(n), dev)
    _lib.check(lib.sv_nms(_lib.ptr(boxes) if n else None, n, float(thresh), int(normal), _lib.ptr(scratch), _lib.ptr(keep_dev), _lib.ptr(num_out),
                          _lib.stream()), "sv_nms")
    k = int(num_out.item())
    if not keep.is_cuda:
        keep[:k] = keep_dev[:k].cpu()
    return k


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, True)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou, device="cuda"):
    assert not (boxes_a.is_cuda or boxes_b.is_cuda or ans_iou.is_cuda), "Only support CPU tensors"
    a, b = boxes_a.float().contiguous().to(device), boxes_b.float().contiguous().to(device)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _overlap(a, b, out, True)
    ans_iou.copy_(out.cpu())
    return 1
