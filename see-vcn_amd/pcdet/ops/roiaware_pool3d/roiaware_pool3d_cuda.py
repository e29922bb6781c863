"""Entry points of the reference's compiled extension `roiaware_pool3d_cuda` (detector3d/pcdet/ops/roiaware_pool3d/src/
roiaware_pool3d.cpp:172-177) that the hot path uses, same names / argument order / caller-allocated outputs / return value 1:

  points_in_boxes_gpu(boxes (B,N,7) cuda, pts (B,npoints,3) cuda, box_idx_of_points (B,npoints) int32 cuda, pre-filled with -1) -> 1   :96-117
  points_in_boxes_cpu(boxes (N,7) cpu, pts (npoints,3) cpu, pts_indices (N,npoints) int32 cpu) -> 1                                       :147-168

`forward` / `backward` (RoI-aware feature pooling) serve the PartA2 head only, which no SEE-VCN configuration builds: they
raise instead of silently doing something else."""
import torch

from .... import _lib


def points_in_boxes_gpu(boxes_tensor, pts_tensor, box_idx_of_points_tensor):
    _lib.require_cuda(boxes_tensor, pts_tensor, box_idx_of_points_tensor)
    assert boxes_tensor.dtype == pts_tensor.dtype == torch.float32 and box_idx_of_points_tensor.dtype == torch.int32
    assert boxes_tensor.is_contiguous() and pts_tensor.is_contiguous() and box_idx_of_points_tensor.is_contiguous()
    batch, nbox, npts = boxes_tensor.shape[0], boxes_tensor.shape[1], pts_tensor.shape[1]
    assert boxes_tensor.shape[2] == 7 and pts_tensor.shape[2] == 3 and tuple(box_idx_of_points_tensor.shape) == (batch, npts)
    lib = _lib.load()
    _lib.check(lib.sv_points_in_boxes(_lib.ptr(boxes_tensor) if boxes_tensor.numel() else None, _lib.ptr(pts_tensor) if pts_tensor.numel() else None,
                                      batch, nbox, npts, _lib.ptr(box_idx_of_points_tensor) if box_idx_of_points_tensor.numel() else None, _lib.stream()),
               "sv_points_in_boxes")
    return 1


def points_in_boxes_cpu(boxes_tensor, pts_tensor, pts_indices_tensor, device="cuda"):
    assert not (boxes_tensor.is_cuda or pts_tensor.is_cuda or pts_indices_tensor.is_cuda), "CPU tensors expected (the test itself runs on the GPU)"
    assert pts_indices_tensor.dtype == torch.int32 and tuple(pts_indices_tensor.shape) == (boxes_tensor.shape[0], pts_tensor.shape[0])
    b, p = boxes_tensor.float().contiguous().to(device), pts_tensor.float().contiguous().to(device)
    out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32, device=p.device)
    lib = _lib.load()
    _lib.check(lib.sv_points_in_boxes_matrix(_lib.ptr(b) if b.numel() else None, _lib.ptr(p) if p.numel() else None, b.shape[0], p.shape[0],
                                             _lib.ptr(out) if out.numel() else None, _lib.stream()), "sv_points_in_boxes_matrix")
    pts_indices_tensor.copy_(out.cpu())
    return 1


def forward(*args, **kwargs):
    raise NotImplementedError("roiaware_pool3d_cuda.forward (PartA2 RoI-aware pooling) is outside the SEE-VCN hot path and not built")


def backward(*args, **kwargs):
    raise NotImplementedError("roiaware_pool3d_cuda.backward (PartA2 RoI-aware pooling) is outside the SEE-VCN hot path and not built")
