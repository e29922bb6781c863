"""Import harness for the *reference* Python (only usable where /root/reference exists).

Golden-vector generators (tests/golden/make_*.py) use this to import the reference's own
modules on CPU with stub modules standing in for third-party packages this image lacks
(SURVEY.md §8c lists them).  Nothing under tests/ that runs in pytest imports this file:
the GPU box has no /root/reference, so tests consume only the committed .npz fixtures.

The stubs carry no reference code: they are empty placeholders (or, for torch_scatter,
a functional restatement over torch index ops) so that `import` statements succeed.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SEEVCN_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _EasyDict(dict):
    """Minimal attribute dict (stands in for the `easydict` package)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(_EasyDict(x) if isinstance(x, dict) and not isinstance(x, _EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _scatter_mean(src, index, dim=0):
    n = int(index.max().item()) + 1 if index.numel() else 0
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    cnt = torch.bincount(index, minlength=n).clamp_min(1).to(src.dtype)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def _scatter_max(src, index, dim=0):
    n = int(index.max().item()) + 1 if index.numel() else 0
    out = src.new_full((n,) + tuple(src.shape[1:]), float("-inf"))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)
    return out, None


def install_stubs():
    """Register placeholder modules and make `.cuda()` an identity (reference calls it unconditionally)."""
    if getattr(install_stubs, "_done", False):
        return
    install_stubs._done = True
    _mod("easydict", EasyDict=_EasyDict)
    o3d = _mod("open3d")
    o3d.geometry = _mod("open3d.geometry")
    o3d.utility = _mod("open3d.utility")
    p2 = _mod("pointnet2_ops")
    p2.pointnet2_utils = _mod("pointnet2_ops.pointnet2_utils")
    _mod("chamfer")
    _mod("transforms3d")
    _mod("SharedArray")
    _mod("numba", jit=lambda *a, **k: (lambda f: f), njit=lambda *a, **k: (lambda f: f),
         cuda=types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f)))
    sk = _mod("skimage")
    sk.transform = _mod("skimage.transform")
    sk.io = _mod("skimage.io")
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.models._utils = _mod("torchvision.models._utils", IntermediateLayerGetter=object)
    tv.models.segmentation = _mod("torchvision.models.segmentation")
    tv.ops = _mod("torchvision.ops")
    tv.ops.boxes = _mod("torchvision.ops.boxes")
    tv.transforms = _mod("torchvision.transforms")
    k = _mod("kornia")
    k.utils = _mod("kornia.utils")
    k.geometry = _mod("kornia.geometry")
    k.geometry.conversions = _mod("kornia.geometry.conversions")
    k.losses = _mod("kornia.losses")
    k.losses.focal = _mod("kornia.losses.focal", FocalLoss=object)
    _mod("tensorboardX", SummaryWriter=object)
    _mod("cv2")
    _mod("torch_scatter", scatter_mean=_scatter_mean, scatter_max=_scatter_max)

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    sp = _mod("spconv")
    spt = _mod("spconv.pytorch", SparseConvTensor=_Dummy, SubMConv3d=_Dummy, SparseConv3d=_Dummy,
               SparseInverseConv3d=_Dummy, SparseSequential=_Dummy, SparseModule=_Dummy,
               SparseConvTensor_=_Dummy)
    sp.pytorch = spt
    spt.conv = _mod("spconv.pytorch.conv", SparseConvolution=_Dummy)
    sp.__version__ = "2.1.0"
    for name in ("SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d",
                 "SparseSequential", "SparseModule"):
        setattr(sp, name, _Dummy)
    sp.utils = _mod("spconv.utils")
    sp.pytorch.utils = _mod("spconv.pytorch.utils")
    _mod("cumm")
    _mod("cumm.tensorview")

    # reference code calls .cuda() unconditionally (e.g. VCN_VC.py:15, dynamic_mean_vfe.py:19)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def import_vcn():
    """Return the reference's vcn `models` package (VCN_VC / VCN_CN registered)."""
    install_stubs()
    root = os.path.join(REF, "see", "surface_completion")
    if root not in sys.path:
        sys.path.insert(0, root)
    import importlib
    return importlib.import_module("models.vcn.models")


def _install_oracle_ops():
    """The reference's compiled CUDA extensions cannot run here; the Python layers above them can.  These modules give those
    layers CPU implementations backed by the ORACLE (oracle/pointnet2.py, oracle/boxes.py) with the extension's own call
    signatures, so that goldens of the Python-level logic (VoxelSetAbstraction, PointHeadSimple, PVRCNNHead,
    ProposalTargetLayer, post-processing) can be produced by the reference's own classes.  The kernels themselves are pinned
    separately (tests/test_pointnet2.py, tests/test_boxes.py)."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import boxes as ob
    from oracle import pointnet2 as op2

    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        idx.copy_(torch.from_numpy(op2.ball_query(radius, nsample, xyz.numpy(), xyz_batch_cnt.numpy(), new_xyz.numpy(), new_xyz_batch_cnt.numpy())))
        return 1

    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        out.copy_(torch.from_numpy(op2.group_points(features.detach().numpy(), features_batch_cnt.numpy(), idx.numpy(), idx_batch_cnt.numpy())))
        return 1

    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        g = op2.group_points_grad(grad_out.numpy(), idx.numpy(), idx_batch_cnt.numpy(), features_batch_cnt.numpy(), N)
        grad_features.copy_(torch.from_numpy(g.astype(np.float32)))
        return 1

    def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
        for i in range(b):
            idx[i] = torch.from_numpy(op2.farthest_point_sampling(points[i].numpy(), m))
        return 1

    _mod("pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda", ball_query_wrapper=ball_query_wrapper,
         group_points_wrapper=group_points_wrapper, group_points_grad_wrapper=group_points_grad_wrapper,
         farthest_point_sampling_wrapper=farthest_point_sampling_wrapper)

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(ob.boxes_overlap_bev(a.numpy(), b.numpy())))
        return 1

    def boxes_iou_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(ob.boxes_iou_bev(a.numpy(), b.numpy())))
        return 1

    def nms_gpu(boxes, keep, thresh):
        k = ob.nms(boxes.numpy(), thresh)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    def nms_normal_gpu(boxes, keep, thresh):
        k = ob.nms(boxes.numpy(), thresh, normal=True)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    _mod("pcdet.ops.iou3d_nms.iou3d_nms_cuda", boxes_overlap_bev_gpu=boxes_overlap_bev_gpu, boxes_iou_bev_gpu=boxes_iou_bev_gpu, nms_gpu=nms_gpu,
         nms_normal_gpu=nms_normal_gpu)

    def points_in_boxes_gpu(boxes, pts, out):
        out.copy_(torch.from_numpy(ob.points_in_boxes(pts.numpy(), boxes.numpy())))
        return 1

    _mod("pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda", points_in_boxes_gpu=points_in_boxes_gpu)
    # the reference allocates with the legacy torch.cuda.*Tensor constructors
    torch.cuda.IntTensor = lambda *a: torch.IntTensor(*a)
    torch.cuda.FloatTensor = lambda *a: torch.FloatTensor(*a)
    torch.cuda.LongTensor = lambda *a: torch.LongTensor(*a)


def import_pcdet():
    install_stubs()
    root = os.path.join(REF, "detector3d")
    if root not in sys.path:
        sys.path.insert(0, root)
    for ext in ("pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda", "pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"):
        _mod(ext)
    _install_oracle_ops()
    import importlib
    return importlib.import_module("pcdet")


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from seeding import seeded_state_dict  # noqa: E402,F401  (shared with the tests)
