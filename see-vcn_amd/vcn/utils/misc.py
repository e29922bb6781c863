import torch

from ... import _lib


def fps(data, number):
    """data (B,N,3) -> (B,number,3): farthest point sampling from point 0 + gather (reference utils/misc.py:29-36, which calls the
    third-party pointnet2_ops; same algorithm as the in-tree pointnet2_batch sampling kernel) on sv_farthest_point_sampling."""
    lib = _lib.load()
    _lib.require_cuda(data)
    x = data.detach().float().contiguous()
    B, N, _ = x.shape
    idx = torch.empty((B, number), dtype=torch.int32, device=x.device)
    from ...pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda import fps_bucketed
    if not fps_bucketed(x, None, None, B, N, N, int(number), idx):
        temp = torch.empty((B, N), dtype=torch.float32, device=x.device)
        _lib.check(lib.sv_farthest_point_sampling(_lib.ptr(x), B, N, int(number), _lib.ptr(temp), _lib.ptr(idx), _lib.stream()), "sv_farthest_point_sampling")
    return torch.gather(data, 1, idx.long().unsqueeze(-1).expand(-1, -1, data.shape[2])).contiguous()
