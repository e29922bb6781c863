"""Small helpers with the reference's names (detector3d/pcdet/utils/common_utils.py)."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def limit_period(val, offset=0.5, period=np.pi):
    """val - floor(val / period + offset) * period   (common_utils.py:22-25)"""
    val, is_numpy = check_numpy_to_torch(val)
    ans = val - torch.floor(val / period + offset) * period
    return ans.numpy() if is_numpy else ans


def cfg_get(cfg, key, default=None):
    """`.get` for EasyDict / dict / attribute-style configs."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def rotate_points_along_z(points, angle):
    """points (B,N,3+C), angle (B) -> rotated about z, angle increases x ==> y (common_utils.py:35-57)."""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N,3) [z,y,x] -> centres (N,3) [x,y,z] (common_utils.py:144-161)."""
    assert voxel_coords.shape[1] == 3
    centers = voxel_coords[:, [2, 1, 0]].float()
    vs = torch.tensor(voxel_size, device=centers.device).float() * downsample_times
    pc = torch.tensor(point_cloud_range[0:3], device=centers.device).float()
    return (centers + 0.5) * vs + pc


def init_dist_pytorch(tcp_port, local_rank, backend='nccl'):
    """One process per GPU under `python -m torch.distributed.launch` / torchrun (common_utils.py:164-182, called by tools/train.py:70-76):
    binds the process to its device and joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  backend 'nccl'
    is RCCL on ROCm (gradients over xGMI); 'gloo' for CPU tests and for several ranks sharing one device.  -> (GPUs on this node, rank).
    Like the reference, tcp_port is accepted and unused when the launcher already exported MASTER_PORT; without one it is used."""
    import os
    import torch.distributed as dist
    import torch.multiprocessing as mp
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method('spawn')
    num_gpus = torch.cuda.device_count()
    if num_gpus > 0:
        torch.cuda.set_device(local_rank % num_gpus)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(tcp_port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # this pool's driver only supports dmabuf IPC (RCCL needs it)
    dist.init_process_group(backend=backend)
    return num_gpus, dist.get_rank()


def get_dist_info(return_gpu_per_machine=False):
    """(rank, world_size[, GPUs per machine]) -- (0, 1) outside a process group (common_utils.py:185-205)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world_size = dist.get_rank(), dist.get_world_size()
    else:
        rank, world_size = 0, 1
    if return_gpu_per_machine:
        return rank, world_size, torch.cuda.device_count()
    return rank, world_size
