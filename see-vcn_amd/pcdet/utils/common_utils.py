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


def batch_counts(batch_indices, batch_size):
    """rows per scene (batch_size,) int32 from the scene-index column -- torch.bincount(..., minlength=B) without its device -> host read
    (bincount sizes its output from the largest index)."""
    b = torch.arange(batch_size, device=batch_indices.device, dtype=batch_indices.dtype).view(1, -1)
    return (batch_indices.view(-1, 1) == b).sum(dim=0, dtype=torch.int32)


_const_cache = {}


def const_tensor(values, device, dtype=torch.float32):
    """A small constant as a device tensor, made once per (values, device, dtype): torch.tensor(list, device=cuda) is a blocking host -> device
    copy that waits for everything queued on the stream, and the reference's helpers make such constants on every call."""
    import numpy as np
    arr = np.asarray(values, dtype=np.float64)
    key = (arr.tobytes(), arr.shape, str(device), dtype)
    t = _const_cache.get(key)
    if t is None:
        t = _const_cache[key] = torch.tensor(arr, dtype=dtype, device=device)
    return t


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N,3) [z,y,x] -> centres (N,3) [x,y,z] (common_utils.py:144-161)."""
    assert voxel_coords.shape[1] == 3
    centers = voxel_coords.flip(1).float()                              # [z,y,x] -> [x,y,z]; indexing with a Python list copies an index tensor host -> device
    vs = const_tensor(voxel_size, centers.device) * downsample_times
    pc = const_tensor(list(point_cloud_range[0:3]), centers.device)
    return (centers + 0.5) * vs + pc


def init_dist_pytorch(tcp_port, local_rank, backend='nccl'):
    """One process per GPU under `python -m torch.distributed.launch` / torchrun (common_utils.py:164-182, called by tools/train.py:70-76):
    binds the process to its device and joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  backend 'nccl'
    is RCCL on ROCm (gradients over xGMI); 'gloo' for CPU tests and for several ranks sharing one device.  -> (GPUs on this node, rank).
    Like the reference, tcp_port is accepted and unused when the launcher already exported MASTER_PORT; without one it is used."""
    import os
    import torch.distributed as dist
    import torch.multiprocessing as mp
    # Environment FIRST, before anything below touches the GPU: ROCr reads HSA_ENABLE_IPC_MODE_LEGACY once, at hsa_init (the first HIP call of
    # the process -- torch.cuda.set_device below).  This pool's driver only supports dmabuf IPC and RCCL needs it.  If the caller has already
    # initialised the GPU the setting comes too late for this process: the LAUNCHER must export it then (bench.py launch_ranks does); never
    # re-exec a GPU-initialised process to apply it.  The nccl path of this helper is exercised on a node only (tests use gloo): unverified here.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(tcp_port))
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method('spawn')
    num_gpus = torch.cuda.device_count()
    if num_gpus > 0:
        torch.cuda.set_device(local_rank % num_gpus)
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


# ---- loss scalars for tb_dict without one device -> host read each --------------------------------------------------------------------
# The reference fills tb_dict with `loss.item()` at every term (anchor_head_template.py:87-143, roi_head_template.py:114-150, ...): each
# .item() waits for the whole stream.  Inside `deferred_tb()` (the detectors' get_training_loss) tb_value() keeps the 0-dim tensor and
# materialize_tb() turns all of them into Python floats with ONE read at the end; called outside, tb_value() is `.item()` as before.
_tb_deferred = [0]


class deferred_tb:
    def __enter__(self):
        _tb_deferred[0] += 1
        return self

    def __exit__(self, *exc):
        _tb_deferred[0] -= 1
        return False


def tb_value(t):
    if _tb_deferred[0] and torch.is_tensor(t):
        return t.detach()
    return t.item() if torch.is_tensor(t) else t


def materialize_tb(*dicts):
    keys = [(d, k) for d in dicts for k, v in d.items() if torch.is_tensor(v)]
    if keys:
        vals = torch.stack([d[k].reshape(()).float() for d, k in keys]).tolist()
        for (d, k), v in zip(keys, vals):
            d[k] = v
    return dicts[0] if len(dicts) == 1 else dicts
