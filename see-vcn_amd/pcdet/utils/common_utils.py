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
