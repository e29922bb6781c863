"""VCN inference wrapper with the reference's interface (see/surface_completion/models/VCN.py:14-103)."""
from pathlib import Path

import numpy as np
import torch

from .datasets.data_transforms import ResamplePoints
from .models import build_model_from_cfg
from .utils.sampling import get_largest_cluster_batch_device, get_partial_mesh_batch_device


def _get(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, 'get') else getattr(cfg, key, default)


class VCN:
    """cfg keys: MODEL, NORM_WITH_GT, SEL_K_NEAREST, CLUSTER_EPS, BATCH_SIZE_LIMIT, CKPT_PATH (VCN.py:27-32).
    `inference` resamples every object to `resample_num` points, pads the object count to a multiple of
    `batch_size_limit` with zero clouds, runs the HIP forward per chunk and returns numpy arrays.
    'surface' (k-NN selection of coarse points around the input, sampling.py:8-41) and 'clustered' (largest DBSCAN cluster,
    sampling.py:83-100) are computed on the GPU by sv_vcn_surface_select / sv_vcn_largest_cluster; the reference does both on
    the CPU (VCN.py:89-93).  `inference_device` returns the same dict as CUDA tensors without any host copy."""

    def __init__(self, cfg, gpu_id=0, state_dict=None):
        self.cfg = cfg
        self.device = f'cuda:{gpu_id}'
        torch.cuda.set_device(gpu_id)
        self.model_init(state_dict)

    def model_init(self, state_dict=None):
        self.norm_with_gt = _get(self.cfg, 'NORM_WITH_GT')
        self.surface_sel_k = _get(self.cfg, 'SEL_K_NEAREST')
        self.cluster_eps = _get(self.cfg, 'CLUSTER_EPS')
        self.batch_size_limit = _get(self.cfg, 'BATCH_SIZE_LIMIT', None)
        self.model = build_model_from_cfg({'NAME': _get(self.cfg, 'MODEL')})
        if state_dict is None:
            ckpt = _get(self.cfg, 'CKPT_PATH')
            assert Path(ckpt).exists(), f"No ckpt found at {ckpt}"
            state_dict = torch.load(ckpt, map_location=self.device)['base_model']
        self.model.load_state_dict({k.replace("module.", ""): v for k, v in state_dict.items()})
        self.model.to(self.device)
        self.model.eval()

    def inference(self, pts, gtboxes=None, batch_size_limit=None, resample_num=1024, k=30, eps=0.4):
        """numpy dict like the reference: input/surface/coarse float32, clustered float64 (open3d points are doubles)."""
        out = self.inference_device(pts, gtboxes, batch_size_limit, resample_num, k, eps)
        ret = {key: v.cpu().numpy() for key, v in out.items()}
        ret['clustered'] = ret['clustered'].astype(np.float64)
        return ret

    def inference_device(self, pts, gtboxes=None, batch_size_limit=None, resample_num=1024, k=30, eps=0.4):
        resample = ResamplePoints({'n_points': resample_num})
        if type(pts) == list:
            resampled = np.concatenate([resample(pc)[np.newaxis, ...] for pc in pts], axis=0)
        else:
            resampled = resample(pts)[np.newaxis, ...]
        num_objs = resampled.shape[0]
        gt = np.vstack(gtboxes)[:, :7] if self.norm_with_gt else None
        in_pc = torch.from_numpy(resampled).float().to(self.device)
        if batch_size_limit is not None and type(pts) == list:
            pad_num = int(batch_size_limit * np.ceil(num_objs / batch_size_limit) - num_objs)
            padded = torch.cat([in_pc, in_pc.new_zeros((pad_num, resample_num, 3))], dim=0)
            gt_t = None
            if self.norm_with_gt:
                gt_t = torch.from_numpy(np.concatenate([gt, np.zeros((pad_num, 7))], axis=0)).float().to(self.device)
            coarse = []
            for s in range(0, padded.shape[0], batch_size_limit):
                in_dict = {'input': padded[s:s + batch_size_limit].contiguous()}
                if self.norm_with_gt:
                    in_dict['gt_boxes'] = gt_t[s:s + batch_size_limit].contiguous()
                coarse.append(self.model(in_dict)['coarse'])
            output = torch.cat(coarse, dim=0)[:num_objs]
        else:
            in_dict = {'input': in_pc}
            if self.norm_with_gt:
                in_dict['gt_boxes'] = torch.from_numpy(gt).float().to(self.device)
            output = self.model(in_dict)['coarse']
        surface, n_sel = get_partial_mesh_batch_device(in_pc, output, k=k)
        clustered, _ = get_largest_cluster_batch_device(surface, eps=eps, min_points=2, total_pts=output.shape[1], period=n_sel)
        return {'input': in_pc, 'surface': surface, 'clustered': clustered, 'coarse': output.detach()}
