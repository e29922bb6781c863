import torch
import torch.nn as nn

from ..... import _lib
from ....ops.pointnet2.pointnet2_stack import pointnet2_modules as pointnet2_stack_modules
from ....ops.pointnet2.pointnet2_stack import pointnet2_utils as pointnet2_stack_utils
from ....utils import common_utils
from ....utils.common_utils import cfg_get


class _BevInterp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bev, keypoints, x0, y0, vx, vy, stride):
        lib = _lib.load()
        _lib.require_cuda(bev, keypoints)
        bev = bev.contiguous().float()
        kp = keypoints.contiguous().float()
        B, C, H, W = bev.shape
        out = torch.empty((kp.shape[0], C), dtype=torch.float32, device=bev.device)
        rc = lib.sv_bev_interpolate(_lib.ptr(kp), kp.shape[0], _lib.ptr(bev), B, C, H, W, x0, y0, vx, vy, float(stride), _lib.ptr(out), _lib.stream())
        _lib.check(rc, "sv_bev_interpolate")
        ctx.save_for_backward(kp)
        ctx.meta = (B, C, H, W, x0, y0, vx, vy, float(stride))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (kp,) = ctx.saved_tensors
        B, C, H, W, x0, y0, vx, vy, stride = ctx.meta
        g = grad_out.contiguous().float()
        gbev = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
        rc = lib.sv_bev_interpolate_grad(_lib.ptr(kp), kp.shape[0], _lib.ptr(g), B, C, H, W, x0, y0, vx, vy, stride, _lib.ptr(gbev), _lib.stream())
        _lib.check(rc, "sv_bev_interpolate_grad")
        return gbev, None, None, None, None, None, None


class VoxelSetAbstraction(nn.Module):
    """Drop-in for the reference VoxelSetAbstraction (backbones_3d/pfe/voxel_set_abstraction.py:122-411), FPS keypoints
    (POINT_SOURCE raw_points | voxel_centers, SAMPLE_METHOD FPS): all scenes are sampled in ONE stacked FPS launch, BEV
    features are interpolated in place from the NCHW map, every SA source runs the HIP ball query / grouping."""

    def __init__(self, model_cfg, voxel_size, point_cloud_range, num_bev_features=None, num_rawpoint_features=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        SA_cfg = cfg_get(model_cfg, 'SA_LAYER')
        self.SA_layers = nn.ModuleList()
        self.SA_layer_names = []
        self.downsample_times_map = {}
        c_in = 0
        self.sources = list(cfg_get(model_cfg, 'FEATURES_SOURCE'))
        for src in self.sources:
            if src in ['bev', 'raw_points']:
                continue
            sc = SA_cfg[src]
            self.downsample_times_map[src] = cfg_get(sc, 'DOWNSAMPLE_FACTOR')
            inp = cfg_get(sc, 'INPUT_CHANNELS', None)
            if inp is None:
                m0 = cfg_get(sc, 'MLPS')[0]
                inp = m0[0] if isinstance(m0, list) else m0
            layer, c_out = pointnet2_stack_modules.build_local_aggregation_module(input_channels=inp, config=sc)
            self.SA_layers.append(layer)
            self.SA_layer_names.append(src)
            c_in += c_out
        if 'bev' in self.sources:
            c_in += num_bev_features
        if 'raw_points' in self.sources:
            self.SA_rawpoints, c_out = pointnet2_stack_modules.build_local_aggregation_module(
                input_channels=num_rawpoint_features - 3, config=SA_cfg['raw_points'])
            c_in += c_out
        nout = cfg_get(model_cfg, 'NUM_OUTPUT_FEATURES')
        self.vsa_point_feature_fusion = nn.Sequential(nn.Linear(c_in, nout, bias=False), nn.BatchNorm1d(nout), nn.ReLU())
        self.num_point_features = nout
        self.num_point_features_before_fusion = c_in

    def interpolate_from_bev_features(self, keypoints, bev_features, batch_size, bev_stride):
        return _BevInterp.apply(bev_features, keypoints, self.point_cloud_range[0], self.point_cloud_range[1], self.voxel_size[0],
                                self.voxel_size[1], bev_stride)

    def get_sampled_points(self, batch_dict):
        """(B*M, 4) [bs_idx, x, y, z] keypoints by farthest point sampling of every scene (reference :227-281)."""
        batch_size = batch_dict['batch_size']
        src = cfg_get(self.model_cfg, 'POINT_SOURCE')
        if src == 'raw_points':
            src_points = batch_dict['points'][:, 1:4]
            batch_indices = batch_dict['points'][:, 0].long()
        elif src == 'voxel_centers':
            src_points = common_utils.get_voxel_centers(batch_dict['voxel_coords'][:, 1:4], downsample_times=1, voxel_size=self.voxel_size,
                                                        point_cloud_range=self.point_cloud_range)
            batch_indices = batch_dict['voxel_coords'][:, 0].long()
        else:
            raise NotImplementedError
        if cfg_get(self.model_cfg, 'SAMPLE_METHOD') != 'FPS':
            raise NotImplementedError("only FPS keypoint sampling is built (SPC is PV-RCNN++)")
        m = cfg_get(self.model_cfg, 'NUM_KEYPOINTS')
        cnt = torch.bincount(batch_indices, minlength=batch_size).int()
        # points are stacked scene by scene (collate_batch), so the stacked FPS can index them directly
        xyz = src_points.contiguous().float()
        idx = pointnet2_stack_utils.stack_farthest_point_sample(xyz, cnt, m)                       # (B, m) global rows
        cnt_l = cnt.tolist()
        if min(cnt_l) < m:  # fewer points than keypoints: repeat the valid picks (reference :258-261)
            starts = (torch.cumsum(cnt, 0) - cnt).tolist()
            for b, n in enumerate(cnt_l):
                if n < m:
                    valid = idx[b, :n]
                    idx[b] = valid.repeat(int(m / n) + 1)[:m]
        keypoints = xyz[idx.long().view(-1)]
        bcol = torch.arange(batch_size, device=xyz.device).view(-1, 1).repeat(1, m).view(-1, 1).float()
        return torch.cat((bcol, keypoints), dim=1)

    @staticmethod
    def aggregate_keypoint_features_from_one_source(batch_size, aggregate_func, xyz, xyz_features, xyz_bs_idxs, new_xyz, new_xyz_batch_cnt):
        xyz_batch_cnt = torch.bincount(xyz_bs_idxs.long(), minlength=batch_size).int()
        _, pooled = aggregate_func(xyz=xyz.contiguous(), xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz, new_xyz_batch_cnt=new_xyz_batch_cnt,
                                   features=xyz_features.contiguous() if xyz_features is not None else None)
        return pooled

    def forward(self, batch_dict):
        keypoints = self.get_sampled_points(batch_dict)
        batch_size = batch_dict['batch_size']
        feats = []
        if 'bev' in self.sources:
            feats.append(self.interpolate_from_bev_features(keypoints, batch_dict['spatial_features'], batch_size,
                                                            bev_stride=batch_dict['spatial_features_stride']))
        new_xyz = keypoints[:, 1:4].contiguous()
        new_xyz_batch_cnt = torch.bincount(keypoints[:, 0].long(), minlength=batch_size).int()
        if 'raw_points' in self.sources:
            raw = batch_dict['points']
            feats.append(self.aggregate_keypoint_features_from_one_source(
                batch_size, self.SA_rawpoints, raw[:, 1:4], raw[:, 4:].contiguous() if raw.shape[1] > 4 else None, raw[:, 0], new_xyz,
                new_xyz_batch_cnt))
        for k, src in enumerate(self.SA_layer_names):
            t = batch_dict['multi_scale_3d_features'][src]
            xyz = common_utils.get_voxel_centers(t.indices[:, 1:4], downsample_times=self.downsample_times_map[src], voxel_size=self.voxel_size,
                                                 point_cloud_range=self.point_cloud_range)
            feats.append(self.aggregate_keypoint_features_from_one_source(batch_size, self.SA_layers[k], xyz.contiguous(), t.features.contiguous(),
                                                                          t.indices[:, 0], new_xyz, new_xyz_batch_cnt))
        point_features = torch.cat(feats, dim=-1)
        batch_dict['point_features_before_fusion'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_features'] = self.vsa_point_feature_fusion(point_features.view(-1, point_features.shape[-1]))
        batch_dict['point_coords'] = keypoints
        return batch_dict
