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
        scratch = _lib.workspace.scratch("bev_interp_grad", lib.sv_bev_interpolate_grad_scratch_bytes(B, C, H, W), g.device)
        rc = lib.sv_bev_interpolate_grad(_lib.ptr(kp), kp.shape[0], _lib.ptr(g), B, C, H, W, x0, y0, vx, vy, stride, _lib.ptr(scratch), _lib.ptr(gbev),
                                         _lib.stream())
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

    def _fps_inputs(self, batch_dict):
        batch_size = batch_dict['batch_size']
        src = cfg_get(self.model_cfg, 'POINT_SOURCE')
        if src == 'raw_points':
            src_points = batch_dict['points'][:, 1:4]
            batch_indices = batch_dict['points'][:, 0].long()
            counts = batch_dict.get('points_per_scene')                  # host list from collate_batch (seevcn extension), or absent
        elif src == 'voxel_centers':
            src_points = common_utils.get_voxel_centers(batch_dict['voxel_coords'][:, 1:4], downsample_times=1, voxel_size=self.voxel_size,
                                                        point_cloud_range=self.point_cloud_range)
            batch_indices = batch_dict['voxel_coords'][:, 0].long()
            counts = None
        else:
            raise NotImplementedError
        if cfg_get(self.model_cfg, 'SAMPLE_METHOD') != 'FPS':
            raise NotImplementedError("only FPS keypoint sampling is built (SPC is PV-RCNN++)")
        cnt = common_utils.batch_counts(batch_indices, batch_size)
        return src_points.contiguous().float(), cnt, counts

    def prefetch_keypoints(self, batch_dict):
        """Start the farthest point sampling on a side stream (seevcn extension; Detector3DTemplate.run_modules calls it before the first
        module).  The keypoints depend on the raw points only (voxel_set_abstraction.py:227-281), the sampling is M dependent rounds on 16
        workgroups per scene (7.6 ms for 4 x 20 k points -> 4096): it runs beside the VFE / sparse backbone / BEV backbone instead of in front
        of the set abstraction.  Needs the per-scene point counts on the host (batch_dict['points_per_scene']): without them nothing is
        prefetched and get_sampled_points samples in line."""
        if cfg_get(self.model_cfg, 'POINT_SOURCE') != 'raw_points' or batch_dict.get('points_per_scene') is None or not batch_dict['points'].is_cuda:
            return
        xyz, cnt, counts = self._fps_inputs(batch_dict)
        main = torch.cuda.current_stream()
        if getattr(self, '_fps_stream', None) is None or self._fps_stream.device != xyz.device:
            self._fps_stream = torch.cuda.Stream(device=xyz.device, priority=-1)
        side = self._fps_stream
        side.wait_stream(main)                                           # the points were produced on the main stream
        with torch.cuda.stream(side):
            handle = pointnet2_stack_utils.pointnet2.stack_farthest_point_sampling_async(xyz, cnt, cfg_get(self.model_cfg, 'NUM_KEYPOINTS'), max(counts))
            done = side.record_event()
        for t in (xyz, cnt):
            t.record_stream(side)
        batch_dict['_fps_prefetch'] = (handle, done, xyz, cnt, counts)

    def get_sampled_points(self, batch_dict):
        """(B*M, 4) [bs_idx, x, y, z] keypoints by farthest point sampling of every scene (reference :227-281)."""
        batch_size = batch_dict['batch_size']
        m = cfg_get(self.model_cfg, 'NUM_KEYPOINTS')
        pre = batch_dict.pop('_fps_prefetch', None)
        if pre is not None:
            handle, done, xyz, cnt, counts = pre
            idx = handle.result()                                        # waits for the side stream only
            torch.cuda.current_stream().wait_event(done)
            idx.record_stream(torch.cuda.current_stream())
            cnt_l = list(counts)
        else:
            xyz, cnt, counts = self._fps_inputs(batch_dict)
            cnt_l = list(counts) if counts is not None else cnt.tolist()
            # points are stacked scene by scene (collate_batch), so the stacked FPS can index them directly
            idx = pointnet2_stack_utils.stack_farthest_point_sample(xyz, cnt, m, max(cnt_l))     # (B, m) global rows
        if min(cnt_l) < m:  # fewer points than keypoints: repeat the valid picks (reference :258-261)
            for b, n in enumerate(cnt_l):
                if n < m:
                    valid = idx[b, :n]
                    idx[b] = valid.repeat(int(m / n) + 1)[:m]
        keypoints = xyz[idx.long().view(-1)]
        bcol = torch.arange(batch_size, device=xyz.device).view(-1, 1).repeat(1, m).view(-1, 1).float()
        batch_dict['point_coords_per_scene'] = m                         # host-side fact for the heads: every scene has exactly m keypoints
        return torch.cat((bcol, keypoints), dim=1)

    @staticmethod
    def aggregate_keypoint_features_from_one_source(batch_size, aggregate_func, xyz, xyz_features, xyz_bs_idxs, new_xyz, new_xyz_batch_cnt):
        xyz_batch_cnt = common_utils.batch_counts(xyz_bs_idxs, batch_size)
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
        new_xyz_batch_cnt = torch.full((batch_size,), keypoints.shape[0] // batch_size, dtype=torch.int32, device=keypoints.device)   # NUM_KEYPOINTS each
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
        from .....dense_ops import run_sequential                       # Linear + BatchNorm1d + ReLU on the library's own GEMM / BatchNorm kernels
        batch_dict['point_features'] = run_sequential(self.vsa_point_feature_fusion, point_features.view(-1, point_features.shape[-1]))
        batch_dict['point_coords'] = keypoints
        return batch_dict
