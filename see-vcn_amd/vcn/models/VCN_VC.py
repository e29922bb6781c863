import torch
import torch.nn as nn

from ... import _lib
from ..extensions.chamfer_dist import ChamferDistanceL2
from ..utils import misc
from ..utils.bbox_utils import get_bbox_from_keypoints
from ..utils.losses import geodesic_distance
from ..utils.sampling import get_partial_mesh_batch_device
from ..utils.transform import rot_from_heading, rotate_points_along_z
from . import layers as L
from .build import MODELS


import os

TRAIN_ON_TORCH = os.environ.get("SEEVCN_VCN_TRAIN_TORCH", "0") == "1"   # 1: the training-mode forward on torch modules (cuBLAS-role library GEMMs) instead of dense_ops
LAZY_ROWS = os.environ.get("SEEVCN_VCN_LAZY_ROWS", "1") != "0"     # 0: the distinct-row count is read on the host and the layers get exact sizes (A/B, tests)


def normalize_vector(v):
    v_mag = torch.sqrt(v.pow(2).sum(1)).clamp_min(1e-8)          # VCN_VC.py:12-21 (max with 1e-8)
    return v / v_mag.view(-1, 1)


def cross_product(u, v):
    return torch.stack((u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2], u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]), 1)


def compute_rotation_matrix_from_ortho6d(ortho6d):
    """Gram-Schmidt 6-D -> rotation matrix with columns x, y, z (VCN_VC.py:37-49)."""
    x = normalize_vector(ortho6d[:, 0:3])
    z = normalize_vector(cross_product(x, ortho6d[:, 3:6]))
    y = cross_product(z, x)
    return torch.cat((x.view(-1, 3, 1), y.view(-1, 3, 1), z.view(-1, 3, 1)), 2)


@MODELS.register_module()
class VCN_VC(nn.Module):
    """Drop-in for the reference VCN_VC (see/surface_completion/models/vcn/models/VCN_VC.py:109-214).

    Same constructor (`VCN_VC(config)`), same state_dict keys (including the unused `final_conv`, :133-141),
    same forward contract: in_dict['input'] (B,n,3) -> {'coarse' (B,1024,3), 'reg_rot' (B,3,3), 'reg_centre' (B,3)}.
    eval(): the forward runs entirely on libseevcn_hip.so (fp32 MFMA GEMMs with fused bias/BN/activation/max-pool epilogues,
    BatchNorm folded).  train(): the same layer stack runs as plain torch ops with autograd (batch-statistics BatchNorm), and
    get_loss (:150-176) uses the HIP Chamfer distance, farthest point sampling and surface selection.
    """

    def __init__(self, config):
        super().__init__()
        self.sel_k = 30
        self.number_coarse = 1024
        self.pose_encoder = nn.Sequential(
            nn.Conv1d(3, 64, 1), nn.LeakyReLU(), nn.Conv1d(64, 128, 1), nn.LeakyReLU(), nn.Conv1d(128, 1024, 1),
            nn.AdaptiveMaxPool1d(output_size=1))
        self.pose_fc = nn.Sequential(nn.Linear(1024, 512), nn.LeakyReLU(), nn.Linear(512, 9))
        self.encoder = L.FeatureEncoder([3, 128, 256, 512, 512, self.number_coarse])
        self.shape_fc = L.fc_layers([1024, 1024, 1024, 3 * self.number_coarse], last_as_linear=True)
        self.final_conv = nn.Sequential(
            nn.Conv1d(1024 + 3 + 2, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 512, 1),
            nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 3, 1))
        self._prepared = L.PreparedCache(self, self._prepare)
        self.dedup_points = True     # eval: run the per-point layers on each object's distinct rows only (bit-identical output)
        self.build_loss_func()

    def _prepare(self):
        pe = self.pose_encoder
        return dict(pose=[L.conv_wb(pe[0]), L.conv_wb(pe[2]), L.conv_wb(pe[4])], pose_fc=L.prepare_fc(self.pose_fc),
                    enc=L.prepare_encoder(self.encoder), shape_fc=L.prepare_fc(self.shape_fc))

    def build_loss_func(self):
        self.loss_coarse = ChamferDistanceL2()
        self.loss_partial = ChamferDistanceL2()
        self.loss_translation = nn.SmoothL1Loss(reduction='none')
        self.loss_dims = nn.SmoothL1Loss(reduction='none')

    def get_loss(self, ret_dict, in_dict):
        """dims / translation / rotation / coarse (Chamfer vs the FPS-downsampled complete cloud) / partial (Chamfer between the
        surfaces selected around the input from prediction and ground truth) — reference VCN_VC.py:150-176.  The reference
        feeds the numpy output of get_partial_mesh_batch to the Chamfer module; here both surfaces stay on the GPU (values only,
        no gradient: the index selection is not differentiable there either)."""
        gt_boxes = in_dict['gt_boxes']
        dev = ret_dict['coarse'].device
        loss_dict = {}
        pred_box = get_bbox_from_keypoints(ret_dict['coarse'], gt_boxes)
        loss_dict['dims'] = self.loss_dims(gt_boxes[:, 3:6].to(dev), pred_box[:, 3:6]).mean()
        loss_dict['translation'] = self.loss_translation(gt_boxes[:, :3].to(dev), ret_dict['reg_centre']).mean()
        loss_dict['rotation'] = geodesic_distance(ret_dict['reg_rot'], rot_from_heading(gt_boxes[:, -1]).to(dev)).mean()
        if in_dict['training']:
            ds_complete = misc.fps(in_dict['complete'], ret_dict['coarse'].shape[1])
            loss_dict['coarse'] = self.loss_coarse(ret_dict['coarse'], ds_complete)
            pred_surface, _ = get_partial_mesh_batch_device(in_dict['input'], ret_dict['coarse'], k=self.sel_k)
            gt_surface, _ = get_partial_mesh_batch_device(in_dict['input'], ds_complete, k=self.sel_k)
            loss_dict['partial'] = self.loss_partial(pred_surface, gt_surface)
        return loss_dict

    def _forward_train(self, in_dict):
        """The reference forward (VCN_VC.py:178-214) in training mode, differentiable, on the library's own kernels: every Conv1d(k=1) / Linear is
        seevcn_amd.dense_ops.linear on channel-last rows (forward GEMM on the matrix cores, hand-written data / weight / bias gradients), the
        training-mode BatchNorm1d + ReLU pairs are the fused kernels of the sparse backbone (spconv.norm.batch_norm_relu: batch statistics, running
        statistics updated like torch's), the max-pools over the points are dense_ops.segment_max; only the 3x3 geometry of a few values per object
        (rotations, Gram-Schmidt) stays on torch ops.  SEEVCN_VCN_TRAIN_TORCH=1: the same graph on torch modules (A/B, tests)."""
        if TRAIN_ON_TORCH:
            return self._forward_train_torch(in_dict)
        from ... import dense_ops as D
        x = in_dict['input']
        _lib.require_cuda(x)
        x = x.float()
        bs, n, _ = x.shape
        frustum_angle = torch.atan2(x[:, :, 1].mean(dim=1), x[:, :, 0].mean(dim=1))
        pc_fview = rotate_points_along_z(x, -frustum_angle)
        pts_mean = pc_fview.mean(dim=1).unsqueeze(1)
        pe = self.pose_encoder

        def conv(m, rows, act=D.ACT_NONE, **kw):
            return D.linear(rows, m.weight.squeeze(-1), m.bias, act, L.LRELU_SLOPE, **kw)

        h = conv(pe[0], (pc_fview - pts_mean).reshape(bs * n, 3), D.ACT_LRELU)
        L._tap("pose.act0", h)
        h = conv(pe[2], h, D.ACT_LRELU)
        L._tap("pose.act1", h)
        z = conv(pe[4], h)
        pose_feat = D.segment_max(z, n)                                                  # AdaptiveMaxPool1d(1)
        L._tap("pose.max", z, pose_feat)
        rel_pose = L.run_fc_train(self.pose_fc, pose_feat, tap="pose_fc")
        centre = pts_mean + rel_pose[:, :3].unsqueeze(1)
        rot_mat = compute_rotation_matrix_from_ortho6d(rel_pose[:, 3:9])
        pc_cn = torch.matmul(pc_fview - centre, rot_mat.permute(0, 2, 1))
        g2 = L.encode_train(self.encoder, pc_cn.reshape(bs * n, 3), bs, n)                # (B, 1024)
        t = L.run_fc_train(self.shape_fc, g2, tap="shape_fc")
        coarse = t.reshape(-1, self.number_coarse, 3)
        coarse_vc = torch.matmul(coarse, rot_mat) + centre
        return {'coarse': rotate_points_along_z(coarse_vc.contiguous(), frustum_angle),
                'reg_rot': torch.matmul(rot_mat, rot_from_heading(frustum_angle)),
                'reg_centre': rotate_points_along_z(centre, frustum_angle).squeeze(1)}

    def _forward_train_torch(self, in_dict):
        """The reference forward line by line (VCN_VC.py:178-214) on torch modules, differentiable (the cross-check of _forward_train)."""
        x = in_dict['input']
        bs, n, _ = x.shape
        frustum_angle = torch.atan2(x[:, :, 1].mean(dim=1), x[:, :, 0].mean(dim=1))
        pc_fview = rotate_points_along_z(x, -frustum_angle)
        pts_mean = pc_fview.mean(dim=1).unsqueeze(1)
        pose_feat = self.pose_encoder((pc_fview - pts_mean).permute(0, 2, 1)).view(bs, -1)
        rel_pose = self.pose_fc(pose_feat)
        centre = pts_mean + rel_pose[:, :3].unsqueeze(1)
        rot_mat = compute_rotation_matrix_from_ortho6d(rel_pose[:, 3:9])
        pc_cn = torch.matmul(pc_fview - centre, rot_mat.permute(0, 2, 1))
        enc = self.encoder
        feature = enc.mlp_conv1(pc_cn.permute(0, 2, 1))
        feature_global = torch.max(feature, dim=2, keepdim=True)[0]
        feature = enc.mlp_conv2(torch.cat([feature_global.expand(-1, -1, n), feature], dim=1))
        feature_global = torch.max(feature, dim=2)[0]
        coarse = self.shape_fc(feature_global).reshape(-1, self.number_coarse, 3)
        coarse_vc = torch.matmul(coarse, rot_mat) + centre
        return {'coarse': rotate_points_along_z(coarse_vc.contiguous(), frustum_angle),
                'reg_rot': torch.matmul(rot_mat, rot_from_heading(frustum_angle)),
                'reg_centre': rotate_points_along_z(centre, frustum_angle).squeeze(1)}

    def train(self, mode=True):
        self._prepared.invalidate()          # see PreparedCache.invalidate
        return super().train(mode)

    def forward(self, in_dict):
        if self.training:
            return self._forward_train(in_dict)
        with torch.no_grad():
            return self._forward_eval(in_dict)

    def _forward_eval(self, in_dict):
        lib = _lib.load()
        x = in_dict['input']
        _lib.require_cuda(x)
        x = x.float().contiguous()
        bs, n, _ = x.shape
        dev = x.device
        p = self._prepared.get()
        st = _lib.stream()
        fview = torch.empty_like(x)
        centred = torch.empty_like(x)
        state = torch.zeros((bs, 32), dtype=torch.float32, device=dev)
        _lib.check(lib.sv_vcn_vc_prep(_lib.ptr(x), bs, n, _lib.ptr(fview), _lib.ptr(centred), _lib.ptr(state), st), "sv_vcn_vc_prep")
        # pose encoder: 3->64 LReLU, 64->128 LReLU, 128->1024, max over n   (VCN_VC.py:116-123,193)
        (w0, b0), (w1, b1), (w2, b2) = p["pose"]
        sel = rg = u_dev = None
        if self.dedup_points and n > 1:
            # the number of distinct rows stays on the device (no host read in the forward): the row-wise layers are launched for the capacity and
            # compute the first *u_dev rows (sv_gemm_bias_act_ragged_dev, sv_pointwise_conv3_gather)
            sel, rg, u_dev = L.distinct_rows(x, sync=False) if LAZY_ROWS else L.distinct_rows(x) + (None,)
        pts = centred.view(bs * n, 3)
        h = L.pointwise3(pts, w0, b0, L.ACT_LRELU, sel=sel, m_dev=u_dev, tag="pose_h0")
        h = L.gemm(h, w1, b1, L.ACT_LRELU, row_group=rg if u_dev is not None else None, m_dev=u_dev, tag="pose_h1")
        pool = L.NegInfPool(bs * (w2.shape[0] + p["enc"]["w1b"].shape[0] + p["enc"]["w2b"].shape[0]), dev)
        pose_feat = pool.take((bs, w2.shape[0]))
        L.gemm(h, w2, b2, L.ACT_NONE, rows_per_group=n, store=False, group_max=pose_feat, row_group=rg, m_dev=u_dev)
        rel_pose = L.run_fc(p["pose_fc"], pose_feat, L.ACT_LRELU)                     # (B, 9)   :194
        pc_cn = torch.empty_like(x)
        _lib.check(lib.sv_vcn_vc_pose(_lib.ptr(fview), bs, n, _lib.ptr(rel_pose), _lib.ptr(state), _lib.ptr(pc_cn), st), "sv_vcn_vc_pose")
        pts = pc_cn.view(bs * n, 3)
        feat = L.encode(p["enc"], pts, bs, n, row_group=rg, sel=sel, m_dev=u_dev, pool=pool)      # (B, 1024) :203
        coarse_cn = L.run_fc(p["shape_fc"], feat, L.ACT_RELU)                         # (B, 3072) :204
        nc = self.number_coarse
        coarse = torch.empty((bs, nc, 3), dtype=torch.float32, device=dev)
        reg_rot = torch.empty((bs, 3, 3), dtype=torch.float32, device=dev)
        reg_centre = torch.empty((bs, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.sv_vcn_vc_finish(_lib.ptr(coarse_cn), bs, nc, _lib.ptr(state), _lib.ptr(coarse), _lib.ptr(reg_rot),
                                        _lib.ptr(reg_centre), st), "sv_vcn_vc_finish")
        return {'coarse': coarse, 'reg_rot': reg_rot, 'reg_centre': reg_centre}
