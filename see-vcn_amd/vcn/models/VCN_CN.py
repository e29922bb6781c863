import torch
import torch.nn as nn

from ... import _lib
from ..extensions.chamfer_dist import ChamferDistanceL2
from ..utils import misc
from ..utils.sampling import get_partial_mesh_batch_device
from ..utils.transform import cn_to_vc, normalize_scale, restore_scale, vc_to_cn
from . import layers as L
from .build import MODELS


@MODELS.register_module()
class VCN_CN(nn.Module):
    """Drop-in for the reference VCN_CN (see/surface_completion/models/vcn/models/VCN_CN.py:110-156):
    GT-box canonicalisation instead of the pose branch; in_dict needs 'input' (B,n,3) and 'gt_boxes' (B,7)."""

    def __init__(self, config):
        super().__init__()
        self.sel_k = 30
        self.number_coarse = 1024
        self.encoder = L.FeatureEncoder([3, 128, 256, 512, 512, self.number_coarse])
        self.shape_fc = L.fc_layers([1024, 1024, 1024, 3 * self.number_coarse], last_as_linear=True)
        self._prepared = L.PreparedCache(self, lambda: dict(enc=L.prepare_encoder(self.encoder), shape_fc=L.prepare_fc(self.shape_fc)))
        self.dedup_points = True     # run the per-point layers on each object's distinct rows only (bit-identical output)

        self.build_loss_func()

    def build_loss_func(self):
        self.loss_coarse = ChamferDistanceL2()
        self.loss_partial = ChamferDistanceL2()
        self.loss_translation = nn.SmoothL1Loss(reduction='none')
        self.loss_dims = nn.SmoothL1Loss(reduction='none')

    def get_loss(self, ret_dict, in_dict):
        """coarse (Chamfer vs the FPS-downsampled complete cloud) and partial (Chamfer between the surfaces selected around the
        input) -- reference VCN_CN.py:125-140; HIP Chamfer distance, farthest point sampling and surface selection."""
        loss_dict = {}
        if in_dict['training']:
            ds_complete = misc.fps(in_dict['complete'], ret_dict['coarse'].shape[1])
            loss_dict['coarse'] = self.loss_coarse(ret_dict['coarse'], ds_complete)
            pred_surface, _ = get_partial_mesh_batch_device(in_dict['input'], ret_dict['coarse'], k=self.sel_k)
            gt_surface, _ = get_partial_mesh_batch_device(in_dict['input'], ds_complete, k=self.sel_k)
            loss_dict['partial'] = self.loss_partial(pred_surface, gt_surface)
        return loss_dict

    def _forward_train(self, in_dict):
        """The reference forward (VCN_CN.py:142-156) in training mode, differentiable, on the library's own kernels (layers.encode_train /
        run_fc_train: fp32 MFMA GEMMs with hand-written backward, fused batch-statistics BatchNorm + ReLU, segment max); the box
        canonicalisation stays on torch ops.  SEEVCN_VCN_TRAIN_TORCH=1: the same graph on torch modules."""
        from .VCN_VC import TRAIN_ON_TORCH
        x, boxes = in_dict['input'], in_dict['gt_boxes']
        bs, n = x.shape[0], x.shape[1]
        pc = normalize_scale(vc_to_cn(x, boxes), boxes)
        if TRAIN_ON_TORCH:
            enc = self.encoder
            feature = enc.mlp_conv1(pc.permute(0, 2, 1))
            feature_global = torch.max(feature, dim=2, keepdim=True)[0]
            feature = enc.mlp_conv2(torch.cat([feature_global.expand(-1, -1, n), feature], dim=1))
            feature_global = torch.max(feature, dim=2)[0]
            coarse = self.shape_fc(feature_global).reshape(-1, self.number_coarse, 3)
        else:
            _lib.require_cuda(x)
            g2 = L.encode_train(self.encoder, pc.float().reshape(bs * n, 3), bs, n)
            coarse = L.run_fc_train(self.shape_fc, g2, tap="shape_fc").reshape(-1, self.number_coarse, 3)
        return {'coarse': cn_to_vc(restore_scale(coarse.contiguous(), boxes), boxes)}

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
        x, boxes = in_dict['input'], in_dict['gt_boxes']
        _lib.require_cuda(x, boxes)
        x = x.float().contiguous()
        boxes = boxes.float()[:, :7].contiguous()
        assert boxes.shape[1] == 7, f'gt_label wrong shape, should be (B 7) but given shape is {boxes.shape}'
        bs, n, _ = x.shape
        p = self._prepared.get()
        st = _lib.stream()
        pc = torch.empty_like(x)
        _lib.check(lib.sv_vcn_cn_transform(_lib.ptr(x), bs, n, _lib.ptr(boxes), 0, _lib.ptr(pc), st), "sv_vcn_cn_transform")
        pts, rg, sel, u_dev = pc.view(bs * n, 3), None, None, None
        if self.dedup_points and n > 1:
            sel, rg, u_dev = L.distinct_rows(x, sync=False)       # the number of distinct rows stays on the device (see VCN_VC._forward_eval)
        feat = L.encode(p["enc"], pts, bs, n, row_group=rg, sel=sel, m_dev=u_dev)
        coarse_cn = L.run_fc(p["shape_fc"], feat, L.ACT_RELU)
        nc = self.number_coarse
        coarse = torch.empty((bs, nc, 3), dtype=torch.float32, device=x.device)
        _lib.check(lib.sv_vcn_cn_transform(_lib.ptr(coarse_cn), bs, nc, _lib.ptr(boxes), 1, _lib.ptr(coarse), st), "sv_vcn_cn_transform")
        return {'coarse': coarse}
