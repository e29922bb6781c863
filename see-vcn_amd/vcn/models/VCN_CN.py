import torch
import torch.nn as nn

from ... import _lib
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

    def get_loss(self, ret_dict, in_dict):
        raise NotImplementedError("VCN training loss (VCN_CN.py:125-140) is outside the built hot path")

    @torch.no_grad()
    def forward(self, in_dict):
        if self.training:
            raise RuntimeError("seevcn_amd VCN_CN implements the eval-mode forward (BatchNorm folded); call .eval()")
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
        pts, rg = pc.view(bs * n, 3), None
        if self.dedup_points and n > 1:
            sel, rg = L.distinct_rows(x)
            pts = pts[sel]
        feat = L.encode(p["enc"], pts, bs, n, row_group=rg)
        coarse_cn = L.run_fc(p["shape_fc"], feat, L.ACT_RELU)
        nc = self.number_coarse
        coarse = torch.empty((bs, nc, 3), dtype=torch.float32, device=x.device)
        _lib.check(lib.sv_vcn_cn_transform(_lib.ptr(coarse_cn), bs, nc, _lib.ptr(boxes), 1, _lib.ptr(coarse), st), "sv_vcn_cn_transform")
        return {'coarse': coarse}
