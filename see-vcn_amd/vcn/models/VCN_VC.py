import torch
import torch.nn as nn

from ... import _lib
from . import layers as L
from .build import MODELS


@MODELS.register_module()
class VCN_VC(nn.Module):
    """Drop-in for the reference VCN_VC (see/surface_completion/models/vcn/models/VCN_VC.py:109-214).

    Same constructor (`VCN_VC(config)`), same state_dict keys (including the unused `final_conv`, :133-141),
    same forward contract: in_dict['input'] (B,n,3) -> {'coarse' (B,1024,3), 'reg_rot' (B,3,3), 'reg_centre' (B,3)}.
    The forward runs entirely on libseevcn_hip.so (fp32 MFMA GEMMs with fused bias/BN/activation/max-pool
    epilogues); inference only — the reference's training loss (get_loss, :150-176) is out of scope.
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

    def _prepare(self):
        pe = self.pose_encoder
        return dict(pose=[L.conv_wb(pe[0]), L.conv_wb(pe[2]), L.conv_wb(pe[4])], pose_fc=L.prepare_fc(self.pose_fc),
                    enc=L.prepare_encoder(self.encoder), shape_fc=L.prepare_fc(self.shape_fc))

    def get_loss(self, ret_dict, in_dict):
        raise NotImplementedError("VCN training loss (Chamfer + FPS, VCN_VC.py:150-176) is outside the built hot path")

    @torch.no_grad()
    def forward(self, in_dict):
        if self.training:
            raise RuntimeError("seevcn_amd VCN_VC implements the eval-mode forward (BatchNorm folded); call .eval()")
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
        h = L.pointwise3(centred.view(bs * n, 3), w0, b0, L.ACT_LRELU)
        h = L.gemm(h, w1, b1, L.ACT_LRELU)
        pose_feat = L.neg_inf((bs, w2.shape[0]), dev)
        L.gemm(h, w2, b2, L.ACT_NONE, rows_per_group=n, store=False, group_max=pose_feat)
        rel_pose = L.run_fc(p["pose_fc"], pose_feat, L.ACT_LRELU)                     # (B, 9)   :194
        pc_cn = torch.empty_like(x)
        _lib.check(lib.sv_vcn_vc_pose(_lib.ptr(fview), bs, n, _lib.ptr(rel_pose), _lib.ptr(state), _lib.ptr(pc_cn), st), "sv_vcn_vc_pose")
        feat = L.encode(p["enc"], pc_cn.view(bs * n, 3), bs, n)                       # (B, 1024) :203
        coarse_cn = L.run_fc(p["shape_fc"], feat, L.ACT_RELU)                         # (B, 3072) :204
        nc = self.number_coarse
        coarse = torch.empty((bs, nc, 3), dtype=torch.float32, device=dev)
        reg_rot = torch.empty((bs, 3, 3), dtype=torch.float32, device=dev)
        reg_centre = torch.empty((bs, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.sv_vcn_vc_finish(_lib.ptr(coarse_cn), bs, nc, _lib.ptr(state), _lib.ptr(coarse), _lib.ptr(reg_rot),
                                        _lib.ptr(reg_centre), st), "sv_vcn_vc_finish")
        return {'coarse': coarse, 'reg_rot': reg_rot, 'reg_centre': reg_centre}
