"""Oracle: VCN_VC / VCN_CN forward as plain fp32 tensor algebra on the CPU (torch-CPU, no nn.Modules).
Test infrastructure only.  Takes a state_dict with the reference's key names."""
import torch
import torch.nn.functional as F


def _rotz(points, angle):
    """rotate_points_along_z, see/surface_completion/models/vcn/utils/transform.py:33-57 (p @ [[c,s,0],[-s,c,0],[0,0,1]])."""
    c, s = torch.cos(angle), torch.sin(angle)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    rot = torch.stack([c, s, z, -s, c, z, z, z, o], dim=1).view(-1, 3, 3)
    return points @ rot, rot


def _conv(x, sd, key):
    """Conv1d(k=1) on (B,C,n): weight (Cout,Cin,1)."""
    return torch.einsum('oc,bcn->bon', sd[key + '.weight'][:, :, 0], x) + sd[key + '.bias'][None, :, None]


def _bn(x, sd, key, eps=1e-5):
    """Eval-mode BatchNorm1d (default eps 1e-5, VCN_VC.py:88,94)."""
    g, b, m, v = (sd[f'{key}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var'))
    return (x - m[None, :, None]) / torch.sqrt(v[None, :, None] + eps) * g[None, :, None] + b[None, :, None]


def _lin(x, sd, key):
    return x @ sd[key + '.weight'].t() + sd[key + '.bias']


def feature_encoder(sd, x, prefix='encoder'):
    """FeatureEncoder.forward, VCN_VC.py:97-106. x: (B,3,n) -> (B,1024)."""
    n = x.shape[2]
    f = F.relu(_bn(_conv(x, sd, f'{prefix}.mlp_conv1.0'), sd, f'{prefix}.mlp_conv1.1'))
    f = _conv(f, sd, f'{prefix}.mlp_conv1.3')                         # B 256 n
    g = f.max(dim=2, keepdim=True)[0]                                 # :100
    f = torch.cat([g.expand(-1, -1, n), f], dim=1)                    # :101
    f = F.relu(_bn(_conv(f, sd, f'{prefix}.mlp_conv2.0'), sd, f'{prefix}.mlp_conv2.1'))
    f = _conv(f, sd, f'{prefix}.mlp_conv2.3')                         # B 1024 n
    return f.max(dim=2)[0]                                            # :104


def shape_fc(sd, feat, nc=1024):
    h = F.relu(_lin(feat, sd, 'shape_fc.0'))
    h = F.relu(_lin(h, sd, 'shape_fc.2'))
    return _lin(h, sd, 'shape_fc.4').reshape(-1, nc, 3)               # VCN_VC.py:204


def ortho6d_to_rot(r6):
    """compute_rotation_matrix_from_ortho6d, VCN_VC.py:36-49 (normalise with clamp 1e-8, :12-22)."""
    def nrm(v):
        return v / torch.clamp(v.pow(2).sum(1).sqrt(), min=1e-8)[:, None]
    x = nrm(r6[:, 0:3])
    z = nrm(torch.cross(x, r6[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


@torch.no_grad()
def vcn_vc_forward(sd, inp):
    """VCN_VC.forward, VCN_VC.py:178-214. inp (B,n,3) float32."""
    sd = {k: v.float() for k, v in sd.items()}
    bs, n, _ = inp.shape
    ang = torch.atan2(inp[:, :, 1].mean(dim=1), inp[:, :, 0].mean(dim=1))      # :185
    fview, _ = _rotz(inp, -ang)                                                 # :186
    mean = fview.mean(dim=1, keepdim=True)                                      # :189
    x = (fview - mean).permute(0, 2, 1)
    h = F.leaky_relu(_conv(x, sd, 'pose_encoder.0'))
    h = F.leaky_relu(_conv(h, sd, 'pose_encoder.2'))
    pose_feat = _conv(h, sd, 'pose_encoder.4').max(dim=2)[0]                    # :193
    rel = _lin(F.leaky_relu(_lin(pose_feat, sd, 'pose_fc.0')), sd, 'pose_fc.2')  # :194
    centre = mean + rel[:, :3].unsqueeze(1)                                     # :195-196
    rot = ortho6d_to_rot(rel[:, 3:9])                                           # :197-198
    pc_cn = (fview - centre) @ rot.permute(0, 2, 1)                             # :200
    feat = feature_encoder(sd, pc_cn.permute(0, 2, 1))                          # :203
    coarse = shape_fc(sd, feat) @ rot + centre                                  # :204-205
    out, rz = _rotz(coarse, ang)                                                # :208
    return {'coarse': out, 'reg_rot': rot @ rz, 'reg_centre': _rotz(centre, ang)[0].squeeze(1)}  # :211-212


@torch.no_grad()
def vcn_cn_forward(sd, inp, gt_boxes):
    """VCN_CN.forward, VCN_CN.py:142-156 with transform.py:91-160."""
    sd = {k: v.float() for k, v in sd.items()}
    centre = gt_boxes[:, :3].unsqueeze(1)
    pc = _rotz(inp - centre, -gt_boxes[:, 6])[0] / gt_boxes[:, 3].view(-1, 1, 1)
    feat = feature_encoder(sd, pc.permute(0, 2, 1))
    coarse = shape_fc(sd, feat) * gt_boxes[:, 3].view(-1, 1, 1)
    return {'coarse': _rotz(coarse, gt_boxes[:, 6])[0] + centre}
