"""ResidualCoder with the reference's interface (detector3d/pcdet/utils/box_coder_utils.py:5-77)."""
import torch


class ResidualCoder(object):
    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        super().__init__()
        self.code_size = code_size
        self.encode_angle_by_sincos = encode_angle_by_sincos
        if self.encode_angle_by_sincos:
            self.code_size += 1

    def encode_torch(self, boxes, anchors):
        """boxes (N,7+C) ground truth, anchors (N,7+C) -> residual targets (N, code_size+C)"""
        a_dims = torch.clamp_min(anchors[:, 3:6], min=1e-5)
        g_dims = torch.clamp_min(boxes[:, 3:6], min=1e-5)
        diagonal = torch.sqrt(a_dims[:, 0:1] ** 2 + a_dims[:, 1:2] ** 2)
        xyz_t = torch.cat([(boxes[:, 0:1] - anchors[:, 0:1]) / diagonal, (boxes[:, 1:2] - anchors[:, 1:2]) / diagonal,
                           (boxes[:, 2:3] - anchors[:, 2:3]) / a_dims[:, 2:3]], dim=-1)
        dims_t = torch.log(g_dims / a_dims)
        rg, ra = boxes[:, 6:7], anchors[:, 6:7]
        rts = [torch.cos(rg) - torch.cos(ra), torch.sin(rg) - torch.sin(ra)] if self.encode_angle_by_sincos else [rg - ra]
        extra = boxes[:, 7:] - anchors[:, 7:]
        return torch.cat([xyz_t, dims_t, *rts, extra], dim=-1)

    def decode_torch(self, box_encodings, anchors):
        """box_encodings (...,code_size+C), anchors (...,7+C) -> boxes (...,7+C)"""
        xa, ya, za, dxa, dya, dza, ra = (anchors[..., i:i + 1] for i in range(7))
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        xg = box_encodings[..., 0:1] * diagonal + xa
        yg = box_encodings[..., 1:2] * diagonal + ya
        zg = box_encodings[..., 2:3] * dza + za
        dxg = torch.exp(box_encodings[..., 3:4]) * dxa
        dyg = torch.exp(box_encodings[..., 4:5]) * dya
        dzg = torch.exp(box_encodings[..., 5:6]) * dza
        if self.encode_angle_by_sincos:
            rg = torch.atan2(box_encodings[..., 7:8] + torch.sin(ra), box_encodings[..., 6:7] + torch.cos(ra))
            rest = box_encodings[..., 8:] + anchors[..., 7:]
        else:
            rg = box_encodings[..., 6:7] + ra
            rest = box_encodings[..., 7:] + anchors[..., 7:]
        return torch.cat([xg, yg, zg, dxg, dyg, dzg, rg, rest], dim=-1)
